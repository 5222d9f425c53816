"""Training-step orchestration on one GPU: the counterpart of the reference's
training loop (main.cu:612-805) over the C ABI.

Per step (stage order of main.cu, host round trips removed):
  rays -> trace(count) -> scan -> trace(write packed CSR)        main.cu:463-543,631-673
       -> sampler                                                :704
       -> encoding + network->forward (activations kept)         :715-728
       -> volume render forward                                  :737
       -> L2 loss                                                :759
       -> volume render backward                                 :767
       -> network->backward (+ hash-grid scatter)                :781
       -> Adam on the fp32 master copy, fp16 params re-packed    :787
The reference's per-batch cudaMalloc/cudaFree (:667-694,794-801) and host re-pack
(:646-673) do not exist here: every buffer is allocated once at capacity.

`mode`: "compat" reproduces the reference's arithmetic (REGULAR samples,
RTXN_VR_COMPAT forward/backward -- whose backward is not the gradient of its
forward); "nerf" is the corrected path (midpoint samples, world-space
steps, exact gradients) that actually converges.  All arithmetic is in librtxn.so;
torch only owns buffers and a few trivial elementwise glue ops.
"""
import json
import os
import struct

import numpy as np
import torch
import torch.distributed as dist

from . import api


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class _Stage:
    """HIP-event bracket around one stage of Trainer.step on the current stream (Trainer.time_stages)."""

    def __init__(self, tr, name):
        self.tr, self.name = tr, name

    def __enter__(self):
        if self.tr._stage_ev is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.tr._stage_ev is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.tr._stage_ev.append((self.name, self.e0, e1))
        return False


class Trainer:
    def __init__(self, grid_res, occupancy=None, encoding="hash", n_neurons=64, n_hidden_layers=4,
                 hashgrid=None, n_dir_freqs=4, batch_rays=4096, max_segments=None, lr=1e-3, loss_scale=128.0,
                 density_scale=1.0, mode="nerf", seed=1337, device="cuda", deterministic=None):
        self.R = grid_res
        self.dev = torch.device(device)
        self.occ = occupancy
        self.coarse = api.build_occupancy_mip(occupancy, grid_res) if (occupancy is not None and grid_res % 4 == 0) else None
        self.bricks = api.build_occupancy_bricks(occupancy, grid_res) if self.coarse is not None else None
        self.super_mip = api.build_occupancy_mip(self.coarse, grid_res // 4) if (self.coarse is not None and grid_res % 16 == 0) else None
        self.B = batch_rays
        self.mode = mode
        self.lr, self.loss_scale, self.density_scale = lr, loss_scale, density_scale
        self.step_count = 0
        self._stage_ev = None      # list of (stage, start event, end event) while time_stages() is collecting
        d = self.dev
        # ---- model -------------------------------------------------------------------------------
        self.encoding = encoding
        if encoding == "hash":
            self.hg = api.HashGrid(n_dir_freqs=n_dir_freqs, **(hashgrid or {}))
            E = self.hg.encoded_width()
            self.net = api.Network(n_neurons=n_neurons, n_hidden_layers=n_hidden_layers, n_encoded_features=E)
            g = torch.Generator().manual_seed(seed + 1)
            self.table_master = ((torch.rand(self.hg.n_params(), generator=g) * 2 - 1) * 1e-4).to(d)   # tcnn: U(-1e-4, 1e-4)
            self.table = self.table_master.half()
            self.table_m = torch.zeros_like(self.table_master)
            self.table_v = torch.zeros_like(self.table_master)
            # tiny-cuda-nn's Adam treats the table as "non-matrix" parameters: entries with a zero gradient are skipped and
            # each entry counts its own updates for the bias correction (librtxn: rtxn_adam_step_sparse).  Also the cheap
            # form -- a batch touches 0.2 .. 25 % of a hashed level.  RTXN_TABLE_ADAM=dense: every entry every step, global
            # step count (rounds 1-2; A/B).
            self.table_steps = torch.zeros(self.hg.n_params(), dtype=torch.int32, device=d)
            self.table_adam_sparse = os.environ.get("RTXN_TABLE_ADAM", "sparse") != "dense"
            self.dtable = torch.zeros_like(self.table_master)
            # The hashed levels' gradient lives in fp16 (dtable_h = parameters hashed_lo..): the scatter writes both features
            # of a corner with one packed fp16 atomic (tiny-cuda-nn's grid gradient is __half2 too), the data-parallel
            # exchange sends it as it is, and it is widened into dtable[hashed_lo:] for Adam.  The densely stored leading
            # levels (thousands of contributions per entry) stay fp32.  RTXN_HASH_GRAD_FP16=0: everything fp32.
            self.hashed_lo = self.hg.hashed_offset()
            self.hash_fp16 = (self.hg.cfg.n_features == 2 and self.hashed_lo < self.hg.n_params()
                              and os.environ.get("RTXN_HASH_GRAD_FP16", "1") != "0")
            self.dtable_h = (torch.zeros(self.hg.n_params() - self.hashed_lo, dtype=torch.float16, device=d)
                             if self.hash_fp16 else None)
        else:
            self.hg = None
            self.net = api.Network(n_neurons=n_neurons, n_hidden_layers=n_hidden_layers)
            E = self.net.encoded_width()
        self.E = E
        self.master = self.net.initialize_params(seed).to(d)       # fp32 master (main.cu:328-349)
        self.params = self.master.half()                           # fp16 params
        self.adam_m = torch.zeros_like(self.master)
        self.adam_v = torch.zeros_like(self.master)
        self.dparams = torch.zeros_like(self.master)
        self.net.set_params(self.params)
        # 64-wide models (configs[2]): forward without saved activations + ONE fused backward kernel that recomputes them and
        # keeps every weight gradient on the chip (librtxn: mlp_bwd_fused64_kernel).  RTXN_TRAIN_RECOMPUTE=0 selects the
        # three-kernel path (saved activations, dgrad chain, weight-gradient GEMM) for A/B runs.
        self.recompute = self.net.recompute_supported() and os.environ.get("RTXN_TRAIN_RECOMPUTE", "1") != "0"
        # launchSampler folded into the encoders / the hash scatter (they form the samples from the packed segments): the
        # float[S][5] samples and the separate sampler launch exist only for teacher rendering (render_rays(radiance_fn)) and
        # for tests (materialize_samples()).  RTXN_TRAIN_FOLD_SAMPLER=0: the reference's stage order, sampler first.
        self.fold_sampler = os.environ.get("RTXN_TRAIN_FOLD_SAMPLER", "1") != "0"
        # "nerf" mode: compositor forward + L2 + compositor backward as one launch (RTXN_TRAIN_FUSE_COMPOSITOR=0: three)
        self.fuse_compositor = os.environ.get("RTXN_TRAIN_FUSE_COMPOSITOR", "1") != "0"
        # The backward visits only the segments that carry a loss gradient (librtxn: rtxn_live_segments; most of a NeRF batch
        # lies behind the first surface, where dL/d(radiance) is exactly 0).  A column tile of the backward kernels is one
        # segment, so a block tile can be any eight of them.  Needs the folded sampler.  RTXN_TRAIN_LIVE_SEGMENTS=0: every segment.
        self.live_segments = self.fold_sampler and os.environ.get("RTXN_TRAIN_LIVE_SEGMENTS", "1") != "0"
        # Saved-activation models (128 wide, or RTXN_TRAIN_RECOMPUTE=0) in "nerf" mode: the forward of a step runs OUTPUTS-ONLY over
        # the whole batch, and the activations are saved afterwards for the live segments alone (rtxn_mlp_train_forward_live) --
        # the full saving forward is HBM-bound on activations of which 70-90 % are never read.  ("compat" mode: the reference's
        # backward has no transmittance factor, every segment is live, one saving pass is the cheaper form.)
        # RTXN_TRAIN_TWO_PASS=0: one saving pass over everything.
        # The reference's own model (8 x 128, 112 encoded features): the LEAN path (librtxn: rtxn_mlp_train_forward_lean /
        # _backward_lean) -- the forward keeps outputs + sign masks only, the weight gradient recomputes the activations from the
        # encoding: 5.1 instead of 8.8 KB moved per sample and half the workspace.  RTXN_TRAIN_LEAN=0: the saved-activation path (A/B).
        self.lean = (not self.recompute and encoding != "hash" and self.net.lean_supported()
                     and os.environ.get("RTXN_TRAIN_LEAN", "1") != "0")
        # ... and with the reference's Composite-Frequency(3 x 10, 2 x 12) encoding the sampler and the encoder are folded into both lean
        # kernels (rtxn_mlp_train_forward_lean_segments / _backward_lean_segments): no encoder launch, encT never written -- the same
        # values bit for bit.  RTXN_TRAIN_LEAN_FUSED=0: the staged encoder (A/B; librtxn's one-call step reads the same switch).
        self.lean_fused = (self.lean and self.fold_sampler and self.net.lean_fused_supported()
                           and os.environ.get("RTXN_TRAIN_LEAN_FUSED", "1") != "0")
        self.two_pass = (not self.recompute and not self.lean and self.live_segments and mode == "nerf"
                         and os.environ.get("RTXN_TRAIN_TWO_PASS", "1") != "0")
        # ---- per-step buffers at capacity ---------------------------------------------------------
        B = batch_rays
        self.max_segments = int(max_segments) if max_segments else 64 * B
        M, K = self.max_segments, api.NUM_SAMPLES_PER_SEGMENT
        self.view_dirs = torch.empty((B, 2), device=d)
        self.num_hits = torch.empty(B, dtype=torch.int32, device=d)
        self.indices = torch.empty(B, dtype=torch.int32, device=d)
        self.num_stored = torch.empty(B, dtype=torch.int32, device=d)   # segments actually written per ray (= num_hits unless cut off)
        self.truncated_steps = 0
        self.total = torch.zeros(1, dtype=torch.int32, device=d)
        # a batch is a small launch: several lanes walk each ray (bit-identical segments, shorter critical path)
        self.sub_rays = api.auto_sub_rays(B)              # for a full batch_rays launch; smaller launches take more lanes per ray
        self.sub_hits = torch.zeros(max(api.auto_sub_rays(n) * n for n in {min(B, t) for t in (11_999, 29_999, 119_999, 299_999, B)}),
                                    dtype=torch.int32, device=d)           # the largest lanes x rays of any launch of <= B rays
        self.scan_ws = torch.empty((api._lib.lib().rtxn_scan_workspace_bytes(B) + 3) // 4, dtype=torch.int32, device=d)
        self.start = torch.empty((M, 3), device=d)
        self.end = torch.empty((M, 3), device=d)
        self.seg_view = torch.empty((M, 2), device=d)      # (theta, phi) of each segment's ray, written by the traversal
        self.samples = torch.empty((M * K, 5), device=d)
        self.t_vals = torch.empty(M * K, device=d)
        Sp = api.padded_samples(M * K)
        self.encT = torch.empty((E, Sp), dtype=torch.float16, device=d)
        self.dencT = torch.empty((E, Sp), dtype=torch.float16, device=d) if encoding == "hash" else None
        if self.recompute:
            self.ws = None
        elif self.lean:
            self.ws = self.net.train_lean_workspace(M * K, device=d)                        # dZ | masks
        else:
            self.ws = self.net.train_workspace(M * K, device=d)                             # saved activations | dZ | masks
        self.out = torch.empty((M * K, 16), dtype=torch.float16, device=d)
        self.radiance = torch.empty((M * K, 4), device=d)
        self.dout = torch.empty((M * K, 4), dtype=torch.float16, device=d)
        self.live_ws = api.live_segments_workspace(M, device=d) if self.live_segments else None
        self.pixels = torch.empty((B, 3), device=d)
        self.loss_grads = torch.empty((B, 3), dtype=torch.float16, device=d)
        self.loss = torch.zeros(1, device=d)
        # data parallel (world > 1): the MLP gradient's all-reduce is issued from inside gradients(), right behind the MLP
        # backward, and runs beside the hash scatter; the hashed levels go through dp.Half2GradExchange (lists where a level
        # is sparse, RTXN_DP_SPARSE=0: always the dense fp16 level)
        self._dp_pending = None
        self._dp_table = None
        self.dp_sparse = os.environ.get("RTXN_DP_SPARSE", "1") != "0"         # "force": lists for every level (tests)
        # deterministic=True (or RTXN_DETERMINISTIC=1): every cross-workgroup gradient sum in 64-bit fixed point instead of float
        # atomics (librtxn: rtxn_set_deterministic_workspace): two runs, or two ranks, of the same steps end with identical bits.
        # The library's switch is process-wide, so every stepping method re-asserts this trainer's choice first (_det_select).
        self.deterministic = (os.environ.get("RTXN_DETERMINISTIC", "0") == "1") if deterministic is None else bool(deterministic)
        self._det_mlp = api.deterministic_shadow(self.master.numel(), device=d) if self.deterministic else None
        self._det_table = (api.deterministic_shadow(self.table_master.numel(), device=d)
                           if (self.deterministic and encoding == "hash") else None)

    # ------------------------------------------------------------------------------------------
    def _det_select(self):
        api.set_deterministic(self._det_mlp, self._det_table)

    def _segments(self, rays_o, rays_d, n):
        kw = dict(grid_res=self.R, rays_o=rays_o, rays_d=rays_d, width=n, height=1, ray_begin=0, ray_count=n,
                  occupancy=self.occ, occupancy_coarse=self.coarse, occupancy_bricks=self.bricks, occupancy_super=self.super_mip, mode=api.TRACE_DDA,
                  viewing_direction=self.view_dirs, num_hits=self.num_hits, sub_rays=api.auto_sub_rays(n), sub_hits=self.sub_hits)
        with _Stage(self, "trace_count"):
            api.trace_grid(None, **kw)
        with _Stage(self, "scan"):
            api.scan_hits(self.num_hits[:n], self.indices[:n], self.total, self.scan_ws)
        with _Stage(self, "trace_write"):
            api.trace_grid(None, indices=self.indices, start_points=self.start, end_points=self.end, seg_view=self.seg_view,
                           num_stored=self.num_stored, segment_capacity=self.max_segments, **kw)
        P = int(self.total.item())            # the reference synchronises here too (thrust::reduce, main.cu:632)
        if P > self.max_segments:
            # Rays whose segments do not fit are cut off ON THE DEVICE (num_stored < num_hits, never out of bounds) and
            # every later stage reads num_stored.  Raising here instead would leave the other ranks of a data-parallel
            # job waiting in the gradient all-reduce; the cut is counted and reported once.
            if self.truncated_steps == 0:
                import warnings
                warnings.warn(f"Trainer: batch needs {P} segments, capacity {self.max_segments}: rays truncated "
                              f"(raise max_segments); further truncations are counted in Trainer.truncated_steps")
            self.truncated_steps += 1
            P = self.max_segments
        return P

    def _stype(self):
        return api.SAMPLING_MIDPOINT_WORLD if self.mode == "nerf" else api.SAMPLING_REGULAR

    def _sample(self, n, P):
        """The standalone sampler (sampler/sampler.h:19-30): float[S][5] samples + t_vals."""
        with _Stage(self, "sampler"):
            api.launchSampler(self.start, self.end, self.view_dirs, self.t_vals, self.samples, n, self.R, self.num_stored,
                              self.indices, self._stype())
            if self.mode == "nerf" and self.density_scale != 1.0:
                self.t_vals[:P * api.NUM_SAMPLES_PER_SEGMENT].mul_(self.density_scale)

    def materialize_samples(self, n):
        """Tests / inspection: run the standalone sampler over the current batch's segments (fills self.samples; t_vals are
        rewritten with the same values the folded path produced)."""
        self._sample(n, min(int(self.total.item()), self.max_segments))

    def _forward(self, S, from_samples=False, save=True):
        """encoding + network->forward over the batch's S samples; save=False: outputs only (rendering; the first pass of the
        two-pass step)."""
        P = S // api.NUM_SAMPLES_PER_SEGMENT
        t_scale = self.density_scale if self.mode == "nerf" else 1.0
        if self.lean_fused and not from_samples and save:
            with _Stage(self, "mlp_fwd"):       # sampler + encoder + forward: one kernel, t_vals beside the outputs
                self.net.train_forward_lean_segments(self.start, self.end, self.seg_view, P, self._stype(), self.ws, self.out, self.radiance,
                                                     t_vals=self.t_vals, t_scale=t_scale)
            self._fused_batch = True
            return
        self._fused_batch = False
        with _Stage(self, "encode"):
            if self.fold_sampler and not from_samples:      # sampler + encoder in one pass over the segments; writes t_vals too
                if self.encoding == "hash":
                    self.hg.encode_segments(self.table, self.start, self.end, self.seg_view, P, self._stype(), self.encT,
                                            self.t_vals, t_scale)
                else:
                    self.net.encode_frequency_segments(self.start, self.end, self.seg_view, P, self._stype(), self.encT,
                                                       self.t_vals, t_scale)
            elif self.encoding == "hash":
                self.hg.encode(self.table, self.samples[:S], self.encT)
            else:
                self.net.encode_frequency(self.samples[:S], self.encT)
        with _Stage(self, "mlp_fwd"):
            if self.lean:
                self.net.train_forward_lean(self.encT, S, self.ws, self.out, self.radiance)
            elif self.recompute or not save:
                self.net.train_forward_outputs(self.encT, S, self.out, self.radiance)
            else:
                self.net.train_forward(self.encT, S, self.ws, self.out, self.radiance)

    def render_rays(self, rays_o, rays_d, radiance_fn=None):
        """Forward only.  radiance_fn(samples[S,5]) -> float[S,4] replaces the network (teacher rendering)."""
        n = rays_o.shape[0]
        P = self._segments(rays_o, rays_d, n)
        S = P * api.NUM_SAMPLES_PER_SEGMENT
        if radiance_fn is not None or not self.fold_sampler:
            self._sample(n, P)
        if S:
            if radiance_fn is None:
                self._forward(S, save=False)
            else:
                self.radiance[:S] = radiance_fn(self.samples[:S])
        vr = api.VR_NERF if self.mode == "nerf" else api.VR_COMPAT
        api.launch_volrender_cuda(None, self.radiance, self.num_stored, self.indices, self.t_vals, n,
                                  api.NUM_SAMPLES_PER_SEGMENT, self.pixels[:n], mode=vr)
        return self.pixels[:n]

    def render_pipeline(self, width, height, focal_length, max_segments=None, **kw):
        """A render.RenderPipeline over this trainer's LIVE model: the fast inference path (rtxn_render_frame: traversal, the fused
        frequency / hash-encode + MLP kernel, compact compositor; pipelined and ray-shardable) drawing whatever the parameters are
        at the time of each frame -- the pipeline holds the trainer's own model handle, hash table and occupancy tensors, and
        rtxn_mlp_set_params[_training] / Adam update them in place.  Compositor and sampling follow the trainer's mode ("nerf":
        midpoint samples, world-space steps x density_scale, exclusive transmittance; "compat": the reference's).  The
        frequency model renders through its fused inference kernel, whose weights a training-only update leaves stale:
        call sync_inference_weights() before rendering.  update_occupancy() refreshes the pipelines made here."""
        from . import render
        nerf = self.mode == "nerf"
        if self.encoding == "hash":
            if not api.hashmlp_supported(self.net, self.hg):
                raise api._lib.RtxnError("render_pipeline: no fused hash-grid inference kernel for this model (built: 64 wide, <= 8 hidden "
                                         "layers, 2 features per level, an even number of levels); use render_rays()")
            kw.update(hashgrid=self.hg, table=self.table)
        pipe = render.RenderPipeline(self.net, self.R, width, height, focal_length, occupancy=self.occ, max_segments=max_segments,
                                     vr_mode=api.VR_NERF if nerf else api.VR_COMPAT, step_scale=self.density_scale if nerf else 1.0, **kw)
        import weakref
        self._pipelines = [r for r in getattr(self, "_pipelines", []) if r() is not None] + [weakref.ref(pipe)]
        return pipe

    def sync_inference_weights(self):
        """Frequency models: re-pack the fused inference kernel's weights from the current parameters (rtxn_mlp_set_params); the
        per-step update only refreshes the training layouts.  (Hash models need nothing: their one 16x16x32 packing is refreshed
        by every update.)"""
        self.net.set_params(self.params)

    def gradients(self, rays_o, rays_d, targets):
        """Everything of a step up to (not including) the optimizer: traversal ... backward.  Leaves the loss-scaled
        gradient SUMS of this batch in self.dparams (MLP, tcnn layout) and self.dtable / self.dtable_h (hash grid: fp32 for the
        densely stored levels, fp16 for the hashed ones; table_grad() assembles them) and returns the number of samples.  The
        loss is the mean over THIS batch's 3n pixel components (tcnn L2, main.cu:759)."""
        n = rays_o.shape[0]
        K = api.NUM_SAMPLES_PER_SEGMENT
        vr = api.VR_NERF if self.mode == "nerf" else api.VR_COMPAT
        self._det_select()
        P = self._segments(rays_o, rays_d, n)
        S = P * K
        self._grads_clean = False          # the eager optimizer leaves the gradients in place (the captured one clears them)
        with _Stage(self, "zero_grads"):
            self.dparams.zero_()
            if self.encoding == "hash":
                if self.hash_fp16:
                    self.dtable[:self.hashed_lo].zero_()
                    self.dtable_h.zero_()
                else:
                    self.dtable.zero_()
        self._dp_pending = None
        if S == 0:
            self.loss.zero_()
            return 0
        if not self.fold_sampler:
            self._sample(n, P)
        self._forward(S, save=not self.two_pass)
        if self.mode == "nerf" and self.fuse_compositor:
            with _Stage(self, "composite_fwd+l2+bwd"):   # one launch: the backward's first sweep IS the forward
                api.volrender_l2_train(self.radiance, self.t_vals, self.num_stored, self.indices, n, K, targets, self.loss_scale,
                                       self.pixels[:n], self.loss_grads[:n], self.loss, self.dout)
        else:
            with _Stage(self, "composite_fwd"):
                api.launch_volrender_cuda(None, self.radiance, self.num_stored, self.indices, self.t_vals, n, K,
                                          self.pixels[:n], mode=vr)
            with _Stage(self, "l2_loss"):
                api.l2_loss(self.pixels[:n], targets, self.loss_scale, None, self.loss_grads[:n], self.loss)
            with _Stage(self, "composite_bwd"):
                api.launch_volrender_backward_cuda(None, self.loss_grads, self.radiance, self.t_vals, self.num_stored,
                                                   self.indices, n, K, self.dout, mode=vr)
        if self.live_segments:
            with _Stage(self, "live_segments"):
                api.live_segments(self.dout, P, self.max_segments, self.live_ws)
        if self.two_pass:
            with _Stage(self, "mlp_fwd_live"):       # the activations of the segments the backward is about to visit
                self.net.train_forward_live(self.encT, S, self.ws, self.live_ws)
        with _Stage(self, "mlp_bwd+wgrad"):
            if self.lean and getattr(self, "_fused_batch", False):
                self.net.train_backward_lean_segments(self.start, self.end, self.seg_view, P, self._stype(), self.out, self.dout, self.ws,
                                                      self.dparams, live_ws=self.live_ws if self.live_segments else None)
            elif self.lean:
                self.net.train_backward_lean(self.encT, self.out, self.dout, S, self.ws, self.dparams,
                                             live_ws=self.live_ws if self.live_segments else None)
            elif self.live_segments and self.recompute:
                self.net.train_backward_recompute_live(self.encT, self.out, self.dout, S, self.live_ws, self.dparams, self.dencT)
            elif self.live_segments:
                self.net.train_backward_live(self.encT, self.out, self.dout, S, self.ws, self.live_ws, self.dparams, self.dencT)
            elif self.recompute:
                self.net.train_backward_recompute(self.encT, self.out, self.dout, S, self.dparams, self.dencT)
            else:
                self.net.train_backward(self.encT, self.out, self.dout, S, self.ws, self.dparams, self.dencT)
        if _world() > 1:                       # the MLP gradient is complete: its all-reduce runs beside the hash scatter
            self._dp_pending = [dist.all_reduce(self.dparams, async_op=True)]
        if self.encoding == "hash":
            with _Stage(self, "hash_bwd"):
                if self.fold_sampler:
                    self.hg.backward_segments(self.start, self.end, P, self._stype(), self.dencT, self.dtable,
                                              self.dtable_h if self.hash_fp16 else None, live_ws=self.live_ws if self.live_segments else None)
                elif self.hash_fp16:
                    self.hg.backward_mixed(self.samples[:S], self.dencT, self.dtable, self.dtable_h)
                else:
                    self.hg.backward(self.samples[:S], self.dencT, self.dtable)
        return S

    def apply_gradients(self, grad_divisor=1.0):
        """optimizer->step (main.cu:787) on self.dparams / self.dtable; grad_divisor: ranks summed into them."""
        self.step_count += 1
        with _Stage(self, "adam"):
            api.adam_step(self.master, self.params, self.dparams, self.adam_m, self.adam_v, self.step_count, lr=self.lr,
                          loss_scale=self.loss_scale * grad_divisor)
            self.net.set_params_training(self.params)     # the Trainer only ever runs the training kernels
            if self.encoding == "hash":
                kw = dict(lr=self.lr * 10.0, eps=1e-15, loss_scale=self.loss_scale * grad_divisor)
                lo = self.hashed_lo
                if self.table_adam_sparse:
                    self._table_adam_sparse(**kw)
                elif self.hash_fp16:     # dense levels from the fp32 gradient, hashed levels straight from the fp16 one
                    if lo > 0:
                        api.adam_step(self.table_master[:lo], self.table[:lo], self.dtable[:lo], self.table_m[:lo], self.table_v[:lo],
                                      self.step_count, **kw)
                    api.adam_step_half_grads(self.table_master[lo:], self.table[lo:], self.dtable_h, self.table_m[lo:],
                                             self.table_v[lo:], self.step_count, **kw)
                else:
                    api.adam_step(self.table_master, self.table, self.dtable, self.table_m, self.table_v, self.step_count, **kw)

    def _table_adam_sparse(self, **kw):
        """rtxn_adam_step_sparse over the table: the densely stored levels from the fp32 gradient, the hashed ones from the fp16 one"""
        lo = self.hashed_lo
        parts = [(slice(0, lo), self.dtable[:lo]), (slice(lo, None), self.dtable_h)] if self.hash_fp16 else [(slice(None), self.dtable)]
        for sl, g in parts:
            if g.numel():
                api.adam_step_sparse(self.table_master[sl], self.table[sl], g, self.table_m[sl], self.table_v[sl], self.table_steps[sl], **kw)

    def table_grad(self):
        """The hash grid's gradient as ONE fp32 vector in table layout (a copy; tests and tools).  In the default mixed form
        the hashed levels' part is held in fp16 (self.dtable_h) and only self.dtable[:hashed_lo] of the fp32 buffer is live."""
        if not self.hash_fp16:
            return self.dtable.clone()
        out = self.dtable.clone()
        out[self.hashed_lo:] = self.dtable_h.float()
        return out

    def step(self, rays_o, rays_d, targets):
        """One optimisation step on a batch of rays; returns the (device) loss scalar.

        Data parallel (torch.distributed initialised): every rank holds B_local rays of the global batch;
        gradients are SUMMED across ranks and the mean over ranks is folded into Adam's loss_scale divisor.  EVERY rank
        takes part in the all-reduces and runs Adam in every step -- also a rank whose rays all miss the grid (its
        gradients are zero) -- so the ranks can neither deadlock nor drift apart in step count."""
        world = _world()
        S = self.gradients(rays_o, rays_d, targets)
        if S == 0 and world == 1:
            return self.loss                  # nothing to learn from: no Adam step, step_count unchanged
        if world > 1:
            with _Stage(self, "allreduce"):
                # issued by gradients() behind the MLP backward; a rank without samples returned before that and joins here
                pending, self._dp_pending = (self._dp_pending or [dist.all_reduce(self.dparams, async_op=True)]), None
                if self.encoding == "hash":
                    pending.extend(self._allreduce_table_grad())
                for w in pending:
                    w.wait()
                if self.encoding == "hash" and not self.hash_fp16:
                    self._finish_table_grad()
        self.apply_gradients(float(world))
        return self.loss

    def _allreduce_table_grad(self):
        """Sum the hash-grid gradient across ranks.  What a rank's 4096/N rays touch is NOT sparse in the hashed levels
        (83 k samples x 8 corners = 664 k touches per level against 2^19 entries at N = 8), so index + value lists would be
        larger than the table; the lever is the element size.  The densely stored leading levels (hashed_lo parameters, a few
        hundred KB) are all-reduced in fp32; the hashed levels go through an fp16 staging buffer -- tiny-cuda-nn keeps that
        gradient in fp16 throughout, and the loss scale keeps it in range -- which halves the bytes on the wire
        (config 3: 2 x 7/8 x 25.2 MB per rank and step at N = 8 instead of 2 x 7/8 x 52 MB).  Returns work handles; the
        caller waits and then calls _finish_table_grad()."""
        lo = self.hashed_lo
        n = self.dtable.numel()
        pending = []
        if lo > 0:
            pending.append(dist.all_reduce(self.dtable[:lo], async_op=True))
        if n > lo and self.hash_fp16 and self.dp_sparse:
            # Round 3: with the scatter restricted to the live segments a level IS sparse (0.2 .. 25 % of its entries per
            # rank and step): per level, lists of (index, half2) where they are smaller than the level -- see dp.py
            if self._dp_table is None:
                from .dp import Half2GradExchange
                self._dp_table = Half2GradExchange(self.dtable_h, 1 << self.hg.cfg.log2_hashmap_size,
                                                   force_lists=os.environ.get("RTXN_DP_SPARSE") == "force")
            pending.extend(self._dp_table.exchange())
            return pending
        if n > lo:
            if not self.hash_fp16:      # fp32 scatter (F != 2 or switched off): stage the hashed part in fp16 for the wire
                if self.dtable_h is None:
                    self.dtable_h = torch.empty(n - lo, dtype=torch.float16, device=self.dev)
                api.convert_f32_to_f16(self.dtable[lo:], self.dtable_h)
            pending.append(dist.all_reduce(self.dtable_h, async_op=True))
        return pending

    def _finish_table_grad(self):
        """fp32 scatter only: widen the exchanged fp16 staging copy back into dtable[hashed_lo:]."""
        if self.dtable_h is not None:
            api.convert_f16_to_f32(self.dtable_h, self.dtable[self.hashed_lo:])

    def dp_bytes_per_step(self, world):
        """Bytes one rank sends (= receives) per step in a ring all-reduce of its gradients, for DESIGN.md 6."""
        f = 2.0 * (world - 1) / world
        b = f * 4 * self.dparams.numel()
        if self.encoding == "hash":
            b += f * 4 * self.hashed_lo
            last = self._dp_table.last if self._dp_table is not None else None
            if last is not None and last["world"] == world:
                b += last["bytes"]             # what the last step's exchange of the hashed levels moved (lists + dense levels)
            else:
                b += f * 2 * (self.dtable.numel() - self.hashed_lo)      # before any exchange ran: every level dense
        return b

    def time_stages(self, rays_o, rays_d, targets, steps=5):
        """Run `steps` optimisation steps with HIP events around every stage (on the stream the kernels are launched on);
        returns {stage: mean ms}.  Diagnostic: the events add launches, so the sum exceeds an untimed step slightly."""
        acc = {}
        for _ in range(steps):
            self._stage_ev = []
            self.step(rays_o, rays_d, targets)
            torch.cuda.synchronize()
            for name, e0, e1 in self._stage_ev:
                acc.setdefault(name, []).append(e0.elapsed_time(e1))
            self._stage_ev = None
        return {k: float(np.mean(v)) for k, v in acc.items()}

    # ------------------------------------------------------------------------------------------ checkpoint
    # Layout (little endian): b"RTXNCKPT", u32 version, u32 json_len, json header, then for every entry of
    # header["arrays"] its raw bytes, 64-byte aligned.  The MLP block is the reference's params_buffer
    # (main.cu:328-342): fp32 master | fp16 params (the fp16 gradient third is not persisted) plus Adam's
    # moments; the hash grid adds its table the same way.
    def state_arrays(self):
        arrs = {"mlp_master": self.master, "mlp_params": self.params, "mlp_adam_m": self.adam_m, "mlp_adam_v": self.adam_v}
        if self.encoding == "hash":
            arrs.update(table_master=self.table_master, table_params=self.table, table_adam_m=self.table_m,
                        table_adam_v=self.table_v, table_adam_steps=self.table_steps)
        return arrs

    def save_checkpoint(self, path):
        arrs = {k: v.detach().cpu().numpy() for k, v in self.state_arrays().items()}
        cfg = self.net.cfg
        header = {"step": self.step_count, "encoding": self.encoding, "grid_res": self.R, "mode": self.mode,
                  "mlp": {f: getattr(cfg, f) for f, _ in cfg._fields_},
                  "hashgrid": None if self.hg is None else {**{f: getattr(self.hg.cfg, f) for f, _ in self.hg.cfg._fields_},
                                                            "n_dir_freqs": self.hg.n_dir_freqs},
                  "arrays": [{"name": k, "dtype": str(a.dtype), "shape": list(a.shape)} for k, a in arrs.items()]}
        blob = json.dumps(header).encode()
        with open(path, "wb") as f:
            f.write(b"RTXNCKPT" + struct.pack("<II", self._CKPT_VERSION, len(blob)) + blob)
            for a in arrs.values():
                f.write(b"\0" * ((-f.tell()) % 64))
                f.write(np.ascontiguousarray(a).tobytes())

    # version 2 (round 3's format + a guarantee): every array of state_arrays() is in the file.  Version 1 files written before
    # the table's per-entry Adam step counts existed (or with RTXN_TABLE_ADAM=dense) lack `table_adam_steps`; loading one used
    # to leave the counts at zero beside warmed moments -- every entry's bias correction restarted at t = 1, an effective table
    # learning rate of ~0.3x for thousands of steps, silently.  Now: a missing array is an error, except that a
    # version-1 file's missing step counts are filled with the file's global step (what the dense rule would have used), loudly.
    _CKPT_VERSION = 2

    def load_checkpoint(self, path):
        with open(path, "rb") as f:
            if f.read(8) != b"RTXNCKPT":
                raise ValueError(f"{path}: not an RTXN checkpoint")
            version, n = struct.unpack("<II", f.read(8))
            header = json.loads(f.read(n))
            if version not in (1, self._CKPT_VERSION) or header["encoding"] != self.encoding:
                raise ValueError(f"{path}: version {version} / encoding {header['encoding']} does not match this trainer")
            dst = self.state_arrays()
            missing = set(dst) - {ent["name"] for ent in header["arrays"]}
            if missing == {"table_adam_steps"} and version == 1:
                import warnings
                warnings.warn(f"{path}: version-1 checkpoint without per-entry table step counts: every entry's count is set to the "
                              f"checkpoint's step ({header['step']})")
                dst["table_adam_steps"].fill_(int(header["step"]))
            elif missing:
                raise ValueError(f"{path}: checkpoint lacks {sorted(missing)}, which this trainer's state needs")
            for ent in header["arrays"]:
                f.seek((-f.tell()) % 64, 1)
                a = np.frombuffer(f.read(int(np.prod(ent["shape"])) * np.dtype(ent["dtype"]).itemsize), dtype=ent["dtype"])
                t = dst[ent["name"]]
                if list(t.shape) != ent["shape"]:
                    raise ValueError(f"{path}: {ent['name']} has shape {ent['shape']}, model needs {list(t.shape)}")
                t.copy_(torch.from_numpy(a.reshape(ent["shape"]).copy()))
        self.step_count = header["step"]
        self.net.set_params(self.params)
        return header

    # ------------------------------------------------------------------------------------------ captured step (hipGraph)
    _LR_TABLE = 1 << 16     # beyond ~2^16 steps the bias correction is exactly 1.0f for beta2 <= 0.999

    def capture_step(self, n_rays, launch_segments=None, prefetch=False):
        """Capture the whole optimisation step for batches of exactly `n_rays` rays as a hipGraph: traversal (count, scan,
        write) -> rtxn_train_gradients (sampler ... backward with the segment count read ON THE DEVICE) -> Adam -> weight
        re-pack, with no host round trip anywhere (the reference synchronises for the segment count, main.cu:632, and so
        does step()).  Afterwards fill graph_rays_o / graph_rays_d / graph_targets (static device buffers) and call
        step_captured().

        launch_segments: the segment count the captured launches are SIZED for (grids and the row stride of the
        feature-major workspaces); default: the trainer's max_segments.  A batch that needs more is cut off on the device
        exactly as in step() and counted in truncated_steps; blocks past the live samples exit at once, so an estimate
        ~1.5x the typical count costs nothing measurable while a 20x one costs a few empty-block dispatches per kernel.

        prefetch=True: the traversal runs ONE BATCH AHEAD, as a parallel branch of the graph.  Traversal does not depend on
        the parameters, and on a 4096-ray batch it is pure latency (the longest ray's walk, ~80 us with the chip almost
        empty), so step_captured() then traverses the batch it is given beside the gradient kernels of the batch given to
        the PREVIOUS call: the first call only traverses (returns None), every later call returns the previous batch's loss,
        and flush_captured() trains on the last batch submitted.  Two segment-buffer sets alternate.  An occupancy grid
        updated between two calls takes effect one batch later.

        Data parallel: gradients and optimizer are captured as two graphs with the all-reduces between them."""
        if not self.fold_sampler:
            raise RuntimeError("capture_step needs the folded sampler (RTXN_TRAIN_FOLD_SAMPLER=0 is set)")
        n = int(n_rays)
        if n < 1 or n > self.B:
            raise ValueError(f"capture_step: n_rays = {n} outside [1, batch_rays = {self.B}]")
        cap = int(launch_segments) if launch_segments else self.max_segments
        cap = max(1, min(cap, self.max_segments))
        d = self.dev
        self._det_select()                      # read by the entry points as they are captured: baked into the graphs
        self.graph_rays_o = torch.zeros((n, 3), device=d)
        self.graph_rays_d = torch.zeros((n, 3), device=d)
        self.graph_rays_d[:, 2] = 1.0
        self.graph_targets = torch.zeros((n, 3), device=d)
        self._g_n, self._g_cap, self._g_prefetch = n, cap, bool(prefetch)
        # the traversal's outputs, once per buffer set (set 0 = the trainer's own buffers)
        names = ("view_dirs", "num_hits", "indices", "num_stored", "sub_hits", "total", "start", "end", "seg_view")
        set0 = {k: getattr(self, k) for k in names}
        sets = [set0]
        if prefetch:
            s1 = {k: torch.zeros_like(v) for k, v in set0.items()}
            for k in ("start", "end", "seg_view"):       # only the launch capacity is ever written
                s1[k] = torch.zeros((cap,) + tuple(set0[k].shape[1:]), device=d)
            sets.append(s1)
        for st in sets:
            st["targets"] = torch.zeros((n, 3), device=d)
            st["total_host"] = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._g_sets = sets
        self._g_pending = None          # prefetch: the set that holds a traversed, not yet trained batch
        self._g_next = 0
        # bias-corrected rates per step number, computed by the library's own host function (bit-identical to step());
        # the graph looks its entry up with a device-side step counter
        T = self._LR_TABLE
        tab = np.zeros((T, 2), np.float32)
        for t in range(1, T):
            tab[t, 0] = api.adam_effective_lr(self.lr, 0.9, 0.999, t)
            tab[t, 1] = api.adam_effective_lr(self.lr * 10.0, 0.9, 0.999, t)
            if t > 64 and tab[t, 0] == tab[t - 1, 0] and tab[t, 1] == tab[t - 1, 1] and tab[t - 1, 0] == tab[t - 2, 0]:
                tab[t:] = tab[t]          # converged: the rest of the table is this value
                break
        self._g_lr_table = torch.from_numpy(tab).to(d)
        self._g_step = torch.full((1,), self.step_count, dtype=torch.int64, device=d)
        self._g_idx = torch.zeros(1, dtype=torch.int64, device=d)
        self._g_lr = torch.zeros((1, 2), device=d)
        world = _world()
        self._g_world = world
        # data parallel + hash grid: the scatter is captured as a graph of its own (rtxn_train_batch.skip_table_backward), so
        # that the MLP gradient's all-reduce can be issued between the two and run beside it
        self._g_split = world > 1 and self.encoding == "hash" and self.live_segments
        self._g_split_table = None

        # one eager pass on a side stream (kernel attributes, lazy module state), then capture
        side = torch.cuda.Stream(device=d)
        self._g_side = side
        side.wait_stream(torch.cuda.current_stream())
        state = (self.master.clone(), self.params.clone(), self.adam_m.clone(), self.adam_v.clone())
        tstate = (self.table_master.clone(), self.table.clone(), self.table_m.clone(), self.table_v.clone(),
                  self.table_steps.clone()) if self.encoding == "hash" else None
        clear_grads = self._clear_grads
        with torch.cuda.stream(side):
            clear_grads()
            for k in range(len(sets)):
                self._captured_traverse(k)
            self._captured_gradients(0)
            if self._g_split:
                self._captured_table_bwd(0)
            self._captured_apply(float(world))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # undo the warm-up step: same parameters and step number as before capture_step()
        for dst, src in zip((self.master, self.params, self.adam_m, self.adam_v), state):
            dst.copy_(src)
        if tstate is not None:
            for dst, src in zip((self.table_master, self.table, self.table_m, self.table_v, self.table_steps), tstate):
                dst.copy_(src)
        self.net.set_params_training(self.params)
        self._g_step.fill_(self.step_count)
        self._g_step_host = self.step_count     # host mirror of the device counter (see _sync_device_step)
        clear_grads()
        torch.cuda.synchronize()
        for st in sets:
            st["total_host"][0] = 0        # the warm-up pass is not a step

        def capture(fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            return g

        def forked(k_traverse, then):
            """traversal into set k_traverse on the side stream, `then` on the capturing stream, joined at the end"""
            def body():
                main = torch.cuda.current_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self._captured_traverse(k_traverse)
                then()
                main.wait_stream(side)
            return body

        def train(k, with_apply):
            def body():
                self._captured_gradients(k)
                if with_apply:
                    self._captured_apply(float(world))
            return body

        one = world == 1
        if not prefetch:
            def whole():
                self._captured_traverse(0)
                train(0, one)()
            self._graphs = {"step": [capture(whole)]}
        else:
            # step[k]: traverse into set k beside training on set 1-k; flush[k]: train on set k; prime: traverse into set 0
            self._graphs = {"step": [capture(forked(k, train(1 - k, one))) for k in (0, 1)],
                            "flush": [capture(train(k, one)) for k in (0, 1)],
                            "prime": [capture(lambda k=k: self._captured_traverse(k)) for k in (0, 1)]}
        if not one:
            self._graphs["apply"] = capture(lambda: self._captured_apply(float(world)))
            if self._g_split:
                self._graphs["table_bwd"] = [capture(lambda k=k: self._captured_table_bwd(k)) for k in range(len(sets))]
        return self

    # ------------------------------------------------------------------------------------------ the step as ONE C call
    def entry_args(self, n_rays, launch_segments=None):
        """struct rtxn_train_step_args over this trainer's buffers: what a C++ host passes to rtxn_train_step (include/rtxn.h)
        -- traversal, rtxn_train_gradients, optimizer and weight re-pack of one batch in one call.  Inputs are read from
        graph_rays_o / graph_rays_d / graph_targets (created here if capture_step() has not).  The step counter the call
        advances lives on the device (entry_step)."""
        if _world() > 1:
            raise RuntimeError("entry_args: rtxn_train_step steps the optimizer inside the call; data-parallel training exchanges the "
                               "gradients between rtxn_train_gradients and the optimizer (step() / capture_step())")
        if self.encoding == "hash" and not self.table_adam_sparse:
            raise RuntimeError("entry_args: rtxn_train_step steps the table by tiny-cuda-nn's non-matrix rule (RTXN_TABLE_ADAM=dense is set)")
        n = int(n_rays)
        cap = max(1, min(int(launch_segments) if launch_segments else self.max_segments, self.max_segments))
        d = self.dev
        if getattr(self, "graph_rays_o", None) is None or self.graph_rays_o.shape[0] != n:
            self.graph_rays_o = torch.zeros((n, 3), device=d)
            self.graph_rays_d = torch.zeros((n, 3), device=d)
            self.graph_rays_d[:, 2] = 1.0
            self.graph_targets = torch.zeros((n, 3), device=d)
        self.entry_step = torch.full((1,), self.step_count, dtype=torch.int32, device=d)
        self._entry_lr = torch.zeros(1, device=d)
        hash_ = self.encoding == "hash"
        a = api._lib.TrainStepArgs()
        a.trace = api.trace_params(grid_res=self.R, rays_o=self.graph_rays_o, rays_d=self.graph_rays_d, width=n, height=1, ray_begin=0,
                                   ray_count=n, occupancy=self.occ, occupancy_coarse=self.coarse, occupancy_bricks=self.bricks,
                                   occupancy_super=self.super_mip, mode=api.TRACE_DDA, viewing_direction=self.view_dirs,
                                   num_hits=self.num_hits, sub_rays=api.auto_sub_rays(n), sub_hits=self.sub_hits)
        a.scan_workspace = self.scan_ws.data_ptr()
        a.scan_workspace_bytes = self.scan_ws.numel() * 4
        a.batch = api.train_batch(self.net, grid=self.hg if hash_ else None, n_dir_freqs=self.hg.n_dir_freqs if hash_ else 0,
                                  table=self.table if hash_ else None, start_points=self.start, end_points=self.end, seg_view=self.seg_view,
                                  num_stored=self.num_stored, indices=self.indices, total_segments=self.total, segment_capacity=cap,
                                  n_rays=n, sample_type=self._stype(), t_scale=self.density_scale if self.mode == "nerf" else 1.0,
                                  vr_mode=api.VR_NERF if self.mode == "nerf" else api.VR_COMPAT, targets=self.graph_targets,
                                  loss_scale=self.loss_scale, encT=self.encT, dencT=self.dencT, workspace=self.ws,
                                  output_half=self.out, radiance=self.radiance, t_vals=self.t_vals, radiance_gradients=self.dout,
                                  pixels=self.pixels, loss_gradients=self.loss_grads, loss_sum=self.loss, dparams=self.dparams,
                                  dtable=self.dtable if hash_ else None,
                                  dtable_hashed_half=self.dtable_h if (hash_ and self.hash_fp16) else None,
                                  live_ws=self.live_ws if self.live_segments else None, workspace_lean=self.lean)
        o = a.opt
        o.mlp_master, o.mlp_params_fp16 = self.master.data_ptr(), self.params.data_ptr()
        o.mlp_m, o.mlp_v = self.adam_m.data_ptr(), self.adam_v.data_ptr()
        if hash_:
            o.table_master, o.table_params_fp16 = self.table_master.data_ptr(), self.table.data_ptr()
            o.table_m, o.table_v, o.table_steps = self.table_m.data_ptr(), self.table_v.data_ptr(), self.table_steps.data_ptr()
        o.step, o.effective_lr = self.entry_step.data_ptr(), self._entry_lr.data_ptr()
        o.lr, o.beta1, o.beta2, o.eps = self.lr, 0.9, 0.999, 1e-8
        o.table_lr, o.table_eps, o.loss_scale_divisor = self.lr * 10.0, 1e-15, 1.0
        self._entry_args, self._entry_cap = a, cap
        self._entry_step_host = self.step_count      # what the device counter holds now (an eager step() in between is noticed by step_entry)
        self._entry_inputs = (self.graph_rays_o.data_ptr(), self.graph_rays_d.data_ptr(), self.graph_targets.data_ptr())
        self._clear_grads()            # the call's optimizer clears what it consumes; it must start from zeros
        return a

    def step_entry(self):
        """One optimisation step on graph_rays_o / graph_rays_d / graph_targets through rtxn_train_step (entry_args() first).
        Returns the (device) loss scalar; like step_captured() it never reads the segment count on the host."""
        self._det_select()
        if getattr(self, "_entry_args", None) is None:
            raise RuntimeError("step_entry: call entry_args() first")
        if self._entry_inputs != (self.graph_rays_o.data_ptr(), self.graph_rays_d.data_ptr(), self.graph_targets.data_ptr()):
            # capture_step() re-created the input buffers after entry_args(): the argument block still points at the old ones
            raise RuntimeError("step_entry: graph_rays_o / graph_rays_d / graph_targets were re-allocated since entry_args(); call entry_args() again")
        if not getattr(self, "_grads_clean", False):
            self._clear_grads()
        if int(self.step_count) != getattr(self, "_entry_step_host", self.step_count):
            self.entry_step.fill_(self.step_count)
        api.train_step(self._entry_args)
        self._grads_clean = True       # the call's optimizer cleared every gradient it consumed
        self.step_count += 1
        self._entry_step_host = self.step_count
        return self.loss

    def _clear_grads(self):
        self.dparams.zero_()
        if self.encoding == "hash":
            self.dtable.zero_()
            if self.dtable_h is not None:
                self.dtable_h.zero_()
        self._grads_clean = True

    def _captured_traverse(self, k):
        st, n, cap = self._g_sets[k], self._g_n, self._g_cap
        kw = dict(grid_res=self.R, rays_o=self.graph_rays_o, rays_d=self.graph_rays_d, width=n, height=1, ray_begin=0, ray_count=n,
                  occupancy=self.occ, occupancy_coarse=self.coarse, occupancy_bricks=self.bricks, occupancy_super=self.super_mip,
                  mode=api.TRACE_DDA, viewing_direction=st["view_dirs"], num_hits=st["num_hits"], sub_rays=api.auto_sub_rays(n),
                  sub_hits=st["sub_hits"])
        st["targets"].copy_(self.graph_targets)
        api.trace_grid(None, **kw)
        api.scan_hits(st["num_hits"][:n], st["indices"][:n], st["total"], self.scan_ws)
        api.trace_grid(None, indices=st["indices"], start_points=st["start"], end_points=st["end"], seg_view=st["seg_view"],
                       num_stored=st["num_stored"], segment_capacity=cap, **kw)
        st["total_host"].copy_(st["total"], non_blocking=True)      # 4 bytes for a later call's truncation check

    def _captured_gradients(self, k):
        st, n, cap = self._g_sets[k], self._g_n, self._g_cap
        # no fills: the captured Adam clears every gradient as it consumes it (zero_grads), capture_step() clears them once
        hash_ = self.encoding == "hash"
        api.train_gradients(self.net, grid=self.hg if hash_ else None, n_dir_freqs=self.hg.n_dir_freqs if hash_ else 0,
                            table=self.table if hash_ else None, start_points=st["start"], end_points=st["end"], seg_view=st["seg_view"],
                            num_stored=st["num_stored"], indices=st["indices"], total_segments=st["total"], segment_capacity=cap,
                            n_rays=n, sample_type=self._stype(), t_scale=self.density_scale if self.mode == "nerf" else 1.0,
                            vr_mode=api.VR_NERF if self.mode == "nerf" else api.VR_COMPAT, targets=st["targets"],
                            loss_scale=self.loss_scale, encT=self.encT, dencT=self.dencT, workspace=self.ws,
                            output_half=self.out, radiance=self.radiance, t_vals=self.t_vals, radiance_gradients=self.dout,
                            pixels=self.pixels, loss_gradients=self.loss_grads, loss_sum=self.loss, dparams=self.dparams,
                            dtable=self.dtable if hash_ else None, dtable_hashed_half=self.dtable_h if (hash_ and self.hash_fp16) else None,
                            live_ws=self.live_ws if self.live_segments else None, skip_table_backward=self._g_split,
                            workspace_lean=self.lean)

    def _captured_table_bwd(self, k):
        """the hash scatter of set k's batch over the live list the gradient graph left (its count is on the device)"""
        st = self._g_sets[k]
        self.hg.backward_segments(st["start"], st["end"], self._g_cap, self._stype(), self.dencT, self.dtable,
                                  self.dtable_h if self.hash_fp16 else None, live_ws=self.live_ws)

    def _captured_apply(self, grad_divisor):
        self._g_step.add_(1)
        torch.clamp(self._g_step, max=self._LR_TABLE - 1, out=self._g_idx)
        torch.index_select(self._g_lr_table, 0, self._g_idx, out=self._g_lr)
        lr_mlp, lr_tab = self._g_lr[0, 0:1], self._g_lr[0, 1:2]
        ls = self.loss_scale * grad_divisor
        api.adam_step_captured(self.master, self.params, self.dparams, self.adam_m, self.adam_v, lr_mlp, loss_scale=ls, zero_grads=True)
        self.net.set_params_training(self.params)
        if self.encoding == "hash" and self.table_adam_sparse:
            self._table_adam_sparse(lr=self.lr * 10.0, eps=1e-15, loss_scale=ls, zero_grads=True)    # no step number: per-entry counts
        elif self.encoding == "hash":
            lo = self.hashed_lo
            if self.hash_fp16:
                if lo > 0:
                    api.adam_step_captured(self.table_master[:lo], self.table[:lo], self.dtable[:lo], self.table_m[:lo], self.table_v[:lo],
                                           lr_tab, eps=1e-15, loss_scale=ls, zero_grads=True)
                api.adam_step_captured(self.table_master[lo:], self.table[lo:], self.dtable_h, self.table_m[lo:], self.table_v[lo:],
                                       lr_tab, eps=1e-15, loss_scale=ls, zero_grads=True)
            else:
                api.adam_step_captured(self.table_master, self.table, self.dtable, self.table_m, self.table_v, lr_tab, eps=1e-15,
                                       loss_scale=ls, zero_grads=True)

    def _sync_device_step(self):
        """The captured Adam looks its bias-corrected rate up with a DEVICE step counter that only replays advance.  An eager
        step() between two captured ones, or load_checkpoint(), moves step_count on the host alone: bring the device counter
        back in line before the next replay (host mirror compared, no synchronisation)."""
        if self._g_step_host != self.step_count:
            self._g_step.fill_(self.step_count)
            self._g_step_host = self.step_count

    def _check_truncation(self, k):
        th = self._g_sets[k]["total_host"]
        need = int(th[0])                  # written by an earlier replay (a stale read only delays the report)
        if need > self._g_cap:
            if self.truncated_steps == 0:
                import warnings
                warnings.warn(f"Trainer: a captured batch needed {need} segments, the graph is sized for {self._g_cap}: rays "
                              f"truncated (capture_step(launch_segments=...)); further truncations are counted in truncated_steps")
            self.truncated_steps += 1
            th[0] = 0

    def _finish_dp(self):
        """world > 1: sum the gradients the gradient graph left, then replay the optimizer graph"""
        pending = [dist.all_reduce(self.dparams, async_op=True)]
        if self.encoding == "hash":
            if self._g_split_table is not None:          # the scatter was left out of the gradient graph: it runs now, beside
                self._graphs["table_bwd"][self._g_split_table].replay()      # the MLP gradient's all-reduce
            pending.extend(self._allreduce_table_grad())
        for w in pending:
            w.wait()
        if self.encoding == "hash" and not self.hash_fp16:
            self._finish_table_grad()
        self._graphs["apply"].replay()

    def step_captured(self):
        """Replay the captured step on graph_rays_o / graph_rays_d / graph_targets; returns the (device) loss scalar -- with
        prefetch, of the batch submitted by the previous call (None on the first).  Unlike step() it never looks at the
        segment count on the host: a batch without any sample still runs Adam (on a zero gradient), and a truncated batch
        is noticed a call later (truncated_steps)."""
        if getattr(self, "_graphs", None) is None:
            raise RuntimeError("step_captured: call capture_step() first (also after update_occupancy() on a trainer that "
                               "was created without an occupancy grid: its graphs were dropped)")
        if not getattr(self, "_grads_clean", False):
            self._clear_grads()                       # an eager step() ran in between: its gradients are still in the buffers
        self._sync_device_step()
        if not self._g_prefetch:
            self._check_truncation(0)
            self.step_count += 1
            self._g_step_host += 1
            self._graphs["step"][0].replay()
            if self._g_world > 1:
                self._g_split_table = 0 if self._g_split else None
                self._finish_dp()
            return self.loss
        k = self._g_next
        self._g_next = 1 - k
        if self._g_pending is None:                     # nothing traversed yet: this call only traverses
            self._graphs["prime"][k].replay()
            self._g_pending = k
            return None
        self._check_truncation(self._g_pending)
        self.step_count += 1
        self._g_step_host += 1
        self._graphs["step"][k].replay()                # traverse into set k || train on set 1-k (the pending one)
        self._g_pending = k
        if self._g_world > 1:
            self._g_split_table = 1 - k if self._g_split else None
            self._finish_dp()
        return self.loss

    def flush_captured(self):
        """prefetch only: train on the batch the last step_captured() call submitted; returns its loss (None if there is none)."""
        if not getattr(self, "_g_prefetch", False) or self._g_pending is None:
            return None
        if getattr(self, "_graphs", None) is None:
            raise RuntimeError("flush_captured: the captured graphs were dropped (update_occupancy on a dense trainer); capture_step() again")
        if not getattr(self, "_grads_clean", False):
            self._clear_grads()
        self._sync_device_step()
        k = self._g_pending
        self._check_truncation(k)
        self.step_count += 1
        self._g_step_host += 1
        self._graphs["flush"][k].replay()
        self._g_pending = None
        if self._g_world > 1:
            self._g_split_table = k if self._g_split else None
            self._finish_dp()
        return self.loss

    # ------------------------------------------------------------------------------------------ occupancy maintenance
    @torch.no_grad()
    def update_occupancy(self, threshold=0.01, chunk=1 << 20):
        """Density-driven refresh of the occupancy grid (the reference only ever builds the dense grid once,
        main.cu:393-399): evaluate sigma at every cell centre and keep cells whose optical thickness over one
        cell, sigma * density_scale * cell_size, exceeds `threshold`.  Returns the occupied fraction."""
        R = self.R
        n = R ** 3
        chunk = min(chunk, self.samples.shape[0])
        sigma = torch.empty(n, device=self.dev)
        ax = (torch.arange(R, device=self.dev, dtype=torch.float32) + 0.5) * (2.0 / R) - 1.0
        for s0 in range(0, n, chunk):
            idx = torch.arange(s0, min(s0 + chunk, n), device=self.dev)
            z, y, x = idx % R, (idx // R) % R, idx // (R * R)                  # bit index = (x*R + y)*R + z
            pts = torch.stack([ax[x], ax[y], ax[z], torch.zeros_like(ax[x]), torch.zeros_like(ax[x])], dim=1).contiguous()
            m = pts.shape[0]
            if self.encoding == "hash":
                self.hg.encode(self.table, pts, self.encT)
            else:
                self.net.encode_frequency(pts, self.encT)
            self.net.train_forward_outputs(self.encT, m, self.out, self.radiance)
            sigma[s0:s0 + m] = self.radiance[:m, 3]
        thick = sigma * (self.density_scale * 2.0 / R)
        self._set_occupancy(api.occupancy_from_density(thick, threshold, R))
        return float((thick > threshold).float().mean().item())

    def _set_occupancy(self, occ):
        """Install a new occupancy bitfield.  The four traversal inputs (fine bits, 4^3 mip, bricks, 16^3 mip) keep their
        ADDRESSES once they exist: capture_step() bakes those device pointers into the traversal nodes of its graphs, so the
        refresh copies into the buffers the graphs read instead of re-binding the attributes (a dropped tensor's memory
        goes back to the caching allocator and may be reused under a replay).  A trainer that started dense (occupancy=None)
        has no buffer for the graphs to read, so its captured graphs are dropped and capture_step() must be called again."""
        R = self.R
        new = {"occ": occ,
               "coarse": api.build_occupancy_mip(occ, R) if R % 4 == 0 else None}
        new["bricks"] = api.build_occupancy_bricks(occ, R) if R % 4 == 0 else None
        new["super_mip"] = api.build_occupancy_mip(new["coarse"], R // 4) if R % 16 == 0 else None
        rebound = False
        for name, t in new.items():
            cur = getattr(self, name)
            if cur is not None and t is not None and cur.shape == t.shape and cur.dtype == t.dtype:
                cur.copy_(t)
            else:
                setattr(self, name, t)
                rebound = rebound or (cur is not None or t is not None)
        if rebound and getattr(self, "_graphs", None) is not None:
            self._graphs = None            # pointers changed under the captured traversal: step_captured() asks for a re-capture
        for ref in getattr(self, "_pipelines", []):
            pipe = ref()
            if pipe is not None and pipe.occ is not None:
                pipe.set_occupancy(self.occ)     # the renderer keeps its own mip / brick hierarchy: rebuild it from the new bits


def camera_rays(look_at, focal, width, height, device="cuda", origin_scale=0.1):
    """Pinhole rays of optixPrograms.cu:43-82 as explicit (o, d) tensors (host-side helper for building
    training batches from several poses; the arithmetic that matters happens in rtxn_trace_grid)."""
    la = torch.as_tensor(look_at, dtype=torch.float64).reshape(4, 4)
    xs = (2 * (torch.arange(width, dtype=torch.float64) + 0.5) / width - 1) * (width / height)
    ys = 2 * (torch.arange(height, dtype=torch.float64) + 0.5) / height - 1
    v, u = torch.meshgrid(ys, xs, indexing="ij")
    dirs = torch.stack([u, v, -torch.full_like(u, focal)], dim=-1).reshape(-1, 3)
    d = dirs @ la[:3, :3].T
    d = d / d.norm(dim=1, keepdim=True)
    o = (la[:3, 3] * origin_scale).expand_as(d)
    return o.float().contiguous().to(device), d.float().contiguous().to(device)


class RayDataset:
    """Dataset of rays resident on the device: the counterpart of the reference's dataset build (main.cu:463-543: one
    traversal launch per training image, every buffer copied to the host and four mallocs per ray) and of its
    std::random_shuffle + host batch gather (:612-629).  Here the rays (origin, direction) and their ground-truth pixels
    stay in HBM as three flat tensors and a batch is an on-device random gather; traversal happens per batch inside
    Trainer.step, so nothing per-ray is ever materialised on the host."""

    def __init__(self, rays_o, rays_d, pixels):
        self.rays_o, self.rays_d, self.pixels = rays_o, rays_d, pixels
        self.n = rays_o.shape[0]

    @classmethod
    def from_images(cls, dataset, corrected_focal=True, origin_scale=0.1, device="cuda"):
        """dataset: rtx_nerf_amd.loader.ImageDataset.  corrected_focal: 1/tan(camera_angle_x/2) in image half-widths
        instead of the reference's 1/tan(0.5*focal_px) (quirk Q1); origin_scale 0.1 is the reference's origin/10 (Q2)."""
        import math
        W, H = dataset.image_width, dataset.image_height
        if corrected_focal:
            focal = 1.0 / math.tan(0.5 * dataset.camera_angle_x)
        else:
            focal = float(np.float32(1.0) / np.tan(np.float32(0.5) * np.float32(dataset.focal)))
        ro, rd = [], []
        for pose in dataset.poses:
            o, d = camera_rays(pose, focal, W, H, device=device, origin_scale=origin_scale)
            ro.append(o)
            rd.append(d)
        pix = torch.from_numpy(np.ascontiguousarray(dataset.images, dtype=np.float32).reshape(-1, 3)).to(device)
        return cls(torch.cat(ro), torch.cat(rd), pix), focal

    def sample_batch(self, batch, generator=None):
        idx = torch.randint(0, self.n, (batch,), device=self.rays_o.device, generator=generator)
        return self.rays_o[idx].contiguous(), self.rays_d[idx].contiguous(), self.pixels[idx].contiguous()


def psnr(pred, target):
    mse = float(((pred - target) ** 2).mean())
    return 10.0 * np.log10(1.0 / max(mse, 1e-12))
