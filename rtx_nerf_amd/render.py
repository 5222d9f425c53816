"""The render hot path on one GPU, as a caller of librtxn's frame entry points (include/rtxn.h: rtxn_render_*).

Counterpart of the stage order of the reference's main loop (main.cu:463-543 traversal, :631-637 compaction, :704
sampler, :721 MLP, :728 glue, :737 compositing), minus its host round trips: the reference copies every traversal
buffer to the host, mallocs four arrays per ray, re-packs the batch on the CPU and cudaMalloc/cudaFrees six buffers per
batch.  Here the whole frame

    trace (count) -> scan -> trace (write packed CSR) -> sampler+encode+MLP (one kernel) -> volume render

is ONE C call (rtxn_render_frame; rtxn_render_frame_async for the three-stream pipelined form) that enqueues every stage
without a host synchronisation; a C++ host gets exactly the same frame (examples/render_host.cpp).  This module only owns
the workspace tensor, maps torch views onto the slot buffers for tests and tools, and applies the overflow policy.

Radiance model: an api.Network with the reference's Composite-Frequency encoding (fused frequency kernels), or -- with
hashgrid= / table= -- a pre-encoded 64-wide api.Network behind an api.HashGrid (the fused hash-encode + MLP kernel): what
train.Trainer(encoding="hash") trains.
"""
import ctypes as C

import torch

from . import _lib, api

RENDER_FLOAT4 = 1     # enum rtxn_render_flags
RENDER_STABLE_INPUTS = 2


class RenderPipeline:
    def __init__(self, network, grid_res, width, height, focal_length, aspect_ratio=None, occupancy=None,
                 max_rays=None, max_segments=None, trace_mode=api.TRACE_DDA, vr_mode=api.VR_COMPAT,
                 device="cuda", window=(0, 0), step_scale=1.0, sub_rays=None, compact=None, on_overflow="raise",
                 hashgrid=None, table=None, sample_type=None, n_slots=None, stable_inputs=False):
        self.net = network
        self.hg, self.table = hashgrid, table
        if (hashgrid is None) != (table is None):
            raise ValueError("hashgrid= and table= go together")
        self.R = grid_res
        self.W, self.H = width, height
        self.focal = float(focal_length)
        self.aspect = float(width) / float(height) if aspect_ratio is None else float(aspect_ratio)
        self.trace_mode, self.vr_mode = trace_mode, vr_mode
        self.window = window   # (chunk, stride) ray interleave of this shard, see rtxn_trace_params
        # sub_rays = Q in {2,4,8,16,..}: Q lanes walk consecutive pieces of each ray (rtxn_trace_params.sub_rays): same
        # segments bit for bit, a shorter critical path -- worth it when the launch is small (a shard, a training batch)
        n_rays = width * height if max_rays is None else max_rays
        self.sub_rays = (api.auto_sub_rays(n_rays) if sub_rays is None else int(sub_rays)) if trace_mode == api.TRACE_DDA else 0
        self.step_scale = float(step_scale)
        # sample placement: the frequency model's fused kernel samples REGULAR (sampler.cu:52-66); a hash model defaults to
        # what its compositor wants (RTXN_VR_NERF: sub-interval midpoints + world-space steps, as train.Trainer's "nerf" mode)
        if sample_type is None:
            sample_type = api.SAMPLING_MIDPOINT_WORLD if (hashgrid is not None and vr_mode == api.VR_NERF) else api.SAMPLING_REGULAR
        self.sample_type = sample_type
        # compact (default wherever it applies): the MLP kernel hands the compositor the network's own half outputs
        # (8 B/sample) and no per-sample t_vals -- bit-identical pixels at 40 % of the intermediate's bytes (0.8 instead of
        # 2.0 GB per bench frame).  compact=False: the reference's convertHalfToFloat layout (float4 + t_vals), frequency model.
        if compact is None:
            compact = hashgrid is not None or vr_mode == api.VR_COMPAT
        self.compact = bool(compact)
        if self.compact and hashgrid is None and vr_mode != api.VR_COMPAT:
            raise ValueError("the frequency model's compact hand-over is RTXN_VR_COMPAT only (REGULAR t_vals are implicit)")
        self.dev = torch.device(device)
        # A pose that needs more segments than the buffers hold is cut off ON THE DEVICE (never out of bounds).  So that
        # this cannot pass unnoticed outside the calibrated pose set, every frame copies its segment count to pinned host
        # memory (4 bytes, async, inside rtxn_render_frame) and the NEXT call -- or finish() -- looks at the counts that have
        # arrived (rtxn_render_status): on_overflow = "raise" (default): RuntimeError; "grow": re-create the renderer at 1.5x
        # the need (a pipeline stall, once) and go on -- the cut frame was delivered truncated either way and is counted in
        # overflow_frames; "ignore": count only.
        if on_overflow not in ("raise", "grow", "ignore"):
            raise ValueError("on_overflow must be 'raise', 'grow' or 'ignore'")
        self.on_overflow = on_overflow
        # stable_inputs: the caller's promise for render_async (RTXN_RENDER_STABLE_INPUTS, include/rtxn.h): device poses are
        # uploaded before the call and left alone until their frame has been traversed.  Without it the traversal of every
        # pipelined frame waits for the caller's stream (correct for a pose buffer rewritten per frame, ~3 % slower).
        self.stable_inputs = bool(stable_inputs)
        self.overflow_frames = 0
        self.occ = occupancy
        self.max_rays = n_rays
        # default capacity: every ray crossing a full grid diagonal's worth of occupied cells is far too
        # pessimistic; callers size it from a counting pass (see calibrate())
        self.max_segments = int(max_segments) if max_segments is not None else 16 * n_rays
        import os
        self.n_slots = int(n_slots) if n_slots else int(os.environ.get("RTXN_ASYNC_SLOTS", "3"))
        self.pixels = torch.empty((n_rays, 3), device=self.dev)
        self.look_at = torch.zeros(16, device=self.dev)
        self._h = None
        self._comp_stream = None
        self._create()

    # ------------------------------------------------------------------------------------------ the C renderer
    def _config(self):
        c = _lib.RenderConfig()
        c.mlp = self.net._h
        c.grid = self.hg._h if self.hg is not None else None
        c.table_fp16 = api._ptr(self.table, torch.float16, "table") if self.table is not None else None
        c.n_dir_freqs = self.hg.n_dir_freqs if self.hg is not None else 0
        c.width, c.height = self.W, self.H
        c.focal_length, c.aspect_ratio = self.focal, self.aspect
        c.max_rays = self.max_rays
        c.window_chunk, c.window_stride = self.window
        c.grid_res = self.R
        c.occupancy = api._ptr(self.occ, torch.int32, "occupancy") if self.occ is not None else None
        c.trace_mode, c.sub_rays = self.trace_mode, self.sub_rays
        c.vr_mode, c.sample_type, c.step_scale = self.vr_mode, self.sample_type, self.step_scale
        c.max_segments = self.max_segments
        c.n_slots = self.n_slots
        c.flags = (0 if self.compact else RENDER_FLOAT4) | (RENDER_STABLE_INPUTS if self.stable_inputs else 0)
        return c

    def _destroy(self):
        h, self._h = self._h, None
        if h and _lib is not None and getattr(_lib, "_lib", None) is not None:
            _lib._lib.rtxn_render_destroy(h)

    def __del__(self):
        self._destroy()

    def _create(self):
        """(Re)create the C renderer for the current max_segments: one workspace tensor, laid out by librtxn."""
        self._destroy()
        lib = _lib.lib()
        cfg = self._config()
        need = lib.rtxn_render_workspace_bytes(C.byref(cfg))
        if need == 0:
            msg = lib.rtxn_last_error()
            raise _lib.RtxnError(f"rtxn_render_workspace_bytes: {msg.decode() if msg else '?'}")
        self._ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        h = C.c_void_p()
        _lib.check(lib.rtxn_render_create(C.byref(cfg), C.c_void_p(self._ws.data_ptr()), need, C.byref(h)), "rtxn_render_create")
        self._h = h
        self._seen_overflows = 0
        self._slots = [self._slot_views(i) for i in range(self.n_slots)]
        for f in ("num_hits", "num_hits_c", "indices", "total", "start", "end", "seg_view", "radiance", "t_vals", "seg_step", "view_dirs"):
            setattr(self, f, getattr(self._slots[0], f))

    def _slot_views(self, i):
        """torch views of slot i's device buffers inside the workspace (tests, tools; the C side owns the layout)."""
        from types import SimpleNamespace
        ptrs = [C.c_void_p() for _ in range(11)]
        _lib.check(_lib.lib().rtxn_render_slot_buffers(self._h, i, *[C.byref(p) for p in ptrs]), "rtxn_render_slot_buffers")
        base, n, m, K = self._ws.data_ptr(), self.max_rays, self.max_segments, api.NUM_SAMPLES_PER_SEGMENT

        def view(p, count, dtype, shape):
            off = p.value - base
            nbytes = count * torch.empty((), dtype=dtype).element_size()
            return self._ws[off:off + nbytes].view(dtype).view(shape)

        g = SimpleNamespace()
        g.num_hits = view(ptrs[0], n, torch.int32, (n,))
        g.num_hits_c = view(ptrs[1], n, torch.int32, (n,))      # segments actually stored per ray
        g.indices = view(ptrs[2], n, torch.int32, (n,))
        g.total = view(ptrs[3], 1, torch.int32, (1,))
        g.start = view(ptrs[4], 3 * m, torch.float32, (m, 3))
        g.end = view(ptrs[5], 3 * m, torch.float32, (m, 3))
        g.seg_view = view(ptrs[6], 2 * m, torch.float32, (m, 2))
        g.radiance = (view(ptrs[7], m * K * 4, torch.float16, (m * K, 4)) if self.compact
                      else view(ptrs[7], m * K * 4, torch.float32, (m * K, 4)))
        g.t_vals = view(ptrs[8], m * K, torch.float32, (m * K,)) if ptrs[8].value else None
        g.seg_step = view(ptrs[9], m, torch.float32, (m,)) if ptrs[9].value else None
        g.view_dirs = view(ptrs[10], 2 * n, torch.float32, (n, 2))
        return g

    def _status(self, wait=False):
        st = _lib.RenderStats()
        _lib.check(_lib.lib().rtxn_render_status(self._h, 1 if wait else 0, C.byref(st)), "rtxn_render_status")
        return st

    def _check_overflow(self, wait=False):
        """Apply the overflow policy to the frames whose segment count has reached the host (no synchronisation unless
        `wait`: the caller then wants every enqueued frame looked at)."""
        if torch.cuda.is_current_stream_capturing():
            return
        st = self._status(wait)
        new = st.overflow_frames - self._seen_overflows
        if new <= 0:
            return
        self._seen_overflows = st.overflow_frames
        self.overflow_frames += new
        need = int(st.max_segments_needed)
        if self.on_overflow == "raise":
            raise RuntimeError(f"RenderPipeline: a frame needed {need} segments but the buffers hold {self.max_segments}: its rays "
                               f"were truncated.  calibrate() with that pose, raise max_segments, or use on_overflow='grow'")
        if self.on_overflow == "grow":
            torch.cuda.synchronize()                      # frames in flight still use the old buffers
            self.max_segments = int(need * 1.5) + 1024
            self._create()

    # ------------------------------------------------------------------------------------------ frames
    def set_pose(self, look_at):
        """look_at: 16 floats (host or device), row-major camera-to-world (params.h:17)."""
        self.look_at.copy_(torch.as_tensor(look_at, dtype=torch.float32).reshape(16), non_blocking=True)

    def set_occupancy(self, occupancy):
        """Point the renderer at an updated occupancy bitfield of the same resolution (rebuilds the mip / brick hierarchy)."""
        self.occ = occupancy
        _lib.check(_lib.lib().rtxn_render_set_occupancy(self._h, api._ptr(occupancy, torch.int32, "occupancy"), api._stream()),
                   "rtxn_render_set_occupancy")

    def count_segments(self, ray_begin=0, ray_count=None):
        """Counting pass + scan for the current pose; returns the number of segments (host int, synchronises)."""
        n = self.max_rays if ray_count is None else ray_count
        out = C.c_long(0)
        _lib.check(_lib.lib().rtxn_render_count_segments(self._h, C.c_void_p(self.look_at.data_ptr()), ray_begin, n, C.byref(out),
                                                         api._stream()), "rtxn_render_count_segments")
        return int(out.value)

    def calibrate(self, poses, ray_begin=0, ray_count=None, margin=1.10):
        """Size the segment buffers for a set of poses (outside any timed region)."""
        worst = 0
        for p in poses:
            self.set_pose(p)
            worst = max(worst, self.count_segments(ray_begin, ray_count))
        need = int(worst * margin) + 1024
        if need > self.max_segments or need < self.max_segments // 2:
            torch.cuda.synchronize()
            self.max_segments = need
            self._create()
        return worst

    def render(self, ray_begin=0, ray_count=None, out=None):
        """Enqueue one frame (or one shard of it) on the current stream; returns the
        pixel buffer view float[ray_count, 3] (or `out`).  No host synchronisation."""
        n = self.max_rays if ray_count is None else ray_count
        pixels = self.pixels[:n] if out is None else out
        self._check_overflow()
        _lib.check(_lib.lib().rtxn_render_frame(self._h, 0, C.c_void_p(self.look_at.data_ptr()), ray_begin, n,
                                                api._ptr(pixels, torch.float32, "pixels"), api._stream()), "rtxn_render_frame")
        return pixels

    def finish(self):
        """Wait for every enqueued frame and apply the overflow policy to all of them (raises under "raise")."""
        self.drain_async()
        torch.cuda.synchronize()
        self._check_overflow(wait=True)

    # A frame is three dependent stages with very different bounds: traversal (two latency-bound passes + scan, ~0.45 ms
    # whatever the ray count), the MLP kernel (MFMA-bound, 96 % of the time) and the compositor (HBM-bound).
    # rtxn_render_frame_async puts them on three HIP streams and rotates through n_slots buffer slots, so that while the MLP
    # kernel of frame i runs, frame i+1 is traversed and frame i-1 composited (DESIGN.md 5.1 has the measurements).  The host
    # never synchronises; it simply runs ahead.  This matters most when the frame is sharded over N GPUs (the fixed
    # traversal latency is 16 % of a rank's frame at N = 8).
    def render_async(self, look_at, ray_begin=0, ray_count=None, out=None):
        """Enqueue one frame, software-pipelined against its neighbours.  look_at: 16 floats -- a DEVICE tensor (copied into
        the slot on the traversal stream, which waits for the current stream first unless the pipeline was built with
        stable_inputs=True) or a HOST array / CPU tensor (rtxn_render_frame_async_host: staged through pinned memory, fully
        overlapped).  Returns (pixels, None, comp_stream): `pixels` is complete on `comp_stream`; follow-up work on it (a
        gather, a copy) is best enqueued there, or call drain_async()."""
        n = self.max_rays if ray_count is None else ray_count
        pixels = self.pixels[:n] if out is None else out
        self._check_overflow()
        comp = C.c_void_p()
        if isinstance(look_at, torch.Tensor) and look_at.is_cuda:
            _lib.check(_lib.lib().rtxn_render_frame_async(self._h, api._ptr(look_at, torch.float32, "look_at"), ray_begin, n,
                                                          api._ptr(pixels, torch.float32, "pixels"), api._stream(), C.byref(comp)),
                       "rtxn_render_frame_async")
        else:
            import numpy as np
            host = np.ascontiguousarray(np.asarray(look_at, dtype=np.float32).reshape(16))
            _lib.check(_lib.lib().rtxn_render_frame_async_host(self._h, host.ctypes.data_as(C.POINTER(C.c_float)), ray_begin, n,
                                                               api._ptr(pixels, torch.float32, "pixels"), api._stream(), C.byref(comp)),
                       "rtxn_render_frame_async_host")
        if self._comp_stream is None or self._comp_stream.cuda_stream != comp.value:
            self._comp_stream = torch.cuda.ExternalStream(comp.value, device=self.dev)
        return pixels, None, self._comp_stream

    def drain_async(self):
        """Make the current stream wait for every frame enqueued with render_async."""
        if self._h:
            _lib.check(_lib.lib().rtxn_render_drain(self._h, api._stream()), "rtxn_render_drain")

    def capture(self, ray_begin=0, ray_count=None, out=None):
        """Capture one frame into a hipGraph (torch.cuda.CUDAGraph over rtxn_render_frame: nothing in it allocates,
        synchronises or touches the host).  Returns (graph, pixels): update the pose with set_pose() and call graph.replay().
        The frame must have been rendered once eagerly (kernel attributes)."""
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            pixels = self.render(ray_begin, ray_count, out)
        return g, pixels

    def overflowed(self):
        """True if a frame was ever truncated (synchronises: also looks at the frames still in flight)."""
        if self.on_overflow == "raise":
            st = self._status(wait=True)
            return self.overflow_frames > 0 or st.overflow_frames > 0
        self._check_overflow(wait=True)
        return self.overflow_frames > 0

    # ------------------------------------------------------------------------------------------ diagnostics
    def shade_again(self, slot=0):
        """Re-issue the sampler+encode+MLP launch of a frame over slot `slot`'s current segments, on the current stream -- the
        very launch rtxn_render_frame makes (same entry point, same buffers) -- so that tools can bracket the dominant kernel
        alone with HIP events (bench.py's roofline)."""
        g = self._slots[slot]
        if self.hg is not None:
            api.hashmlp_forward_segments(self.net, self.hg, self.table, g.start, g.end, g.seg_view, g.total, self.max_segments,
                                         g.radiance, self.sample_type, self.step_scale, g.seg_step)
        elif self.compact:
            self.net.forward_segments_compact(g.start, g.end, g.seg_view, g.total, self.max_segments, g.radiance)
        else:
            raise RuntimeError("shade_again: compact pipelines only")
