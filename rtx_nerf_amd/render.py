"""Stage orchestration of the render hot path on one GPU.

Counterpart of the stage order of the reference's main loop
(main.cu:463-543 traversal, :631-637 compaction, :704 sampler, :721 MLP,
:728 glue, :737 compositing), minus its host round trips: the reference copies
every traversal buffer to the host, mallocs four arrays per ray, re-packs the
batch on the CPU and cudaMalloc/cudaFrees six buffers per batch.  Here every
buffer is allocated once, sized for the GPU's HBM, and the whole frame is
enqueued on one stream without a host synchronisation:

    trace (count) -> scan -> trace (write packed CSR) -> sampler+encode+MLP+glue (one kernel)
    -> volume render        [fused=True: per-segment compositing in the MLP epilogue + per-ray combine]

All arithmetic happens in librtxn.so; this module only owns buffers and calls.
"""
import torch

from . import api


class RenderPipeline:
    def __init__(self, network, grid_res, width, height, focal_length, aspect_ratio=None, occupancy=None,
                 max_rays=None, max_segments=None, trace_mode=api.TRACE_DDA, vr_mode=api.VR_COMPAT,
                 device="cuda", window=(0, 0), fused=False, step_scale=1.0, sub_rays=None, compact=None, on_overflow="raise"):
        self.net = network
        self.R = grid_res
        self.W, self.H = width, height
        self.focal = float(focal_length)
        self.aspect = float(width) / float(height) if aspect_ratio is None else float(aspect_ratio)
        self.trace_mode, self.vr_mode = trace_mode, vr_mode
        self.window = window   # (chunk, stride) ray interleave of this shard, see rtxn_trace_params
        # sub_rays = Q in {2,4,8,16}: Q lanes walk consecutive pieces of each ray (rtxn_trace_params.sub_rays): same
        # segments bit for bit, a shorter critical path -- worth it when the launch is small (a shard, a training batch)
        n_rays = width * height if max_rays is None else max_rays
        self.sub_rays = (api.auto_sub_rays(n_rays) if sub_rays is None else int(sub_rays)) if trace_mode == api.TRACE_DDA else 0
        # fused: the compositor's per-segment half runs in the MLP epilogue (16 B/segment leave the kernel);
        # unfused (default): per-sample radiance + t_vals (20 B/sample) and the reference-shaped launch_volrender_cuda.
        # Measured on MI355X (same run, 800x800 bench frame): fused 22.9 ms/frame vs unfused 21.7 -- the frame is
        # MFMA/issue-bound, not HBM-bound, and the scan in the epilogue costs the MLP kernel 6.5 % while the
        # per-sample round trip it removes is only 2 % of the frame.  Fusion stays available for HBM-capacity reasons.
        self.fused, self.step_scale = fused, step_scale
        # compact (default wherever it applies: unfused RTXN_VR_COMPAT): the MLP kernel hands the compositor the network's
        # own half outputs (8 B/sample) and no t_vals (REGULAR sampling makes them (i+1)/32) -- bit-identical pixels at
        # 40 % of the intermediate's bytes (0.8 instead of 2.0 GB per bench frame)
        self.compact = (not fused and vr_mode == api.VR_COMPAT) if compact is None else bool(compact)
        if self.compact and (fused or vr_mode != api.VR_COMPAT):
            raise ValueError("compact needs the unfused RTXN_VR_COMPAT pipeline")
        self.dev = torch.device(device)
        # A pose that needs more segments than the buffers hold is cut off ON THE DEVICE (never out of bounds).  So that
        # this cannot pass unnoticed outside the calibrated pose set, every frame copies its segment count to pinned host
        # memory (4 bytes, async, no synchronisation) and the NEXT call on that slot -- or drain_async() / finish() --
        # looks at it: on_overflow = "raise" (default): RuntimeError naming the frame; "grow": re-allocate the segment
        # buffers at 1.5x the need (a pipeline stall, once) and go on -- the cut frame was delivered truncated either way
        # and is counted in overflow_frames; "ignore": count only.
        if on_overflow not in ("raise", "grow", "ignore"):
            raise ValueError("on_overflow must be 'raise', 'grow' or 'ignore'")
        self.on_overflow = on_overflow
        self.overflow_frames = 0
        self.occ = occupancy
        self.coarse = self.bricks = self.super_mip = None
        if occupancy is not None and trace_mode == api.TRACE_DDA and grid_res % 4 == 0:
            self.coarse = api.build_occupancy_mip(occupancy, grid_res)
            self.bricks = api.build_occupancy_bricks(occupancy, grid_res)
            if grid_res % 16 == 0:
                self.super_mip = api.build_occupancy_mip(self.coarse, grid_res // 4)
        n = width * height if max_rays is None else max_rays
        self.max_rays = n
        # default capacity: every ray crossing a full grid diagonal's worth of occupied cells is far too
        # pessimistic; callers size it from a counting pass (see calibrate())
        self.max_segments = int(max_segments) if max_segments is not None else 16 * n
        ws = api._lib.lib().rtxn_scan_workspace_bytes(n)
        self.scan_ws = torch.empty((ws + 3) // 4, dtype=torch.int32, device=self.dev)
        self.pixels = torch.empty((n, 3), device=self.dev)
        # per-frame buffers live in "slots": slot 0 is the one render() uses (and what the attributes self.start,
        # self.num_hits, ... alias); render_async() alternates between two so that consecutive frames overlap
        self._slots = [self._alloc_slot()]
        self._bind_slot0()
        self._async = None

    _SLOT_FIELDS = ("look_at", "view_dirs", "num_hits", "num_hits_c", "indices", "total", "start", "end", "seg_view",
                    "seg_first", "seg_out", "radiance", "t_vals")

    def _alloc_slot(self):
        from types import SimpleNamespace
        d, n = self.dev, self.max_rays
        g = SimpleNamespace()
        g.look_at = torch.zeros(16, device=d)
        g.view_dirs = torch.empty((n, 2), device=d)
        g.num_hits = torch.empty(n, dtype=torch.int32, device=d)
        g.num_hits_c = torch.empty(n, dtype=torch.int32, device=d)
        g.indices = torch.empty(n, dtype=torch.int32, device=d)
        g.total = torch.zeros(1, dtype=torch.int32, device=d)
        g.total_host = torch.zeros(1, dtype=torch.int32).pin_memory()   # last frame's segment count, copied asynchronously
        g.total_ev = torch.cuda.Event()
        g.total_pending = False
        g.sub_hits = torch.zeros(n * self.sub_rays, dtype=torch.int32, device=d) if self.sub_rays > 1 else None
        self._alloc_slot_segments(g)
        return g

    def _alloc_slot_segments(self, g):
        d, m = self.dev, self.max_segments
        g.start = torch.empty((m, 3), device=d)
        g.end = torch.empty((m, 3), device=d)
        g.seg_view = torch.empty((m, 2), device=d)
        if self.fused:
            g.seg_first = torch.empty(m, dtype=torch.uint8, device=d)
            g.seg_out = torch.empty((m, 4), device=d)
            g.radiance = g.t_vals = None
        elif self.compact:
            g.seg_first = g.seg_out = g.t_vals = None
            g.radiance = torch.empty((m * api.NUM_SAMPLES_PER_SEGMENT, 4), dtype=torch.float16, device=d)
        else:
            g.seg_first = g.seg_out = None
            g.radiance = torch.empty((m * api.NUM_SAMPLES_PER_SEGMENT, 4), device=d)
            g.t_vals = torch.empty(m * api.NUM_SAMPLES_PER_SEGMENT, device=d)

    def _bind_slot0(self):
        for f in self._SLOT_FIELDS:
            setattr(self, f, getattr(self._slots[0], f))

    def _alloc_segments(self):
        """(Re)allocate the segment-sized buffers of every slot for the current max_segments (calibrate()); the pose and
        the per-ray buffers stay."""
        for g in self._slots:
            self._alloc_slot_segments(g)
        self._bind_slot0()

    def set_pose(self, look_at):
        """look_at: 16 floats (host or device), row-major camera-to-world (params.h:17)."""
        self.look_at.copy_(torch.as_tensor(look_at, dtype=torch.float32).reshape(16), non_blocking=True)

    def _trace(self, ray_begin, ray_count, write, slot=None):
        g = self._slots[0] if slot is None else slot
        kw = dict(grid_res=self.R, ray_begin=ray_begin, ray_count=ray_count, occupancy=self.occ,
                  occupancy_coarse=self.coarse, occupancy_bricks=self.bricks, occupancy_super=self.super_mip, mode=self.trace_mode,
                  viewing_direction=g.view_dirs,
                  num_hits=g.num_hits, window_chunk=self.window[0], window_stride=self.window[1],
                  sub_rays=self.sub_rays, sub_hits=g.sub_hits)
        if write:
            kw.update(indices=g.indices, start_points=g.start, end_points=g.end, seg_view=g.seg_view,
                      seg_first=g.seg_first, num_stored=g.num_hits_c, segment_capacity=self.max_segments)
        api.trace_grid(g.look_at, self.focal, self.aspect, self.W, self.H, **kw)

    def _geometry(self, g, ray_begin, n):
        """count -> scan -> write of one frame into slot g (current stream)."""
        self._trace(ray_begin, n, write=False, slot=g)
        api.scan_hits(g.num_hits[:n], g.indices[:n], g.total, self.scan_ws)
        self._trace(ray_begin, n, write=True, slot=g)
        g.total_host.copy_(g.total, non_blocking=True)    # 4 bytes to pinned memory: the overflow check of the next call
        if not torch.cuda.is_current_stream_capturing():
            g.total_ev.record()
            g.total_pending = True

    def _check_overflow(self, g, wait=False):
        """Look at the segment count slot g's LAST frame reported (no synchronisation unless `wait`: the caller has then
        synchronised the device, which also covers frames replayed from a captured hipGraph)."""
        if torch.cuda.is_current_stream_capturing():
            return
        if not wait and not (g.total_pending and g.total_ev.query()):
            return
        g.total_pending = False
        need = int(g.total_host[0])
        if need <= self.max_segments:
            return
        self.overflow_frames += 1
        if self.on_overflow == "raise":
            raise RuntimeError(f"RenderPipeline: a frame needed {need} segments but the buffers hold {self.max_segments}: its rays "
                               f"were truncated.  calibrate() with that pose, raise max_segments, or use on_overflow='grow'")
        if self.on_overflow == "grow":
            torch.cuda.synchronize()                      # frames in flight still use the old buffers
            self.max_segments = int(need * 1.5) + 1024
            self._alloc_segments()

    def _shade(self, g):
        """sampler + encode + MLP (+ per-segment compositing when fused) over slot g's packed segments."""
        if self.fused:
            self.net.forward_segments_composite(g.start, g.end, g.seg_view, g.seg_first, g.total,
                                                self.max_segments, g.seg_out, self.vr_mode, self.step_scale)
        elif self.compact:
            self.net.forward_segments_compact(g.start, g.end, g.seg_view, g.total, self.max_segments, g.radiance)
        else:
            self.net.forward_segments(g.start, g.end, g.seg_view, g.total, self.max_segments, g.radiance, g.t_vals)

    def _composite(self, g, n, pixels):
        # rays whose segments would overflow the capacity are truncated on the device (never out of bounds):
        # the write pass reports how many segments it actually stored per ray (num_hits_c)
        if self.fused:
            api.composite_segments(g.seg_out, g.num_hits_c[:n], g.indices[:n], n, pixels)
        elif self.compact:
            api.volrender_compact(g.radiance, g.num_hits_c[:n], g.indices[:n], n, api.NUM_SAMPLES_PER_SEGMENT, pixels)
        else:
            api.launch_volrender_cuda(None, g.radiance, g.num_hits_c[:n], g.indices[:n], g.t_vals, n,
                                      api.NUM_SAMPLES_PER_SEGMENT, pixels, mode=self.vr_mode)

    def count_segments(self, ray_begin=0, ray_count=None):
        """Counting pass + scan; returns the number of segments (host int, synchronises)."""
        n = self.max_rays if ray_count is None else ray_count
        self._trace(ray_begin, n, write=False)
        api.scan_hits(self.num_hits[:n], self.indices[:n], self.total, self.scan_ws)
        return int(self.total.item())

    def calibrate(self, poses, ray_begin=0, ray_count=None, margin=1.10):
        """Size the segment buffers for a set of poses (outside any timed region)."""
        worst = 0
        for p in poses:
            self.set_pose(p)
            worst = max(worst, self.count_segments(ray_begin, ray_count))
        need = int(worst * margin) + 1024
        if need > self.max_segments or need < self.max_segments // 2:
            self.max_segments = need
            self._alloc_segments()
        return worst

    def render(self, ray_begin=0, ray_count=None, out=None):
        """Enqueue one frame (or one shard of it) on the current stream; returns the
        pixel buffer view float[ray_count, 3] (or `out`).  No host synchronisation."""
        n = self.max_rays if ray_count is None else ray_count
        pixels = self.pixels[:n] if out is None else out
        g = self._slots[0]
        self._check_overflow(g)
        self._geometry(g, ray_begin, n)
        self._shade(g)
        self._composite(g, n, pixels)
        return pixels

    def finish(self):
        """Wait for every enqueued frame and apply the overflow policy to all of them (raises under "raise")."""
        self.drain_async()
        torch.cuda.synchronize()
        for g in self._slots:
            self._check_overflow(g, wait=True)

    # ------------------------------------------------------------------------------------------ frame pipelining
    # A frame is three dependent stages with very different bounds: traversal (two latency-bound passes + scan, ~0.45 ms
    # whatever the ray count), the MLP kernel (MFMA-bound, 96 % of the time) and the compositor (HBM-bound).  render_async
    # puts them on three HIP streams and rotates through three buffer slots, so that while the MLP kernel of frame i runs,
    # frame i+1 is traversed and frame i-1 composited: the compositor (33 VGPRs, no LDS) and the scan fit beside a resident
    # MLP block and run underneath it; the traversal kernel (58 VGPRs, 4 KiB LDS) does so only beside the 214-VGPR MLP
    # variants -- beside the 227-VGPR segment variant it runs in the gap between two MLP kernels, two frames ahead thanks to
    # the third slot (DESIGN.md 5.1 has the measurements).  The host never synchronises; it simply runs ahead.  This
    # matters most when the frame is sharded over N GPUs (the fixed traversal latency is 16 % of a rank's frame at N = 8).
    def _async_state(self):
        if self._async is None:
            import os
            from types import SimpleNamespace
            n_slots = int(os.environ.get("RTXN_ASYNC_SLOTS", "3"))
            while len(self._slots) < n_slots:
                self._slots.append(self._alloc_slot())
            a = SimpleNamespace()
            a.n = n_slots
            a.geo, a.comp = torch.cuda.Stream(), torch.cuda.Stream()
            a.ev_geo = [torch.cuda.Event() for _ in range(n_slots)]
            a.ev_mlp = [torch.cuda.Event() for _ in range(n_slots)]
            a.ev_comp = [torch.cuda.Event() for _ in range(n_slots)]
            a.used = [False] * n_slots
            a.frame = 0
            self._async = a
        return self._async

    def render_async(self, look_at, ray_begin=0, ray_count=None, out=None):
        """Enqueue one frame, software-pipelined against its neighbours.  look_at: 16 floats on the DEVICE (copied into the
        slot on the traversal stream).  Returns (pixels, done_event, comp_stream): `pixels` is complete once `done_event`
        has fired; follow-up work on it (a gather, a copy) is best enqueued on `comp_stream`."""
        a = self._async_state()
        n = self.max_rays if ray_count is None else ray_count
        pixels = self.pixels[:n] if out is None else out
        b = a.frame % a.n
        a.frame += 1
        g = self._slots[b]
        self._check_overflow(g)                           # the frame that used this slot three frames ago
        main = torch.cuda.current_stream()
        with torch.cuda.stream(a.geo):
            if a.used[b]:
                a.geo.wait_event(a.ev_comp[b])        # frame i-2 (MLP and compositor) is done with this slot
            else:
                a.geo.wait_stream(main)               # first use: whatever set the pipeline up
            g.look_at.copy_(look_at, non_blocking=True)
            self._geometry(g, ray_begin, n)
            a.ev_geo[b].record(a.geo)
        main.wait_event(a.ev_geo[b])
        if a.used[b]:
            main.wait_event(a.ev_comp[b])             # radiance / seg_out of this slot were last read by compositor i-2
        self._shade(g)
        a.ev_mlp[b].record(main)
        with torch.cuda.stream(a.comp):
            a.comp.wait_event(a.ev_mlp[b])
            self._composite(g, n, pixels)
            a.ev_comp[b].record(a.comp)
        a.used[b] = True
        return pixels, a.ev_comp[b], a.comp

    def drain_async(self):
        """Make the current stream wait for every frame enqueued with render_async."""
        if self._async is not None:
            main = torch.cuda.current_stream()
            main.wait_stream(self._async.geo)
            main.wait_stream(self._async.comp)

    def capture(self, ray_begin=0, ray_count=None, out=None):
        """Capture one frame into a hipGraph (torch.cuda.CUDAGraph over the C-ABI launches: none of them
        allocates, synchronises or touches the host).  Returns (graph, pixels): update the pose with set_pose()
        and call graph.replay().  The frame must have been rendered once eagerly (module/attribute setup)."""
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            pixels = self.render(ray_begin, ray_count, out)
        return g, pixels

    def overflowed(self):
        """True if a frame was ever truncated (synchronises: also looks at the frames still in flight)."""
        return self.overflow_frames > 0 or any(int(g.total.item()) > self.max_segments for g in self._slots)
