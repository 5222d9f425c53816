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
                 device="cuda", window=(0, 0), fused=False, step_scale=1.0):
        self.net = network
        self.R = grid_res
        self.W, self.H = width, height
        self.focal = float(focal_length)
        self.aspect = float(width) / float(height) if aspect_ratio is None else float(aspect_ratio)
        self.trace_mode, self.vr_mode = trace_mode, vr_mode
        self.window = window   # (chunk, stride) ray interleave of this shard, see rtxn_trace_params
        # fused: the compositor's per-segment half runs in the MLP epilogue (16 B/segment leave the kernel);
        # unfused (default): per-sample radiance + t_vals (20 B/sample) and the reference-shaped launch_volrender_cuda.
        # Measured on MI355X (same run, 800x800 bench frame): fused 22.9 ms/frame vs unfused 21.7 -- the frame is
        # MFMA/issue-bound, not HBM-bound, and the scan in the epilogue costs the MLP kernel 6.5 % while the
        # per-sample round trip it removes is only 2 % of the frame.  Fusion stays available for HBM-capacity reasons.
        self.fused, self.step_scale = fused, step_scale
        self.dev = torch.device(device)
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
        d = self.dev
        self.look_at = torch.zeros(16, device=d)
        self.view_dirs = torch.empty((n, 2), device=d)
        self.num_hits = torch.empty(n, dtype=torch.int32, device=d)
        self.num_hits_c = torch.empty(n, dtype=torch.int32, device=d)
        self.indices = torch.empty(n, dtype=torch.int32, device=d)
        self.total = torch.zeros(1, dtype=torch.int32, device=d)
        ws = api._lib.lib().rtxn_scan_workspace_bytes(n)
        self.scan_ws = torch.empty((ws + 3) // 4, dtype=torch.int32, device=d)
        self.pixels = torch.empty((n, 3), device=d)
        self._alloc_segments()

    def _alloc_segments(self):
        d, m = self.dev, self.max_segments
        self.start = torch.empty((m, 3), device=d)
        self.end = torch.empty((m, 3), device=d)
        self.seg_view = torch.empty((m, 2), device=d)
        if self.fused:
            self.seg_first = torch.empty(m, dtype=torch.uint8, device=d)
            self.seg_out = torch.empty((m, 4), device=d)
            self.radiance = self.t_vals = None
        else:
            self.seg_first = self.seg_out = None
            self.radiance = torch.empty((m * api.NUM_SAMPLES_PER_SEGMENT, 4), device=d)
            self.t_vals = torch.empty(m * api.NUM_SAMPLES_PER_SEGMENT, device=d)

    def set_pose(self, look_at):
        """look_at: 16 floats (host or device), row-major camera-to-world (params.h:17)."""
        self.look_at.copy_(torch.as_tensor(look_at, dtype=torch.float32).reshape(16), non_blocking=True)

    def _trace(self, ray_begin, ray_count, write):
        kw = dict(grid_res=self.R, ray_begin=ray_begin, ray_count=ray_count, occupancy=self.occ,
                  occupancy_coarse=self.coarse, occupancy_bricks=self.bricks, occupancy_super=self.super_mip, mode=self.trace_mode,
                  viewing_direction=self.view_dirs,
                  num_hits=self.num_hits, window_chunk=self.window[0], window_stride=self.window[1])
        if write:
            kw.update(indices=self.indices, start_points=self.start, end_points=self.end, seg_view=self.seg_view,
                      seg_first=self.seg_first, num_stored=self.num_hits_c, segment_capacity=self.max_segments)
        api.trace_grid(self.look_at, self.focal, self.aspect, self.W, self.H, **kw)

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
        nh, idx = self.num_hits[:n], self.indices[:n]
        self._trace(ray_begin, n, write=False)
        api.scan_hits(nh, idx, self.total, self.scan_ws)
        self._trace(ray_begin, n, write=True)
        # rays whose segments would overflow the capacity are truncated on the device (never out of bounds):
        # the write pass reports how many segments it actually stored per ray
        nhc = self.num_hits_c[:n]
        if self.fused:
            self.net.forward_segments_composite(self.start, self.end, self.seg_view, self.seg_first, self.total,
                                                self.max_segments, self.seg_out, self.vr_mode, self.step_scale)
            api.composite_segments(self.seg_out, nhc, idx, n, pixels)
        else:
            self.net.forward_segments(self.start, self.end, self.seg_view, self.total,
                                      self.max_segments, self.radiance, self.t_vals)
            api.launch_volrender_cuda(None, self.radiance, nhc, idx, self.t_vals, n, api.NUM_SAMPLES_PER_SEGMENT,
                                      pixels, mode=self.vr_mode)
        return pixels

    def capture(self, ray_begin=0, ray_count=None, out=None):
        """Capture one frame into a hipGraph (torch.cuda.CUDAGraph over the C-ABI launches: none of them
        allocates, synchronises or touches the host).  Returns (graph, pixels): update the pose with set_pose()
        and call graph.replay().  The frame must have been rendered once eagerly (module/attribute setup)."""
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            pixels = self.render(ray_begin, ray_count, out)
        return g, pixels

    def overflowed(self):
        return int(self.total.item()) > self.max_segments
