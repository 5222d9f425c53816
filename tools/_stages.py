"""Stage-by-stage driver of the render path on its OWN buffers, for the measurement tools in this directory (stage_bench.py,
trace_bench.py, co_run.py): the per-stage C entry points in the order rtxn_render_frame issues them, each callable alone so
that a tool can time or overlap it.  The product path is rtxn_render_frame (rtx_nerf_amd/render.py); this is not used by it."""
import torch

from rtx_nerf_amd import api


class Stages:
    def __init__(self, net, grid_res, width, height, focal, occupancy, max_rays=None, window=(0, 0), sub_rays=None, compact=True):
        self.net, self.R, self.W, self.H, self.focal = net, grid_res, width, height, float(focal)
        self.aspect = width / height
        self.occ = occupancy
        self.coarse = api.build_occupancy_mip(occupancy, grid_res) if grid_res % 4 == 0 else None
        self.bricks = api.build_occupancy_bricks(occupancy, grid_res) if grid_res % 4 == 0 else None
        self.super_mip = api.build_occupancy_mip(self.coarse, grid_res // 4) if grid_res % 16 == 0 else None
        n = width * height if max_rays is None else max_rays
        self.n, self.window, self.compact = n, window, compact
        self.sub_rays = api.auto_sub_rays(n) if sub_rays is None else int(sub_rays)
        d = "cuda"
        self.look_at = torch.zeros(16, device=d)
        self.view_dirs = torch.empty((n, 2), device=d)
        self.num_hits = torch.empty(n, dtype=torch.int32, device=d)
        self.num_hits_c = torch.empty(n, dtype=torch.int32, device=d)
        self.indices = torch.empty(n, dtype=torch.int32, device=d)
        self.total = torch.zeros(1, dtype=torch.int32, device=d)
        self.sub_hits = torch.zeros(n * max(self.sub_rays, 1), dtype=torch.int32, device=d)
        self.scan_ws = torch.empty((api._lib.lib().rtxn_scan_workspace_bytes(n) + 3) // 4, dtype=torch.int32, device=d)
        self.pixels = torch.empty((n, 3), device=d)
        self.max_segments = 0

    def set_pose(self, la):
        self.look_at.copy_(torch.as_tensor(la, dtype=torch.float32).reshape(16))

    def trace(self, ray_begin, n, write):
        kw = dict(grid_res=self.R, ray_begin=ray_begin, ray_count=n, occupancy=self.occ, occupancy_coarse=self.coarse,
                  occupancy_bricks=self.bricks, occupancy_super=self.super_mip, mode=api.TRACE_DDA, viewing_direction=self.view_dirs,
                  num_hits=self.num_hits, window_chunk=self.window[0], window_stride=self.window[1], sub_rays=self.sub_rays,
                  sub_hits=self.sub_hits)
        if write:
            kw.update(indices=self.indices, start_points=self.start, end_points=self.end, seg_view=self.seg_view,
                      num_stored=self.num_hits_c, segment_capacity=self.max_segments)
        api.trace_grid(self.look_at, self.focal, self.aspect, self.W, self.H, **kw)

    def scan(self, n):
        api.scan_hits(self.num_hits[:n], self.indices[:n], self.total, self.scan_ws)

    def size_for(self, la, ray_begin=0, n=None, margin=1.1):
        """counting pass for pose la, then (re)allocate the segment buffers; returns the segment count"""
        n = self.n if n is None else n
        self.set_pose(la)
        self.trace(ray_begin, n, False)
        self.scan(n)
        P = int(self.total.item())
        m = self.max_segments = int(P * margin) + 1024
        self.start, self.end = torch.empty((m, 3), device="cuda"), torch.empty((m, 3), device="cuda")
        self.seg_view = torch.empty((m, 2), device="cuda")
        K = api.NUM_SAMPLES_PER_SEGMENT
        self.radiance = torch.empty((m * K, 4), dtype=torch.float16 if self.compact else torch.float32, device="cuda")
        self.t_vals = None if self.compact else torch.empty(m * K, device="cuda")
        return P

    def geometry(self, ray_begin=0, n=None):
        n = self.n if n is None else n
        self.trace(ray_begin, n, False)
        self.scan(n)
        self.trace(ray_begin, n, True)

    def shade(self):
        if self.compact:
            self.net.forward_segments_compact(self.start, self.end, self.seg_view, self.total, self.max_segments, self.radiance)
        else:
            self.net.forward_segments(self.start, self.end, self.seg_view, self.total, self.max_segments, self.radiance, self.t_vals)

    def composite(self, n=None):
        n = self.n if n is None else n
        if self.compact:
            api.volrender_compact(self.radiance, self.num_hits_c[:n], self.indices[:n], n, api.NUM_SAMPLES_PER_SEGMENT, self.pixels[:n])
        else:
            api.launch_volrender_cuda(None, self.radiance, self.num_hits_c[:n], self.indices[:n], self.t_vals, n,
                                      api.NUM_SAMPLES_PER_SEGMENT, self.pixels[:n])
