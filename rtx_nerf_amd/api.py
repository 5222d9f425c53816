"""Host-side mirror of the reference's operator interface over the C ABI.

Function names, argument order and argument meaning follow the reference
headers (sampler/sampler.h:19-30, vol_render/vol_render.h:5-25,
rtx/include/params.h:14-42, and the tiny-cuda-nn calls of main.cu:325-349,721)
so the parity tests read like the reference's call sites.  Arguments are torch
CUDA tensors (plumbing: device memory + streams); all work is enqueued on
torch's current stream.  Nothing here computes on the CPU.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import MlpConfig, TraceParams, check

NUM_SAMPLES_PER_SEGMENT = 32           # sampler/sampler.h:4
SAMPLING_REGULAR = 0                   # sampler/sampler.h:5-9
SAMPLING_STRATIFIED_JITTERING = 1
SAMPLING_UNIFORM = 2
SAMPLING_MIDPOINT_WORLD = 3               # this build: midpoints + world step lengths (for VR_NERF)
TRACE_COMPAT, TRACE_DDA = 0, 1
VR_COMPAT, VR_NERF = 0, 1
ACT_NONE, ACT_SIGMOID = 0, 1


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t, dtype=None, name="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.RtxnError(f"{name} must be a CUDA tensor (librtxn has no CPU path)")
    if not t.is_contiguous():
        raise _lib.RtxnError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise _lib.RtxnError(f"{name} must be {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


# --------------------------------------------------------------------------- traversal
def trace_grid(look_at=None, focal_length=1.0, aspect_ratio=1.0, width=0, height=0, *, grid_res,
               rays_o=None, rays_d=None, ray_begin=0, ray_count=None, occupancy=None,
               occupancy_coarse=None, occupancy_bricks=None, occupancy_super=None, mode=TRACE_COMPAT, ray_origins=None, viewing_direction=None,
               num_hits=None, intersection_arr_size=0, indices=None, start_points=None,
               end_points=None, t_start=None, t_end=None, seg_ray=None, seg_view=None, seg_first=None, num_stored=None, segment_capacity=0,
               window_chunk=0, window_stride=0, sub_rays=0, sub_hits=None):
    """optixLaunch(pipeline_ray_march, ..., width, height, 1) with Params (main.cu:481-508)."""
    p = trace_params(**{k: v for k, v in locals().items()})
    check(_lib.lib().rtxn_trace_grid(C.byref(p), _stream()), "rtxn_trace_grid")


def trace_params(look_at=None, focal_length=1.0, aspect_ratio=1.0, width=0, height=0, *, grid_res,
                 rays_o=None, rays_d=None, ray_begin=0, ray_count=None, occupancy=None,
                 occupancy_coarse=None, occupancy_bricks=None, occupancy_super=None, mode=TRACE_COMPAT, ray_origins=None, viewing_direction=None,
                 num_hits=None, intersection_arr_size=0, indices=None, start_points=None,
                 end_points=None, t_start=None, t_end=None, seg_ray=None, seg_view=None, seg_first=None, num_stored=None, segment_capacity=0,
                 window_chunk=0, window_stride=0, sub_rays=0, sub_hits=None):
    """struct rtxn_trace_params over the given tensors (which the caller keeps alive)."""
    p = TraceParams()
    p.look_at = _ptr(look_at, torch.float32, "look_at")
    p.focal_length, p.aspect_ratio = focal_length, aspect_ratio
    p.width, p.height = width, height
    p.rays_o = _ptr(rays_o, torch.float32, "rays_o")
    p.rays_d = _ptr(rays_d, torch.float32, "rays_d")
    if look_at is None and rays_o is not None and width == 0:
        p.width, p.height = rays_o.shape[0], 1
    p.ray_begin = ray_begin
    p.ray_count = (p.width * p.height - ray_begin) if ray_count is None else ray_count
    p.grid_res = grid_res
    p.occupancy = _ptr(occupancy, torch.int32, "occupancy")
    p.occupancy_coarse = _ptr(occupancy_coarse, torch.int32, "occupancy_coarse")
    p.occupancy_bricks = _ptr(occupancy_bricks, torch.int64, "occupancy_bricks")
    p.occupancy_super = _ptr(occupancy_super, torch.int32, "occupancy_super")
    p.mode = mode
    p.ray_origins = _ptr(ray_origins, torch.float32, "ray_origins")
    p.viewing_direction = _ptr(viewing_direction, torch.float32, "viewing_direction")
    p.num_hits = _ptr(num_hits, torch.int32, "num_hits")
    p.intersection_arr_size = intersection_arr_size
    p.indices = _ptr(indices, torch.int32, "indices")
    p.start_points = _ptr(start_points, torch.float32, "start_points")
    p.end_points = _ptr(end_points, torch.float32, "end_points")
    p.t_start = _ptr(t_start, torch.float32, "t_start")
    p.t_end = _ptr(t_end, torch.float32, "t_end")
    p.seg_ray = _ptr(seg_ray, torch.int32, "seg_ray")
    p.seg_view = _ptr(seg_view, torch.float32, "seg_view")
    p.seg_first = _ptr(seg_first, torch.uint8, "seg_first")
    p.num_stored = _ptr(num_stored, torch.int32, "num_stored")
    p.segment_capacity = segment_capacity
    p.window_chunk, p.window_stride = window_chunk, window_stride
    p.sub_rays = sub_rays
    p.sub_hits = _ptr(sub_hits, torch.int32, "sub_hits")
    return p


def auto_sub_rays(n_rays):
    """Lanes per ray (rtxn_trace_params.sub_rays) for a launch of n_rays: measured on MI355X (tools/trace_bench.py), the
    smaller the launch the more the longest ray's walk dominates it -- 640 k rays: 2 (0.20 -> 0.15 ms), 80 k: 8 (0.18 ->
    0.07 ms), 16 k rays: 16, a 4096-ray training batch: 32."""
    import os
    if os.environ.get("RTXN_SUB_RAYS"):          # experiments (tools/trace_bench.py, tools/probe)
        return int(os.environ["RTXN_SUB_RAYS"])
    if n_rays >= 300_000:
        return 2
    if n_rays >= 120_000:
        return 4
    if n_rays >= 30_000:
        return 8
    if n_rays >= 12_000:
        return 16
    return 32          # a 4096-ray training batch: count + scan + write 81 -> 67 us; 64 lanes per ray: no further gain


def build_occupancy_mip(occupancy, grid_res):
    rc = grid_res // 4
    words = (rc ** 3 + 31) // 32
    coarse = torch.empty(words, dtype=torch.int32, device=occupancy.device)
    check(_lib.lib().rtxn_build_occupancy_mip(_ptr(occupancy, torch.int32), grid_res, _ptr(coarse), _stream()),
          "rtxn_build_occupancy_mip")
    return coarse


def build_occupancy_bricks(occupancy, grid_res):
    rc = grid_res // 4
    bricks = torch.empty(rc ** 3, dtype=torch.int64, device=occupancy.device)
    check(_lib.lib().rtxn_build_occupancy_bricks(_ptr(occupancy, torch.int32), grid_res, _ptr(bricks), _stream()),
          "rtxn_build_occupancy_bricks")
    return bricks


def occupancy_from_density(density, threshold, grid_res):
    words = (grid_res ** 3 + 31) // 32
    occ = torch.empty(words, dtype=torch.int32, device=density.device)
    check(_lib.lib().rtxn_occupancy_from_density(_ptr(density, torch.float32, "density"), threshold, grid_res, _ptr(occ),
                                                 _stream()), "rtxn_occupancy_from_density")
    return occ


# --------------------------------------------------------------------------- CSR compaction
def scan_hits(num_hits, indices=None, total=None, workspace=None):
    """thrust::reduce + thrust::exclusive_scan (main.cu:631-637); total stays on the device."""
    n = num_hits.numel()
    dev = num_hits.device
    if indices is None:
        indices = torch.empty(n, dtype=torch.int32, device=dev)
    if total is None:
        total = torch.empty(1, dtype=torch.int32, device=dev)
    need = _lib.lib().rtxn_scan_workspace_bytes(n)
    if workspace is None:
        workspace = torch.empty((need + 3) // 4, dtype=torch.int32, device=dev)
    check(_lib.lib().rtxn_scan_hits(_ptr(num_hits, torch.int32, "num_hits"), _ptr(indices, torch.int32),
                                    _ptr(total, torch.int32), n, _ptr(workspace), workspace.numel() * 4, _stream()),
          "rtxn_scan_hits")
    return indices, total


# --------------------------------------------------------------------------- sampler
def launchSampler(d_start_points, d_end_points, d_view_dirs, d_t_vals, d_sampled_points, batch_size,
                  grid_res, d_num_hits, d_indices, sample_type=SAMPLING_REGULAR):
    """sampler/sampler.h:19-30 (the stream argument is torch's current stream)."""
    check(_lib.lib().rtxn_sample(_ptr(d_start_points, torch.float32, "d_start_points"),
                                 _ptr(d_end_points, torch.float32, "d_end_points"),
                                 _ptr(d_view_dirs, torch.float32, "d_view_dirs"),
                                 _ptr(d_t_vals, torch.float32, "d_t_vals"),
                                 _ptr(d_sampled_points, torch.float32, "d_sampled_points"),
                                 batch_size, grid_res, _ptr(d_num_hits, torch.int32, "d_num_hits"),
                                 _ptr(d_indices, torch.int32, "d_indices"), sample_type, _stream()),
          "rtxn_sample")


# --------------------------------------------------------------------------- volume rendering
def launch_volrender_cuda(network_inputs, network_outputs, num_hits, indices, ray_hit, batch_size,
                          num_samples_per_hit, pixels, mode=VR_COMPAT):
    """vol_render/vol_render.h:5-13."""
    check(_lib.lib().rtxn_volrender_fwd(_ptr(network_inputs), _ptr(network_outputs, torch.float32, "network_outputs"),
                                        _ptr(num_hits, torch.int32, "num_hits"), _ptr(indices, torch.int32, "indices"),
                                        _ptr(ray_hit, torch.float32, "ray_hit"), batch_size, num_samples_per_hit,
                                        _ptr(pixels, torch.float32, "pixels"), mode, _stream()),
          "rtxn_volrender_fwd")


def launch_volrender_backward_cuda(loss_values, loss_gradients, sampled_points_radiance, t_hit, num_hits,
                                   indices, batch_size, num_samples_per_hit, radiance_gradients, mode=VR_COMPAT):
    """vol_render/vol_render.h:15-25."""
    check(_lib.lib().rtxn_volrender_bwd(_ptr(loss_values), _ptr(loss_gradients, torch.float16, "loss_gradients"),
                                        _ptr(sampled_points_radiance, torch.float32, "sampled_points_radiance"),
                                        _ptr(t_hit, torch.float32, "t_hit"), _ptr(num_hits, torch.int32, "num_hits"),
                                        _ptr(indices, torch.int32, "indices"), batch_size, num_samples_per_hit,
                                        _ptr(radiance_gradients, torch.float16, "radiance_gradients"), mode, _stream()),
          "rtxn_volrender_bwd")


def volrender_l2_train(network_outputs, ray_hit, num_hits, indices, batch_size, num_samples_per_hit, target, loss_scale, pixels,
                       loss_gradients, loss_sum, radiance_gradients):
    """launch_volrender_cuda + L2 loss->evaluate + launch_volrender_backward_cuda (main.cu:737-767) in one launch (VR_NERF)."""
    check(_lib.lib().rtxn_volrender_l2_train(_ptr(network_outputs, torch.float32, "network_outputs"), _ptr(ray_hit, torch.float32, "ray_hit"),
                                             _ptr(num_hits, torch.int32, "num_hits"), _ptr(indices, torch.int32, "indices"), batch_size,
                                             num_samples_per_hit, _ptr(target, torch.float32, "target"), loss_scale,
                                             _ptr(pixels, torch.float32, "pixels"), _ptr(loss_gradients, torch.float16, "loss_gradients"),
                                             _ptr(loss_sum, torch.float32, "loss_sum"), _ptr(radiance_gradients, torch.float16, "radiance_gradients"),
                                             _stream()), "rtxn_volrender_l2_train")


# --------------------------------------------------------------------------- MLP
class Network:
    """tcnn::create_from_config(n_input_dims=5, n_output_dims=4, config) (main.cu:35-69,325)."""

    def __init__(self, n_neurons=128, n_hidden_layers=8, n_pos_freqs=10, n_dir_freqs=12, n_pos_dims=3,
                 n_dir_dims=2, n_output_dims=4, output_activation=ACT_SIGMOID, n_encoded_features=0):
        """n_encoded_features > 0: the model takes pre-encoded input of that width (RTXN_ENC_EXTERNAL)."""
        self.cfg = MlpConfig(n_pos_dims, n_pos_freqs, n_dir_dims, n_dir_freqs, n_neurons, n_hidden_layers,
                             n_output_dims, output_activation, 1 if n_encoded_features else 0, n_encoded_features)
        h = C.c_void_p()
        check(_lib.lib().rtxn_mlp_create(C.byref(self.cfg), C.byref(h)), "rtxn_mlp_create")
        self._h = h
        self.params = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and getattr(_lib, "_lib", None) is not None:   # not during interpreter shutdown
            _lib._lib.rtxn_mlp_destroy(h)

    def n_params(self):
        return int(_lib.lib().rtxn_mlp_n_params(self._h))

    def padded_output_width(self):
        return int(_lib.lib().rtxn_mlp_padded_output_width(self._h))

    def set_reserved_cus(self, n_cus):
        """Keep n_cus CUs free of the persistent inference grid (for a collective library's kernels on other streams)."""
        check(_lib.lib().rtxn_mlp_set_reserved_cus(self._h, int(n_cus)), "rtxn_mlp_set_reserved_cus")

    def encoded_width(self):
        return int(_lib.lib().rtxn_mlp_encoded_width(self._h))

    def mfma_shape(self):
        """16: the fused inference kernels (v_mfma_f32_16x16x32_f16); 0: no fused inference kernel (pre-encoded input)."""
        return int(_lib.lib().rtxn_mlp_mfma_shape(self._h))

    def flops_per_sample(self):
        w, p, nh = self.cfg.n_neurons, self.encoded_width(), self.cfg.n_hidden_layers
        return 2 * (p * w + (nh - 1) * w * w + 16 * w)

    def initialize_params(self, seed=1337):
        """network->initialize_params(rng, params_full_precision) (main.cu:344-349): host fp32."""
        out = torch.empty(self.n_params(), dtype=torch.float32)
        check(_lib.lib().rtxn_mlp_initialize_params(self._h, seed, C.c_void_p(out.data_ptr())),
              "rtxn_mlp_initialize_params")
        return out

    def set_params(self, params_fp16):
        """network->set_params(params, params_inference, gradients) (main.cu:342)."""
        if params_fp16.numel() != self.n_params():
            raise _lib.RtxnError(f"params has {params_fp16.numel()} elements, model needs {self.n_params()}")
        self.params = params_fp16
        check(_lib.lib().rtxn_mlp_set_params(self._h, _ptr(params_fp16, torch.float16, "params"), _stream()),
              "rtxn_mlp_set_params")

    def set_params_training(self, params_fp16):
        """Per-step update inside a training loop: re-packs the training kernels' weights only (set_params() again before
        rendering with the fused inference kernels)."""
        if params_fp16.numel() != self.n_params():
            raise _lib.RtxnError(f"params has {params_fp16.numel()} elements, model needs {self.n_params()}")
        self.params = params_fp16
        check(_lib.lib().rtxn_mlp_set_params_training(self._h, _ptr(params_fp16, torch.float16, "params"), _stream()),
              "rtxn_mlp_set_params_training")

    def forward(self, input_batch, output=None):
        """network->forward(stream, input(5xN), &output(16xN)) (main.cu:715-721)."""
        n = input_batch.numel() // 5
        if output is None:
            output = torch.empty((n, 16), dtype=torch.float16, device=input_batch.device)
        check(_lib.lib().rtxn_mlp_forward(self._h, _ptr(input_batch, torch.float32, "input"), _ptr(output, torch.float16),
                                          n, _stream()), "rtxn_mlp_forward")
        return output

    def forward_radiance(self, input_batch, radiance=None):
        """forward + convertHalfToFloat of rows 0..3 (main.cu:721-728)."""
        n = input_batch.numel() // 5
        if radiance is None:
            radiance = torch.empty((n, 4), dtype=torch.float32, device=input_batch.device)
        check(_lib.lib().rtxn_mlp_forward_radiance(self._h, _ptr(input_batch, torch.float32, "input"),
                                                   _ptr(radiance, torch.float32), n, _stream()),
              "rtxn_mlp_forward_radiance")
        return radiance

    def forward_segments(self, start_points, end_points, seg_view, total_segments, max_segments, radiance,
                         t_vals=None):
        """launchSampler(REGULAR) + forward + glue fused over packed segments."""
        check(_lib.lib().rtxn_mlp_forward_segments(
            self._h, _ptr(start_points, torch.float32, "start_points"), _ptr(end_points, torch.float32, "end_points"),
            _ptr(seg_view, torch.float32, "seg_view"), _ptr(total_segments, torch.int32, "total_segments"),
            max_segments, _ptr(radiance, torch.float32, "radiance"), _ptr(t_vals, torch.float32, "t_vals"), _stream()),
            "rtxn_mlp_forward_segments")
        return radiance


def _net_forward_segments_compact(self, start_points, end_points, seg_view, total_segments, max_segments, radiance_half4):
    """As forward_segments, but stores the network's four half outputs per sample (half[N, 4]) and no t_vals."""
    check(_lib.lib().rtxn_mlp_forward_segments_compact(
        self._h, _ptr(start_points, torch.float32, "start_points"), _ptr(end_points, torch.float32, "end_points"),
        _ptr(seg_view, torch.float32, "seg_view"), _ptr(total_segments, torch.int32, "total_segments"),
        max_segments, _ptr(radiance_half4, torch.float16, "radiance_half4"), _stream()), "rtxn_mlp_forward_segments_compact")
    return radiance_half4


Network.forward_segments_compact = _net_forward_segments_compact


def volrender_compact(radiance_half4, num_hits, indices, batch_size, num_samples_per_hit, pixels):
    """RTXN_VR_COMPAT compositing of half[N, 4] radiance with the implicit REGULAR t_vals (i + 1) / K."""
    check(_lib.lib().rtxn_volrender_fwd_compact(_ptr(radiance_half4, torch.float16, "radiance_half4"),
                                                _ptr(num_hits, torch.int32, "num_hits"), _ptr(indices, torch.int32, "indices"),
                                                batch_size, num_samples_per_hit, _ptr(pixels, torch.float32, "pixels"), _stream()),
          "rtxn_volrender_fwd_compact")


def volrender_compact_nerf(radiance_half4, segment_step, num_hits, indices, batch_size, num_samples_per_hit, pixels):
    """RTXN_VR_NERF compositing of half[N, 4] radiance with ONE step length per segment (float[P])."""
    check(_lib.lib().rtxn_volrender_fwd_compact_nerf(_ptr(radiance_half4, torch.float16, "radiance_half4"),
                                                     _ptr(segment_step, torch.float32, "segment_step"),
                                                     _ptr(num_hits, torch.int32, "num_hits"), _ptr(indices, torch.int32, "indices"),
                                                     batch_size, num_samples_per_hit, _ptr(pixels, torch.float32, "pixels"), _stream()),
          "rtxn_volrender_fwd_compact_nerf")


def hashmlp_supported(net, grid):
    """True if the fused hash-encode + MLP inference kernel is built for this model / grid pair."""
    return bool(_lib.lib().rtxn_hashmlp_supported(net._h, grid._h, grid.n_dir_freqs))


def hashmlp_forward_segments(net, grid, table_fp16, start_points, end_points, seg_view, total_segments, max_segments, radiance_half4,
                             sample_type=SAMPLING_REGULAR, t_scale=1.0, segment_step=None):
    """launchSampler + HashGrid/Frequency encoding + network->forward + glue as one kernel over packed segments (the hash-grid
    counterpart of Network.forward_segments_compact): half[N, 4] radiance, optionally the per-segment world step."""
    check(_lib.lib().rtxn_hashmlp_forward_segments(
        net._h, grid._h, grid.n_dir_freqs, _ptr(table_fp16, torch.float16, "table"), _ptr(start_points, torch.float32, "start_points"),
        _ptr(end_points, torch.float32, "end_points"), _ptr(seg_view, torch.float32, "seg_view"),
        _ptr(total_segments, torch.int32, "total_segments"), int(max_segments), int(sample_type), float(t_scale),
        _ptr(radiance_half4, torch.float16, "radiance_half4"), _ptr(segment_step, torch.float32, "segment_step"), _stream()),
        "rtxn_hashmlp_forward_segments")
    return radiance_half4


# --------------------------------------------------------------------------- training path
def padded_samples(n):
    return int(_lib.lib().rtxn_padded_samples(n))


class HashGrid:
    """Multiresolution hash grid for positions (+ Frequency for directions)."""

    def __init__(self, n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.5,
                 n_dir_freqs=4):
        self.cfg = _lib.HashGridConfig(n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale)
        self.n_dir_freqs = n_dir_freqs
        h = C.c_void_p()
        check(_lib.lib().rtxn_hashgrid_create(C.byref(self.cfg), C.byref(h)), "rtxn_hashgrid_create")
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and getattr(_lib, "_lib", None) is not None:
            _lib._lib.rtxn_hashgrid_destroy(h)

    def n_params(self):
        return int(_lib.lib().rtxn_hashgrid_n_params(self._h))

    def encoded_width(self):
        return int(_lib.lib().rtxn_hashgrid_encoded_width(self._h, self.n_dir_freqs))

    def level_offset(self, level):
        """Offset of `level` in table parameters (entries x features); level == n_levels: the total."""
        return int(_lib.lib().rtxn_hashgrid_level_offset(self._h, level))

    def hashed_offset(self):
        """Parameters before the first hashed level: the leading, densely stored levels (a few hundred KB)."""
        for l in range(self.cfg.n_levels):
            if _lib.lib().rtxn_hashgrid_level_is_hashed(self._h, l):
                return self.level_offset(l)
        return self.n_params()

    def encode(self, table_fp16, inputs, encT=None):
        n = inputs.numel() // 5
        if encT is None:
            encT = torch.empty((self.encoded_width(), padded_samples(n)), dtype=torch.float16, device=inputs.device)
        check(_lib.lib().rtxn_hashgrid_encode(self._h, self.n_dir_freqs, _ptr(table_fp16, torch.float16, "table"),
                                              _ptr(inputs, torch.float32, "inputs"), _ptr(encT, torch.float16, "encT"),
                                              n, _stream()), "rtxn_hashgrid_encode")
        return encT

    def encode_segments(self, table_fp16, start_points, end_points, seg_view, n_segments, sample_type, encT, t_vals=None, t_scale=1.0):
        """launchSampler + encode in one pass over the packed segments (the float[S][5] samples are never written)."""
        check(_lib.lib().rtxn_hashgrid_encode_segments(self._h, self.n_dir_freqs, _ptr(table_fp16, torch.float16, "table"),
                                                       _ptr(start_points, torch.float32, "start_points"), _ptr(end_points, torch.float32, "end_points"),
                                                       _ptr(seg_view, torch.float32, "seg_view"), n_segments, sample_type, t_scale,
                                                       _ptr(encT, torch.float16, "encT"), _ptr(t_vals, torch.float32, "t_vals"), _stream()),
              "rtxn_hashgrid_encode_segments")
        return encT

    def backward_segments(self, start_points, end_points, n_segments, sample_type, dencT, dtable, dtable_hashed_half=None, live_ws=None):
        """live_ws (from live_segments): visit only the segments that carry a loss gradient."""
        if live_ws is not None:
            check(_lib.lib().rtxn_hashgrid_backward_segments_live(self._h, _ptr(start_points, torch.float32, "start_points"),
                                                                  _ptr(end_points, torch.float32, "end_points"), n_segments, sample_type,
                                                                  _ptr(dencT, torch.float16, "dencT"), _ptr(live_ws, None, "live_ws"),
                                                                  _ptr(dtable, torch.float32, "dtable"),
                                                                  _ptr(dtable_hashed_half, torch.float16, "dtable_hashed_half"), _stream()),
                  "rtxn_hashgrid_backward_segments_live")
            return dtable
        check(_lib.lib().rtxn_hashgrid_backward_segments(self._h, _ptr(start_points, torch.float32, "start_points"),
                                                         _ptr(end_points, torch.float32, "end_points"), n_segments, sample_type,
                                                         _ptr(dencT, torch.float16, "dencT"), _ptr(dtable, torch.float32, "dtable"),
                                                         _ptr(dtable_hashed_half, torch.float16, "dtable_hashed_half"), _stream()),
              "rtxn_hashgrid_backward_segments")
        return dtable

    def backward_mixed(self, inputs, dencT, dtable, dtable_hashed_half):
        """backward with the hashed levels' gradient accumulated in fp16 (packed atomics; n_features == 2)."""
        n = inputs.numel() // 5
        check(_lib.lib().rtxn_hashgrid_backward_mixed(self._h, _ptr(inputs, torch.float32, "inputs"), _ptr(dencT, torch.float16, "dencT"),
                                                      n, _ptr(dtable, torch.float32, "dtable"),
                                                      _ptr(dtable_hashed_half, torch.float16, "dtable_hashed_half"), _stream()),
              "rtxn_hashgrid_backward_mixed")
        return dtable

    def backward(self, inputs, dencT, dtable):
        n = inputs.numel() // 5
        check(_lib.lib().rtxn_hashgrid_backward(self._h, _ptr(inputs, torch.float32, "inputs"),
                                                _ptr(dencT, torch.float16, "dencT"), n,
                                                _ptr(dtable, torch.float32, "dtable"), _stream()),
              "rtxn_hashgrid_backward")
        return dtable


def _net_encode_frequency(self, inputs, encT=None):
    n = inputs.numel() // 5
    if encT is None:
        encT = torch.empty((self.encoded_width(), padded_samples(n)), dtype=torch.float16, device=inputs.device)
    check(_lib.lib().rtxn_encode_frequency(self._h, _ptr(inputs, torch.float32, "inputs"), _ptr(encT, torch.float16),
                                           n, _stream()), "rtxn_encode_frequency")
    return encT


def _net_encode_frequency_segments(self, start_points, end_points, seg_view, n_segments, sample_type, encT, t_vals=None, t_scale=1.0):
    """launchSampler + Composite-Frequency encoding in one pass over the packed segments."""
    check(_lib.lib().rtxn_encode_frequency_segments(self._h, _ptr(start_points, torch.float32, "start_points"),
                                                    _ptr(end_points, torch.float32, "end_points"), _ptr(seg_view, torch.float32, "seg_view"),
                                                    n_segments, sample_type, t_scale, _ptr(encT, torch.float16, "encT"),
                                                    _ptr(t_vals, torch.float32, "t_vals"), _stream()), "rtxn_encode_frequency_segments")
    return encT


def _net_train_workspace(self, n, device="cuda"):
    nbytes = _lib.lib().rtxn_mlp_train_workspace_bytes(self._h, n)
    return torch.empty(nbytes // 2, dtype=torch.float16, device=device)


def _net_train_forward(self, encT, n, workspace, output=None, radiance=None):
    """network->forward(stream, input, &output, use_inference_params=false, prepare_input_gradients) (main.cu:721)."""
    if output is None:
        output = torch.empty((n, 16), dtype=torch.float16, device=encT.device)
    check(_lib.lib().rtxn_mlp_train_forward(self._h, _ptr(encT, torch.float16, "encT"), n, _ptr(workspace, torch.float16),
                                            _ptr(output, torch.float16), _ptr(radiance, torch.float32, "radiance"),
                                            _stream()), "rtxn_mlp_train_forward")
    return output


def _net_train_backward(self, encT, output, dout, n, workspace, dparams, dencT=None):
    """network->backward(stream, ctx, input, output, dL_doutput) (main.cu:781)."""
    check(_lib.lib().rtxn_mlp_train_backward(self._h, _ptr(encT, torch.float16, "encT"), _ptr(output, torch.float16, "output"),
                                             _ptr(dout, torch.float16, "dout"), n, _ptr(workspace, torch.float16),
                                             _ptr(dparams, torch.float32, "dparams"), _ptr(dencT, torch.float16, "dencT"),
                                             _stream()), "rtxn_mlp_train_backward")
    return dparams


def _net_recompute_supported(self):
    """True for models whose backward can run as the fused recompute kernel (64 wide, <= 4 layers, encoded width <= 64)."""
    return bool(_lib.lib().rtxn_mlp_train_recompute_supported(self._h))


def _net_train_forward_outputs(self, encT, n, output=None, radiance=None):
    """network->forward without saved activations (the forward half of the recompute path)."""
    if output is None:
        output = torch.empty((n, 16), dtype=torch.float16, device=encT.device)
    check(_lib.lib().rtxn_mlp_train_forward_outputs(self._h, _ptr(encT, torch.float16, "encT"), n, _ptr(output, torch.float16),
                                                    _ptr(radiance, torch.float32, "radiance"), _stream()),
          "rtxn_mlp_train_forward_outputs")
    return output


def _net_train_backward_recompute(self, encT, output, dout, n, dparams, dencT=None):
    """network->backward as one kernel: activations rebuilt from encT, every weight gradient accumulated on the chip."""
    check(_lib.lib().rtxn_mlp_train_backward_recompute(self._h, _ptr(encT, torch.float16, "encT"), _ptr(output, torch.float16, "output"),
                                                       _ptr(dout, torch.float16, "dout"), n, _ptr(dparams, torch.float32, "dparams"),
                                                       _ptr(dencT, torch.float16, "dencT"), _stream()),
          "rtxn_mlp_train_backward_recompute")
    return dparams


Network.recompute_supported = _net_recompute_supported
Network.train_forward_outputs = _net_train_forward_outputs
Network.train_backward_recompute = _net_train_backward_recompute
Network.encode_frequency = _net_encode_frequency
Network.encode_frequency_segments = _net_encode_frequency_segments
def _net_train_backward_recompute_live(self, encT, output, dout, n, live_ws, dparams, dencT=None):
    check(_lib.lib().rtxn_mlp_train_backward_recompute_live(self._h, _ptr(encT, torch.float16, "encT"), _ptr(output, torch.float16, "output"),
                                                           _ptr(dout, torch.float16, "dout"), n, _ptr(live_ws, None, "live_ws"),
                                                           _ptr(dparams, torch.float32, "dparams"), _ptr(dencT, torch.float16, "dencT"),
                                                           _stream()), "rtxn_mlp_train_backward_recompute_live")


def _net_train_backward_live(self, encT, output, dout, n, workspace, live_ws, dparams, dencT=None):
    check(_lib.lib().rtxn_mlp_train_backward_live(self._h, _ptr(encT, torch.float16, "encT"), _ptr(output, torch.float16, "output"),
                                                 _ptr(dout, torch.float16, "dout"), n, _ptr(workspace, torch.float16, "workspace"),
                                                 _ptr(live_ws, None, "live_ws"), _ptr(dparams, torch.float32, "dparams"),
                                                 _ptr(dencT, torch.float16, "dencT"), _stream()), "rtxn_mlp_train_backward_live")


def _net_train_forward_live(self, encT, n, workspace, live_ws):
    """network->forward's SAVED ACTIVATIONS for the live segments only (the outputs come from train_forward_outputs)."""
    check(_lib.lib().rtxn_mlp_train_forward_live(self._h, _ptr(encT, torch.float16, "encT"), n, _ptr(workspace, torch.float16, "workspace"),
                                                 _ptr(live_ws, None, "live_ws"), _stream()), "rtxn_mlp_train_forward_live")


def _net_lean_supported(self):
    """True for models with the lean training path (the reference's 8 x 128 model): no saved activations, the weight gradient
    recomputes them (rtxn_mlp_train_lean_supported)."""
    return bool(_lib.lib().rtxn_mlp_train_lean_supported(self._h))


def _net_train_lean_workspace(self, n, device="cuda"):
    nbytes = _lib.lib().rtxn_mlp_train_lean_workspace_bytes(self._h, n)
    if nbytes == 0:
        raise _lib.RtxnError("train_lean_workspace: this model has no lean path")
    return torch.empty(nbytes // 2, dtype=torch.float16, device=device)


def _net_train_forward_lean(self, encT, n, workspace, output=None, radiance=None):
    """network->forward keeping outputs + sign masks only (rtxn_mlp_train_forward_lean)."""
    if output is None:
        output = torch.empty((n, 16), dtype=torch.float16, device=encT.device)
    check(_lib.lib().rtxn_mlp_train_forward_lean(self._h, _ptr(encT, torch.float16, "encT"), n, _ptr(workspace, torch.float16, "workspace"),
                                                 _ptr(output, torch.float16), _ptr(radiance, torch.float32, "radiance"), _stream()),
          "rtxn_mlp_train_forward_lean")
    return output


def _net_lean_fused_supported(self):
    """True where the lean forward can encode for itself (the reference's Composite-Frequency(3 x 10, 2 x 12) model)."""
    return bool(_lib.lib().rtxn_mlp_train_forward_lean_fused_supported(self._h))


def _net_train_forward_lean_segments(self, start_points, end_points, seg_view, n_segments, sample_type, workspace, output, radiance=None,
                                     t_vals=None, t_scale=1.0):
    """The lean forward with sampler and encoder folded in (rtxn_mlp_train_forward_lean_segments): encT is not read; t_vals as the
    standalone encoder writes them."""
    check(_lib.lib().rtxn_mlp_train_forward_lean_segments(self._h, _ptr(start_points, torch.float32, "start_points"),
                                                          _ptr(end_points, torch.float32, "end_points"), _ptr(seg_view, torch.float32, "seg_view"),
                                                          n_segments, sample_type, t_scale, _ptr(t_vals, torch.float32, "t_vals"),
                                                          _ptr(workspace, torch.float16, "workspace"), _ptr(output, torch.float16),
                                                          _ptr(radiance, torch.float32, "radiance"), _stream()),
          "rtxn_mlp_train_forward_lean_segments")
    return output


def _net_train_backward_lean_segments(self, start_points, end_points, seg_view, n_segments, sample_type, output, dout, workspace, dparams,
                                      live_ws=None):
    """The lean backward whose weight gradient recomputes the encoding as well (rtxn_mlp_train_backward_lean_segments): no encT."""
    check(_lib.lib().rtxn_mlp_train_backward_lean_segments(self._h, _ptr(start_points, torch.float32, "start_points"),
                                                           _ptr(end_points, torch.float32, "end_points"), _ptr(seg_view, torch.float32, "seg_view"),
                                                           n_segments, sample_type, _ptr(output, torch.float16), _ptr(dout, torch.float16),
                                                           _ptr(workspace, torch.float16, "workspace"), _ptr(live_ws, None, "live_ws"),
                                                           _ptr(dparams, torch.float32), _stream()),
          "rtxn_mlp_train_backward_lean_segments")
    return dparams


def _net_train_backward_lean(self, encT, output, dout, n, workspace, dparams, live_ws=None):
    """network->backward on the lean workspace: dgrad chain + weight gradient with recomputed activations."""
    check(_lib.lib().rtxn_mlp_train_backward_lean(self._h, _ptr(encT, torch.float16, "encT"), _ptr(output, torch.float16, "output"),
                                                  _ptr(dout, torch.float16, "dout"), n, _ptr(workspace, torch.float16, "workspace"),
                                                  _ptr(live_ws, None, "live_ws"), _ptr(dparams, torch.float32, "dparams"), _stream()),
          "rtxn_mlp_train_backward_lean")
    return dparams


Network.lean_supported = _net_lean_supported
Network.train_lean_workspace = _net_train_lean_workspace
Network.train_forward_lean = _net_train_forward_lean
Network.lean_fused_supported = _net_lean_fused_supported
Network.train_forward_lean_segments = _net_train_forward_lean_segments
Network.train_backward_lean_segments = _net_train_backward_lean_segments
Network.train_backward_lean = _net_train_backward_lean
Network.train_forward_live = _net_train_forward_live
Network.train_backward_recompute_live = _net_train_backward_recompute_live
Network.train_backward_live = _net_train_backward_live
Network.train_workspace = _net_train_workspace
Network.train_forward = _net_train_forward
Network.train_backward = _net_train_backward


def l2_loss(pred, target, loss_scale=1.0, values=None, grads=None, loss_sum=None):
    """loss->evaluate(loss_scale, prediction, target, values, gradients) (main.cu:759)."""
    n = pred.numel()
    check(_lib.lib().rtxn_l2_loss(_ptr(pred, torch.float32, "pred"), _ptr(target, torch.float32, "target"), n, loss_scale,
                                  _ptr(values, torch.float32, "values"), _ptr(grads, torch.float16, "grads"),
                                  _ptr(loss_sum, torch.float32, "loss_sum"), _stream()), "rtxn_l2_loss")


def adam_step(master, params_fp16, grads, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, loss_scale=1.0):
    """optimizer->step(stream, loss_scale, params_fp32, params, gradients) (main.cu:787)."""
    check(_lib.lib().rtxn_adam_step(master.numel(), _ptr(master, torch.float32, "master"),
                                    _ptr(params_fp16, torch.float16, "params"), _ptr(grads, torch.float32, "grads"),
                                    _ptr(m, torch.float32, "m"), _ptr(v, torch.float32, "v"), step, lr, beta1, beta2, eps,
                                    loss_scale, _stream()), "rtxn_adam_step")


def adam_step_half_grads(master, params_fp16, grads_fp16, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, loss_scale=1.0):
    """adam_step with an fp16 gradient."""
    check(_lib.lib().rtxn_adam_step_half_grads(master.numel(), _ptr(master, torch.float32, "master"),
                                               _ptr(params_fp16, torch.float16, "params"), _ptr(grads_fp16, torch.float16, "grads"),
                                               _ptr(m, torch.float32, "m"), _ptr(v, torch.float32, "v"), step, lr, beta1, beta2, eps,
                                               loss_scale, _stream()), "rtxn_adam_step_half_grads")


def adam_effective_lr(lr, beta1, beta2, step):
    """The bias-corrected rate rtxn_adam_step uses at `step` (computed by the library, so the captured form is bit-identical)."""
    return float(_lib.lib().rtxn_adam_effective_lr(lr, beta1, beta2, int(step)))


ADAM_GRADS_FP16, ADAM_ZERO_GRADS = 1, 2


def adam_step_captured(master, params_fp16, grads, m, v, effective_lr, beta1=0.9, beta2=0.999, eps=1e-8, loss_scale=1.0, zero_grads=False):
    """adam_step / adam_step_half_grads (by the dtype of `grads`) with the bias-corrected rate read from the device float
    `effective_lr`: the form a hipGraph can replay.  zero_grads: clear the gradient as it is consumed."""
    half = grads.dtype == torch.float16
    flags = (ADAM_GRADS_FP16 if half else 0) | (ADAM_ZERO_GRADS if zero_grads else 0)
    check(_lib.lib().rtxn_adam_step_captured(master.numel(), _ptr(master, torch.float32, "master"),
                                             _ptr(params_fp16, torch.float16, "params"),
                                             _ptr(grads, torch.float16 if half else torch.float32, "grads"), flags,
                                             _ptr(m, torch.float32, "m"), _ptr(v, torch.float32, "v"),
                                             _ptr(effective_lr, torch.float32, "effective_lr"), beta1, beta2, eps, loss_scale,
                                             _stream()), "rtxn_adam_step_captured")


def adam_step_sparse(master, params_fp16, grads, m, v, param_steps, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, loss_scale=1.0,
                     zero_grads=False):
    """rtxn_adam_step_sparse: tiny-cuda-nn's Adam for the hash table -- entries with a zero gradient are skipped, bias correction
    by the entry's own update count (param_steps, int32/uint32[n]).  grads: fp32 or fp16."""
    half = grads.dtype == torch.float16
    flags = (ADAM_GRADS_FP16 if half else 0) | (ADAM_ZERO_GRADS if zero_grads else 0)
    check(_lib.lib().rtxn_adam_step_sparse(master.numel(), _ptr(master, torch.float32, "master"),
                                           _ptr(params_fp16, torch.float16, "params"),
                                           _ptr(grads, torch.float16 if half else torch.float32, "grads"), flags,
                                           _ptr(m, torch.float32, "m"), _ptr(v, torch.float32, "v"),
                                           _ptr(param_steps, torch.int32, "param_steps"), lr, beta1, beta2, eps, loss_scale,
                                           _stream()), "rtxn_adam_step_sparse")


def deterministic_shadow(n_params, device="cuda"):
    """A zeroed 64-bit fixed-point shadow for a gradient buffer of n_params floats (rtxn_deterministic_workspace_bytes)."""
    nbytes = _lib.lib().rtxn_deterministic_workspace_bytes(int(n_params))
    return torch.zeros(nbytes // 8, dtype=torch.int64, device=device)


def set_deterministic(mlp_shadow=None, table_shadow=None):
    """rtxn_set_deterministic_workspace: process-wide; (None, None) restores the float atomics.  The caller keeps the tensors alive."""
    check(_lib.lib().rtxn_set_deterministic_workspace(_ptr(mlp_shadow, torch.int64, "mlp_shadow"), _ptr(table_shadow, torch.int64, "table_shadow")),
          "rtxn_set_deterministic_workspace")


def train_gradients(net, *, grid=None, n_dir_freqs=0, table=None, start_points, end_points, seg_view, num_stored, indices,
                    total_segments, segment_capacity, n_rays, sample_type, t_scale=1.0, vr_mode, targets, loss_scale,
                    encT, dencT=None, workspace=None, output_half, radiance, t_vals, radiance_gradients, pixels, loss_gradients,
                    loss_sum=None, dparams, dtable=None, dtable_hashed_half=None, live_ws=None, skip_table_backward=False,
                    workspace_lean=False):
    """rtxn_train_gradients: sampler ... backward of one batch with the segment count taken on the device (main.cu:703-781)."""
    b = train_batch(**{k: v for k, v in locals().items()})
    check(_lib.lib().rtxn_train_gradients(C.byref(b), _stream()), "rtxn_train_gradients")


def train_batch(net, *, grid=None, n_dir_freqs=0, table=None, start_points, end_points, seg_view, num_stored, indices,
                total_segments, segment_capacity, n_rays, sample_type, t_scale=1.0, vr_mode, targets, loss_scale,
                encT, dencT=None, workspace=None, output_half, radiance, t_vals, radiance_gradients, pixels, loss_gradients,
                loss_sum=None, dparams, dtable=None, dtable_hashed_half=None, live_ws=None, skip_table_backward=False,
                workspace_lean=False):
    """struct rtxn_train_batch over the given tensors (which the caller keeps alive), sizes checked against the capacity."""
    b = _lib.TrainBatch()
    b.mlp, b.grid = net._h, (grid._h if grid is not None else None)
    b.n_dir_freqs = int(n_dir_freqs)
    b.table_fp16 = _ptr(table, torch.float16, "table")
    b.start_points, b.end_points = _ptr(start_points, torch.float32, "start_points"), _ptr(end_points, torch.float32, "end_points")
    b.seg_view = _ptr(seg_view, torch.float32, "seg_view")
    b.num_stored, b.indices = _ptr(num_stored, torch.int32, "num_stored"), _ptr(indices, torch.int32, "indices")
    b.total_segments = _ptr(total_segments, torch.int32, "total_segments")
    b.segment_capacity, b.n_rays, b.sample_type, b.t_scale, b.vr_mode = int(segment_capacity), int(n_rays), int(sample_type), float(t_scale), int(vr_mode)
    b.targets, b.loss_scale = _ptr(targets, torch.float32, "targets"), float(loss_scale)
    b.encT, b.dencT = _ptr(encT, torch.float16, "encT"), _ptr(dencT, torch.float16, "dencT")
    b.workspace = _ptr(workspace, torch.float16, "workspace")
    b.output_half, b.radiance = _ptr(output_half, torch.float16, "output_half"), _ptr(radiance, torch.float32, "radiance")
    b.t_vals, b.radiance_gradients = _ptr(t_vals, torch.float32, "t_vals"), _ptr(radiance_gradients, torch.float16, "radiance_gradients")
    b.pixels, b.loss_gradients_half = _ptr(pixels, torch.float32, "pixels"), _ptr(loss_gradients, torch.float16, "loss_gradients")
    b.loss_sum = _ptr(loss_sum, torch.float32, "loss_sum")
    b.dparams, b.dtable = _ptr(dparams, torch.float32, "dparams"), _ptr(dtable, torch.float32, "dtable")
    b.dtable_hashed_half = _ptr(dtable_hashed_half, torch.float16, "dtable_hashed_half")
    if live_ws is not None and live_ws.numel() * live_ws.element_size() < live_segments_workspace_bytes(segment_capacity):
        raise _lib.RtxnError("train_gradients: live_ws smaller than live_segments_workspace_bytes(segment_capacity)")
    b.live_ws = _ptr(live_ws, None, "live_ws")
    b.skip_table_backward = 1 if skip_table_backward else 0
    b.workspace_lean = 1 if workspace_lean else 0
    for nm, t, need in (("encT", encT, net.encoded_width() * padded_samples(32 * int(segment_capacity))),
                        ("output_half", output_half, 32 * int(segment_capacity) * 16), ("radiance", radiance, 32 * int(segment_capacity) * 4),
                        ("t_vals", t_vals, 32 * int(segment_capacity)), ("radiance_gradients", radiance_gradients, 32 * int(segment_capacity) * 4),
                        ("start_points", start_points, 3 * int(segment_capacity)), ("end_points", end_points, 3 * int(segment_capacity)),
                        ("seg_view", seg_view, 2 * int(segment_capacity)), ("pixels", pixels, 3 * int(n_rays)),
                        ("targets", targets, 3 * int(n_rays)), ("num_stored", num_stored, int(n_rays)), ("indices", indices, int(n_rays))):
        if t.numel() < need:
            raise _lib.RtxnError(f"train_gradients: {nm} holds {t.numel()} elements, {need} needed for capacity {segment_capacity} / {n_rays} rays")
    return b


def train_step(args):
    """rtxn_train_step(args: _lib.TrainStepArgs): traversal -> gradients -> optimizer of one batch, one call, current stream."""
    check(_lib.lib().rtxn_train_step(C.byref(args), _stream()), "rtxn_train_step")


def half2_workspace(values, block_entries):
    """int32 workspace of rtxn_half2_count_nonzero / _pack_nonzero for `values` (fp16, two halves per entry)."""
    nbytes = int(_lib.lib().rtxn_half2_workspace_bytes(values.numel() // 2, int(block_entries)))
    return torch.zeros(nbytes // 4, dtype=torch.int32, device=values.device)


def half2_count_nonzero(values, block_entries, workspace=None):
    """rtxn_half2_count_nonzero: non-zero half2 entries per block of `values` -> workspace (its first `blocks` ints are the
    per-block counts; returns (counts view, workspace))."""
    n = values.numel() // 2
    nb = (n + block_entries - 1) // block_entries
    if workspace is None:
        workspace = half2_workspace(values, block_entries)
    check(_lib.lib().rtxn_half2_count_nonzero(_ptr(values, torch.float16, "values"), n, int(block_entries),
                                              _ptr(workspace, torch.int32, "workspace"), _stream()), "rtxn_half2_count_nonzero")
    return workspace[:nb], workspace


def half2_pack_nonzero(values, block_entries, workspace, block_mask, pairs, count, clear=True):
    """rtxn_half2_pack_nonzero: (index, half2 bits) of the non-zero entries of the masked blocks, ascending -> pairs
    int32[capacity][2]; count (device int32[1]) = entries needed.  workspace: as half2_count_nonzero left it for these values."""
    need = int(_lib.lib().rtxn_half2_workspace_bytes(values.numel() // 2, int(block_entries)))
    if workspace.numel() * 4 < need:
        raise _lib.RtxnError(f"half2_pack_nonzero: workspace of {workspace.numel() * 4} bytes, {need} needed")
    check(_lib.lib().rtxn_half2_pack_nonzero(_ptr(values, torch.float16, "values"), values.numel() // 2, int(block_entries),
                                             _ptr(workspace, torch.int32, "workspace"), int(block_mask), pairs.numel() // 2,
                                             _ptr(pairs, torch.int32, "pairs"), _ptr(count, torch.int32, "count"),
                                             1 if clear else 0, _stream()), "rtxn_half2_pack_nonzero")


def half2_add_pairs(values, pairs, count):
    """rtxn_half2_add_pairs: values[index] += value over the first `count` pairs of one list."""
    if count > pairs.numel() // 2:
        raise _lib.RtxnError(f"half2_add_pairs: count {count} > {pairs.numel() // 2} pairs held")
    check(_lib.lib().rtxn_half2_add_pairs(_ptr(values, torch.float16, "values"), values.numel() // 2, _ptr(pairs, torch.int32, "pairs"),
                                          int(count), _stream()), "rtxn_half2_add_pairs")


def live_segments_workspace_bytes(segment_capacity):
    return int(_lib.lib().rtxn_live_segments_workspace_bytes(int(segment_capacity)))


def live_segments_workspace(segment_capacity, device="cuda"):
    """[int count | pad | int list[capacity] | flags]: see rtxn_live_segments (include/rtxn.h)."""
    return torch.zeros((live_segments_workspace_bytes(segment_capacity) + 3) // 4, dtype=torch.int32, device=device)


def live_segments(radiance_gradients, n_segments, segment_capacity, live_ws):
    """List the 32-sample segments whose radiance gradient is not all zero; count = live_ws[0], list = live_ws[4:4+count]."""
    check(_lib.lib().rtxn_live_segments(_ptr(radiance_gradients, torch.float16, "radiance_gradients"), int(n_segments),
                                        int(segment_capacity), _ptr(live_ws, None, "live_ws"), _stream()), "rtxn_live_segments")


def convert_f32_to_f16(src, dst):
    check(_lib.lib().rtxn_convert_f32_to_f16(_ptr(src, torch.float32, "src"), _ptr(dst, torch.float16, "dst"), src.numel(), _stream()),
          "rtxn_convert_f32_to_f16")


def convert_f16_to_f32(src, dst):
    check(_lib.lib().rtxn_convert_f16_to_f32(_ptr(src, torch.float16, "src"), _ptr(dst, torch.float32, "dst"), src.numel(), _stream()),
          "rtxn_convert_f16_to_f32")
