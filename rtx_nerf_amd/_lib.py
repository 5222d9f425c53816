"""Loader for librtxn.so (the HIP implementation behind include/rtxn.h).

torch is imported FIRST so that the HIP runtime torch ships (soname
libamdhip64.so.7) is the one instance both torch and librtxn use: device
pointers of torch tensors are then valid in librtxn's kernels and streams are
shared.  There is no CPU fallback: a missing library is an ImportError-grade
failure, and compute entry points fail with RTXN_ERR_HIP without a GPU.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTXN_LIB_PATH: load an alternative build of the same ABI (kernel A/B experiments in tools/)
LIB_PATH = os.environ.get("RTXN_LIB_PATH") or os.path.join(_HERE, "librtxn.so")

RTXN_OK = 0


class RtxnError(RuntimeError):
    pass


class TraceParams(C.Structure):
    """struct rtxn_trace_params (include/rtxn.h)."""
    _fields_ = [
        ("look_at", C.c_void_p),
        ("focal_length", C.c_float),
        ("aspect_ratio", C.c_float),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("rays_o", C.c_void_p),
        ("rays_d", C.c_void_p),
        ("ray_begin", C.c_uint32),
        ("ray_count", C.c_uint32),
        ("window_chunk", C.c_uint32),
        ("window_stride", C.c_uint32),
        ("grid_res", C.c_int),
        ("occupancy", C.c_void_p),
        ("occupancy_coarse", C.c_void_p),
        ("mode", C.c_int),
        ("ray_origins", C.c_void_p),
        ("viewing_direction", C.c_void_p),
        ("num_hits", C.c_void_p),
        ("intersection_arr_size", C.c_int),
        ("indices", C.c_void_p),
        ("start_points", C.c_void_p),
        ("end_points", C.c_void_p),
        ("t_start", C.c_void_p),
        ("t_end", C.c_void_p),
        ("seg_ray", C.c_void_p),
        ("seg_view", C.c_void_p),
        ("segment_capacity", C.c_long),
        ("seg_first", C.c_void_p),
        ("occupancy_bricks", C.c_void_p),
        ("occupancy_super", C.c_void_p),
        ("num_stored", C.c_void_p),
        ("sub_rays", C.c_int),
        ("sub_hits", C.c_void_p),
    ]


class HashGridConfig(C.Structure):
    """struct rtxn_hashgrid_config (include/rtxn.h)."""
    _fields_ = [("n_levels", C.c_int), ("n_features", C.c_int), ("log2_hashmap_size", C.c_int),
                ("base_resolution", C.c_int), ("per_level_scale", C.c_float)]


class ImageDataset(C.Structure):
    """struct rtxn_image_dataset (include/rtxn.h)."""
    _fields_ = [("n_images", C.c_int), ("image_width", C.c_uint), ("image_height", C.c_uint), ("image_channels", C.c_uint),
                ("focal", C.c_float), ("camera_angle_x", C.c_float), ("images", C.POINTER(C.c_float)),
                ("poses", C.POINTER(C.c_float))]


class MlpConfig(C.Structure):
    """struct rtxn_mlp_config (include/rtxn.h)."""
    _fields_ = [(n, C.c_int) for n in (
        "n_pos_dims", "n_pos_freqs", "n_dir_dims", "n_dir_freqs",
        "n_neurons", "n_hidden_layers", "n_output_dims", "output_activation", "encoding", "n_encoded_features")]


class TrainBatch(C.Structure):
    """struct rtxn_train_batch (include/rtxn.h)."""
    _fields_ = [("mlp", C.c_void_p), ("grid", C.c_void_p), ("n_dir_freqs", C.c_int), ("table_fp16", C.c_void_p),
                ("start_points", C.c_void_p), ("end_points", C.c_void_p), ("seg_view", C.c_void_p),
                ("num_stored", C.c_void_p), ("indices", C.c_void_p), ("total_segments", C.c_void_p),
                ("segment_capacity", C.c_long), ("n_rays", C.c_int), ("sample_type", C.c_int), ("t_scale", C.c_float),
                ("vr_mode", C.c_int), ("targets", C.c_void_p), ("loss_scale", C.c_float),
                ("encT", C.c_void_p), ("dencT", C.c_void_p), ("workspace", C.c_void_p), ("output_half", C.c_void_p),
                ("radiance", C.c_void_p), ("t_vals", C.c_void_p), ("radiance_gradients", C.c_void_p),
                ("pixels", C.c_void_p), ("loss_gradients_half", C.c_void_p), ("loss_sum", C.c_void_p),
                ("dparams", C.c_void_p), ("dtable", C.c_void_p), ("dtable_hashed_half", C.c_void_p), ("live_ws", C.c_void_p),
                ("skip_table_backward", C.c_int), ("workspace_lean", C.c_int)]


class TrainState(C.Structure):
    """struct rtxn_train_state (include/rtxn.h)."""
    _fields_ = [("mlp_master", C.c_void_p), ("mlp_params_fp16", C.c_void_p), ("mlp_m", C.c_void_p), ("mlp_v", C.c_void_p),
                ("table_master", C.c_void_p), ("table_params_fp16", C.c_void_p), ("table_m", C.c_void_p), ("table_v", C.c_void_p),
                ("table_steps", C.c_void_p), ("step", C.c_void_p), ("effective_lr", C.c_void_p),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("table_lr", C.c_float), ("table_eps", C.c_float), ("loss_scale_divisor", C.c_float)]


class TrainStepArgs(C.Structure):
    """struct rtxn_train_step_args (include/rtxn.h)."""
    _fields_ = [("trace", TraceParams), ("scan_workspace", C.c_void_p), ("scan_workspace_bytes", C.c_size_t),
                ("batch", TrainBatch), ("opt", TrainState)]


class RenderConfig(C.Structure):
    """struct rtxn_render_config (include/rtxn.h)."""
    _fields_ = [("mlp", C.c_void_p), ("grid", C.c_void_p), ("table_fp16", C.c_void_p), ("n_dir_freqs", C.c_int),
                ("width", C.c_uint32), ("height", C.c_uint32), ("focal_length", C.c_float), ("aspect_ratio", C.c_float),
                ("max_rays", C.c_uint32), ("window_chunk", C.c_uint32), ("window_stride", C.c_uint32), ("grid_res", C.c_int),
                ("occupancy", C.c_void_p), ("trace_mode", C.c_int), ("sub_rays", C.c_int), ("vr_mode", C.c_int),
                ("sample_type", C.c_int), ("step_scale", C.c_float), ("max_segments", C.c_long), ("n_slots", C.c_int),
                ("flags", C.c_int)]


class RenderStats(C.Structure):
    """struct rtxn_render_stats (include/rtxn.h)."""
    _fields_ = [(n, C.c_long) for n in ("frames", "frames_checked", "overflow_frames", "max_segments_needed", "last_segments",
                                        "max_segments")]


# every symbol include/rtxn.h declares: name -> (restype, argtypes)
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_long, C.c_float
SYMBOLS = {
    "rtxn_version": (_I, []),
    "rtxn_last_error": (C.c_char_p, []),
    "rtxn_trace_grid": (_I, [C.POINTER(TraceParams), _P]),
    "rtxn_build_occupancy_mip": (_I, [_P, _I, _P, _P]),
    "rtxn_occupancy_from_density": (_I, [_P, _F, _I, _P, _P]),
    "rtxn_build_occupancy_bricks": (_I, [_P, _I, _P, _P]),
    "rtxn_scan_workspace_bytes": (C.c_size_t, [_I]),
    "rtxn_scan_hits": (_I, [_P, _P, _P, _I, _P, C.c_size_t, _P]),
    "rtxn_sample": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _P]),
    "rtxn_volrender_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _I, _P]),
    "rtxn_volrender_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P]),
    "rtxn_volrender_l2_train": (_I, [_P, _P, _P, _P, _I, _I, _P, _F, _P, _P, _P, _P, _P]),
    "rtxn_mlp_create": (_I, [C.POINTER(MlpConfig), C.POINTER(_P)]),
    "rtxn_mlp_destroy": (_I, [_P]),
    "rtxn_mlp_n_params": (_L, [_P]),
    "rtxn_mlp_padded_output_width": (_I, [_P]),
    "rtxn_mlp_set_reserved_cus": (_I, [_P, _I]),
    "rtxn_mlp_mfma_shape": (_I, [_P]),
    "rtxn_mlp_encoded_width": (_I, [_P]),
    "rtxn_mlp_initialize_params": (_I, [_P, C.c_uint64, _P]),
    "rtxn_mlp_set_params": (_I, [_P, _P, _P]),
    "rtxn_mlp_set_params_training": (_I, [_P, _P, _P]),
    "rtxn_mlp_forward": (_I, [_P, _P, _P, _L, _P]),
    "rtxn_mlp_forward_radiance": (_I, [_P, _P, _P, _L, _P]),
    "rtxn_mlp_forward_segments": (_I, [_P, _P, _P, _P, _P, _L, _P, _P, _P]),
    "rtxn_mlp_forward_segments_compact": (_I, [_P, _P, _P, _P, _P, _L, _P, _P]),
    "rtxn_volrender_fwd_compact": (_I, [_P, _P, _P, _I, _I, _P, _P]),
    "rtxn_volrender_fwd_compact_nerf": (_I, [_P, _P, _P, _P, _I, _I, _P, _P]),
    "rtxn_hashmlp_supported": (_I, [_P, _P, _I]),
    "rtxn_hashmlp_forward_segments": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _L, _I, _F, _P, _P, _P]),
    "rtxn_render_workspace_bytes": (C.c_size_t, [C.POINTER(RenderConfig)]),
    "rtxn_render_create": (_I, [C.POINTER(RenderConfig), _P, C.c_size_t, C.POINTER(_P)]),
    "rtxn_render_destroy": (_I, [_P]),
    "rtxn_render_set_occupancy": (_I, [_P, _P, _P]),
    "rtxn_render_count_segments": (_I, [_P, _P, C.c_uint32, C.c_uint32, C.POINTER(C.c_long), _P]),
    "rtxn_render_frame": (_I, [_P, _I, _P, C.c_uint32, C.c_uint32, _P, _P]),
    "rtxn_render_frame_async": (_I, [_P, _P, C.c_uint32, C.c_uint32, _P, _P, C.POINTER(_P)]),
    "rtxn_render_frame_async_host": (_I, [_P, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, _P, _P, C.POINTER(_P)]),
    "rtxn_render_drain": (_I, [_P, _P]),
    "rtxn_render_status": (_I, [_P, _I, C.POINTER(RenderStats)]),
    "rtxn_render_slot_buffers": (_I, [_P, _I] + [C.POINTER(_P)] * 11),
    "rtxn_padded_samples": (_L, [_L]),
    "rtxn_encode_frequency": (_I, [_P, _P, _P, _L, _P]),
    "rtxn_hashgrid_create": (_I, [C.POINTER(HashGridConfig), C.POINTER(_P)]),
    "rtxn_hashgrid_destroy": (_I, [_P]),
    "rtxn_hashgrid_n_params": (_L, [_P]),
    "rtxn_hashgrid_encoded_width": (_I, [_P, _I]),
    "rtxn_hashgrid_level_offset": (_L, [_P, _I]),
    "rtxn_hashgrid_level_is_hashed": (_I, [_P, _I]),
    "rtxn_half2_workspace_bytes": (C.c_size_t, [_L, _L]),
    "rtxn_half2_count_nonzero": (_I, [_P, _L, _L, _P, _P]),
    "rtxn_half2_pack_nonzero": (_I, [_P, _L, _L, _P, C.c_ulonglong, _L, _P, _P, _I, _P]),
    "rtxn_half2_add_pairs": (_I, [_P, _L, _P, _L, _P]),
    "rtxn_convert_f32_to_f16": (_I, [_P, _P, _L, _P]),
    "rtxn_convert_f16_to_f32": (_I, [_P, _P, _L, _P]),
    "rtxn_hashgrid_encode": (_I, [_P, _I, _P, _P, _P, _L, _P]),
    "rtxn_hashgrid_backward": (_I, [_P, _P, _P, _L, _P, _P]),
    "rtxn_hashgrid_backward_mixed": (_I, [_P, _P, _P, _L, _P, _P, _P]),
    "rtxn_encode_frequency_segments": (_I, [_P, _P, _P, _P, _L, _I, _F, _P, _P, _P]),
    "rtxn_hashgrid_encode_segments": (_I, [_P, _I, _P, _P, _P, _P, _L, _I, _F, _P, _P, _P]),
    "rtxn_hashgrid_backward_segments": (_I, [_P, _P, _P, _L, _I, _P, _P, _P, _P]),
    "rtxn_mlp_train_workspace_bytes": (C.c_size_t, [_P, _L]),
    "rtxn_mlp_train_forward": (_I, [_P, _P, _L, _P, _P, _P, _P]),
    "rtxn_mlp_train_backward": (_I, [_P, _P, _P, _P, _L, _P, _P, _P, _P]),
    "rtxn_mlp_train_recompute_supported": (_I, [_P]),
    "rtxn_mlp_train_forward_outputs": (_I, [_P, _P, _L, _P, _P, _P]),
    "rtxn_mlp_train_backward_recompute": (_I, [_P, _P, _P, _P, _L, _P, _P, _P]),
    "rtxn_l2_loss": (_I, [_P, _P, _L, _F, _P, _P, _P, _P]),
    "rtxn_adam_step": (_I, [_L, _P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _F, _P]),
    "rtxn_adam_step_half_grads": (_I, [_L, _P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _F, _P]),
    "rtxn_adam_effective_lr": (_F, [_F, _F, _F, _I]),
    "rtxn_adam_step_captured": (_I, [_L, _P, _P, _P, _I, _P, _P, _P, _F, _F, _F, _F, _P]),
    "rtxn_adam_step_sparse": (_I, [_L, _P, _P, _P, _I, _P, _P, _P, _F, _F, _F, _F, _F, _P]),
    "rtxn_deterministic_workspace_bytes": (C.c_size_t, [_L]),
    "rtxn_set_deterministic_workspace": (_I, [_P, _P]),
    "rtxn_mlp_train_lean_supported": (_I, [_P]),
    "rtxn_mlp_train_lean_workspace_bytes": (C.c_size_t, [_P, _L]),
    "rtxn_mlp_train_forward_lean": (_I, [_P, _P, _L, _P, _P, _P, _P]),
    "rtxn_mlp_train_backward_lean": (_I, [_P, _P, _P, _P, _L, _P, _P, _P, _P]),
    "rtxn_mlp_train_forward_lean_fused_supported": (_I, [_P]),
    "rtxn_mlp_train_forward_lean_segments": (_I, [_P, _P, _P, _P, _L, _I, _F, _P, _P, _P, _P, _P]),
    "rtxn_mlp_train_backward_lean_segments": (_I, [_P, _P, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P]),
    "rtxn_train_gradients": (_I, [C.POINTER(TrainBatch), _P]),
    "rtxn_train_step": (_I, [C.POINTER(TrainStepArgs), _P]),
    "rtxn_live_segments_workspace_bytes": (C.c_size_t, [_L]),
    "rtxn_live_segments": (_I, [_P, _L, _L, _P, _P]),
    "rtxn_mlp_train_backward_recompute_live": (_I, [_P, _P, _P, _P, _L, _P, _P, _P, _P]),
    "rtxn_mlp_train_backward_live": (_I, [_P, _P, _P, _P, _L, _P, _P, _P, _P, _P]),
    "rtxn_mlp_train_forward_live": (_I, [_P, _P, _L, _P, _P, _P]),
    "rtxn_hashgrid_backward_segments_live": (_I, [_P, _P, _P, _L, _I, _P, _P, _P, _P, _P]),
    "rtxn_load_images_json": (_I, [C.c_char_p, C.c_char_p, _I, C.POINTER(ImageDataset)]),
    "rtxn_free_image_dataset": (None, [C.POINTER(ImageDataset)]),
    "rtxn_load_llff": (_I, [C.c_char_p, _I, _I, C.POINTER(ImageDataset), C.POINTER(C.POINTER(C.c_float))]),
    "rtxn_free_llff_bounds": (None, [C.POINTER(C.c_float)]),
    "rtxn_write_png_rgb8": (_I, [C.c_char_p, _P, _I, _I]),
}

_lib = None


def lib():
    """The loaded library; raises if it has not been built (python -c
    'import __graft_entry__ as g; g.build()' or `make`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtxnError(
                f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950). "
                "rtx_nerf_amd has no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the ABI drifted from the header
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what="librtxn"):
    if rc != RTXN_OK:
        msg = lib().rtxn_last_error()
        raise RtxnError(f"{what} failed with status {rc}: {msg.decode() if msg else '?'}")
