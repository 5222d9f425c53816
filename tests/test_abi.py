"""CPU-side checks of the C-ABI boundary: the library loads, exports every
symbol include/rtxn.h declares, the ctypes table covers exactly those symbols,
and (without a GPU) compute entry points fail loudly instead of falling back."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "rtxn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rtxn_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from rtx_nerf_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 18
    lib = _lib.lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rtxn.h but not exported by librtxn.so"
    assert sorted(_lib.SYMBOLS) == names
    assert lib.rtxn_version() == 100


def _header_fields(name):
    src = open(os.path.join(ROOT, "include", "rtxn.h")).read()
    body = src[src.index(f"typedef struct {name} {{"):src.index(f"}} {name};")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.split("{")[-1].strip()
        if not decl:
            continue
        for part in decl.split(","):
            fields.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", part.strip())[0])
    return fields


@pytest.mark.parametrize("c_name,binding", [("rtxn_trace_params", "TraceParams"), ("rtxn_mlp_config", "MlpConfig"),
                                            ("rtxn_hashgrid_config", "HashGridConfig"), ("rtxn_train_batch", "TrainBatch"),
                                            ("rtxn_render_config", "RenderConfig"), ("rtxn_render_stats", "RenderStats"),
                                            ("rtxn_train_state", "TrainState"), ("rtxn_train_step_args", "TrainStepArgs"),
                                            ("rtxn_image_dataset", "ImageDataset")])
def test_struct_layouts_match_header_order(c_name, binding):
    """Every struct that crosses the ABI: the ctypes binding lists the header's fields in the header's order."""
    from rtx_nerf_amd import _lib
    assert _header_fields(c_name) == [f[0] for f in getattr(_lib, binding)._fields_]
    assert C.sizeof(_lib.MlpConfig) == 40


def test_render_entry_validates_before_touching_a_device():
    """rtxn_render_*: argument errors are RTXN_ERR_INVALID / UNSUPPORTED with a message, with or without a GPU."""
    from rtx_nerf_amd import _lib
    lib = _lib.lib()
    cfg = _lib.MlpConfig(3, 10, 2, 12, 64, 2, 4, 1)
    h = C.c_void_p()
    assert lib.rtxn_mlp_create(C.byref(cfg), C.byref(h)) == 0
    rc = _lib.RenderConfig()
    assert lib.rtxn_render_workspace_bytes(C.byref(rc)) == 0 and b"NULL model" in lib.rtxn_last_error()
    rc.mlp, rc.width, rc.height, rc.grid_res, rc.trace_mode, rc.max_segments, rc.n_slots = h, 64, 48, 32, 1, 1000, 3
    rc.focal_length, rc.aspect_ratio = 1.0, 64 / 48
    need = lib.rtxn_render_workspace_bytes(C.byref(rc))
    assert need > 3 * (1000 * 32 * 8 + 1000 * 32) and need % 256 == 0
    rc.flags = 1                                           # float4 + t_vals hand-over: 20 B per sample instead of 8
    assert lib.rtxn_render_workspace_bytes(C.byref(rc)) > need + 3 * 1000 * 32 * 12 - 4096
    rc.flags, rc.n_slots = 0, 9
    assert lib.rtxn_render_workspace_bytes(C.byref(rc)) == 0 and b"n_slots" in lib.rtxn_last_error()
    rc.n_slots, rc.vr_mode = 3, 1                          # NERF compositing of the frequency model needs t_vals
    assert lib.rtxn_render_workspace_bytes(C.byref(rc)) == 0 and b"RTXN_RENDER_FLOAT4" in lib.rtxn_last_error()
    rc.vr_mode = 0
    out = C.c_void_p()
    assert lib.rtxn_render_create(C.byref(rc), None, 0, C.byref(out)) in (1, 2)      # no workspace (1) / no device (2)
    assert lib.rtxn_render_frame(None, 0, None, 0, 0, None, None) == 1
    assert lib.rtxn_render_destroy(None) == 0
    assert lib.rtxn_mlp_destroy(h) == 0


def test_no_cpu_fallback():
    import torch
    from rtx_nerf_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    lib = _lib.lib()
    rc = lib.rtxn_scan_hits(None, None, C.c_void_p(8), 0, None, 0, None)
    assert rc == 2, "compute entry points must fail with RTXN_ERR_HIP when there is no device"
    assert b"no HIP device" in lib.rtxn_last_error()
    from rtx_nerf_amd import api
    with pytest.raises(_lib.RtxnError):
        api.scan_hits(torch.zeros(4, dtype=torch.int32))


def test_argument_validation_precedes_device_use():
    from rtx_nerf_amd import _lib
    lib = _lib.lib()
    assert lib.rtxn_sample(None, None, None, None, None, -1, 8, None, None, 0, None) == 1
    assert b"batch_size" in lib.rtxn_last_error()
    assert lib.rtxn_volrender_fwd(None, None, None, None, None, 4, 0, None, 0, None) == 1
    cfg = _lib.MlpConfig(3, 10, 2, 12, 96, 8, 4, 1)   # 96-wide: not built
    h = C.c_void_p()
    assert lib.rtxn_mlp_create(C.byref(cfg), C.byref(h)) == 3
    cfg = _lib.MlpConfig(3, 10, 2, 12, 128, 8, 4, 1)
    assert lib.rtxn_mlp_create(C.byref(cfg), C.byref(h)) == 0
    assert lib.rtxn_mlp_n_params(h) == 131072
    assert lib.rtxn_mlp_encoded_width(h) == 112 and lib.rtxn_mlp_padded_output_width(h) == 16
    assert lib.rtxn_mlp_destroy(h) == 0


def test_initialize_params_is_seeded_xavier():
    import numpy as np
    import torch  # noqa: F401
    from rtx_nerf_amd import api
    net = api.Network(n_neurons=64, n_hidden_layers=2)
    a = net.initialize_params(1337).numpy()
    b = net.initialize_params(1337).numpy()
    c = net.initialize_params(1338).numpy()
    assert a.shape == (64 * 112 + 64 * 64 + 16 * 64,)
    np.testing.assert_array_equal(a, b)
    assert np.abs(a - c).max() > 0
    lim0 = (6.0 / (64 + 112)) ** 0.5
    assert np.abs(a[:64 * 112]).max() <= lim0 and np.abs(a[:64 * 112]).max() > 0.95 * lim0
    assert abs(a.mean()) < 5e-3


def test_header_is_plain_c99(tmp_path):
    """The boundary is a C ABI: include/rtxn.h must compile as C99 with no C++ or torch types."""
    import subprocess
    src = tmp_path / "c99.c"
    src.write_text('#include "rtxn.h"\nint main(void) { rtxn_trace_params p; rtxn_mlp_config c; rtxn_hashgrid_config h; '
                   'rtxn_image_dataset d; (void)p; (void)c; (void)h; (void)d; return RTXN_VERSION == 100 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{ROOT}/include", "-fsyntax-only", str(src)])
    text = open(os.path.join(ROOT, "include", "rtxn.h")).read()
    assert "torch" not in text and "at::" not in text and "std::" not in text


def test_auto_sub_rays_is_a_valid_lane_count():
    """rtxn_trace_params.sub_rays accepts 0/1 and the powers of two up to 64; the heuristic must stay inside and not grow with the launch."""
    from rtx_nerf_amd import api
    prev = 64
    for n in (1, 100, 4096, 29_999, 30_000, 80_000, 120_000, 299_999, 300_000, 640_000, 10_000_000):
        q = api.auto_sub_rays(n)
        assert q in (2, 4, 8, 16, 32, 64) and q <= prev
        prev = q
