"""ctypes binding of the CPU oracle (oracle/rtxn_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of rtxn_oracle.c.  Importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
package (rtx_nerf_amd) must never import this module.  PARITY UNPINNED: the
reference holds no golden vectors and cannot be built or run here.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "rtxn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_f32_to_f16_bits.restype = C.c_uint16
        _lib.orc_f32_to_f16_bits.argtypes = [C.c_float]
        _lib.orc_f16_bits_to_f32.restype = C.c_float
        _lib.orc_f16_bits_to_f32.argtypes = [C.c_uint16]
        _lib.orc_scan_hits.restype = C.c_int
        _lib.orc_mlp_n_params.restype = C.c_long
        _lib.orc_mlp_enc_padded.restype = C.c_int
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


class MlpCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "n_pos_dims", "n_pos_freqs", "n_dir_dims", "n_dir_freqs",
        "n_neurons", "n_hidden_layers", "n_output_dims", "output_activation")]


def mlp_cfg(n_neurons=128, n_hidden_layers=8, n_pos_freqs=10, n_dir_freqs=12,
            n_pos_dims=3, n_dir_dims=2, n_output_dims=4, output_activation=1):
    """Defaults = the reference model, main.cu:35-69."""
    return MlpCfg(n_pos_dims, n_pos_freqs, n_dir_dims, n_dir_freqs, n_neurons,
                  n_hidden_layers, n_output_dims, output_activation)


def num_threads():
    return lib().orc_num_threads()


def make_ray(look_at, focal, aspect, W, H, px, py):
    la = _f32(look_at).reshape(16)
    o = np.zeros(3, np.float32)
    d = np.zeros(3, np.float32)
    v = np.zeros(2, np.float32)
    lib().orc_make_ray(_p(la), C.c_float(focal), C.c_float(aspect), C.c_uint(W), C.c_uint(H),
                       C.c_uint(px), C.c_uint(py), _p(o), _p(d), _p(v))
    return o, d, v


def trace(look_at=None, focal=1.0, aspect=1.0, W=0, H=0, R=8, occ=None, mode=0,
          ray_begin=0, ray_count=None, S=None, indices=None, rays_o=None, rays_d=None,
          want_t=True, want_seg_ray=False, count_only=False, nslots=None,
          window_chunk=0, window_stride=0):
    """Returns dict(origins, view_dirs, num_hits, start, end, t_start, t_end, seg_ray).

    Strided reference layout when indices is None (slot = ray*S + k), packed
    CSR layout otherwise (slot = indices[ray] + k)."""
    la = None if look_at is None else _f32(look_at).reshape(16)
    ro = None if rays_o is None else _f32(rays_o).reshape(-1, 3)
    rd = None if rays_d is None else _f32(rays_d).reshape(-1, 3)
    if ray_count is None:
        ray_count = (W * H if la is not None else ro.shape[0]) - ray_begin
    if S is None:
        S = 3 * R
    occ_a = None if occ is None else np.ascontiguousarray(occ, dtype=np.uint32)
    idx = _i32(indices)
    origins = np.zeros((ray_count, 3), np.float32)
    views = np.zeros((ray_count, 2), np.float32)
    nh = np.zeros(ray_count, np.int32)
    out = dict(origins=origins, view_dirs=views, num_hits=nh)
    sp = ep = t0 = t1 = sr = None
    if not count_only:
        if nslots is None:
            nslots = ray_count * S
        sp = np.full((nslots, 3), -2.0, np.float32)
        ep = np.full((nslots, 3), -2.0, np.float32)
        if want_t:
            t0 = np.full(nslots, -2.0, np.float32)
            t1 = np.full(nslots, -2.0, np.float32)
        if want_seg_ray:
            sr = np.full(nslots, -1, np.int32)
    lib().orc_trace(_p(la), _p(ro), _p(rd), C.c_float(focal), C.c_float(aspect), C.c_uint(W), C.c_uint(H),
                    C.c_int(R), _p(occ_a), C.c_int(mode), C.c_uint(ray_begin), C.c_uint(ray_count),
                    C.c_uint(window_chunk), C.c_uint(window_stride), C.c_int(S), _p(idx), _p(origins), _p(views), _p(nh), _p(sp), _p(ep), _p(t0), _p(t1), _p(sr))
    out.update(start=sp, end=ep, t_start=t0, t_end=t1, seg_ray=sr)
    return out


def trace_packed(**kw):
    """Two-pass packed trace: count -> scan -> write.  Returns the dict of
    trace() plus indices and total."""
    kw = dict(kw)
    cnt = trace(count_only=True, **kw)
    indices, total = scan_hits(cnt["num_hits"])
    out = trace(indices=indices, want_seg_ray=True, nslots=max(total, 1), **kw)
    for k in ("start", "end", "t_start", "t_end", "seg_ray"):
        if out[k] is not None:
            out[k] = out[k][:total]
    out["indices"] = indices
    out["total"] = total
    return out


def scan_hits(num_hits):
    nh = _i32(num_hits)
    idx = np.zeros_like(nh)
    total = lib().orc_scan_hits(_p(nh), _p(idx), C.c_int(nh.size))
    return idx, int(total)


def sample(start_points, end_points, view_dirs, num_hits, indices, sample_type=0, grid_res=8):
    sp, ep, vd = _f32(start_points), _f32(end_points), _f32(view_dirs)
    nh, idx = _i32(num_hits), _i32(indices)
    P = sp.reshape(-1, 3).shape[0]
    samples = np.zeros((P * 32, 5), np.float32)
    t_vals = np.zeros(P * 32, np.float32)
    lib().orc_sample(_p(sp), _p(ep), _p(vd), _p(t_vals), _p(samples), C.c_int(nh.size), C.c_int(grid_res),
                     _p(nh), _p(idx), C.c_int(sample_type))
    return samples, t_vals


def volrender_fwd(network_outputs, num_hits, indices, ray_hit, K=32):
    no, rh = _f32(network_outputs), _f32(ray_hit)
    nh, idx = _i32(num_hits), _i32(indices)
    pix = np.zeros((nh.size, 3), np.float32)
    lib().orc_volrender_fwd(_p(no), _p(nh), _p(idx), _p(rh), C.c_int(nh.size), C.c_int(K), _p(pix))
    return pix


def volrender_bwd(loss_gradients_f16, radiance, t_hit, num_hits, indices, K=32):
    lg = np.ascontiguousarray(loss_gradients_f16, dtype=np.float16)
    rad, th = _f32(radiance), _f32(t_hit)
    nh, idx = _i32(num_hits), _i32(indices)
    out = np.zeros((th.size, 4), np.float16)
    lib().orc_volrender_bwd(_p(lg), _p(rad), _p(th), _p(nh), _p(idx), C.c_int(nh.size), C.c_int(K), _p(out))
    return out


def mlp_n_params(cfg):
    return int(lib().orc_mlp_n_params(C.byref(cfg)))


def mlp_enc_padded(cfg):
    return int(lib().orc_mlp_enc_padded(C.byref(cfg)))


def freq_encode(cfg, in5):
    x = _f32(in5).reshape(-1)
    enc = np.zeros(mlp_enc_padded(cfg), np.float32)
    lib().orc_freq_encode(C.byref(cfg), _p(x), _p(enc))
    return enc


def mlp_forward(cfg, params_f16, inputs):
    p = np.ascontiguousarray(params_f16, dtype=np.float16)
    assert p.size == mlp_n_params(cfg)
    x = _f32(inputs).reshape(-1, cfg.n_pos_dims + cfg.n_dir_dims)
    out = np.zeros((x.shape[0], 16), np.float16)
    lib().orc_mlp_forward(C.byref(cfg), _p(p), _p(x), _p(out), C.c_long(x.shape[0]))
    return out


def render(look_at, focal, aspect, W, H, R, occ, trace_mode, cfg, params_f16, ray_ids):
    la = _f32(look_at).reshape(16)
    p = np.ascontiguousarray(params_f16, dtype=np.float16)
    ids = np.ascontiguousarray(ray_ids, dtype=np.uint32)
    occ_a = None if occ is None else np.ascontiguousarray(occ, dtype=np.uint32)
    pix = np.zeros((ids.size, 3), np.float32)
    tot = C.c_long(0)
    lib().orc_render(_p(la), C.c_float(focal), C.c_float(aspect), C.c_uint(W), C.c_uint(H), C.c_int(R),
                     _p(occ_a), C.c_int(trace_mode), C.byref(cfg), _p(p), _p(ids), C.c_long(ids.size),
                     _p(pix), C.byref(tot))
    return pix, int(tot.value)


def render_tiled(look_at, focal, aspect, W, H, R, occ, trace_mode, cfg, params_f16, ray_ids):
    """orc_render_tiled: the same ray march with the network evaluated 64 samples at a time by an AVX2 + FMA micro-kernel (the
    cpu_baseline leg of bench.py); agrees with render() up to the single-precision sine of the encoding."""
    la = _f32(look_at).reshape(16)
    p = np.ascontiguousarray(params_f16, dtype=np.float16)
    ids = np.ascontiguousarray(ray_ids, dtype=np.uint32)
    occ_a = None if occ is None else np.ascontiguousarray(occ, dtype=np.uint32)
    pix = np.zeros((ids.size, 3), np.float32)
    tot = C.c_long(0)
    lib().orc_render_tiled(_p(la), C.c_float(focal), C.c_float(aspect), C.c_uint(W), C.c_uint(H), C.c_int(R),
                           _p(occ_a), C.c_int(trace_mode), C.byref(cfg), _p(p), _p(ids), C.c_long(ids.size),
                           _p(pix), C.byref(tot))
    return pix, int(tot.value)


def volrender_fwd_nerf(radiance, num_hits, indices, step, K=32):
    rad, st = _f32(radiance), _f32(step)
    nh, idx = _i32(num_hits), _i32(indices)
    pix = np.zeros((nh.size, 3), np.float32)
    lib().orc_volrender_fwd_nerf(_p(rad), _p(nh), _p(idx), _p(st), C.c_int(nh.size), C.c_int(K), _p(pix))
    return pix


def volrender_bwd_nerf(loss_gradients_f16, radiance, step, num_hits, indices, K=32):
    lg = np.ascontiguousarray(loss_gradients_f16, dtype=np.float16)
    rad, st = _f32(radiance), _f32(step)
    nh, idx = _i32(num_hits), _i32(indices)
    out = np.zeros((st.size, 4), np.float32)
    lib().orc_volrender_bwd_nerf(_p(lg), _p(rad), _p(st), _p(nh), _p(idx), C.c_int(nh.size), C.c_int(K), _p(out))
    return out


# --------------------------------------------------------------------------- training path
class HgCfg(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("n_features", C.c_int), ("log2_hashmap_size", C.c_int),
                ("base_resolution", C.c_int), ("per_level_scale", C.c_float)]


def hg_cfg(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.5):
    return HgCfg(n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale)


def hg_n_params(cfg):
    lib().orc_hg_n_params.restype = C.c_long
    return int(lib().orc_hg_n_params(C.byref(cfg)))


def hg_enc_width(cfg, n_dir_freqs):
    return int(lib().orc_enc_hg_width(C.byref(cfg), C.c_int(n_dir_freqs)))


def encode_hg(cfg, n_dir_freqs, table_f16, in5):
    x = _f32(in5).reshape(-1, 5)
    t = np.ascontiguousarray(table_f16, dtype=np.float16)
    out = np.zeros((x.shape[0], hg_enc_width(cfg, n_dir_freqs)), np.float16)
    lib().orc_encode_hg(C.byref(cfg), C.c_int(n_dir_freqs), _p(t), _p(x), C.c_long(x.shape[0]), _p(out))
    return out


def hg_backward(cfg, in5, denc_f16):
    x = _f32(in5).reshape(-1, 5)
    d = np.ascontiguousarray(denc_f16, dtype=np.float16)
    out = np.zeros(hg_n_params(cfg), np.float32)
    lib().orc_hg_backward(C.byref(cfg), _p(x), C.c_long(x.shape[0]), _p(d), C.c_int(d.shape[1]), _p(out))
    return out


def encode_freq(cfg, in5):
    x = _f32(in5).reshape(-1, 5)
    out = np.zeros((x.shape[0], mlp_enc_padded(cfg)), np.float16)
    lib().orc_encode_freq(C.byref(cfg), _p(x), C.c_long(x.shape[0]), _p(out))
    return out


def mlpe_forward(W, L, out_act, params_f16, enc_f16):
    e = np.ascontiguousarray(enc_f16, dtype=np.float16)
    S, E = e.shape
    p = np.ascontiguousarray(params_f16, dtype=np.float16)
    assert p.size == W * E + (L - 1) * W * W + 16 * W
    acts = np.zeros((L, S, W), np.float16)
    out = np.zeros((S, 16), np.float16)
    lib().orc_mlpe_forward(C.c_int(W), C.c_int(L), C.c_int(E), C.c_int(out_act), _p(p), _p(e), C.c_long(S), _p(acts), _p(out))
    return acts, out


def mlpe_backward(W, L, out_act, params_f16, enc_f16, acts, out, dout_f16, want_denc=True):
    e = np.ascontiguousarray(enc_f16, dtype=np.float16)
    S, E = e.shape
    p = np.ascontiguousarray(params_f16, dtype=np.float16)
    do = np.ascontiguousarray(dout_f16, dtype=np.float16).reshape(S, 4)
    dparams = np.zeros(p.size, np.float32)
    denc = np.zeros((S, E), np.float32) if want_denc else None
    lib().orc_mlpe_backward(C.c_int(W), C.c_int(L), C.c_int(E), C.c_int(out_act), _p(p), _p(e),
                            _p(np.ascontiguousarray(acts)), _p(np.ascontiguousarray(out)), _p(do), C.c_long(S),
                            _p(dparams), _p(denc))
    return dparams, denc


def l2_loss(pred, target, scale=1.0):
    pr, tg = _f32(pred).reshape(-1), _f32(target).reshape(-1)
    values = np.zeros_like(pr)
    g16 = np.zeros(pr.size, np.float16)
    g32 = np.zeros(pr.size, np.float32)
    lib().orc_l2_loss.restype = C.c_double
    tot = lib().orc_l2_loss(_p(pr), _p(tg), C.c_long(pr.size), C.c_float(scale), _p(values), _p(g16), _p(g32))
    return float(tot), values, g16, g32


def adam_step(master, grads, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, loss_scale=1.0):
    """in-place on master/m/v (float32 arrays); returns the fp16 copy."""
    p16 = np.zeros(master.size, np.float16)
    lib().orc_adam_step(C.c_long(master.size), _p(master), _p(p16), _p(_f32(grads)), _p(m), _p(v), C.c_int(step),
                        C.c_float(lr), C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_float(loss_scale))
    return p16


def adam_step_sparse(master, grads, m, v, steps, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, loss_scale=1.0):
    """tiny-cuda-nn's Adam for hash-table entries: zero-gradient entries skipped, per-entry step counts (uint32 array `steps`);
    in-place on master/m/v/steps; returns the fp16 copy of the entries it updated (zeros elsewhere)."""
    p16 = np.zeros(master.size, np.float16)
    assert steps.dtype == np.uint32
    lib().orc_adam_step_sparse(C.c_long(master.size), _p(master), _p(p16), _p(_f32(grads)), _p(m), _p(v), _p(steps),
                               C.c_float(lr), C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_float(loss_scale))
    return p16
