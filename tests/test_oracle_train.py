"""Known-answer / independent-restatement tests pinning the TRAINING half of the
oracle (hash grid, MLP backward, L2, Adam).  tiny-cuda-nn is un-vendored and
unpinned, so these are restated from the published algorithms: PARITY UNPINNED."""
import numpy as np
import pytest

from rtx_nerf_amd import scenes


def _f16(x):
    return np.asarray(x, np.float32).astype(np.float16)


# ------------------------------------------------------------------ hash grid
def test_hashgrid_sizes_and_vertex_lookup(oracle):
    cfg = oracle.hg_cfg(n_levels=3, n_features=2, log2_hashmap_size=19, base_resolution=4, per_level_scale=2.0)
    # scale_l = 4*2^l - 1 = 3, 7, 15 ; res = 4, 8, 16 ; dense sizes 64, 512, 4096 (all < 2^19)
    assert oracle.hg_n_params(cfg) == (64 + 512 + 4096) * 2
    assert oracle.hg_enc_width(cfg, 4) == 32          # 6 hash + 16 dir = 22 -> 32
    rng = np.random.default_rng(0)
    table = _f16(rng.uniform(-1, 1, oracle.hg_n_params(cfg)))
    # level 0: pos = x01*3 + 0.5 ; x01 = 1/6 -> pos = 1.0 exactly: vertex (1,1,1), fraction 0
    x01 = 1.0 / 6.0
    xin = np.array([[2 * x01 - 1] * 3 + [0.3, -0.4]], np.float32)
    enc = oracle.encode_hg(cfg, 4, table, xin).astype(np.float32)[0]
    pos = np.float32(np.float32(xin[0, 0]) * np.float32(0.5) + np.float32(0.5)) * np.float32(3) + np.float32(0.5)
    if abs(float(pos) - 1.0) < 1e-6 and float(pos) >= 1.0:
        idx = 1 + 1 * 4 + 1 * 16
        np.testing.assert_allclose(enc[:2], table[idx * 2:idx * 2 + 2].astype(np.float32), atol=2e-3)
    # direction part: sin/cos of pi*2^f*theta then phi, then ones
    want = []
    for d in (0.3, -0.4):
        for f in range(4):
            a = np.pi * np.float32(d).astype(np.float64) * 2.0 ** f
            want += [np.sin(a), np.cos(a)]
    np.testing.assert_array_equal(enc[6:22], _f16(want).astype(np.float32))
    assert np.all(enc[22:] == 1.0)


def test_hashgrid_is_trilinear_and_hashed_levels_wrap(oracle):
    cfg = oracle.hg_cfg(n_levels=2, n_features=1, log2_hashmap_size=6, base_resolution=4, per_level_scale=2.0)
    # level 0 dense (64 <= 64), level 1: 512 > 64 -> hashed into 64 entries
    assert oracle.hg_n_params(cfg) == 128
    table = np.zeros(128, np.float16)
    table[:64] = _f16(np.arange(64) % 4)      # level-0 value = x index  -> trilinear in x reproduces a linear ramp
    # x01 <= (scale-0.5)/scale keeps the +1 corner inside the dense 4^3 level (beyond it the published
    # indexing wraps into the next row, as tcnn's does)
    xs = np.linspace(-0.9, 0.6, 7, dtype=np.float32)
    xin = np.stack([xs, np.zeros_like(xs), np.zeros_like(xs), np.zeros_like(xs), np.zeros_like(xs)], axis=1)
    enc = oracle.encode_hg(cfg, 0, table, xin).astype(np.float32)
    ramp = (xs * 0.5 + 0.5) * 3 + 0.5
    np.testing.assert_allclose(enc[:, 0], ramp, atol=2e-3)
    # backward: weights of the 8 corners sum to 1 per (sample, level, feature)
    denc = np.zeros((7, oracle.hg_enc_width(cfg, 0)), np.float16)
    denc[:, 0] = 1.0
    denc[:, 1] = 2.0
    g = oracle.hg_backward(cfg, xin, denc)
    np.testing.assert_allclose(g[:64].sum(), 7.0, rtol=1e-6)
    np.testing.assert_allclose(g[64:].sum(), 14.0, rtol=1e-6)


def test_hashgrid_backward_is_the_adjoint_of_encode(oracle):
    cfg = oracle.hg_cfg(n_levels=4, n_features=2, log2_hashmap_size=10, base_resolution=4, per_level_scale=1.7)
    rng = np.random.default_rng(1)
    n = oracle.hg_n_params(cfg)
    xin = np.concatenate([rng.uniform(-1, 1, (50, 3)), rng.uniform(-3, 3, (50, 2))], axis=1).astype(np.float32)
    # <encode(t), d> is linear in t: its gradient is hg_backward(d); compare on a random direction (fp16-exact values)
    t0 = _f16(rng.integers(-8, 9, n) / 8.0)
    dt = _f16(rng.integers(-8, 9, n) / 64.0)
    d = _f16(rng.integers(-4, 5, (50, oracle.hg_enc_width(cfg, 2))) / 4.0)
    nh = cfg.n_levels * cfg.n_features
    e0 = oracle.encode_hg(cfg, 2, t0, xin).astype(np.float64)[:, :nh]
    e1 = oracle.encode_hg(cfg, 2, (t0.astype(np.float32) + dt.astype(np.float32)).astype(np.float16), xin).astype(np.float64)[:, :nh]
    lhs = ((e1 - e0) * d.astype(np.float64)[:, :nh]).sum()
    rhs = (oracle.hg_backward(cfg, xin, d).astype(np.float64) * dt.astype(np.float64)).sum()
    assert abs(lhs - rhs) < 2e-2 * max(1.0, abs(rhs))   # fp16 rounding of the encoded outputs


# ------------------------------------------------------------------ MLP forward/backward on pre-encoded input
def _numpy_mlpe(W, L, out_act, params, enc, dout):
    p = params.astype(np.float64)
    E = enc.shape[1]
    x = enc.astype(np.float64)
    Ws, off, in_w = [], 0, E
    for l in range(L):
        Ws.append(p[off:off + W * in_w].reshape(W, in_w))
        off += W * in_w
        in_w = W
    Wo = p[off:].reshape(16, W)
    acts, h = [], x
    for l in range(L):
        h = np.maximum(h @ Ws[l].T, 0).astype(np.float16).astype(np.float64)
        acts.append(h)
    z = h @ Wo.T
    y = (1 / (1 + np.exp(-z)) if out_act else z).astype(np.float16).astype(np.float64)
    g = np.zeros_like(y)
    g[:, :4] = dout.astype(np.float64)
    dz = (g * y * (1 - y) if out_act else g).astype(np.float16).astype(np.float64)
    grads = [None] * (L + 1)
    grads[L] = dz.T @ acts[-1]
    da = dz @ Wo
    for l in range(L - 1, -1, -1):
        dz = np.where(acts[l] > 0, da, 0).astype(np.float16).astype(np.float64)
        grads[l] = dz.T @ (x if l == 0 else acts[l - 1])
        da = dz @ Ws[l]
    return acts, y, np.concatenate([g_.reshape(-1) for g_ in grads]), da


@pytest.mark.parametrize("W,L,E,act", [(64, 4, 48, 1), (128, 3, 112, 1), (64, 1, 16, 0)])
def test_mlpe_backward_matches_matrix_form(oracle, W, L, E, act):
    rng = np.random.default_rng(W + L)
    params = scenes.xavier_params_fp16(W, L, E, seed=3)
    S = 37
    enc = _f16(rng.uniform(-1, 1, (S, E)))
    dout = _f16(rng.standard_normal((S, 4)) * 0.05)
    acts, out = oracle.mlpe_forward(W, L, act, params, enc)
    dparams, denc = oracle.mlpe_backward(W, L, act, params, enc, acts, out, dout)
    n_acts, n_y, n_dp, n_denc = _numpy_mlpe(W, L, act, params, enc, dout)
    np.testing.assert_allclose(out.astype(np.float32), n_y.astype(np.float32), atol=4e-3)
    scale = np.abs(n_dp).max()
    assert scale > 0
    # same graph, float32-sequential vs float64-matrix accumulation and the resulting fp16 rounding flips
    assert np.abs(dparams - n_dp).max() < 3e-2 * scale
    assert np.linalg.norm(dparams - n_dp) < 2e-2 * np.linalg.norm(n_dp)
    assert np.linalg.norm(denc - n_denc) < 2e-2 * np.linalg.norm(n_denc)
    # rows 4..15 of the output layer receive no gradient
    assert np.all(dparams[-16 * W:].reshape(16, W)[4:] == 0)


# ------------------------------------------------------------------ loss / optimizer
def test_l2_loss_known_answer(oracle):
    pred = np.array([0.5, 0.25, 1.0, 0.0, 0.75, 0.5], np.float32)
    tgt = np.array([0.0, 0.25, 0.5, 1.0, 0.75, 0.0], np.float32)
    tot, values, g16, g32 = oracle.l2_loss(pred, tgt, scale=128.0)
    d = pred - tgt
    np.testing.assert_allclose(values, d * d / 6, rtol=1e-7)
    np.testing.assert_allclose(g32, 128.0 * 2 * d / 6, rtol=1e-7)
    assert abs(tot - (d * d).sum() / 6) < 1e-7
    np.testing.assert_array_equal(g16, g32.astype(np.float16))


def test_adam_first_steps_known_answer(oracle):
    w = np.array([1.0, -2.0, 0.5], np.float32)
    m = np.zeros(3, np.float32)
    v = np.zeros(3, np.float32)
    g = np.array([0.1, -0.2, 0.0], np.float32)
    p16 = oracle.adam_step(w, g, m, v, step=1, lr=1e-3)
    # bias-corrected first step moves every non-zero-gradient weight by ~lr against the gradient's sign
    np.testing.assert_allclose(w, [1.0 - 1e-3, -2.0 + 1e-3, 0.5], atol=2e-7)
    np.testing.assert_allclose(m, 0.1 * g, rtol=1e-6)
    np.testing.assert_allclose(v, 0.001 * g * g, rtol=1e-4)   # (1 - 0.999f) in fp32
    np.testing.assert_array_equal(p16, w.astype(np.float16))
    oracle.adam_step(w, g * 4, m, v, step=2, lr=1e-3, loss_scale=4.0)   # loss_scale divides the gradient back
    mm = 0.9 * 0.1 * g + 0.1 * g
    vv = 0.999 * 0.001 * g * g + 0.001 * g * g
    lr_eff = 1e-3 * np.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    np.testing.assert_allclose(w, np.array([1.0 - 1e-3, -2.0 + 1e-3, 0.5]) - lr_eff * mm / (np.sqrt(vv) + 1e-8), atol=3e-7)


def test_adam_sparse_known_answer(oracle):
    """tiny-cuda-nn's Adam for hash-table entries (oracle.adam_step_sparse): an entry with a zero gradient is untouched, and an
    entry's FIRST update moves it by exactly lr * sign(g) whenever that happens (bias correction by its own count), where the
    dense rule at global step 2 would move it by lr * 0.1 / sqrt(0.001) * sqrt(1 - 0.999^2) / (1 - 0.9^2) != lr."""
    w = np.array([1.0, 2.0, 3.0, -1.0], np.float32)
    m, v = np.zeros(4, np.float32), np.zeros(4, np.float32)
    st = np.zeros(4, np.uint32)
    oracle.adam_step_sparse(w, np.array([0.0, 0.5, 0.0, -0.0], np.float32), m, v, st, lr=1e-2, eps=0.0)
    np.testing.assert_allclose(w, [1.0, 1.99, 3.0, -1.0], rtol=0, atol=1e-7)
    assert st.tolist() == [0, 1, 0, 0] and m[0] == 0 and v[0] == 0
    p16 = oracle.adam_step_sparse(w, np.array([-0.25 * 8, 0.5 * 8, 0.0, 0.0], np.float32), m, v, st, lr=1e-2, eps=0.0, loss_scale=8.0)
    assert st.tolist() == [1, 2, 0, 0]
    assert abs(w[0] - 1.01) < 1e-7                        # first update of entry 0, at the optimizer's second call
    # entry 1, second update with the same gradient: m_hat = v_hat^(1/2) = 0.5 -> moves by lr again
    assert abs(w[1] - 1.98) < 2e-7
    assert w[2] == 3.0 and w[3] == -1.0 and m[2] == 0 and v[3] == 0
    assert p16[0] == np.float16(w[0]) and p16[1] == np.float16(w[1])
