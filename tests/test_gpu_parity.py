"""GPU parity: every HIP kernel, called through the C ABI (rtx_nerf_amd.api ->
librtxn.so), against the CPU oracle on the same seeded inputs.

Stated bars (the oracle itself is "parity unpinned", see its header):
  traversal  num_hits, start/end points, t: bit-exact; theta/phi: 2e-6 (atan2f differs between libms)
  scan       bit-exact
  sampler    bit-exact, all three modes
  volrender  forward 1e-5 abs (wave scan sums in a different order; device expf);
             backward: fp16 outputs within 1 fp16 ulp (device expf vs glibc)
  MLP        fp16 sigmoid outputs within 1e-2 abs, mean abs error < 1e-3 (MFMA summation order)
"""
import numpy as np
import pytest

from rtx_nerf_amd import scenes

pytestmark = pytest.mark.gpu


def _dev(torch, a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _occ_dev(torch, words):
    return torch.from_numpy(words.view(np.int32).copy()).cuda()


# ------------------------------------------------------------------ scan
@pytest.mark.parametrize("n", [0, 1, 63, 4096, 4097, 640000, 5_000_001])
def test_scan_hits(gpu, oracle, n):
    torch = gpu
    from rtx_nerf_amd import api
    rng = np.random.default_rng(n)
    nh = rng.integers(0, 40, n).astype(np.int32)
    idx, total = api.scan_hits(_dev(torch, nh))
    want_idx, want_total = oracle.scan_hits(nh)
    assert int(total.item()) == want_total
    np.testing.assert_array_equal(idx.cpu().numpy(), want_idx)


# ------------------------------------------------------------------ traversal
def _trace_gpu(torch, api, *, R, mode, look_at=None, f=1.0, W=0, H=0, rays_o=None, rays_d=None, occ=None,
               coarse=None, ray_begin=0, ray_count=None, S=None, bricks=None):
    n_all = W * H if look_at is not None else rays_o.shape[0]
    n = n_all - ray_begin if ray_count is None else ray_count
    S = 3 * R if S is None else S
    out = dict(
        origins=torch.full((n, 3), -2.0, device="cuda"), view_dirs=torch.full((n, 2), -2.0, device="cuda"),
        num_hits=torch.zeros(n, dtype=torch.int32, device="cuda"),
        start=torch.full((n * S, 3), -2.0, device="cuda"), end=torch.full((n * S, 3), -2.0, device="cuda"),
        t_start=torch.full((n * S,), -2.0, device="cuda"), t_end=torch.full((n * S,), -2.0, device="cuda"))
    api.trace_grid(None if look_at is None else _dev(torch, look_at.reshape(16)), f, 1.0, W, H, grid_res=R,
                   rays_o=None if rays_o is None else _dev(torch, rays_o),
                   rays_d=None if rays_d is None else _dev(torch, rays_d),
                   ray_begin=ray_begin, ray_count=n, occupancy=occ, occupancy_coarse=coarse, occupancy_bricks=bricks, mode=mode,
                   ray_origins=out["origins"], viewing_direction=out["view_dirs"], num_hits=out["num_hits"],
                   intersection_arr_size=S, start_points=out["start"], end_points=out["end"],
                   t_start=out["t_start"], t_end=out["t_end"])
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def _assert_trace_equal(got, want):
    np.testing.assert_array_equal(got["num_hits"], want["num_hits"])
    np.testing.assert_array_equal(got["origins"], want["origins"])
    np.testing.assert_allclose(got["view_dirs"], want["view_dirs"], rtol=0, atol=2e-6)
    for k in ("start", "end", "t_start", "t_end"):
        np.testing.assert_array_equal(got[k], want[k].reshape(got[k].shape), err_msg=k)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("R,W,H,pose", [(8, 32, 32, (30.0, -30.0, 10.0)), (8, 32, 32, (0.0, 0.0, 1.0)),
                                        (32, 40, 24, (120.0, -60.0, 10.0)), (12, 17, 9, (75.0, -10.0, 10.0))])
def test_trace_dense_bit_exact(gpu, oracle, mode, R, W, H, pose):
    torch = gpu
    from rtx_nerf_amd import api
    la = scenes.pose_spherical(pose[0], pose[1], origin_scale=pose[2])   # scale 1: camera inside the grid (quirk Q2)
    f = scenes.lego_focal_length(True)
    want = oracle.trace(look_at=la, focal=f, aspect=1.0, W=W, H=H, R=R, mode=mode)
    got = _trace_gpu(torch, api, R=R, mode=mode, look_at=la, f=f, W=W, H=H)
    assert want["num_hits"].max() > 0
    _assert_trace_equal(got, want)


def test_trace_q1_literal_focal(gpu, oracle):
    torch = gpu
    from rtx_nerf_amd import api
    la = scenes.pose_spherical(30.0, -30.0, origin_scale=1.0)
    f = scenes.lego_focal_length(False)   # negative: rays point backwards, still must agree
    want = oracle.trace(look_at=la, focal=f, aspect=1.0, W=32, H=32, R=8, mode=0)
    got = _trace_gpu(torch, api, R=8, mode=0, look_at=la, f=f, W=32, H=32)
    _assert_trace_equal(got, want)


@pytest.mark.parametrize("mode", [0, 1])
def test_trace_explicit_rays_edge_cases(gpu, oracle, mode):
    torch = gpu
    from rtx_nerf_amd import api
    s3 = np.float32(1 / np.sqrt(3))
    o = np.array([[-2.0, 0.1, 0.1], [-2.0, 1.5, 0.0], [0.05, 0.05, 0.05], [-2.0, -2.0, -2.0], [2.0, 0.1, 0.1],
                  [0.0, 0.0, 3.0], [-1.0, -1.0, -1.0]], np.float32)
    d = np.array([[1, 0, 0], [1, 0, 0], [1, 0, 0], [s3, s3, s3], [1, 0, 0], [0, 0, -1], [s3, s3, s3]], np.float32)
    rng = np.random.default_rng(11)
    ro = rng.uniform(-3, 3, (500, 3)).astype(np.float32)
    rd = rng.standard_normal((500, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    o, d = np.concatenate([o, ro]), np.concatenate([d, rd.astype(np.float32)])
    want = oracle.trace(rays_o=o, rays_d=d, R=16, mode=mode)
    got = _trace_gpu(torch, api, R=16, mode=mode, rays_o=o, rays_d=d)
    assert got["num_hits"][0] == 16 and got["num_hits"][1] == 0 and got["num_hits"][3] == 16
    _assert_trace_equal(got, want)


@pytest.mark.parametrize("use_coarse", [False, True])
@pytest.mark.parametrize("R", [16, 64])
def test_trace_occupancy_and_hierarchical_skip(gpu, oracle, R, use_coarse):
    """The two-level DDA (coarse mip in LDS) must reproduce the oracle's FLAT walk bit for bit."""
    torch = gpu
    from rtx_nerf_amd import api
    dense = scenes.lego_standin_density(R, seed=1)
    words = scenes.pack_occupancy(dense)
    occ = _occ_dev(torch, words)
    coarse = api.build_occupancy_mip(occ, R) if use_coarse else None
    bricks = None
    if use_coarse:
        want_c = scenes.pack_occupancy(scenes.coarse_occupancy(dense))
        np.testing.assert_array_equal(coarse.cpu().numpy().view(np.uint32), want_c)
        bricks = api.build_occupancy_bricks(occ, R)
        rc = R // 4
        b = dense.reshape(rc, 4, rc, 4, rc, 4).transpose(0, 2, 4, 1, 3, 5).reshape(rc ** 3, 64)   # [block][x&3, y&3, z&3]
        want_b = (b.astype(np.uint64) << np.arange(64, dtype=np.uint64)).sum(axis=1, dtype=np.uint64)
        np.testing.assert_array_equal(bricks.cpu().numpy().view(np.uint64), want_b)
    la = scenes.pose_spherical(50.0, -35.0, origin_scale=10.0)
    f = scenes.lego_focal_length(True)
    for mode in ([1] if use_coarse else [0, 1]):
        want = oracle.trace(look_at=la, focal=f, aspect=1.0, W=48, H=40, R=R, occ=words, mode=mode, S=R)
        got = _trace_gpu(torch, api, R=R, mode=mode, look_at=la, f=f, W=48, H=40, occ=occ, coarse=coarse, S=R)
        assert 0 < want["num_hits"].sum()
        _assert_trace_equal(got, want)
        if bricks is not None:
            got = _trace_gpu(torch, api, R=R, mode=mode, look_at=la, f=f, W=48, H=40, occ=occ, coarse=coarse, S=R, bricks=bricks)
            _assert_trace_equal(got, want)


@pytest.mark.parametrize("sub_rays", [0, 2, 8, 16, 32, 64])
def test_trace_window_and_packed_two_pass(gpu, oracle, sub_rays):
    """Ray window (the multi-GPU shard) + count -> scan -> write into the packed CSR layout; with sub_rays = Q the ray
    is walked by Q lanes in consecutive pieces and must give the same segments, in the same order, bit for bit."""
    torch = gpu
    from rtx_nerf_amd import api
    R, W, H = 32, 64, 48
    dense = scenes.lego_standin_density(R, seed=2)
    words = scenes.pack_occupancy(dense)
    occ = _occ_dev(torch, words)
    coarse = api.build_occupancy_mip(occ, R)
    la = scenes.pose_spherical(-40.0, -20.0, origin_scale=10.0)
    f = scenes.lego_focal_length(True)
    begin, count = 64 * 10 + 7, 64 * 21 + 5
    want = oracle.trace_packed(look_at=la, focal=f, aspect=1.0, W=W, H=H, R=R, occ=words, mode=1,
                               ray_begin=begin, ray_count=count)
    la_d = _dev(torch, la.reshape(16))
    nh = torch.zeros(count, dtype=torch.int32, device="cuda")
    vd = torch.zeros((count, 2), device="cuda")
    sub = torch.zeros(count * max(sub_rays, 1), dtype=torch.int32, device="cuda")
    kw = dict(grid_res=R, ray_begin=begin, ray_count=count, occupancy=occ, occupancy_coarse=coarse, mode=1,
              viewing_direction=vd, num_hits=nh, sub_rays=sub_rays, sub_hits=sub)
    api.trace_grid(la_d, f, 1.0, W, H, **kw)
    idx, total = api.scan_hits(nh)
    P = int(total.item())
    assert P == want["total"] > 0
    cap = P - 3   # capacity guard: the last 3 slots must stay untouched
    sp = torch.full((P, 3), -2.0, device="cuda")
    ep = torch.full((P, 3), -2.0, device="cuda")
    sr = torch.full((P,), -1, dtype=torch.int32, device="cuda")
    sv = torch.full((P, 2), -9.0, device="cuda")
    sf = torch.full((P,), 7, dtype=torch.uint8, device="cuda")
    stored = torch.zeros(count, dtype=torch.int32, device="cuda")
    api.trace_grid(la_d, f, 1.0, W, H, indices=idx, start_points=sp, end_points=ep, seg_ray=sr, seg_view=sv, seg_first=sf,
                   num_stored=stored, segment_capacity=cap, **kw)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(nh.cpu().numpy(), want["num_hits"])
    first = np.zeros(P, np.uint8)
    first[want["indices"][want["num_hits"] > 0]] = 1
    np.testing.assert_array_equal(sf.cpu().numpy()[:cap], first[:cap])
    np.testing.assert_array_equal(stored.cpu().numpy(), np.clip(cap - want["indices"], 0, want["num_hits"]))
    np.testing.assert_array_equal(idx.cpu().numpy(), want["indices"])
    np.testing.assert_array_equal(sp.cpu().numpy()[:cap], want["start"][:cap])
    np.testing.assert_array_equal(ep.cpu().numpy()[:cap], want["end"][:cap])
    np.testing.assert_array_equal(sr.cpu().numpy()[:cap], want["seg_ray"][:cap])
    np.testing.assert_array_equal(sv.cpu().numpy()[:cap], vd.cpu().numpy()[want["seg_ray"][:cap]])
    assert np.all(sp.cpu().numpy()[cap:] == -2.0) and np.all(sr.cpu().numpy()[cap:] == -1)
    assert np.all(sv.cpu().numpy()[cap:] == -9.0)


# ------------------------------------------------------------------ sampler
def _ragged_segments(rng, B, max_hits):
    nh = rng.integers(0, max_hits + 1, B).astype(np.int32)
    nh[rng.integers(0, B, max(1, B // 5))] = 0
    idx = np.concatenate([[0], np.cumsum(nh)[:-1]]).astype(np.int32)
    P = int(nh.sum())
    sp = rng.uniform(-1, 1, (max(P, 1), 3)).astype(np.float32)
    ep = rng.uniform(-1, 1, (max(P, 1), 3)).astype(np.float32)
    vd = rng.uniform(-3.1, 3.1, (B, 2)).astype(np.float32)
    return nh, idx, P, sp, ep, vd


@pytest.mark.parametrize("sample_type", [0, 1, 2, 3])
@pytest.mark.parametrize("B,max_hits", [(64, 7), (1, 1), (1000, 46), (5, 0)])
def test_sampler_bit_exact(gpu, oracle, sample_type, B, max_hits):
    torch = gpu
    from rtx_nerf_amd import api
    rng = np.random.default_rng(B * 10 + sample_type)
    nh, idx, P, sp, ep, vd = _ragged_segments(rng, B, max_hits)
    samples = torch.full((max(P, 1) * 32, 5), -7.0, device="cuda")
    t_vals = torch.full((max(P, 1) * 32,), -7.0, device="cuda")
    api.launchSampler(_dev(torch, sp), _dev(torch, ep), _dev(torch, vd), t_vals, samples, B, 8,
                      _dev(torch, nh), _dev(torch, idx), sample_type)
    torch.cuda.synchronize()
    ws, wt = oracle.sample(sp[:P], ep[:P], vd, nh, idx, sample_type)
    np.testing.assert_array_equal(samples.cpu().numpy()[:P * 32], ws)
    np.testing.assert_array_equal(t_vals.cpu().numpy()[:P * 32], wt)
    if P == 0:
        assert np.all(samples.cpu().numpy() == -7.0)


# ------------------------------------------------------------------ volume rendering
def _vr_inputs(rng, B, max_hits, K):
    nh, idx, P, _, _, _ = _ragged_segments(rng, B, max_hits)
    N = max(P, 1) * K
    rad = rng.uniform(0, 1, (N, 4)).astype(np.float32)
    t = np.tile(((np.arange(K) + 1) / K).astype(np.float32), max(P, 1))
    return nh, idx, P, rad, t


@pytest.mark.parametrize("B,max_hits,K", [(64, 7, 32), (1000, 46, 32), (3, 0, 32), (200, 5, 7), (50, 300, 32)])
def test_volrender_fwd_compat(gpu, oracle, B, max_hits, K):
    torch = gpu
    from rtx_nerf_amd import api
    rng = np.random.default_rng(B + K)
    nh, idx, P, rad, t = _vr_inputs(rng, B, max_hits, K)
    pix = torch.full((B, 3), -1.0, device="cuda")
    api.launch_volrender_cuda(None, _dev(torch, rad), _dev(torch, nh), _dev(torch, idx), _dev(torch, t), B, K, pix)
    torch.cuda.synchronize()
    want = oracle.volrender_fwd(rad, nh, idx, t, K=K)
    np.testing.assert_allclose(pix.cpu().numpy(), want, rtol=0, atol=1e-5)
    assert np.all(pix.cpu().numpy()[nh == 0] == 0.0)


def test_volrender_fwd_jittered_t(gpu, oracle):
    """t_vals of the other sampler modes (non-monotone across segment boundaries)."""
    torch = gpu
    from rtx_nerf_amd import api
    rng = np.random.default_rng(9)
    nh, idx, P, rad, _ = _vr_inputs(rng, 300, 9, 32)
    t = rng.uniform(0, 1, max(P, 1) * 32).astype(np.float32)
    pix = torch.zeros((300, 3), device="cuda")
    api.launch_volrender_cuda(None, _dev(torch, rad), _dev(torch, nh), _dev(torch, idx), _dev(torch, t), 300, 32, pix)
    want = oracle.volrender_fwd(rad, nh, idx, t)
    np.testing.assert_allclose(pix.cpu().numpy(), want, rtol=0, atol=1e-5)


def _half_ulp_close(got_f16, want_f16, ulps=1):
    a = got_f16.view(np.int16).astype(np.int32)
    b = want_f16.view(np.int16).astype(np.int32)
    a = np.where(a < 0, -(a & 0x7fff), a)
    b = np.where(b < 0, -(b & 0x7fff), b)
    return a.size == 0 or np.abs(a - b).max() <= ulps


@pytest.mark.parametrize("B,max_hits,K", [(64, 7, 32), (1000, 30, 32), (3, 0, 32), (100, 5, 7)])
def test_volrender_bwd_compat(gpu, oracle, B, max_hits, K):
    torch = gpu
    from rtx_nerf_amd import api
    rng = np.random.default_rng(B + 3 * K)
    nh, idx, P, rad, t = _vr_inputs(rng, B, max_hits, K)
    g = rng.standard_normal((B, 3)).astype(np.float16)
    out = torch.zeros((max(P, 1) * K, 4), dtype=torch.float16, device="cuda")
    api.launch_volrender_backward_cuda(None, _dev(torch, g), _dev(torch, rad), _dev(torch, t), _dev(torch, nh),
                                       _dev(torch, idx), B, K, out)
    torch.cuda.synchronize()
    want = oracle.volrender_bwd(g, rad, t, nh, idx, K=K)
    got = out.cpu().numpy()[:P * K]
    assert _half_ulp_close(got, want[:P * K])
    assert P == 0 or (got.view(np.uint16) == want[:P * K].view(np.uint16)).mean() > 0.995


def test_volrender_nerf_mode(gpu, oracle):
    torch = gpu
    from rtx_nerf_amd import api
    rng = np.random.default_rng(21)
    B, K = 400, 32
    nh, idx, P, rad, _ = _vr_inputs(rng, B, 12, K)
    rad[:, 3] *= 20.0
    step = rng.uniform(0.001, 0.02, max(P, 1) * K).astype(np.float32)
    g = rng.standard_normal((B, 3)).astype(np.float16)
    pix = torch.zeros((B, 3), device="cuda")
    api.launch_volrender_cuda(None, _dev(torch, rad), _dev(torch, nh), _dev(torch, idx), _dev(torch, step), B, K, pix,
                              mode=api.VR_NERF)
    np.testing.assert_allclose(pix.cpu().numpy(), oracle.volrender_fwd_nerf(rad, nh, idx, step, K=K), rtol=0, atol=2e-5)
    out = torch.zeros((max(P, 1) * K, 4), dtype=torch.float16, device="cuda")
    api.launch_volrender_backward_cuda(None, _dev(torch, g), _dev(torch, rad), _dev(torch, step), _dev(torch, nh),
                                       _dev(torch, idx), B, K, out, mode=api.VR_NERF)
    want = oracle.volrender_bwd_nerf(g, rad, step, nh, idx, K=K)[:P * K]
    got = out.cpu().numpy()[:P * K].astype(np.float32)
    # fp16 output: half an ulp relative (2^-11) plus the fp32-vs-double suffix-sum noise
    np.testing.assert_allclose(got, want, rtol=1.5e-3, atol=2e-5)


# ------------------------------------------------------------------ MLP
@pytest.fixture
def mfma_shape():
    """The fused inference kernels run on v_mfma_f32_16x16x32_f16 (rounds 1-2 also carried 32x32x16 builds; removed)."""
    return "16"


def _mlp_case(oracle, W, nh, n, seed, dir_freqs=12):
    cfg = oracle.mlp_cfg(n_neurons=W, n_hidden_layers=nh, n_dir_freqs=dir_freqs)
    params = scenes.xavier_params_fp16(W, nh, oracle.mlp_enc_padded(cfg), seed=seed)
    rng = np.random.default_rng(seed)
    x = np.concatenate([rng.uniform(-1, 1, (n, 3)), rng.uniform(0, 3.1416, (n, 1)), rng.uniform(-3.1416, 3.1416, (n, 1))],
                       axis=1).astype(np.float32)
    return cfg, params, x


@pytest.mark.parametrize("W,nh,n,dir_freqs", [(128, 8, 1000, 12), (64, 2, 777, 12), (128, 1, 256, 12), (64, 4, 3, 12),
                                              (128, 3, 515, 4), (64, 3, 300, 4), (256, 8, 900, 12), (256, 1, 33, 12),
                                              (256, 2, 256, 12), (256, 3, 2500, 12)])
def test_mlp_forward_half_output(gpu, oracle, W, nh, n, dir_freqs, mfma_shape):
    torch = gpu
    from rtx_nerf_amd import api
    cfg, params, x = _mlp_case(oracle, W, nh, n, seed=W + nh, dir_freqs=dir_freqs)
    net = api.Network(n_neurons=W, n_hidden_layers=nh, n_dir_freqs=dir_freqs)
    assert net.n_params() == params.size
    net.set_params(_dev(torch, params))
    out = torch.full((n, 16), -5.0, dtype=torch.float16, device="cuda")
    net.forward(_dev(torch, x), out)
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float32)
    want = oracle.mlp_forward(cfg, params, x).astype(np.float32)
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-2)
    assert np.abs(got - want).mean() < 1e-3
    assert want[:, :4].std() > 0.01


def test_mlp_forward_radiance_and_no_activation(gpu, oracle, mfma_shape):
    torch = gpu
    from rtx_nerf_amd import api
    cfg, params, x = _mlp_case(oracle, 128, 8, 2049, seed=5)
    net = api.Network()
    net.set_params(_dev(torch, params))
    rad = net.forward_radiance(_dev(torch, x)).cpu().numpy()
    want = oracle.mlp_forward(cfg, params, x).astype(np.float32)[:, :4]
    np.testing.assert_allclose(rad, want, rtol=0, atol=1e-2)
    # radiance is the fp16 network output widened to fp32 (convertHalfToFloat): exactly fp16-representable
    np.testing.assert_array_equal(rad, rad.astype(np.float16).astype(np.float32))
    cfg2 = oracle.mlp_cfg(output_activation=0)
    net2 = api.Network(output_activation=api.ACT_NONE)
    net2.set_params(_dev(torch, params))
    out = net2.forward(_dev(torch, x[:300])).cpu().numpy().astype(np.float32)
    want2 = oracle.mlp_forward(cfg2, params, x[:300]).astype(np.float32)
    np.testing.assert_allclose(out, want2, rtol=0, atol=2e-2)


def test_mlp_identity_weights_expose_layouts(gpu, oracle, mfma_shape):
    """Structured weights (one 1.0 per row at an asymmetric position) make the output an exact copy
    of chosen encoding features: catches any transposed/permuted MFMA fragment map outright."""
    torch = gpu
    from rtx_nerf_amd import api
    W, nh, P = 128, 3, 112
    cfg = oracle.mlp_cfg(n_neurons=W, n_hidden_layers=nh, output_activation=0)
    p = np.zeros(W * P + (nh - 1) * W * W + 16 * W, np.float16)
    w0 = p[:W * P].reshape(W, P)
    perm0 = (np.arange(W) * 37 + 5) % P            # row r of layer 0 copies encoding feature perm0[r]
    w0[np.arange(W), perm0] = 1
    off = W * P
    perms = []
    for l in range(nh - 1):
        wl = p[off:off + W * W].reshape(W, W)
        pl = (np.arange(W) * 29 + 11 * (l + 1)) % W
        wl[np.arange(W), pl] = 1
        perms.append(pl)
        off += W * W
    wo = p[off:].reshape(16, W)
    po = (np.arange(16) * 7 + 3) % W
    wo[np.arange(16), po] = 1
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-1, 1, (512, 3)), rng.uniform(-3, 3, (512, 2))], axis=1).astype(np.float32)
    net = api.Network(n_neurons=W, n_hidden_layers=nh, output_activation=api.ACT_NONE)
    net.set_params(_dev(torch, p))
    got = net.forward(_dev(torch, x)).cpu().numpy().astype(np.float32)
    want = oracle.mlp_forward(cfg, p, x).astype(np.float32)
    # copies of relu(fp16 feature): exact up to the v_sin_f32-vs-libm rounding of the feature itself
    np.testing.assert_allclose(got, want, rtol=0, atol=1.5e-3)
    assert (got == want).mean() > 0.97
    src = perm0
    for pl in perms:
        src = src[pl]
    src = src[po]
    assert len(set(src.tolist())) > 8 and want.std() > 0.1


@pytest.mark.parametrize("W", [64, 128, 256])
def test_mlp_forward_segments_equals_sampler_plus_forward(gpu, oracle, W, mfma_shape):
    torch = gpu
    from rtx_nerf_amd import api
    rng = np.random.default_rng(4)
    B = 700
    nh, idx, P, sp, ep, vd = _ragged_segments(rng, B, 9)
    seg_ray = np.repeat(np.arange(B, dtype=np.int32), nh)
    cfg, params, _ = _mlp_case(oracle, W, 8, 1, seed=9)
    net = api.Network(n_neurons=W)
    net.set_params(_dev(torch, params))
    cap = P + 13
    rad = torch.full((cap * 32, 4), -3.0, device="cuda")
    tv = torch.full((cap * 32,), -3.0, device="cuda")
    sp_d = _dev(torch, np.concatenate([sp[:P], np.zeros((13, 3), np.float32)]))
    ep_d = _dev(torch, np.concatenate([ep[:P], np.zeros((13, 3), np.float32)]))
    sv_d = _dev(torch, np.concatenate([vd[seg_ray], np.zeros((13, 2), np.float32)]))
    total = torch.tensor([P], dtype=torch.int32, device="cuda")
    net.forward_segments(sp_d, ep_d, sv_d, total, cap, rad, tv)
    torch.cuda.synchronize()
    # same samples as the standalone sampler (bit-exact), then the same network
    samples, t_vals = oracle.sample(sp[:P], ep[:P], vd, nh, idx, 0)
    s_d = torch.zeros((P * 32, 5), device="cuda")
    t_d = torch.zeros((P * 32,), device="cuda")
    api.launchSampler(sp_d, ep_d, _dev(torch, vd), t_d, s_d, B, 8, _dev(torch, nh), _dev(torch, idx), 0)
    rad2 = net.forward_radiance(s_d).cpu().numpy()
    got = rad.cpu().numpy()
    np.testing.assert_array_equal(got[:P * 32], rad2)                  # fused == staged, bit for bit
    np.testing.assert_array_equal(tv.cpu().numpy()[:P * 32], t_vals)
    assert np.all(got[P * 32:] == -3.0) and np.all(tv.cpu().numpy()[P * 32:] == -3.0)   # nothing beyond *total
    want = oracle.mlp_forward(cfg, params, samples).astype(np.float32)[:, :4]
    np.testing.assert_allclose(got[:P * 32], want, rtol=0, atol=1e-2)


# ------------------------------------------------------------------ end-to-end render (BASELINE configs)
@pytest.mark.parametrize("variant,trace_mode", [("dense", 0), ("sphere", 0), ("sphere", 1)])
def test_config1_render_matches_host_ray_march(gpu, oracle, variant, trace_mode):
    """configs[0] (32^3 grid, 2x64 MLP, 1024 rays) rendered by the HIP pipeline == the host-CPU ray march."""
    torch = gpu
    from rtx_nerf_amd import api, render
    R, W, H = 32, 32, 32
    cfg = oracle.mlp_cfg(n_neurons=64, n_hidden_layers=2)
    params = scenes.xavier_params_fp16(64, 2, oracle.mlp_enc_padded(cfg), seed=1337)
    words = None if variant == "dense" else scenes.pack_occupancy(scenes.sphere_density(R, 0.5))
    occ = None if words is None else _occ_dev(torch, words)
    net = api.Network(n_neurons=64, n_hidden_layers=2)
    net.set_params(_dev(torch, params))
    la = scenes.pose_spherical(30.0, -30.0, origin_scale=10.0)
    f = scenes.lego_focal_length(True)
    pipe = render.RenderPipeline(net, R, W, H, f, occupancy=occ, max_segments=W * H * 100, trace_mode=trace_mode)
    pipe.set_pose(la)
    pix = pipe.render().cpu().numpy()
    assert not pipe.overflowed()
    want, nsamp = oracle.render(la, f, 1.0, W, H, R, words, trace_mode, cfg, params, np.arange(W * H))
    assert nsamp == int(pipe.total.item()) * 32
    np.testing.assert_allclose(pix, want, rtol=0, atol=2e-3)
    mse = float(((pix - want) ** 2).mean())
    assert 10 * np.log10(1.0 / max(mse, 1e-20)) > 60.0


def test_render_capacity_overflow_is_safe(gpu, oracle):
    """A segment buffer that is too small truncates rays on the device and reports it; nothing is written out of bounds."""
    torch = gpu
    from rtx_nerf_amd import api, render
    R, W, H = 32, 32, 32
    net = api.Network(n_neurons=64, n_hidden_layers=2)
    net.set_params(_dev(torch, scenes.xavier_params_fp16(64, 2, 112, seed=1)))
    la = scenes.pose_spherical(30.0, -30.0, origin_scale=10.0)
    pipe = render.RenderPipeline(net, R, W, H, scenes.lego_focal_length(True), max_segments=2000)
    guard = pipe.radiance.numel()
    pipe.set_pose(la)
    pix = pipe.render().cpu().numpy()
    assert pipe.overflowed() and np.isfinite(pix).all()
    assert pipe.radiance.numel() == guard
    need = pipe.calibrate([la])
    assert pipe.max_segments >= need
    pix2 = pipe.render().cpu().numpy()
    assert not pipe.overflowed() and np.abs(pix2).sum() > np.abs(pix).sum()


def test_config5_forward_facing_256_grid_8x256(gpu, oracle):
    """BASELINE configs[4], reduced image: forward-facing frustum, 256^3 sparse grid (coarse 64^3 mip exactly
    fills the 32-KiB LDS stage), 8x256 MLP -- HIP pipeline vs the host ray march."""
    torch = gpu
    from rtx_nerf_amd import api, render
    R, W, H = 256, 64, 48
    dense = scenes.llff_standin_density(R, seed=3)
    assert 0.005 < dense.mean() < 0.05
    words = scenes.pack_occupancy(dense)
    occ = _occ_dev(torch, words)
    cfg = oracle.mlp_cfg(n_neurons=256, n_hidden_layers=8)
    params = scenes.xavier_params_fp16(256, 8, oracle.mlp_enc_padded(cfg), seed=5)
    assert params.size == 491520
    net = api.Network(n_neurons=256, n_hidden_layers=8)
    net.set_params(_dev(torch, params))
    la = scenes.pose_forward_facing(0.15, -0.1)
    f = 1.6
    pipe = render.RenderPipeline(net, R, W, H, f, occupancy=occ, max_segments=W * H * 64)
    pipe.set_pose(la)
    pix = pipe.render().cpu().numpy()
    assert not pipe.overflowed()
    want, nsamp = oracle.render(la, f, W / H, W, H, R, words, 1, cfg, params, np.arange(W * H))
    assert nsamp == int(pipe.total.item()) * 32 and nsamp > 20000
    np.testing.assert_allclose(pix, want, rtol=0, atol=3e-3)
    mse = float(((pix - want) ** 2).mean())
    assert 10 * np.log10(1.0 / max(mse, 1e-20)) > 55.0
    # the two-level walk and the flat walk agree on this grid too (bit-exact segments)
    tr_c = oracle.trace(look_at=la, focal=f, aspect=W / H, W=W, H=H, R=R, occ=words, mode=1, count_only=True)
    np.testing.assert_array_equal(pipe.num_hits.cpu().numpy(), tr_c["num_hits"])


@pytest.mark.parametrize("mode", [0, 1])
def test_trace_maximum_grid_resolution(gpu, oracle, mode):
    """R = 1024 (the largest grid the ABI accepts), dense: up to 3R-2 = 3070 segments per ray, strided layout."""
    torch = gpu
    from rtx_nerf_amd import api
    R = 1024
    rng = np.random.default_rng(77)
    ro = rng.uniform(-2.5, 2.5, (48, 3)).astype(np.float32)
    tgt = rng.uniform(-0.9, 0.9, (48, 3)).astype(np.float32)
    rd = tgt - ro
    rd = (rd / np.linalg.norm(rd, axis=1, keepdims=True)).astype(np.float32)
    want = oracle.trace(rays_o=ro, rays_d=rd, R=R, mode=mode)
    got = _trace_gpu(torch, api, R=R, mode=mode, rays_o=ro, rays_d=rd)
    assert want["num_hits"].max() > R and want["num_hits"].max() <= 3 * R - 2
    _assert_trace_equal(got, want)


@pytest.mark.parametrize("mode", [0, 1])
def test_trace_axis_parallel_poses(gpu, oracle, mode):
    torch = gpu
    from rtx_nerf_amd import api
    for th, ph in [(0.0, -90.0), (90.0, 0.0), (180.0, -90.0)]:
        la = scenes.pose_spherical(th, ph, radius=2.236169, origin_scale=10.0)
        want = oracle.trace(look_at=la, focal=1.7, aspect=13 / 39, W=13, H=39, R=4, mode=mode)
        got = _trace_gpu(torch, api, R=4, mode=mode, look_at=la, f=1.7, W=13, H=39)
        # aspect differs from _trace_gpu's 1.0: recompute the oracle with aspect 1.0 for an exact comparison
        want = oracle.trace(look_at=la, focal=1.7, aspect=1.0, W=13, H=39, R=4, mode=mode)
        assert np.isfinite(got["start"][got["start"] != -2.0]).all()
        _assert_trace_equal(got, want)


def test_encoder_error_per_octave(gpu, oracle, mfma_shape, capsys):
    """The encoded features themselves, read out through selection weights (layer 0 copies four features per pass, the
    output layer passes them on), against the oracle's double-precision sin/cos rounded to fp16 -- per octave, because the
    16x16x32 kernel derives two of every three octaves by angle doubling from one v_sin_f32 / v_cos_f32 seed.  Bar: every
    feature within 1 fp16 ulp of the oracle's (5e-4 at these magnitudes), < 2 % of them different at all."""
    torch = gpu
    from rtx_nerf_amd import api
    W, E, n = 128, 112, 4096
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-1, 1, (n, 3)), rng.uniform(0, 3.1416, (n, 1)), rng.uniform(-3.1416, 3.1416, (n, 1))], 1).astype(np.float32)
    cfg = oracle.mlp_cfg(n_neurons=W, n_hidden_layers=1, output_activation=0)
    want = oracle.encode_freq(cfg, x).astype(np.float32)                    # [n][112], tcnn feature order
    net = api.Network(n_neurons=W, n_hidden_layers=1, output_activation=api.ACT_NONE)
    got = np.zeros((n, E), np.float32)
    x_d = _dev(torch, x)
    for f0 in range(0, E, 4):
        p = np.zeros(W * E + 16 * W, np.float16)
        w0, wo = p[:W * E].reshape(W, E), p[W * E:].reshape(16, W)
        for k in range(4):
            w0[5 + 9 * k, f0 + k] = 1          # a feature can be negative: pass +f and -f through the ReLU and subtract
            w0[70 + 9 * k, f0 + k] = -1
            wo[k, 5 + 9 * k] = 1
            wo[k, 70 + 9 * k] = -1
        net.set_params(_dev(torch, p))
        got[:, f0:f0 + 4] = net.forward(x_d).cpu().numpy().astype(np.float32)[:, :4]
    err = np.abs(got - want)
    assert err[:, 108:].max() == 0                                          # the 1.0 padding
    lines = []
    for name, base, dims, F in (("position", 0, 3, 10), ("direction", 60, 2, 12)):
        for f in range(F):
            cols = [base + (d * F + f) * 2 + ph for d in range(dims) for ph in range(2)]
            e = err[:, cols]
            lines.append(f"{name} octave {f:2d}: max |delta| {e.max():.2e}  features differing {100.0 * (e > 0).mean():.2f} %")
            assert e.max() <= 1.0e-3, lines[-1]                             # 1 fp16 ulp at |v| in [0.5, 1) is 4.9e-4; 2 at most
    assert (err > 0).mean() < 0.02
    with capsys.disabled():
        print(f"\n[encoder error per octave, mfma shape {mfma_shape}]\n  " + "\n  ".join(lines))
