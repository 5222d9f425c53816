"""Committed fixtures (tests/golden/, generator: tests/golden/make_golden.py).

kat_closed_form.npz -- expected values derived in numpy float64 from the formulas the reference source spells
out, independently of the oracle: the oracle (CPU tests) AND the HIP kernels (gpu tests) are both held to them.
kat_north_star.npz -- the same for the north_star model's own arithmetic (tiny-cuda-nn hash encoding; the corrected
compositor with a central-difference gradient), restated from the published formulas (round 3).
oracle_snapshot.npz -- a regression pin of the oracle itself (the reference holds no vectors: parity unpinned).
"""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = np.load(os.path.join(HERE, "golden", "kat_closed_form.npz"))
SNAP = np.load(os.path.join(HERE, "golden", "oracle_snapshot.npz"))


def _csr(nh):
    return np.concatenate([[0], np.cumsum(nh)[:-1]]).astype(np.int32)


# ------------------------------------------------------------------ oracle vs closed-form fixtures (CPU)
def test_oracle_sampler_regular_and_minstd(oracle):
    nh = KAT["smp_num_hits"]
    s, t = oracle.sample(KAT["smp_start"], KAT["smp_end"], KAT["smp_view"], nh, _csr(nh), 0)
    np.testing.assert_allclose(s, KAT["smp_samples"], rtol=0, atol=1.2e-7)      # sampler.cu:52-66
    np.testing.assert_array_equal(t, KAT["smp_t_vals"].astype(np.float32))
    assert KAT["minstd_10000th"][0] == 399268537                                 # the engine's published check value
    su, _ = oracle.sample(KAT["smp_start"], KAT["smp_end"], KAT["smp_view"], nh, _csr(nh), 2)
    r = (KAT["minstd_first64"].astype(np.float64) - 1) / 2147483648.0            # ray 0, segment 0 and 1: draws 1..64
    d0 = KAT["smp_end"][0].astype(np.float64) - KAT["smp_start"][0]
    np.testing.assert_allclose(su[:32, :3], KAT["smp_start"][0] + r[:32, None] * d0, rtol=0, atol=3e-7)


def test_oracle_volrender_compat(oracle):
    pix = oracle.volrender_fwd(KAT["vr_radiance"], KAT["vr_num_hits"], KAT["vr_indices"], KAT["vr_t"])
    np.testing.assert_allclose(pix, KAT["vr_pixels"], rtol=0, atol=3e-6)
    g = oracle.volrender_bwd(KAT["vr_loss_grads"], KAT["vr_radiance"], KAT["vr_t"], KAT["vr_num_hits"], KAT["vr_indices"])
    np.testing.assert_allclose(g.astype(np.float64), KAT["vr_grads"], rtol=1.2e-3, atol=1e-7)   # fp16 outputs


def test_oracle_encoding_traversal_loss(oracle):
    cfg = oracle.mlp_cfg()
    enc = np.stack([oracle.freq_encode(cfg, x) for x in KAT["enc_in"]])
    np.testing.assert_array_equal(enc, KAT["enc_out"].astype(np.float16).astype(np.float32))
    for mode in (0, 1):
        r = oracle.trace(rays_o=KAT["tr_rays_o"], rays_d=KAT["tr_rays_d"], R=8, mode=mode)
        np.testing.assert_array_equal(r["num_hits"], KAT["tr_num_hits"])
        e = KAT["tr_edges"].astype(np.float32)
        np.testing.assert_array_equal(r["start"][:8, 0], e[:-1])                 # ray 0 walks x cell by cell
        np.testing.assert_array_equal(r["end"][24:32, 1], e[1:])                 # ray 1 walks y
        np.testing.assert_allclose(r["end"][48:56], np.repeat(e[1:, None], 3, 1), rtol=0, atol=3e-7)   # diagonal
    tot, values, _, g32 = oracle.l2_loss(KAT["l2_pred"], KAT["l2_target"], 64.0)
    np.testing.assert_allclose(values, KAT["l2_values"], rtol=2e-6)
    np.testing.assert_allclose(g32, KAT["l2_grads"], rtol=2e-6)


def test_oracle_matches_its_committed_snapshot(oracle):
    from rtx_nerf_amd import scenes  # noqa: F401
    R, W, H = (int(v) for v in SNAP["dims"])
    cfg = oracle.mlp_cfg(n_neurons=64, n_hidden_layers=2)
    for mode in (0, 1):
        pk = oracle.trace_packed(look_at=SNAP["look_at"], focal=float(SNAP["focal"][0]), aspect=W / H, W=W, H=H, R=R,
                                 occ=SNAP["occ"], mode=mode)
        np.testing.assert_array_equal(pk["num_hits"], SNAP[f"m{mode}_num_hits"])
        np.testing.assert_array_equal(pk["start"], SNAP[f"m{mode}_start"])
        np.testing.assert_array_equal(pk["end"], SNAP[f"m{mode}_end"])
        pix, ns = oracle.render(SNAP["look_at"], float(SNAP["focal"][0]), W / H, W, H, R, SNAP["occ"], mode, cfg, SNAP["params"],
                                np.arange(W * H))
        assert ns == int(SNAP[f"m{mode}_samples"][0])
        # libm (expf/sin) may differ by an ulp between hosts: pixels to 1e-6, geometry bit-exact
        np.testing.assert_allclose(pix, SNAP[f"m{mode}_pixels"], rtol=0, atol=1e-6)


# ------------------------------------------------------------------ HIP kernels vs the same fixtures (GPU)
@pytest.mark.gpu
def test_hip_kernels_against_closed_form_fixtures(gpu):
    torch = gpu
    from rtx_nerf_amd import api

    def dev(a, dt=None):
        t = torch.from_numpy(np.ascontiguousarray(a))
        return (t if dt is None else t.to(dt)).cuda()

    nh = KAT["smp_num_hits"]
    P = int(nh.sum())
    samples = torch.zeros((P * 32, 5), device="cuda")
    tv = torch.zeros(P * 32, device="cuda")
    api.launchSampler(dev(KAT["smp_start"]), dev(KAT["smp_end"]), dev(KAT["smp_view"]), tv, samples, nh.size, 8, dev(nh),
                      dev(_csr(nh)), api.SAMPLING_REGULAR)
    np.testing.assert_allclose(samples.cpu().numpy(), KAT["smp_samples"], rtol=0, atol=1.2e-7)
    np.testing.assert_array_equal(tv.cpu().numpy(), KAT["smp_t_vals"].astype(np.float32))
    n = KAT["vr_num_hits"].size
    pix = torch.zeros((n, 3), device="cuda")
    api.launch_volrender_cuda(None, dev(KAT["vr_radiance"]), dev(KAT["vr_num_hits"]), dev(KAT["vr_indices"]), dev(KAT["vr_t"]),
                              n, 32, pix)
    np.testing.assert_allclose(pix.cpu().numpy(), KAT["vr_pixels"], rtol=0, atol=1e-5)
    g = torch.zeros((KAT["vr_t"].size, 4), dtype=torch.float16, device="cuda")
    api.launch_volrender_backward_cuda(None, dev(KAT["vr_loss_grads"]), dev(KAT["vr_radiance"]), dev(KAT["vr_t"]),
                                       dev(KAT["vr_num_hits"]), dev(KAT["vr_indices"]), n, 32, g)
    np.testing.assert_allclose(g.cpu().numpy().astype(np.float64), KAT["vr_grads"], rtol=1.5e-3, atol=1e-7)
    net = api.Network(n_neurons=64, n_hidden_layers=2)
    encT = net.encode_frequency(dev(KAT["enc_in"])).cpu().numpy()[:, :KAT["enc_in"].shape[0]].T.astype(np.float32)
    np.testing.assert_allclose(encT, KAT["enc_out"], rtol=0, atol=8e-4)          # fp16 features
    nh_t = torch.zeros(3, dtype=torch.int32, device="cuda")
    sp = torch.zeros((72, 3), device="cuda")
    ep = torch.zeros((72, 3), device="cuda")
    api.trace_grid(None, grid_res=8, rays_o=dev(KAT["tr_rays_o"]), rays_d=dev(KAT["tr_rays_d"]), mode=api.TRACE_COMPAT,
                   num_hits=nh_t, intersection_arr_size=24, start_points=sp, end_points=ep)
    np.testing.assert_array_equal(nh_t.cpu().numpy(), KAT["tr_num_hits"])
    e = KAT["tr_edges"].astype(np.float32)
    np.testing.assert_array_equal(sp.cpu().numpy()[:8, 0], e[:-1])
    np.testing.assert_array_equal(ep.cpu().numpy()[24:32, 1], e[1:])
    vals = torch.zeros(30, device="cuda")
    gr = torch.zeros(30, dtype=torch.float16, device="cuda")
    api.l2_loss(dev(KAT["l2_pred"]), dev(KAT["l2_target"]), 64.0, vals, gr, None)
    np.testing.assert_allclose(vals.cpu().numpy(), KAT["l2_values"], rtol=2e-6)
    np.testing.assert_allclose(gr.cpu().numpy().astype(np.float64), KAT["l2_grads"], rtol=1e-3)


@pytest.mark.gpu
def test_hip_render_against_oracle_snapshot(gpu):
    torch = gpu
    from rtx_nerf_amd import api, render
    R, W, H = (int(v) for v in SNAP["dims"])
    net = api.Network(n_neurons=64, n_hidden_layers=2)
    net.set_params(torch.from_numpy(SNAP["params"]).cuda())
    occ = torch.from_numpy(SNAP["occ"].view(np.int32).copy()).cuda()
    for mode in (0, 1):
        pipe = render.RenderPipeline(net, R, W, H, float(SNAP["focal"][0]), occupancy=occ, max_segments=W * H * 48, trace_mode=mode)
        pipe.set_pose(SNAP["look_at"])
        pix = pipe.render().cpu().numpy()
        np.testing.assert_array_equal(pipe.num_hits.cpu().numpy(), SNAP[f"m{mode}_num_hits"])
        P = int(pipe.total.item())
        np.testing.assert_array_equal(pipe.start[:P].cpu().numpy(), SNAP[f"m{mode}_start"])
        np.testing.assert_allclose(pix, SNAP[f"m{mode}_pixels"], rtol=0, atol=2e-3)


# ------------------------------------------------------------------ the north_star model's own arithmetic (round 3)
NS = np.load(os.path.join(HERE, "golden", "kat_north_star.npz"))


def _ns_in5():
    p = NS["hg_points"]
    return np.concatenate([p, np.zeros((p.shape[0], 2), np.float32)], axis=1)


def test_oracle_hashgrid_and_nerf_compositor_against_float64_restatements(oracle):
    """kat_north_star.npz: tiny-cuda-nn's hash encoding and the corrected compositor with a NUMERICAL gradient, both restated
    in numpy float64 from the published formulas without touching oracle/rtxn_oracle.c."""
    L, F, T, base = (int(v) for v in NS["hg_cfg"])
    cfg = oracle.hg_cfg(L, F, T, base, float(NS["hg_scale"][0]))
    assert oracle.hg_n_params(cfg) == NS["hg_table"].size == int(NS["hg_level_sizes"].sum()) * F
    enc = oracle.encode_hg(cfg, 0, NS["hg_table"], _ns_in5())[:, :L * F].astype(np.float64)
    np.testing.assert_allclose(enc, NS["hg_enc"], rtol=0, atol=1.5e-3)          # fp16 outputs of values up to ~2
    assert np.abs(enc - NS["hg_enc"]).mean() < 2e-4
    pix = oracle.volrender_fwd_nerf(NS["nerf_radiance"], NS["nerf_num_hits"], NS["nerf_indices"], NS["nerf_step"])
    np.testing.assert_allclose(pix, NS["nerf_pixels"], rtol=0, atol=3e-6)
    g = oracle.volrender_bwd_nerf(NS["nerf_loss_grads"], NS["nerf_radiance"], NS["nerf_step"], NS["nerf_num_hits"], NS["nerf_indices"])
    np.testing.assert_allclose(g.astype(np.float64), NS["nerf_grads"], rtol=2e-3, atol=2e-6)     # fp16 outputs vs central differences


@pytest.mark.gpu
def test_hip_hashgrid_and_nerf_compositor_against_float64_restatements(gpu):
    torch = gpu
    from rtx_nerf_amd import api

    def dev(a, dt=None):
        t = torch.from_numpy(np.ascontiguousarray(a))
        return (t if dt is None else t.to(dt)).cuda()

    L, F, T, base = (int(v) for v in NS["hg_cfg"])
    hg = api.HashGrid(n_levels=L, n_features=F, log2_hashmap_size=T, base_resolution=base, per_level_scale=float(NS["hg_scale"][0]),
                      n_dir_freqs=0)
    assert hg.n_params() == NS["hg_table"].size
    n = NS["hg_points"].shape[0]
    encT = hg.encode(dev(NS["hg_table"]), dev(_ns_in5()))
    enc = encT.cpu().numpy()[:L * F, :n].T.astype(np.float64)
    np.testing.assert_allclose(enc, NS["hg_enc"], rtol=0, atol=1.5e-3)
    assert np.abs(enc - NS["hg_enc"]).mean() < 2e-4
    nr = NS["nerf_num_hits"].size
    pix = torch.zeros((nr, 3), device="cuda")
    api.launch_volrender_cuda(None, dev(NS["nerf_radiance"]), dev(NS["nerf_num_hits"]), dev(NS["nerf_indices"]), dev(NS["nerf_step"]),
                              nr, 32, pix, mode=api.VR_NERF)
    np.testing.assert_allclose(pix.cpu().numpy(), NS["nerf_pixels"], rtol=0, atol=1e-5)
    g = torch.zeros((NS["nerf_step"].size, 4), dtype=torch.float16, device="cuda")
    api.launch_volrender_backward_cuda(None, dev(NS["nerf_loss_grads"]), dev(NS["nerf_radiance"]), dev(NS["nerf_step"]),
                                       dev(NS["nerf_num_hits"]), dev(NS["nerf_indices"]), nr, 32, g, mode=api.VR_NERF)
    np.testing.assert_allclose(g.cpu().numpy().astype(np.float64), NS["nerf_grads"], rtol=2.5e-3, atol=3e-6)
