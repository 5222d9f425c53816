"""Hash-grid inference (rtxn_hashmlp_forward_segments, rtxn_volrender_fwd_compact_nerf, RenderPipeline(hashgrid=...)):
parity against the oracle chain orc_sample -> orc_encode_hg -> orc_mlpe_forward -> compositor, and fused == staged
(rtxn_hashgrid_encode_segments + rtxn_mlp_train_forward_outputs) bit for bit."""
import numpy as np
import pytest

from rtx_nerf_amd import scenes

pytestmark = pytest.mark.gpu


def _dev(torch, a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).cuda()


def _segments(rng, P, extent=0.9):
    """P random segments inside [-extent, extent]^3, about a 128^3 cell long, and their rays' (theta, phi)."""
    start = rng.uniform(-extent, extent, (P, 3)).astype(np.float32)
    d = rng.standard_normal((P, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    end = (start + d * rng.uniform(0.005, 0.03, (P, 1))).astype(np.float32)
    view = np.stack([rng.uniform(0, np.pi, P), rng.uniform(-np.pi, np.pi, P)], axis=1).astype(np.float32)
    return start, np.clip(end, -1, 1), view


def _model(torch, api, oracle, levels, log2, base, scale, n_dir_freqs, L, seed, act=1):
    hg = api.HashGrid(levels, 2, log2, base, scale, n_dir_freqs=n_dir_freqs)
    ocfg = oracle.hg_cfg(levels, 2, log2, base, scale)
    E = hg.encoded_width()
    net = api.Network(n_neurons=64, n_hidden_layers=L, n_encoded_features=E, output_activation=act)
    rng = np.random.default_rng(seed)
    params = scenes.xavier_params_fp16(64, L, E, seed=seed)
    table = rng.uniform(-0.5, 0.5, hg.n_params()).astype(np.float16)
    return hg, ocfg, net, params, table, E


@pytest.mark.parametrize("levels,log2,base,scale,ndf,L,P,stype,act", [
    (16, 19, 16, 1.5, 4, 4, 700, 3, 1),      # BASELINE configs[2]'s model, MIDPOINT_WORLD sampling
    (16, 19, 16, 1.5, 4, 4, 33, 0, 1),       # REGULAR sampling, an odd segment count (half-empty last wave tile)
    (4, 12, 8, 1.6, 4, 2, 257, 3, 0),        # few levels: hash, direction and padding dwords inside ONE k-step, no activation
    (8, 14, 8, 2.0, 2, 1, 64, 0, 1),         # one hidden layer, E = 32 (a single layer-0 k-step)
    (12, 15, 4, 1.7, 6, 8, 1, 3, 1),         # eight hidden layers, one segment
])
def test_hashmlp_fused_equals_staged_and_oracle(gpu, oracle, levels, log2, base, scale, ndf, L, P, stype, act):
    torch = gpu
    from rtx_nerf_amd import api
    hg, ocfg, net, params, table, E = _model(torch, api, oracle, levels, log2, base, scale, ndf, L, seed=levels + L, act=act)
    assert api.hashmlp_supported(net, hg)
    rng = np.random.default_rng(P)
    start, end, view = _segments(rng, P)
    cap = P + 37                                           # launch capacity > live segments: blocks past the count must idle
    sp, ep, sv = (torch.zeros((cap, k), device="cuda") for k in (3, 3, 2))
    sp[:P], ep[:P], sv[:P] = _dev(torch, start), _dev(torch, end), _dev(torch, view)
    total = torch.tensor([P], dtype=torch.int32, device="cuda")
    p16, t16 = _dev(torch, params), _dev(torch, table)
    net.set_params(p16)
    t_scale = 7.5
    rad = torch.full((cap * 32, 4), -1.0, dtype=torch.float16, device="cuda")
    step = torch.full((cap,), -1.0, device="cuda")
    api.hashmlp_forward_segments(net, hg, t16, sp, ep, sv, total, cap, rad, stype, t_scale, step)
    # ---- staged: encoder kernel -> MLP kernel (the training step's forward), same values bit for bit ----
    S = P * 32
    encT = torch.zeros((E, api.padded_samples(S)), dtype=torch.float16, device="cuda")
    tv = torch.zeros(S, device="cuda")
    hg.encode_segments(t16, sp, ep, sv, P, stype, encT, tv, t_scale)
    out16 = net.train_forward_outputs(encT, S)
    torch.cuda.synchronize()
    got = rad.cpu().numpy()
    np.testing.assert_array_equal(got[:S].view(np.uint16), out16[:, :4].contiguous().cpu().numpy().view(np.uint16))
    assert np.all(got[S:] == -1.0)                         # nothing written past the live segments
    if stype == 3:
        np.testing.assert_array_equal(step.cpu().numpy()[:P], tv.cpu().numpy()[::32])     # one step per segment == the sampler's t_vals
        assert np.all(step.cpu().numpy()[P:] == -1.0)
    # ---- oracle chain: sampler -> hash/frequency encoding -> MLP ----
    nh = np.ones(P, np.int32)
    idx = np.arange(P, dtype=np.int32)
    samples, _ = oracle.sample(start, end, view, nh, idx, stype)
    enc = oracle.encode_hg(ocfg, ndf, table, samples)
    want = oracle.mlpe_forward(64, L, act, params, enc)[1][:, :4].astype(np.float32)
    g = got[:S].astype(np.float32)
    assert np.abs(g - want).max() < (1e-2 if act else 3e-2) and np.abs(g - want).mean() < 1e-3
    # the encoder's hash features are bit-exact against the oracle (gathers + trilinear), hence so is the fused kernel's input
    np.testing.assert_array_equal(encT[:2 * levels, :S].cpu().numpy().T, enc[:, :2 * levels])


def test_hashmlp_rejects_what_it_is_not_built_for(gpu):
    torch = gpu
    from rtx_nerf_amd import api, _lib
    hg = api.HashGrid(16, 2, 19, 16, 1.5, n_dir_freqs=4)
    wide = api.Network(n_neurons=128, n_hidden_layers=2, n_encoded_features=hg.encoded_width())
    assert not api.hashmlp_supported(wide, hg)
    odd = api.HashGrid(5, 2, 12, 4, 1.5, n_dir_freqs=4)
    net = api.Network(n_neurons=64, n_hidden_layers=2, n_encoded_features=odd.encoded_width())
    assert not api.hashmlp_supported(net, odd)
    z = torch.zeros((4, 3), device="cuda")
    with pytest.raises(_lib.RtxnError, match="built for"):
        api.hashmlp_forward_segments(wide, hg, torch.zeros(hg.n_params(), dtype=torch.float16, device="cuda"), z, z, z[:, :2].contiguous(),
                                     torch.zeros(1, dtype=torch.int32, device="cuda"), 4, torch.zeros((128, 4), dtype=torch.float16, device="cuda"))
    ok_net = api.Network(n_neurons=64, n_hidden_layers=2, n_encoded_features=hg.encoded_width())
    assert api.hashmlp_supported(ok_net, hg)
    with pytest.raises(_lib.RtxnError, match="set_params"):      # no weights yet
        api.hashmlp_forward_segments(ok_net, hg, torch.zeros(hg.n_params(), dtype=torch.float16, device="cuda"), z, z, z[:, :2].contiguous(),
                                     torch.zeros(1, dtype=torch.int32, device="cuda"), 4, torch.zeros((128, 4), dtype=torch.float16, device="cuda"))


@pytest.mark.parametrize("K", [32, 6])
def test_compact_nerf_compositor_equals_the_float4_form(gpu, oracle, K):
    """rtxn_volrender_fwd_compact_nerf (half4 radiance + ONE step per segment) == rtxn_volrender_fwd(RTXN_VR_NERF) on the
    widened radiance and the per-sample steps, bit for bit; and both match the oracle's NERF compositor."""
    torch = gpu
    from rtx_nerf_amd import api
    rng = np.random.default_rng(K)
    B = 300
    nh = rng.integers(0, 9, B).astype(np.int32)
    nh[::7] = 0
    idx = np.concatenate([[0], np.cumsum(nh)[:-1]]).astype(np.int32)
    P = int(nh.sum())
    rad16 = rng.uniform(0, 1, (P * K, 4)).astype(np.float16)
    rad16[:, 3] = rng.uniform(0, 30, P * K).astype(np.float16)
    seg_step = rng.uniform(1e-3, 2e-2, P).astype(np.float32)
    pix = [torch.zeros((B, 3), device="cuda") for _ in range(2)]
    api.volrender_compact_nerf(_dev(torch, rad16), _dev(torch, seg_step), _dev(torch, nh), _dev(torch, idx), B, K, pix[0])
    tv = np.repeat(seg_step, K).astype(np.float32)
    api.launch_volrender_cuda(None, _dev(torch, rad16.astype(np.float32)), _dev(torch, nh), _dev(torch, idx), _dev(torch, tv), B, K, pix[1],
                              mode=api.VR_NERF)
    torch.cuda.synchronize()
    assert torch.equal(pix[0], pix[1]) and float(pix[0].abs().sum()) > 0
    want = oracle.volrender_fwd_nerf(rad16.astype(np.float32), nh, idx, tv, K=K)
    np.testing.assert_allclose(pix[0].cpu().numpy(), want, rtol=0, atol=1e-4)


def _hash_scene(torch, R=32, seed=5):
    from rtx_nerf_amd import api
    words = scenes.pack_occupancy(scenes.sphere_density(R, 0.7))
    occ = torch.from_numpy(words.view(np.int32).copy()).cuda()
    hg = api.HashGrid(8, 2, 14, 8, 1.6, n_dir_freqs=4)
    E = hg.encoded_width()
    net = api.Network(n_neurons=64, n_hidden_layers=3, n_encoded_features=E)
    params = scenes.xavier_params_fp16(64, 3, E, seed=seed)
    table = np.random.default_rng(seed).uniform(-0.5, 0.5, hg.n_params()).astype(np.float16)
    net.set_params(torch.from_numpy(params).cuda())
    return words, occ, hg, net, params, table, E


@pytest.mark.parametrize("vr_mode", [1, 0])
def test_hash_render_pipeline_matches_the_oracle_chain(gpu, oracle, vr_mode):
    """A whole frame of a hash-grid model through rtxn_render_frame (RenderPipeline(hashgrid=..., table=...)): traversal, scan,
    fused hash kernel, compact compositor -- against oracle trace -> sample -> encode_hg -> mlpe_forward -> compositor, in the
    corrected (MIDPOINT_WORLD + RTXN_VR_NERF, what Trainer's "nerf" mode trains) and the reference's (REGULAR + RTXN_VR_COMPAT)
    arithmetic; pipelined frames equal serial ones."""
    torch = gpu
    from rtx_nerf_amd import api, render
    R, W, H = 32, 44, 36
    words, occ, hg, net, params, table, E = _hash_scene(torch, R)
    t16 = torch.from_numpy(table).cuda()
    la = scenes.pose_spherical(35.0, -28.0, origin_scale=10.0)
    f = scenes.lego_focal_length(True)
    scale = 40.0
    pipe = render.RenderPipeline(net, R, W, H, f, occupancy=occ, max_segments=W * H * 50, vr_mode=vr_mode, step_scale=scale,
                                 hashgrid=hg, table=t16)
    assert pipe.compact and pipe.sample_type == (3 if vr_mode == 1 else 0)
    pipe.set_pose(la)
    pix = pipe.render().clone()
    torch.cuda.synchronize()
    assert not pipe.overflowed()
    # oracle
    tr = oracle.trace(look_at=la, focal=f, aspect=W / H, W=W, H=H, R=R, occ=words, mode=1)
    nh = tr["num_hits"].astype(np.int32)
    np.testing.assert_array_equal(pipe.num_hits.cpu().numpy(), nh)
    idx = np.concatenate([[0], np.cumsum(nh)[:-1]]).astype(np.int32)
    P = int(nh.sum())
    assert P == int(pipe.total.item()) and P > 500
    S_ = tr["start"].shape[0] // (W * H)
    keep = (np.arange(S_)[None, :] < nh[:, None]).reshape(-1)
    start, end = tr["start"].reshape(-1, 3)[keep], tr["end"].reshape(-1, 3)[keep]
    np.testing.assert_array_equal(pipe.start.cpu().numpy()[:P], start)
    samples, tv = oracle.sample(start, end, tr["view_dirs"], nh, idx, 3 if vr_mode == 1 else 0)
    enc = oracle.encode_hg(oracle.hg_cfg(8, 2, 14, 8, 1.6), 4, table, samples)
    out = oracle.mlpe_forward(64, 3, 1, params, enc)[1][:, :4].astype(np.float32)
    if vr_mode == 1:
        want = oracle.volrender_fwd_nerf(out, nh, idx, tv * np.float32(scale), K=32)
    else:
        want = oracle.volrender_fwd(out, nh, idx, tv, K=32)
    np.testing.assert_allclose(pix.cpu().numpy(), want, rtol=0, atol=3e-3)
    assert float(pix.std()) > 0.01
    mse = float(((pix.cpu().numpy() - want) ** 2).mean())
    assert 10 * np.log10(1.0 / max(mse, 1e-20)) > 60.0
    # pipelined == serial, bit for bit, over more frames than slots
    la_d = torch.from_numpy(la.reshape(16).astype(np.float32)).cuda()
    outs = [torch.empty((W * H, 3), device="cuda") for _ in range(5)]
    for o in outs:
        pipe.render_async(la_d, out=o)
    pipe.finish()
    for o in outs:
        assert torch.equal(o, pix)


def test_trained_hash_model_renders_through_the_frame_entry(gpu):
    """train -> render: a Trainer's hash model drawn by RenderPipeline on the trainer's OWN live tensors (table, model handle,
    occupancy), after every step without re-packing anything by hand; agrees with Trainer.render_rays (staged training kernels)
    on the same camera, and follows the parameters as they train."""
    torch = gpu
    from rtx_nerf_amd import api, render
    from rtx_nerf_amd.train import Trainer, camera_rays
    R, W, H = 32, 40, 40
    occ = torch.from_numpy(scenes.pack_occupancy(scenes.sphere_density(R, 0.72)).view(np.int32).copy()).cuda()
    tr = Trainer(R, occ, encoding="hash", n_neurons=64, n_hidden_layers=4,
                 hashgrid=dict(n_levels=8, n_features=2, log2_hashmap_size=13, base_resolution=8, per_level_scale=1.5),
                 batch_rays=W * H, max_segments=W * H * 40, lr=1e-2, density_scale=100.0, seed=1)
    la = scenes.pose_spherical(25.0, -30.0, origin_scale=10.0)
    f = scenes.lego_focal_length(True)
    o, d = camera_rays(la, f, W, H)
    pipe = tr.render_pipeline(W, H, f, max_segments=W * H * 40)
    pipe.set_pose(la)
    frames = []
    for it in range(3):
        a = pipe.render().clone()
        b = tr.render_rays(o, d).clone()
        torch.cuda.synchronize()
        # same kernels' values, but render_rays takes explicit float64-built rays where the frame generates them in the traversal
        # kernel (optixPrograms.cu:56-69 arithmetic): endpoints differ in the last ulps, a grazing ray may gain or lose a cell
        diff = (a - b).abs().cpu().numpy()
        assert np.median(diff) < 2e-4 and (diff < 1e-2).mean() > 0.99, (np.median(diff), (diff < 1e-2).mean())
        frames.append(a)
        tgt = torch.full((W * H, 3), 0.25 + 0.2 * it, device="cuda")
        for _ in range(4):
            tr.step(o, d, tgt)
    assert not torch.equal(frames[0], frames[2]) and float(frames[2].std()) > 0
