"""BASELINE.json configs[2] (training: 4096 rays/batch, hash grid L=16/T=2^19 + 4x64, 128^3) and configs[4] (1008x756
forward-facing frame, 256^3 sparse grid, 8x256) at FULL size.  The oracle cannot run these sizes in seconds, so it checks a
strided sub-sample of the rays; everything else is a size-independent property: the loss falls, the batch splits into
two data-parallel shards whose summed gradients equal the whole batch's, row shards recombine to the full frame bit for
bit, and a re-render is bit-identical."""
import numpy as np
import pytest

from rtx_nerf_amd import scenes

pytestmark = pytest.mark.gpu

HGD = dict(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.5)


# ----------------------------------------------------------------------------------------------- configs[2]
@pytest.fixture(scope="module")
def config3(gpu):
    torch = gpu
    from rtx_nerf_amd.train import Trainer, camera_rays
    R, B = 128, 4096
    dense = scenes.lego_standin_density(R, seed=0)
    words = scenes.pack_occupancy(dense)
    occ = torch.from_numpy(words.view(np.int32).copy()).cuda()

    def make(seed=1337):
        return Trainer(R, occ, encoding="hash", n_neurons=64, n_hidden_layers=4, hashgrid=HGD, n_dir_freqs=4,
                       batch_rays=128 * 128, max_segments=128 * 128 * 24, lr=1e-2, loss_scale=128.0, density_scale=300.0,
                       mode="nerf", seed=seed)

    tr = make()
    focal = scenes.lego_focal_length(True)
    ro, rd, tg = [], [], []
    for i in range(4):
        o, d = camera_rays(scenes.pose_spherical(90.0 * i + 15.0, -30.0, origin_scale=10.0), focal, 128, 128)
        ro.append(o); rd.append(d); tg.append(tr.render_rays(o, d, radiance_fn=scenes.teacher_field).clone())
    ro, rd, tg = torch.cat(ro), torch.cat(rd), torch.cat(tg)
    g = torch.Generator(device="cuda").manual_seed(42)
    idx = torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g)
    batch = (ro[idx].contiguous(), rd[idx].contiguous(), tg[idx].contiguous())
    return dict(torch=torch, tr=tr, make=make, R=R, B=B, words=words, batch=batch, pool=(ro, rd, tg), gen=g)


def test_config3_forward_matches_oracle_on_a_ray_subsample(config3, oracle):
    """Full-size model and batch on the GPU; the oracle chain (trace -> sample -> hash encode -> 4x64 MLP -> NeRF
    compositor) on every 16th ray of the same batch with the same table and weights."""
    torch, tr, O = config3["torch"], config3["tr"], oracle
    o, d, tgt = config3["batch"]
    # non-trivial table so that the encoding is not ~0 (tcnn initialises it to U(-1e-4, 1e-4))
    g = torch.Generator().manual_seed(5)
    tr.table_master.copy_(((torch.rand(tr.hg.n_params(), generator=g) * 2 - 1) * 0.5).cuda())
    tr.table.copy_(tr.table_master.half())
    pix = tr.render_rays(o, d).cpu().numpy()
    P = int(tr.total.item())
    assert P * 32 > 400_000 and tr.truncated_steps == 0          # a full-size batch: several hundred thousand samples
    sub = np.arange(0, config3["B"], 16)
    on, dn = o.cpu().numpy()[sub], d.cpu().numpy()[sub]
    pk = O.trace_packed(rays_o=on, rays_d=dn, R=config3["R"], occ=config3["words"], mode=1)
    nh = tr.num_hits[:config3["B"]].cpu().numpy()
    np.testing.assert_array_equal(pk["num_hits"], nh[sub])       # traversal: bit-exact counts
    samples, steps = O.sample(pk["start"], pk["end"], pk["view_dirs"], pk["num_hits"], pk["indices"], 3)
    steps = steps * np.float32(tr.density_scale)
    enc = O.encode_hg(O.hg_cfg(**HGD), 4, tr.table.cpu().numpy(), samples)
    _, out = O.mlpe_forward(64, 4, 1, tr.params.cpu().numpy(), enc)
    want = O.volrender_fwd_nerf(out[:, :4].astype(np.float32), pk["num_hits"], pk["indices"], steps)
    np.testing.assert_allclose(pix[sub], want, rtol=0, atol=4e-3)   # fp16 activations through 5 layers + 160-sample compositing
    assert np.abs(want).max() > 0.05


def test_config3_loss_decreases_at_full_size(config3):
    torch = config3["torch"]
    tr = config3["make"](seed=7)
    ro, rd, tg = config3["pool"]
    g = torch.Generator(device="cuda").manual_seed(1)
    losses = []
    for it in range(60):
        idx = torch.randint(0, ro.shape[0], (config3["B"],), device="cuda", generator=g)
        loss = tr.step(ro[idx].contiguous(), rd[idx].contiguous(), tg[idx].contiguous())
        if it % 10 == 0 or it == 59:
            losses.append(float(loss.item()))
    assert np.isfinite(losses).all() and losses[-1] < 0.5 * losses[0], losses
    assert tr.step_count == 60 and tr.truncated_steps == 0


def test_config3_two_shard_gradients_sum_to_the_batch_gradient(config3):
    """Data-parallel decomposition at full size on one GPU: the loss-scaled gradient of the whole 4096-ray batch equals
    the mean of the gradients of its two 2048-ray halves (each half's L2 divides by ITS pixel count), up to the fp16
    rounding of the loss gradients and fp32 atomics order."""
    torch = config3["torch"]
    tr = config3["make"](seed=11)
    g = torch.Generator().manual_seed(3)
    tr.table_master.copy_(((torch.rand(tr.hg.n_params(), generator=g) * 2 - 1) * 0.3).cuda())
    tr.table.copy_(tr.table_master.half())
    o, d, tgt = config3["batch"]
    h = config3["B"] // 2
    tr.gradients(o, d, tgt)
    full_p, full_t = tr.dparams.clone(), tr.table_grad()
    acc_p, acc_t = torch.zeros_like(full_p), torch.zeros_like(full_t)
    for sl in (slice(0, h), slice(h, 2 * h)):
        tr.gradients(o[sl].contiguous(), d[sl].contiguous(), tgt[sl].contiguous())
        acc_p += tr.dparams
        acc_t += tr.table_grad()
    acc_p *= 0.5
    acc_t *= 0.5
    for full, acc in ((full_p, acc_p), (full_t, acc_t)):
        assert float(full.abs().max()) > 0
        assert float((full - acc).norm()) < 2e-2 * float(full.norm())


def test_train_path_rejects_the_256_wide_model(gpu):
    """The 256-wide model (config 5) has an inference kernel only: the training entry points must refuse it instead of
    running the 128-wide kernels over its packing."""
    torch = gpu
    from rtx_nerf_amd import _lib, api
    net = api.Network(n_neurons=256, n_hidden_layers=2)
    net.set_params(torch.from_numpy(scenes.xavier_params_fp16(256, 2, net.encoded_width())).cuda())
    enc = torch.zeros((net.encoded_width(), 256), dtype=torch.float16, device="cuda")
    ws = torch.zeros(1 << 20, dtype=torch.float16, device="cuda")
    with pytest.raises(_lib.RtxnError, match="no training kernels"):
        net.train_forward(enc, 256, ws)


# ----------------------------------------------------------------------------------------------- configs[4]
@pytest.fixture(scope="module")
def config5(gpu):
    torch = gpu
    from rtx_nerf_amd import api, render
    R, W, H = 256, 1008, 756
    dense = scenes.llff_standin_density(R, seed=3)
    words = scenes.pack_occupancy(dense)
    occ = torch.from_numpy(words.view(np.int32).copy()).cuda()
    net = api.Network(n_neurons=256, n_hidden_layers=8)
    params = scenes.xavier_params_fp16(256, 8, net.encoded_width(), seed=1337)
    net.set_params(torch.from_numpy(params).cuda())
    la = scenes.pose_forward_facing(0.3, 0.0)
    pipe = render.RenderPipeline(net, R, W, H, 1.6, occupancy=occ, max_segments=1024)
    pipe.calibrate([la])
    pipe.set_pose(la)
    pix = pipe.render().clone()
    torch.cuda.synchronize()
    return dict(torch=torch, render=render, net=net, occ=occ, words=words, params=params, la=la, pipe=pipe, pix=pix,
                R=R, W=W, H=H, f=1.6)


def test_config5_csr_determinism_and_range(config5):
    torch, pipe = config5["torch"], config5["pipe"]
    n = config5["W"] * config5["H"]
    nh = pipe.num_hits[:n].cpu().numpy().astype(np.int64)
    idx = pipe.indices[:n].cpu().numpy().astype(np.int64)
    total = int(pipe.total.item())
    assert total == nh.sum() and total > 300_000 and not pipe.overflowed()
    np.testing.assert_array_equal(idx, np.concatenate([[0], np.cumsum(nh)[:-1]]))
    assert nh.max() <= 3 * config5["R"] - 2
    pix2 = pipe.render().clone()
    torch.cuda.synchronize()
    assert torch.equal(pix2, config5["pix"])
    p = config5["pix"].cpu().numpy()
    assert np.isfinite(p).all() and np.all(p[nh == 0] == 0) and p.min() >= 0 and p.max() < 1.0 and p.max() > 0.1


@pytest.mark.parametrize("world", [2, 8])
def test_config5_row_shards_recombine_to_the_full_frame(config5, world):
    torch, render = config5["torch"], config5["render"]
    from rtx_nerf_amd.shard import RowShard
    W, H = config5["W"], config5["H"]
    bufs = []
    for rank in range(world):
        sh = RowShard(W, H, rank, world)
        pipe = render.RenderPipeline(config5["net"], config5["R"], W, H, config5["f"], occupancy=config5["occ"],
                                     max_rays=sh.n_local, max_segments=config5["pipe"].max_segments, window=sh.window)
        pipe.set_pose(config5["la"])
        out = torch.zeros((sh.n_max, 3), device="cuda")
        pipe.render(ray_begin=sh.ray_begin, ray_count=sh.n_local, out=out[:sh.n_local])
        assert not pipe.overflowed()
        bufs.append(out)
    img = RowShard(W, H, 0, world).assemble(bufs)
    assert torch.equal(img.reshape(-1, 3), config5["pix"])


def test_config5_oracle_spot_check_on_a_strided_sample(config5, oracle):
    W, H = config5["W"], config5["H"]
    nh = config5["pipe"].num_hits[:W * H].cpu().numpy()
    hit = np.nonzero(nh > 0)[0]
    ids = np.concatenate([hit[:: max(1, len(hit) // 600)][:600], (np.arange(200, dtype=np.int64) * 3803 + 17) % (W * H)]).astype(np.uint32)
    cfg = oracle.mlp_cfg(n_neurons=256, n_hidden_layers=8)
    want, ns = oracle.render(config5["la"], config5["f"], W / H, W, H, config5["R"], config5["words"], 1, cfg,
                             config5["params"], ids)
    got = config5["pix"].cpu().numpy()[ids]
    assert ns > 0 and np.abs(want).max() > 0.05
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-3)
    mse = float(((got - want) ** 2).mean())
    assert 10 * np.log10(1.0 / max(mse, 1e-20)) > 70.0
