"""The C++ host (examples/render_host.cpp).  `stages`: the drop-in headers in the stage order of the reference's main.cu
must produce the same image as the same stages driven through the Python mirror of the interface, and both must agree
with the oracle.  `frame`: BASELINE configs[1] (800x800, 128^3, 8x128) through the frame entry rtxn_render_frame from C++
-- no host re-pack, no Python -- must be bit-identical to RenderPipeline.render() and within tolerance of the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_matches_python_api_and_oracle(gpu, oracle, tmp_path):
    torch = gpu
    from rtx_nerf_amd import api
    exe = os.path.join(ROOT, "examples", "render_host")
    if not os.path.exists(exe):      # normally built by `make` / __graft_entry__.build(); hipcc is on the GPU box too
        subprocess.check_call(["make", "-C", ROOT, "examples/render_host"])
    assert os.path.exists(exe)
    W = H = 40
    out = str(tmp_path / "host.ppm")
    res = subprocess.run([exe, "stages", str(W), str(H), out], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = open(out, "rb").read()
    header = f"P6\n{W} {H}\n255\n".encode()
    assert raw.startswith(header)
    img = np.frombuffer(raw[len(header):], np.uint8).reshape(H * W, 3)

    R, S = 8, 24
    net = api.Network()
    p32 = net.initialize_params(1337)
    p16 = p32.half()
    net.set_params(p16.cuda())
    la = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 40, 0, 0, 0, 1], np.float32)
    f = float(np.float32(1.0) / np.tan(np.float32(0.5) * np.float32(0.6911112)))
    n = W * H
    nh = torch.zeros(n, dtype=torch.int32, device="cuda")
    vd = torch.zeros((n, 2), device="cuda")
    sp = torch.zeros((n * S, 3), device="cuda")
    ep = torch.zeros((n * S, 3), device="cuda")
    api.trace_grid(torch.from_numpy(la).cuda(), f, 1.0, W, H, grid_res=R, mode=api.TRACE_COMPAT, num_hits=nh,
                   viewing_direction=vd, intersection_arr_size=S, start_points=sp, end_points=ep)
    idx, total = api.scan_hits(nh)
    P = int(total.item())
    keep = (torch.arange(S, device="cuda")[None, :] < nh[:, None]).reshape(-1)
    psp, pep = sp[keep].contiguous(), ep[keep].contiguous()
    assert psp.shape[0] == P
    samples = torch.zeros((P * 32, 5), device="cuda")
    tv = torch.zeros(P * 32, device="cuda")
    api.launchSampler(psp, pep, vd, tv, samples, n, R, nh, idx, api.SAMPLING_REGULAR)
    rad = net.forward_radiance(samples)
    pix = torch.zeros((n, 3), device="cuda")
    api.launch_volrender_cuda(samples, rad, nh, idx, tv, n, 32, pix)
    pix = pix.cpu().numpy()
    q = np.rint(255.0 * np.clip(pix, 0, 1)).astype(np.int32)
    assert np.abs(q - img.astype(np.int32)).max() <= 1 and (q == img).mean() > 0.99
    assert img.std() > 1.0

    # and the oracle on the same rays / weights
    cfg = oracle.mlp_cfg()
    want, _ = oracle.render(la, f, 1.0, W, H, R, None, 0, cfg, p16.numpy(), np.arange(n))
    np.testing.assert_allclose(pix, want, rtol=0, atol=2e-3)


def _exe():
    exe = os.path.join(ROOT, "examples", "render_host")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", ROOT, "examples/render_host"])
    return exe


@pytest.mark.parametrize("W,H,R,occupied", [(800, 800, 128, True), (96, 64, 8, False)])
def test_cpp_host_frame_entry_is_the_python_pipelines_frame(gpu, oracle, tmp_path, W, H, R, occupied):
    torch = gpu
    from rtx_nerf_amd import api, render, scenes
    la = scenes.pose_spherical(15.0, -30.0, origin_scale=10.0).astype(np.float32).reshape(16)
    words = scenes.pack_occupancy(scenes.lego_standin_density(R, seed=0)) if occupied else None
    occ_path = "-"
    if occupied:
        occ_path = str(tmp_path / "occ.u32")
        words.astype("<u4").tofile(occ_path)
    out, raw = str(tmp_path / "frame.ppm"), str(tmp_path / "frame.f32")
    f = float(np.float32(1.0) / np.tan(np.float32(0.5) * np.float32(0.6911112)))
    res = subprocess.run([_exe(), "frame", str(W), str(H), str(R), out, raw, occ_path, "4", ",".join(f"{v:.9g}" for v in la), f"{f:.9g}"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "ms/frame" in res.stdout
    got = np.fromfile(raw, np.float32).reshape(W * H, 3)
    # the same frame through the Python caller of the same entry point
    net = api.Network()
    p16 = net.initialize_params(1337).half()
    net.set_params(p16.cuda())
    occ = torch.from_numpy(words.view(np.int32).copy()).cuda() if occupied else None
    pipe = render.RenderPipeline(net, R, W, H, f, occupancy=occ, max_segments=1024, sub_rays=2 if W * H >= 300000 else 8)
    pipe.calibrate([la])
    pipe.set_pose(la)
    want = pipe.render().cpu().numpy()
    assert np.array_equal(got, want) and got.std() > 1e-3
    assert f"{int(pipe.total.item())} segments" in res.stdout
    # and the oracle on a strided ray sample
    ids = (np.arange(1500, dtype=np.int64) * ((W * H) // 1500) + 7).astype(np.uint32)
    cfg = oracle.mlp_cfg()
    ref, _ = oracle.render(la, f, W / H, W, H, R, words, 1, cfg, p16.numpy(), ids)
    np.testing.assert_allclose(got[ids], ref, rtol=0, atol=1e-3)


def test_cpp_training_host_runs_the_reference_iteration_as_one_call_per_step(gpu, tmp_path):
    """examples/train_host.cpp: the reference's own training iteration (8x128 + Composite-Frequency, REGULAR sampler, the
    reference compositor and its backward, L2, Adam 1e-3, 8^3 dense grid) with rtxn_train_step as the whole loop body, captured
    into a hipGraph after the first step.  The program exits non-zero on a non-finite loss or a device step counter that is not
    the number of steps; here additionally: with the corrected (NeRF) compositor the loss halves, and the parameters either mode
    leaves are finite and have moved from the PCG32 initialisation."""
    torch = gpu
    from rtx_nerf_amd import api
    exe = os.path.join(ROOT, "examples", "train_host")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", ROOT, "examples/train_host"])
    init = api.Network().initialize_params(1337).numpy()
    for mode in ("compat", "nerf"):
        out = str(tmp_path / f"master_{mode}.f32")
        res = subprocess.run([exe, "151", "2048", "8", out, mode], capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout + res.stderr
        losses = [float(line.split("loss")[1].split(",")[0]) for line in res.stdout.splitlines() if line.startswith("step")]
        assert len(losses) >= 4 and all(np.isfinite(losses)), res.stdout
        if mode == "nerf":       # the reference's own backward is not the gradient of its forward: only this one learns
            assert losses[-1] < 0.5 * losses[0], res.stdout
        assert "device step counter 151" in res.stdout and "ms per step" in res.stdout
        got = np.fromfile(out, np.float32)
        assert got.shape == init.shape and np.isfinite(got).all()
        assert 1e-3 < np.abs(got - init).max() < 1.0          # 150 Adam steps of 1e-3


@pytest.mark.parametrize("W,H,R,occupied,devices", [(320, 203, 64, True, "0,0"), (320, 203, 64, True, "0,0,0"), (96, 64, 8, False, "0,0,0,0,0")])
def test_cpp_multi_gpu_host_shards_rehearsed_on_one_gpu(gpu, tmp_path, W, H, R, occupied, devices):
    """examples/render_host_mgpu.cpp: ONE C++ process, one rtxn_render + stream per shard, rows dealt
    round-robin (window_chunk = W, window_stride = N W), each shard's rows copied to the root behind its compositor and
    interleaved there.  With every shard on device 0 ("0,0": the one-GPU rehearsal; H is not a multiple of N: ragged shards) the
    assembled frame must be the single-device frame of examples/render_host bit for bit.  Unmeasured on more than one GPU."""
    from rtx_nerf_amd import scenes
    exe = os.path.join(ROOT, "examples", "render_host_mgpu")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", ROOT, "examples/render_host_mgpu"])
    la = scenes.pose_spherical(15.0, -30.0, origin_scale=10.0).astype(np.float32).reshape(16)
    occ_path = "-"
    if occupied:
        occ_path = str(tmp_path / "occ.u32")
        scenes.pack_occupancy(scenes.lego_standin_density(R, seed=0)).astype("<u4").tofile(occ_path)
    f = float(np.float32(1.0) / np.tan(np.float32(0.5) * np.float32(0.6911112)))
    pose = ",".join(f"{v:.9g}" for v in la)
    one, one_raw = str(tmp_path / "one.ppm"), str(tmp_path / "one.f32")
    res = subprocess.run([_exe(), "frame", str(W), str(H), str(R), one, one_raw, occ_path, "1", pose, f"{f:.9g}"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    out, raw = str(tmp_path / "mgpu.ppm"), str(tmp_path / "mgpu.f32")
    res = subprocess.run([exe, str(W), str(H), str(R), out, raw, occ_path, devices, "3", pose, f"{f:.9g}"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    n = devices.count(",") + 1
    assert f"over {n} shards" in res.stdout and "0 overflowed frames" in res.stdout and "ms/frame" in res.stdout
    want = np.fromfile(one_raw, np.float32)
    got = np.fromfile(raw, np.float32)
    assert want.shape == got.shape == (W * H * 3,) and want.std() > 1e-3
    assert np.array_equal(got, want)
    assert open(out, "rb").read() == open(one, "rb").read()


def test_cpp_data_parallel_training_host_rehearsed_on_one_gpu(gpu, tmp_path):
    """examples/train_host_mgpu.cpp: ONE C++ process, the reference's 8 x 128 iteration with the ray batch split over N shards, each
    with its own replica, stream and buffers; per step every shard computes its gradient (rtxn_trace_grid / rtxn_scan_hits /
    rtxn_train_gradients), the gradients are summed on the root in shard order (peer copies + one add kernel of the host's own), the
    sum is copied back and every replica takes the same Adam step (loss scale x N).  With the shards on device 0 ("0,0", "0,0,0,0":
    the one-GPU rehearsal): the replicas end bit-identical to each other (the program checks and says so); the first step's summed
    gradient is the one-shard gradient of the same batch (each shard scales its loss by 1 / N, so every ray's fp16 loss gradient is the
    very number the one-shard step rounds to: what differs is the order of the fp32 sums, 1e-4 of the norm); and the parameters after
    eight Adam steps agree to what Adam makes of that (its first steps are lr x sign(g): entries with a gradient at rounding level
    can go either way).
    Unmeasured on more than one GPU."""
    exe = os.path.join(ROOT, "examples", "train_host_mgpu")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", ROOT, "examples/train_host_mgpu"])
    outs, grads = {}, {}
    for devices in ("0", "0,0", "0,0,0,0"):
        n = devices.count(",") + 1
        out, gout = str(tmp_path / f"dp_{n}.f32"), str(tmp_path / f"dpg_{n}.f32")
        res = subprocess.run([exe, "8", "2048", "8", devices, out, gout], capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout + res.stderr
        assert f"over {n} shard(s)" in res.stdout and "replicas bit-identical" in res.stdout, res.stdout
        outs[n], grads[n] = np.fromfile(out, np.float32), np.fromfile(gout, np.float32)
        assert np.isfinite(outs[n]).all() and np.isfinite(grads[n]).all()
    assert np.linalg.norm(grads[1]) > 0
    for n in (2, 4):
        assert np.linalg.norm(grads[n] - grads[1]) <= 1e-4 * np.linalg.norm(grads[1]), (n, np.linalg.norm(grads[n] - grads[1]) / np.linalg.norm(grads[1]))
        assert np.linalg.norm(outs[n] - outs[1]) <= 6e-3 * np.linalg.norm(outs[1])
