"""N>1 path on CPU: two gloo ranks each render their row-interleaved ray shard
(the oracle stands in for the GPU stages; the window arithmetic, the gather and
the reassembly are the product's rtx_nerf_amd.shard), and rank 0's assembled
image must equal the single-process frame bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rtx_nerf_amd import scenes
from rtx_nerf_amd.shard import RowShard

W, H, R = 20, 13, 16     # H not divisible by world: ragged shards + padding


def _scene():
    import oracle as O
    cfg = O.mlp_cfg(n_neurons=64, n_hidden_layers=2)
    params = scenes.xavier_params_fp16(64, 2, O.mlp_enc_padded(cfg), seed=11)
    occ = scenes.pack_occupancy(scenes.sphere_density(R, 0.7))
    la = scenes.pose_spherical(35.0, -25.0, origin_scale=10.0)
    return O, cfg, params, occ, la, scenes.lego_focal_length(True)


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        O, cfg, params, occ, la, f = _scene()
        sh = RowShard(W, H, rank, world)
        # the traversal's own window arithmetic (oracle.trace implements rtxn_trace_params' formula)
        tr = O.trace(look_at=la, focal=f, aspect=W / H, W=W, H=H, R=R, occ=occ, mode=1, ray_begin=sh.ray_begin,
                     ray_count=sh.n_local, window_chunk=sh.window[0], window_stride=sh.window[1], count_only=True)
        ids = np.array([sh.local_to_global(i) for i in range(sh.n_local)], np.uint32)
        full = O.trace(look_at=la, focal=f, aspect=W / H, W=W, H=H, R=R, occ=occ, mode=1, count_only=True)
        np.testing.assert_array_equal(tr["num_hits"], full["num_hits"][ids])
        pix, _ = O.render(la, f, W / H, W, H, R, occ, 1, cfg, params, ids)
        local = torch.zeros((sh.n_max, 3))
        local[:sh.n_local] = torch.from_numpy(pix)
        glist = [torch.empty((sh.n_max, 3)) for _ in range(world)] if rank == 0 else None
        work = sh.gather(local, glist, async_op=True)
        work.wait()
        dist.barrier()
        if rank == 0:
            img = sh.assemble(glist).numpy()
            np.save(out_path, img)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_render_equals_single_process(tmp_path, world):
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    img = np.load(out)
    O, cfg, params, occ, la, f = _scene()
    want, _ = O.render(la, f, W / H, W, H, R, occ, 1, cfg, params, np.arange(W * H))
    np.testing.assert_array_equal(img.reshape(-1, 3), want)
    assert img.std() > 0


def test_rowshard_partition_is_exact():
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            sh = RowShard(W, H, r, world)
            assert sh.n_local <= sh.n_max
            seen += [sh.local_to_global(i) for i in range(sh.n_local)]
        assert sorted(seen) == list(range(W * H))


def test_ray_dataset_matches_oracle_raygen(oracle):
    """RayDataset (device-resident counterpart of main.cu:463-543) must hold exactly the rays the traversal's own
    ray generation would produce for each pose, in pose-major pixel order, with the frames' pixels alongside."""
    import numpy as np
    from rtx_nerf_amd import loader, scenes
    from rtx_nerf_amd.train import RayDataset
    W, H = 12, 8
    poses = np.stack([scenes.pose_spherical(40.0 * i, -25.0, origin_scale=10.0) for i in range(3)]).astype(np.float32)
    imgs = np.random.default_rng(0).random((3, H, W, 3), dtype=np.float32)
    ds = loader.ImageDataset(imgs, poses, 100.0, W, H, 3, scenes.LEGO_CAMERA_ANGLE_X)
    rays, focal = RayDataset.from_images(ds, device="cpu")
    assert rays.n == 3 * W * H
    for i in range(3):
        r = oracle.trace(look_at=poses[i], focal=focal, aspect=W / H, W=W, H=H, R=4, mode=1)
        sl = slice(i * W * H, (i + 1) * W * H)
        np.testing.assert_allclose(rays.rays_o[sl].numpy(), r["origins"], rtol=0, atol=1e-7)
        d = rays.rays_d[sl].numpy()
        theta_phi = np.stack([np.arccos(np.clip(d[:, 2], -1, 1)), np.arctan2(d[:, 1], d[:, 0])], 1)
        np.testing.assert_allclose(theta_phi, r["view_dirs"], rtol=0, atol=2e-6)
        np.testing.assert_array_equal(rays.pixels[sl].numpy(), imgs[i].reshape(-1, 3))
    o, d, p = rays.sample_batch(64)
    assert o.shape == (64, 3) and d.shape == (64, 3) and p.shape == (64, 3)
