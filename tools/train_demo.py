#!/usr/bin/env python3
"""Train the hash-grid (or frequency) model against an analytic teacher field and report
PSNR on a held-out pose -- exercises the whole training path of rtx_nerf_amd.train.Trainer.
  python tools/train_demo.py [--steps 300] [--encoding hash|freq] [--grid 32] [--res 64]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rtx_nerf_amd import scenes
from rtx_nerf_amd.train import Trainer, camera_rays


teacher_field = scenes.teacher_field   # kept under this name for the tests that import it from here


def run(steps=300, encoding="hash", grid=32, res=64, batch=4096, n_poses=12, seed=0, verbose=True, neurons=64, layers=2,
        hash_levels=8, hash_log2=15, hash_base=8):
    torch.cuda.set_device(0)
    dense = scenes.sphere_density(grid, 0.72)
    occ = torch.from_numpy(scenes.pack_occupancy(dense).view(np.int32).copy()).cuda()
    hashgrid = dict(n_levels=hash_levels, n_features=2, log2_hashmap_size=hash_log2, base_resolution=hash_base,
                    per_level_scale=1.5)
    tr = Trainer(grid, occ, encoding=encoding, n_neurons=neurons, n_hidden_layers=layers, hashgrid=hashgrid,
                 batch_rays=max(batch, res * res), max_segments=max(batch, res * res) * 40, lr=1e-2 if encoding == "hash" else 2e-3,
                 loss_scale=128.0, density_scale=150.0, mode="nerf", seed=seed)
    focal = scenes.lego_focal_length(True)
    rays_o, rays_d, targets = [], [], []
    for i in range(n_poses):
        la = scenes.pose_spherical(360.0 * i / n_poses, -20.0 - 25.0 * (i % 3), origin_scale=10.0)
        o, d = camera_rays(la, focal, res, res)
        pix = tr.render_rays(o, d, radiance_fn=teacher_field).clone()
        rays_o.append(o); rays_d.append(d); targets.append(pix)
    rays_o, rays_d, targets = torch.cat(rays_o), torch.cat(rays_d), torch.cat(targets)
    la_test = scenes.pose_spherical(77.0, -33.0, origin_scale=10.0)
    o_t, d_t = camera_rays(la_test, focal, res, res)
    gt_t = tr.render_rays(o_t, d_t, radiance_fn=teacher_field).clone()

    def psnr():
        pred = tr.render_rays(o_t, d_t)
        mse = float(((pred - gt_t) ** 2).mean())
        return 10 * np.log10(1.0 / max(mse, 1e-12))

    g = torch.Generator(device="cuda").manual_seed(seed)
    p0 = psnr()
    losses = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        idx = torch.randint(0, rays_o.shape[0], (batch,), device="cuda", generator=g)
        loss = tr.step(rays_o[idx].contiguous(), rays_d[idx].contiguous(), targets[idx].contiguous())
        if it % 50 == 0 or it == steps - 1:
            losses.append(float(loss.item()))
            if verbose:
                print(f"step {it:4d} loss {losses[-1]:.6f}", flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p1 = psnr()
    if verbose:
        print(f"encoding={encoding} steps={steps} batch={batch}: PSNR {p0:.2f} -> {p1:.2f} dB; "
              f"{steps * batch / dt / 1e6:.3f} Mrays/s trained ({1e3 * dt / steps:.2f} ms/step)")
    return p0, p1, losses


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--encoding", default="hash")
    ap.add_argument("--grid", type=int, default=32)
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--neurons", type=int, default=64)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--hash-levels", type=int, default=8)
    ap.add_argument("--hash-log2", type=int, default=15)
    ap.add_argument("--hash-base", type=int, default=8)
    a = ap.parse_args()
    run(a.steps, a.encoding, a.grid, a.res, a.batch, neurons=a.neurons, layers=a.layers, hash_levels=a.hash_levels,
        hash_log2=a.hash_log2, hash_base=a.hash_base)
