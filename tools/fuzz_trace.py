#!/usr/bin/env python3
"""Fuzz the traversal kernel against the oracle: random poses, grid sizes, occupancies and modes; every
num_hits / start / end / t must match bit for bit (theta/phi to 2e-6).  python tools/fuzz_trace.py [--iters 200]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oracle as O
from rtx_nerf_amd import api, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
torch.cuda.set_device(0)
bad = 0
for it in range(a.iters):
    R = int(rng.choice([4, 8, 12, 16, 20, 32, 48, 64, 100, 128]))
    W, H = int(rng.integers(8, 64)), int(rng.integers(8, 64))
    mode = int(rng.integers(0, 2))
    la = scenes.pose_spherical(rng.uniform(0, 360), rng.uniform(-89, 20), radius=rng.uniform(0.2, 6.0), origin_scale=10.0)
    if rng.random() < 0.2:       # axis-aligned views: rays parallel to grid planes, exact ties
        la = scenes.pose_spherical(float(rng.choice([0, 90, 180, 270])), float(rng.choice([0, -90])), radius=rng.uniform(1.5, 5), origin_scale=10.0)
    f = float(rng.uniform(0.5, 4.0))
    dense = None
    words = None
    r = rng.random()
    if r < 0.6:
        dense = rng.random((R, R, R)) < rng.uniform(0.0, 0.3)
    elif r < 0.8:
        dense = scenes.sphere_density(R, rng.uniform(0.2, 0.9))
    if dense is not None:
        words = scenes.pack_occupancy(dense)
    use_coarse = dense is not None and mode == 1 and R % 4 == 0 and rng.random() < 0.7
    S = 3 * R
    want = O.trace(look_at=la, focal=f, aspect=W / H, W=W, H=H, R=R, occ=words, mode=mode, S=S)
    n = W * H
    occ = None if words is None else torch.from_numpy(words.view(np.int32).copy()).cuda()
    coarse = api.build_occupancy_mip(occ, R) if use_coarse else None
    bricks = api.build_occupancy_bricks(occ, R) if (use_coarse and rng.random() < 0.7) else None
    sup = api.build_occupancy_mip(coarse, R // 4) if (use_coarse and R % 16 == 0 and rng.random() < 0.8) else None
    nh = torch.zeros(n, dtype=torch.int32, device="cuda")
    vd = torch.zeros((n, 2), device="cuda")
    og = torch.zeros((n, 3), device="cuda")
    sp = torch.full((n * S, 3), -2.0, device="cuda")
    ep = torch.full((n * S, 3), -2.0, device="cuda")
    t0 = torch.full((n * S,), -2.0, device="cuda")
    t1 = torch.full((n * S,), -2.0, device="cuda")
    # sub-ray walk (DDA only): Q lanes per ray, counting pass first (it fills sub_hits), then the write pass
    Q = int(rng.choice([1, 1, 2, 4, 8, 16, 32, 64])) if mode == 1 else 1
    sub = torch.zeros(n * Q, dtype=torch.int32, device="cuda") if Q > 1 else None
    la_d = torch.from_numpy(la.reshape(16)).cuda()
    common = dict(grid_res=R, occupancy=occ, occupancy_coarse=coarse, occupancy_bricks=bricks, occupancy_super=sup, mode=mode,
                  num_hits=nh, intersection_arr_size=S, sub_rays=Q, sub_hits=sub)
    if Q > 1:
        api.trace_grid(la_d, f, W / H, W, H, **common)
        assert np.array_equal(nh.cpu().numpy(), want["num_hits"]), "counting pass"
        nh.zero_()
    api.trace_grid(la_d, f, W / H, W, H, ray_origins=og, viewing_direction=vd, start_points=sp, end_points=ep, t_start=t0, t_end=t1,
                   **common)
    torch.cuda.synchronize()
    finite = all(np.isfinite(x.cpu().numpy()).all() for x in (sp, ep, t0, t1, og, vd))
    ok = (finite and np.array_equal(nh.cpu().numpy(), want["num_hits"]) and np.array_equal(sp.cpu().numpy(), want["start"])
          and np.array_equal(ep.cpu().numpy(), want["end"]) and np.array_equal(t0.cpu().numpy(), want["t_start"])
          and np.array_equal(t1.cpu().numpy(), want["t_end"]) and np.array_equal(og.cpu().numpy(), want["origins"])
          and np.abs(vd.cpu().numpy() - want["view_dirs"]).max() <= 2e-6 and want["num_hits"].max() <= 3 * R - 2)
    if not ok:
        bad += 1
        d = np.nonzero(nh.cpu().numpy() != want["num_hits"])[0]
        print(f"MISMATCH it={it} Q={Q} R={R} {W}x{H} mode={mode} occ={'none' if words is None else 'yes'} coarse={use_coarse} "
              f"rays with different num_hits: {d[:8]}", flush=True)
        for name, g_, w_ in (("start", sp, want["start"]), ("end", ep, want["end"]), ("t_start", t0, want["t_start"]),
                             ("t_end", t1, want["t_end"]), ("origins", og, want["origins"]), ("view", vd, want["view_dirs"])):
            gg = g_.cpu().numpy().reshape(w_.shape)
            neq = np.nonzero(gg != w_)
            if len(neq[0]):
                k = neq[0][0]
                print(f"   {name}: {len(neq[0])} differing entries; first at {k}: gpu {gg[k]} oracle {w_[k]}", flush=True)
                if name in ("start", "end", "t_start", "t_end"):
                    ray = k // S
                    o_, d_, v_ = O.make_ray(la, f, W / H, W, H, ray % W, ray // W)
                    print(f"   ray {ray}: o={o_} d={d_} nh={want['num_hits'][ray]} slot={k % S}", flush=True)
print(f"fuzz_trace: {a.iters} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
