#!/usr/bin/env python3
"""Fuzz the ragged-CSR stages against the oracle: random batch sizes, hit distributions (empty rays, one huge ray,
all-empty batches) and modes for scan, sampler (4 types: bit-exact), compositor forward (2 modes, 1e-5) and backward
(COMPAT: <= 1 half-ulp; NERF: 2e-3 relative).   python tools/fuzz_stages.py [--iters 200] [--seed 0]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oracle as O
from rtx_nerf_amd import api

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
torch.cuda.set_device(0)
K = 32


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def half_ulps(got, want):
    g = got.view(np.int16).astype(np.int32)
    w = want.view(np.int16).astype(np.int32)
    g = np.where(g < 0, -(g & 0x7fff), g)
    w = np.where(w < 0, -(w & 0x7fff), w)
    return 0 if g.size == 0 else int(np.abs(g - w).max())


bad = 0
for it in range(a.iters):
    B = int(rng.choice([1, 2, 3, 63, 64, 65, 255, 256, 1000, 4097, int(rng.integers(1, 20000))]))
    kind = rng.integers(0, 5)
    if kind == 0:
        nh = np.zeros(B, np.int32)                                     # nothing hit
    elif kind == 1:
        nh = (rng.random(B) < 0.1).astype(np.int32) * rng.integers(1, 4, B).astype(np.int32)   # sparse
    elif kind == 2:
        nh = rng.integers(0, 12, B).astype(np.int32)
    elif kind == 3:
        nh = np.zeros(B, np.int32)
        nh[rng.integers(0, B)] = int(rng.integers(1, 400))             # one long ray
    else:
        nh = rng.integers(0, 60, B).astype(np.int32) * (rng.random(B) < 0.5)
        nh = nh.astype(np.int32)
    idx = np.concatenate([[0], np.cumsum(nh)[:-1]]).astype(np.int32)
    P = int(nh.sum())
    Pm = max(P, 1)
    errs = []
    # scan
    gi, gt = api.scan_hits(dev(nh))
    if not (np.array_equal(gi.cpu().numpy(), idx) and int(gt.item()) == P):
        errs.append("scan")
    sp = rng.uniform(-1, 1, (Pm, 3)).astype(np.float32)
    ep = (sp + rng.uniform(-0.05, 0.05, (Pm, 3))).astype(np.float32)
    vd = rng.uniform(-3.1, 3.1, (B, 2)).astype(np.float32)
    # sampler
    stype = int(rng.integers(0, 4))
    samples = torch.full((Pm * K, 5), -7.0, device="cuda")
    tv = torch.full((Pm * K,), -7.0, device="cuda")
    api.launchSampler(dev(sp), dev(ep), dev(vd), tv, samples, B, 8, dev(nh), dev(idx), stype)
    ws, wt = O.sample(sp[:P], ep[:P], vd, nh, idx, stype)
    if not (np.array_equal(samples.cpu().numpy()[:P * K], ws) and np.array_equal(tv.cpu().numpy()[:P * K], wt)
            and np.all(samples.cpu().numpy()[P * K:] == -7.0)):
        errs.append(f"sampler type {stype}")
    # compositor
    rad = rng.uniform(0, 1, (Pm * K, 4)).astype(np.float32)
    g = rng.standard_normal((B, 3)).astype(np.float16)
    t_compat = np.tile(((np.arange(K) + 1) / K).astype(np.float32), Pm)
    pix = torch.full((B, 3), -1.0, device="cuda")
    api.launch_volrender_cuda(None, dev(rad), dev(nh), dev(idx), dev(t_compat), B, K, pix)
    want = O.volrender_fwd(rad, nh, idx, t_compat, K=K)
    if not (np.abs(pix.cpu().numpy() - want).max() <= 1e-5 and np.all(pix.cpu().numpy()[nh == 0] == 0.0)):
        errs.append("volrender fwd compat")
    out = torch.zeros((Pm * K, 4), dtype=torch.float16, device="cuda")
    api.launch_volrender_backward_cuda(None, dev(g), dev(rad), dev(t_compat), dev(nh), dev(idx), B, K, out)
    wb = O.volrender_bwd(g, rad, t_compat, nh, idx, K=K)
    if half_ulps(out.cpu().numpy()[:P * K], wb[:P * K]) > 1:
        errs.append("volrender bwd compat")
    rad_n = rad.copy()
    rad_n[:, 3] *= 20.0
    step = rng.uniform(0.001, 0.02, Pm * K).astype(np.float32)
    api.launch_volrender_cuda(None, dev(rad_n), dev(nh), dev(idx), dev(step), B, K, pix, mode=api.VR_NERF)
    want = O.volrender_fwd_nerf(rad_n, nh, idx, step, K=K)
    if not np.abs(pix.cpu().numpy() - want).max() <= 2e-5:
        errs.append("volrender fwd nerf")
    api.launch_volrender_backward_cuda(None, dev(g), dev(rad_n), dev(step), dev(nh), dev(idx), B, K, out, mode=api.VR_NERF)
    wb = O.volrender_bwd_nerf(g, rad_n, step, nh, idx, K=K).astype(np.float32)
    gb = out.cpu().numpy()[:P * K].astype(np.float32)
    if P and not np.all(np.abs(gb - wb[:P * K]) <= 2e-3 * np.abs(wb[:P * K]) + 2e-4):
        errs.append("volrender bwd nerf")
    if errs:
        bad += 1
        print(f"MISMATCH it={it} B={B} kind={kind} P={P}: {errs}", flush=True)
print(f"fuzz_stages: {a.iters} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
