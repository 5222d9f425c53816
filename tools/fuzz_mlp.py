#!/usr/bin/env python3
"""Fuzz the fused forward kernels against the oracle: random widths (64/128/256), hidden-layer counts 1..9, direction
frequencies 4/12, batch sizes around tile boundaries, both input modes.  Checks: half output vs the oracle (1e-2, mean
1e-3), segment path == sampler + forward bit for bit, untouched memory beyond the batch.
  python tools/fuzz_mlp.py [--iters 60] [--seed 0]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oracle as O
from rtx_nerf_amd import api, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
torch.cuda.set_device(0)


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


bad = 0
for it in range(a.iters):
    W = int(rng.choice([64, 128, 256]))
    nhid = int(rng.integers(1, 10))
    df = int(rng.choice([4, 12])) if W != 256 else 12
    n = int(rng.choice([1, 31, 32, 33, 255, 256, 257, 511, 512, 513, 1023, 1025, int(rng.integers(1, 3000))]))
    cfg = O.mlp_cfg(n_neurons=W, n_hidden_layers=nhid, n_dir_freqs=df)
    params = scenes.xavier_params_fp16(W, nhid, O.mlp_enc_padded(cfg), seed=it)
    x = np.concatenate([rng.uniform(-1, 1, (n, 3)), rng.uniform(0, 3.1416, (n, 1)), rng.uniform(-3.1416, 3.1416, (n, 1))],
                       axis=1).astype(np.float32)
    net = api.Network(n_neurons=W, n_hidden_layers=nhid, n_dir_freqs=df)
    net.set_params(dev(params))
    errs = []
    out = torch.full((n + 7, 16), -5.0, dtype=torch.float16, device="cuda")
    net.forward(dev(x), out[:n])
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float32)
    want = O.mlp_forward(cfg, params, x).astype(np.float32)
    if not (np.isfinite(got).all() and np.abs(got[:n] - want).max() <= 1e-2 and np.abs(got[:n] - want).mean() < 1e-3):
        errs.append(f"half output: max {np.abs(got[:n] - want).max():.4g}")
    if not np.all(got[n:] == -5.0):
        errs.append("wrote beyond the batch")
    # segment path vs sampler + forward
    B = int(rng.integers(1, 120))
    nh = rng.integers(0, 6, B).astype(np.int32)
    idx = np.concatenate([[0], np.cumsum(nh)[:-1]]).astype(np.int32)
    P = int(nh.sum())
    if P:
        sp = rng.uniform(-1, 1, (P, 3)).astype(np.float32)
        ep = (sp + rng.uniform(-0.05, 0.05, (P, 3))).astype(np.float32)
        vd = rng.uniform(-3.1, 3.1, (B, 2)).astype(np.float32)
        seg_ray = np.repeat(np.arange(B, dtype=np.int32), nh)
        cap = P + 5
        rad = torch.full((cap * 32, 4), -3.0, device="cuda")
        tv = torch.full((cap * 32,), -3.0, device="cuda")
        pad3, pad2 = np.zeros((5, 3), np.float32), np.zeros((5, 2), np.float32)
        net.forward_segments(dev(np.concatenate([sp, pad3])), dev(np.concatenate([ep, pad3])), dev(np.concatenate([vd[seg_ray], pad2])),
                             torch.tensor([P], dtype=torch.int32, device="cuda"), cap, rad, tv)
        s_d = torch.zeros((P * 32, 5), device="cuda")
        t_d = torch.zeros((P * 32,), device="cuda")
        api.launchSampler(dev(sp), dev(ep), dev(vd), t_d, s_d, B, 8, dev(nh), dev(idx), 0)
        rad2 = net.forward_radiance(s_d)
        torch.cuda.synchronize()
        if not torch.equal(rad[:P * 32], rad2):
            errs.append("segment path != sampler + forward")
        if not (torch.all(rad[P * 32:] == -3.0) and torch.all(tv[P * 32:] == -3.0)):
            errs.append("segment path wrote beyond *total")
    if errs:
        bad += 1
        print(f"MISMATCH it={it} W={W} hidden={nhid} dir_freqs={df} n={n} P={P}: {errs}", flush=True)
print(f"fuzz_mlp: {a.iters} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
