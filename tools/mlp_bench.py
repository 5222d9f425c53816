#!/usr/bin/env python3
"""Isolated timing of the fused sampler+encode+MLP kernel (rtxn_mlp_forward_segments)
on random packed segments -- for A/B-ing kernel variants and for PMC runs.
  python tools/mlp_bench.py [--segments N] [--neurons 128] [--layers 8] [--iters 10]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rtx_nerf_amd import api, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--segments", type=int, default=3_000_000)
ap.add_argument("--neurons", type=int, default=128)
ap.add_argument("--layers", type=int, default=8)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--compact", action="store_true", help="the render pipeline's launch: half4 outputs, no t_vals (rtxn_mlp_forward_segments_compact)")
ap.add_argument("--weights", choices=["xavier", "zero", "small"], default="xavier", help="power experiment: all-zero weights (no operand toggling "
                "in the matrix core: what the kernel does when the clock is not held down) or tiny ones; timing only")
ap.add_argument("--samples", action="store_true", help="time rtxn_mlp_forward_radiance on a materialised float[N][5] batch (the bench.py default path)")
args = ap.parse_args()

torch.cuda.set_device(0)
P = args.segments
g = torch.Generator(device="cuda").manual_seed(0)
sp = torch.rand((P, 3), device="cuda", generator=g) * 2 - 1
ep = sp + (torch.rand((P, 3), device="cuda", generator=g) - 0.5) * 0.03
n_rays = max(1, P // 5)
vd = torch.rand((n_rays, 2), device="cuda", generator=g) * 3.0
seg_ray = (torch.arange(P, device="cuda", dtype=torch.int32) // 5).clamp_(max=n_rays - 1)
sv = vd[seg_ray.long()].contiguous()
total = torch.tensor([P], dtype=torch.int32, device="cuda")
net = api.Network(n_neurons=args.neurons, n_hidden_layers=args.layers)
w_np = scenes.xavier_params_fp16(args.neurons, args.layers, net.encoded_width())
if args.weights == "zero":
    w_np = np.zeros_like(w_np)
elif args.weights == "small":
    w_np = (w_np.astype(np.float32) * 1e-3).astype(np.float16)
net.set_params(torch.from_numpy(w_np).cuda())
if args.compact:
    rad16 = torch.empty((P * 32, 4), dtype=torch.float16, device="cuda")

    def run():
        net.forward_segments_compact(sp, ep, sv, total, P, rad16)
elif args.samples:
    t = (torch.arange(32, device="cuda", dtype=torch.float32) / 32)[None, :, None]
    xyz = sp[:, None, :] + t * (ep - sp)[:, None, :]
    batch = torch.cat([xyz, sv[:, None, :].expand(P, 32, 2)], dim=2).reshape(P * 32, 5).contiguous()
    rad = torch.empty((P * 32, 4), device="cuda")

    def run():
        net.forward_radiance(batch, rad)
else:
    rad = torch.empty((P * 32, 4), device="cuda")
    tv = torch.empty(P * 32, device="cuda")

    def run():
        net.forward_segments(sp, ep, sv, total, P, rad, tv)
for _ in range(2):
    run()
torch.cuda.synchronize()
ms = []
for _ in range(args.iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
ms = np.array(ms)
fl = net.flops_per_sample() * P * 32
print(f"segments {P} samples {P*32} | ms median {np.median(ms):.3f} min {ms.min():.3f} | "
      f"TFLOP/s median {fl/np.median(ms)/1e9:.1f} best {fl/ms.min()/1e9:.1f}")
