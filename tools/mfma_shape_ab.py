#!/usr/bin/env python3
"""A/B of the fused inference kernel's two MFMA shapes in ONE process, interleaved rounds (cdna_hip_programming.md rule 24):
  32: mlp_fwd_kernel   (v_mfma_f32_32x32x16_f16)      16: mlp_fwd16_kernel (v_mfma_f32_16x16x32_f16)
on the render pipeline's compact path (rtxn_mlp_forward_segments_compact) over random packed segments -- random data: the chip
is power-limited under this kernel and zero data would rank the shapes by cycles only (MI355X_MICROARCH.md, give-back 7).
  python tools/mfma_shape_ab.py [--segments 3000000] [--neurons 128] [--layers 8] [--rounds 8]
Prints per-shape median / min kernel ms and PFLOP/s, and the max |difference| of the two kernels' outputs."""
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
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--samples", action="store_true", help="materialised float[N][5] input (rtxn_mlp_forward_radiance) instead of segments")
args = ap.parse_args()
torch.cuda.set_device(0)
P = args.segments
g = torch.Generator(device="cuda").manual_seed(0)
sp = torch.rand((P, 3), device="cuda", generator=g) * 2 - 1
ep = sp + (torch.rand((P, 3), device="cuda", generator=g) - 0.5) * 0.03
sv = (torch.rand((max(1, P // 5), 2), device="cuda", generator=g) * 3.0).repeat_interleave(5, dim=0)[:P].contiguous()
total = torch.tensor([P], dtype=torch.int32, device="cuda")
params = torch.from_numpy(scenes.xavier_params_fp16(args.neurons, args.layers, 112)).cuda()
nets, outs = {}, {}
for shape in ("32", "16"):
    os.environ["RTXN_MFMA_SHAPE"] = shape          # read by rtxn_mlp_create
    nets[shape] = api.Network(n_neurons=args.neurons, n_hidden_layers=args.layers)
    nets[shape].set_params(params)
    outs[shape] = torch.empty((P * 32, 4), dtype=torch.float16, device="cuda")
if args.samples:
    t = (torch.arange(32, device="cuda", dtype=torch.float32) / 32)[None, :, None]
    batch = torch.cat([sp[:, None, :] + t * (ep - sp)[:, None, :], sv[:, None, :].expand(P, 32, 2)], dim=2).reshape(P * 32, 5).contiguous()
    rad = {k: torch.empty((P * 32, 4), device="cuda") for k in nets}


def run(shape):
    if args.samples:
        nets[shape].forward_radiance(batch, rad[shape])
    else:
        nets[shape].forward_segments_compact(sp, ep, sv, total, P, outs[shape])


for s in nets:
    for _ in range(2):
        run(s)
torch.cuda.synchronize()
ms = {s: [] for s in nets}
for r in range(args.rounds):
    for s in (("32", "16") if r % 2 == 0 else ("16", "32")):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(s)
        e1.record()
        torch.cuda.synchronize()
        ms[s].append(e0.elapsed_time(e1))
fl = nets["32"].flops_per_sample() * P * 32
for s in ("32", "16"):
    m = np.array(ms[s])
    print(f"shape {s}: ms median {np.median(m):.3f} min {m.min():.3f} | PFLOP/s median {fl / np.median(m) / 1e12:.3f} best {fl / m.min() / 1e12:.3f}")
a, b = (rad["32"], rad["16"]) if args.samples else (outs["32"].float(), outs["16"].float())
print(f"ratio 16/32 (median time): {np.median(ms['16']) / np.median(ms['32']):.4f}   max |out16 - out32| = {float((a - b).abs().max()):.3e}"
      f"   mean |diff| = {float((a - b).abs().mean()):.3e}")
