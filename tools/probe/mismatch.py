#!/usr/bin/env python3
"""Where do two builds of the fused kernel disagree?  Runs rtxn_mlp_forward_segments on the same random segments under the
library given by RTXN_LIB_PATH and prints the mismatch pattern against the reference output saved by a run with --save."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rtx_nerf_amd import api, scenes
ap = argparse.ArgumentParser()
ap.add_argument("--save"); ap.add_argument("--ref"); ap.add_argument("--segments", type=int, default=3200)
a = ap.parse_args()
P = a.segments
g = torch.Generator(device="cuda").manual_seed(0)
sp = torch.rand((P, 3), device="cuda", generator=g) * 2 - 1
ep = sp + (torch.rand((P, 3), device="cuda", generator=g) - 0.5) * 0.03
sv = torch.rand((P, 2), device="cuda", generator=g) * 3.0
total = torch.tensor([P], dtype=torch.int32, device="cuda")
net = api.Network(n_neurons=128, n_hidden_layers=8)
net.set_params(torch.from_numpy(scenes.xavier_params_fp16(128, 8, net.encoded_width())).cuda())
outs = []
for it in range(3):
    rad = torch.empty((P * 32, 4), device="cuda")
    net.forward_segments(sp, ep, sv, total, P, rad, None)
    torch.cuda.synchronize()
    outs.append(rad.cpu().numpy())
print("run-to-run identical:", [bool((outs[0] == o).all()) for o in outs[1:]])
if a.save:
    np.save(a.save, outs[0])
if a.ref:
    ref = np.load(a.ref)
    bad = (outs[0] != ref).any(axis=1)
    print("mismatching samples:", int(bad.sum()), "of", bad.size, "max |d|", float(np.abs(outs[0] - ref).max()))
    idx = np.nonzero(bad)[0]
    if idx.size:
        print("by sample % 64 // 16 (column tile):", np.bincount((idx % 64) // 16, minlength=4))
        print("by wave (sample // 64 % 8):", np.bincount((idx // 64) % 8, minlength=8))
        print("by tile (sample // 512), first 20 tiles with errors:", np.unique(idx // 512)[:20], "n tiles:", np.unique(idx // 512).size)
        print("by channel:", (outs[0] != ref).sum(axis=0))
