#!/usr/bin/env python3
"""Run the fused sampler+encode+MLP kernel several times on the same 3 M random segments and count values that differ
between runs (a hazard in hand-scheduled code shows up as rare, timing-dependent differences)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rtx_nerf_amd import api, scenes

torch.cuda.set_device(0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 128
g = torch.Generator(device="cuda").manual_seed(0)
sp = torch.rand((P, 3), device="cuda", generator=g) * 2 - 1
ep = sp + (torch.rand((P, 3), device="cuda", generator=g) - 0.5) * 0.03
sv = torch.rand((P, 2), device="cuda", generator=g) * 3.0
total = torch.tensor([P], dtype=torch.int32, device="cuda")
net = api.Network(n_neurons=W, n_hidden_layers=8)
net.set_params(torch.from_numpy(scenes.xavier_params_fp16(W, 8, net.encoded_width())).cuda())
outs = []
for i in range(4):
    rad = torch.empty((P * 32, 4), device="cuda")
    net.forward_segments(sp, ep, sv, total, P, rad, None)
    torch.cuda.synchronize()
    outs.append(rad)
bad = 0
for o in outs[1:]:
    d = (o != outs[0])
    n = int(d.sum())
    bad += n
    if n:
        idx = d.nonzero()[:5].cpu().numpy()
        print("differs:", n, "first", idx.tolist(), "max abs diff", float((o - outs[0]).abs().max()))
        rows = d.any(dim=1).nonzero().flatten().cpu().numpy()
        import numpy as np
        tiles = np.unique(rows // 256)
        grid = int(os.environ.get("GRID", "512"))
        print("  samples", rows.size, "tiles", tiles.size, "block ids", (tiles % grid)[:12].tolist(), "iterations", (tiles // grid)[:12].tolist())
        for t in tiles[:6]:
            r = rows[rows // 256 == t] % 256
            print("   tile", int(t), "cols", r.size, "waves", np.unique(r // 64).tolist(), "ct", np.unique(r % 64 // 32).tolist(), "min/max col", int(r.min()), int(r.max()))
print("DETERMINISTIC" if bad == 0 else f"NONDETERMINISTIC ({bad} values)")
sys.exit(0 if bad == 0 else 1)
