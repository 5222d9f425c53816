#!/usr/bin/env python3
"""The lean backward (dgrad chain + the three weight-gradient passes in one launch) against the CU split of the passes, one process,
splits interleaved, median of 9 single calls each (HIP events):  lean_split.py [samples] [split ...]   split = "s1,s2" of 32 slots"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rtx_nerf_amd import api, scenes
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4_695_827
splits = sys.argv[2:] or ["8,18", "8,20", "9,20", "9,21", "8,19", "0"]
W, L, E = 128, 8, 112
net = api.Network(n_neurons=W, n_hidden_layers=L)
wts = scenes.xavier_params_fp16(W, L, E, seed=3)
if os.environ.get("LEAN_ZERO_WEIGHTS"):      # same instruction stream on zeros (activations, masks and hidden dZ all zero): is the kernel power-limited?
    wts = np.zeros_like(wts)
net.set_params(torch.from_numpy(wts).cuda())
Sp = api.padded_samples(S)
g = torch.Generator(device="cuda").manual_seed(5)
encT = (torch.rand((E, Sp), device="cuda", generator=g) * 2 - 1).half()
encT[:, S:] = 0
out = torch.empty((S, 16), dtype=torch.float16, device="cuda")
dout = ((torch.rand((S, 4), device="cuda", generator=g) - 0.5) * 1e-3).half()
dparams = torch.zeros(net.n_params(), dtype=torch.float32, device="cuda")
ws = net.train_lean_workspace(S)
FOLDED = bool(os.environ.get("LEAN_FOLDED"))      # the kernels with sampler + encoder folded in, on random segments
if FOLDED:
    P = S // 32
    S = P * 32
    start = torch.rand((P, 3), device="cuda", generator=g) * 2 - 1
    end = start + (torch.rand((P, 3), device="cuda", generator=g) - 0.5) * 0.3
    view = torch.rand((P, 2), device="cuda", generator=g) * 3
    net.train_forward_lean_segments(start, end, view, P, 0, ws, out)
else:
    net.train_forward_lean(encT, S, ws, out)
times = {s: [] for s in splits}
for rep in range(10):
    for s in splits:
        os.environ["RTXN_LEAN_SPLIT"] = s
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if FOLDED:
            net.train_backward_lean_segments(start, end, view, P, 0, out, dout[:S], ws, dparams)
        else:
            net.train_backward_lean(encT, out, dout, S, ws, dparams)
        e1.record()
        torch.cuda.synchronize()
        if rep:
            times[s].append(e0.elapsed_time(e1))
for s in splits:
    v = np.array(times[s])
    print(f"split {s:6s}: backward median {np.median(v):.3f} ms, min {v.min():.3f}, max {v.max():.3f}", flush=True)
