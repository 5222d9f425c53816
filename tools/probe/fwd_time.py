#!/usr/bin/env python3
"""Kernel time of the 128-wide training forward at a large batch: saving (activations + masks written) against outputs-only
(the same matrix work, 48 B per sample written), by HIP events.  fwd_time.py [samples]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rtx_nerf_amd import api, scenes
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4_700_000
W, L, E = 128, 8, 112
net = api.Network(n_neurons=W, n_hidden_layers=L)
net.set_params(torch.from_numpy(scenes.xavier_params_fp16(W, L, E, seed=3)).cuda())
Sp = api.padded_samples(S)
encT = (torch.rand((E, Sp), device="cuda") * 2 - 1).half()
ws = net.train_workspace(S)
out = torch.empty((S, 16), dtype=torch.float16, device="cuda")


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t_save = timed(lambda: net.train_forward(encT, S, ws, out))
t_out = timed(lambda: net.train_forward_outputs(encT, S, out))
flop = 262144 * S
print(f"{S} samples: saving forward {t_save:.3f} ms ({flop / t_save / 1e9:.0f} TFLOP/s, {2448 * S / t_save / 1e6:.0f} GB/s of its 2,448 B/sample), "
      f"outputs-only {t_out:.3f} ms ({flop / t_out / 1e9:.0f} TFLOP/s)")
