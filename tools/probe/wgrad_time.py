#!/usr/bin/env python3
"""Time of the 8x128 saved-activation backward (mlp_bwd_kernel + wgrad kernels) at a large batch, by HIP events, and a
checksum of the weight gradient (A/B of wgrad_lds_kernel builds: RTXN_LIB_PATH=...).  wgrad_time.py [samples]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rtx_nerf_amd import api, scenes
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4_700_000
W, L, E = 128, 8, 112
net = api.Network(n_neurons=W, n_hidden_layers=L)
net.set_params(torch.from_numpy(scenes.xavier_params_fp16(W, L, E, seed=3)).cuda())
Sp = api.padded_samples(S)
g = torch.Generator(device="cuda").manual_seed(5)
encT = (torch.rand((E, Sp), device="cuda", generator=g) * 2 - 1).half()
ws = net.train_workspace(S)
out = torch.empty((S, 16), dtype=torch.float16, device="cuda")
dout = torch.zeros((S, 4), dtype=torch.float16, device="cuda")
dout.copy_((torch.rand((S, 4), device="cuda", generator=g) - 0.5) * 1e-3)
n_params = E * W + (L - 1) * W * W + 16 * W
dparams = torch.zeros(n_params, dtype=torch.float32, device="cuda")
net.train_forward(encT, S, ws, out)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


dparams.zero_()
net.train_backward(encT, out, dout, S, ws, dparams)
torch.cuda.synchronize()
ref = dparams.double().clone()
t = timed(lambda: net.train_backward(encT, out, dout, S, ws, dparams))
print(f"{os.environ.get('RTXN_LIB_PATH', 'librtxn.so')}: {S} samples: backward (dgrad + wgrad) {t:.3f} ms; "
      f"|dW|_1 = {ref.abs().sum().item():.9e}, sum = {ref.sum().item():.9e}")
