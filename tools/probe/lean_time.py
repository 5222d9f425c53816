#!/usr/bin/env python3
"""8x128 training MLP kernels at a large batch, saved-activation path against the lean path (HIP events), with the relative
difference of the weight gradients.  lean_time.py [samples]   (RTXN_LEAN_PASSES=3: the three-pass weight gradient)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rtx_nerf_amd import api, scenes
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4_695_827
W, L, E = 128, 8, 112
net = api.Network(n_neurons=W, n_hidden_layers=L)
net.set_params(torch.from_numpy(scenes.xavier_params_fp16(W, L, E, seed=3)).cuda())
Sp = api.padded_samples(S)
g = torch.Generator(device="cuda").manual_seed(5)
encT = (torch.rand((E, Sp), device="cuda", generator=g) * 2 - 1).half()
encT[:, S:] = 0
out = torch.empty((S, 16), dtype=torch.float16, device="cuda")
dout = torch.zeros((S, 4), dtype=torch.float16, device="cuda")
dout.copy_((torch.rand((S, 4), device="cuda", generator=g) - 0.5) * 1e-3)
dparams = torch.zeros(net.n_params(), dtype=torch.float32, device="cuda")


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {}
ws = net.train_workspace(S)
t_f = timed(lambda: net.train_forward(encT, S, ws, out))
dparams.zero_(); net.train_backward(encT, out, dout, S, ws, dparams); torch.cuda.synchronize()
ref = dparams.double().clone()
t_b = timed(lambda: net.train_backward(encT, out, dout, S, ws, dparams))
print(f"saved : forward {t_f:.3f} ms, backward (dgrad + wgrad) {t_b:.3f} ms, sum {t_f + t_b:.3f} ms; workspace {ws.numel() * 2 / 2**30:.2f} GiB", flush=True)
del ws
torch.cuda.empty_cache()
wl = net.train_lean_workspace(S)
t_fo = timed(lambda: net.train_forward_outputs(encT, S, out))
t_f = timed(lambda: net.train_forward_lean(encT, S, wl, out))
dparams.zero_(); net.train_backward_lean(encT, out, dout, S, wl, dparams); torch.cuda.synchronize()
got = dparams.double().clone()
t_b = timed(lambda: net.train_backward_lean(encT, out, dout, S, wl, dparams))
rel = float((got - ref).norm() / ref.norm())
print(f"lean  : forward {t_f:.3f} ms (outputs only {t_fo:.3f}), backward (dgrad + recompute wgrad) {t_b:.3f} ms, sum {t_f + t_b:.3f} ms; "
      f"workspace {wl.numel() * 2 / 2**30:.2f} GiB; |dW_lean - dW_saved| / |dW_saved| = {rel:.2e}", flush=True)
