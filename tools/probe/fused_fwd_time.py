#!/usr/bin/env python3
"""The lean forward with the encoder folded in against encode + lean forward, 4.7 M samples of random segments (HIP events, 5 reps);
and the folded forward with the standalone encoder running beside it on a second stream (what Trainer.step does)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rtx_nerf_amd import api, scenes
P = 146_745
n = P * 32
W, L = 128, 8
net = api.Network(n_neurons=W, n_hidden_layers=L)
net.set_params(torch.from_numpy(scenes.xavier_params_fp16(W, L, 112, seed=3)).cuda())
g = torch.Generator(device="cuda").manual_seed(1)
start = torch.rand((P, 3), device="cuda", generator=g) * 2 - 1
end = start + (torch.rand((P, 3), device="cuda", generator=g) - 0.5) * 0.3
view = torch.rand((P, 2), device="cuda", generator=g) * 3
Sp = api.padded_samples(n)
encT = torch.empty((112, Sp), dtype=torch.float16, device="cuda")
tv = torch.empty(n, device="cuda")
ws = net.train_lean_workspace(n)
out = torch.empty((n, 16), dtype=torch.float16, device="cuda")
side = torch.cuda.Stream()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def staged():
    net.encode_frequency_segments(start, end, view, P, 0, encT, tv)
    net.train_forward_lean(encT, n, ws, out)


def folded():
    net.train_forward_lean_segments(start, end, view, P, 0, ws, out)


def folded_beside():
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(cur)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        net.encode_frequency_segments(start, end, view, P, 0, encT, tv)
        done = torch.cuda.Event(); done.record(side)
    net.train_forward_lean_segments(start, end, view, P, 0, ws, out)
    cur.wait_event(done)


print(f"encoder alone {timed(lambda: net.encode_frequency_segments(start, end, view, P, 0, encT, tv)):.3f} ms; lean forward from encT {timed(lambda: net.train_forward_lean(encT, n, ws, out)):.3f} ms")
print(f"encode + forward {timed(staged):.3f} ms; forward with the encoder folded in {timed(folded):.3f} ms; folded, encoder beside it {timed(folded_beside):.3f} ms")

dout = ((torch.rand((n, 4), device="cuda", generator=g) - 0.5) * 1e-3).half()
dp = torch.zeros(net.n_params(), dtype=torch.float32, device="cuda")
net.encode_frequency_segments(start, end, view, P, 0, encT, tv)
net.train_forward_lean(encT, n, ws, out)
tb0 = timed(lambda: net.train_backward_lean(encT, out, dout, n, ws, dp))
tb1 = timed(lambda: net.train_backward_lean_segments(start, end, view, P, 0, out, dout, ws, dp))
print(f"backward (dgrad chain + weight gradient): reading encT {tb0:.3f} ms; recomputing the encoding {tb1:.3f} ms")
