#!/usr/bin/env python3
"""Where a tile's cycles go in the lean weight-gradient kernel (train.hip, wgrad_recompute_pass): s_memtime stamps of the first
block of each pass, tiles 20 and 21 (diagnostic build, -DRTXN_LN_STAMPS).
  ABLATE_SRC=train tools/ablate.sh lnstamps="-DRTXN_LN_STAMPS"
  RTXN_LIB_PATH=rtx_nerf_amd/librtxn_lnstamps.so python tools/probe/lean_stamps.py [samples]
Per pass and layer, mean over the four waves and two tiles, in cycles:
  wt   [W+T]: wait for the layer's weights + barrier + issue of stage 3        fwd  the recomputed forward layer
  v0   first images written, wait for stages 0, 1, barrier, next weights issued   c01, c23  the two 128-sample contractions
  v2   barrier, look-ahead issue, second images written, wait for stages 2, 3, barrier"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from rtx_nerf_amd import _lib, api, scenes

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4_695_827
W, L, E = 128, 8, 112
net = api.Network(n_neurons=W, n_hidden_layers=L)
net.set_params(torch.from_numpy(scenes.xavier_params_fp16(W, L, E, seed=3)).cuda())
Sp = api.padded_samples(S)
g = torch.Generator(device="cuda").manual_seed(5)
encT = (torch.rand((E, Sp), device="cuda", generator=g) * 2 - 1).half()
encT[:, S:] = 0
out = torch.empty((S, 16), dtype=torch.float16, device="cuda")
dout = ((torch.rand((S, 4), device="cuda", generator=g) - 0.5) * 1e-3).half()
dparams = torch.zeros(net.n_params(), dtype=torch.float32, device="cuda")
ws = net.train_lean_workspace(S)
net.train_forward_lean(encT, S, ws, out)
for _ in range(3):
    net.train_backward_lean(encT, out, dout, S, ws, dparams)
torch.cuda.synchronize()
fn = _lib.lib().rtxn_debug_read_lean_stamps
fn.restype = ctypes.c_int
buf = (ctypes.c_uint32 * (3 * 4 * 2 * 92))()
assert fn(buf) == 0
st = np.frombuffer(buf, dtype=np.uint32).reshape(3, 4, 2, 92).astype(np.int64)


def d(p, a, b):
    return float(((st[p, :, :, b] - st[p, :, :, a]) & 0xFFFFFFFF).mean())


fc = _lib.lib().rtxn_debug_read_lean_clock
fc.restype = ctypes.c_int
cb = (ctypes.c_uint64 * 24)()
assert fc(cb) == 0
clk = np.frombuffer(cb, dtype=np.uint64).reshape(3, 8).astype(np.int64)
passes = [("layers 0-2", 0, 3, 2, False), ("layers 3-5", 3, 6, 5, False), ("layers 6-7 + output", 6, 8, 8, True)]
for p, (name, l0, l1, fwd_end, has_out) in enumerate(passes):
    last = 7 if has_out else l1 - 1
    period = float(((st[p, :, 1, 0] - st[p, :, 0, 0]) & 0xFFFFFFFF).mean())
    c = clk[p]
    us = (c[3] - c[2]) / 100.0
    print(f"pass {name}: first block ran {us:.0f} us = {c[1] - c[0]} s_memtime ticks ({(c[1] - c[0]) / us / 1000:.3f} G ticks/s) over {c[4]} tiles: "
          f"{(c[1] - c[0]) / max(c[4], 1):.0f} ticks per tile on average")
    print(f"pass {name}: tile period {period:.0f} cycles; top -> encoding in registers {d(p, 0, 1):.0f}")
    tot = dict(wt=0.0, fwd=0.0, sync=0.0, contract=0.0)
    for l in range(last + 1):
        b = 2 + 11 * l
        has_w, has_f = l0 <= l < l1, l < fwd_end
        if not (has_w or has_f):
            continue
        wt, fw = d(p, b, b + 1), d(p, b + 1, b + 2)
        line = f"  layer {l}: wt {wt:5.0f}  fwd {fw:5.0f}  v0 {d(p, b + 2, b + 3):5.0f}"
        tot["wt"] += wt
        tot["fwd"] += fw
        tot["sync"] += d(p, b + 2, b + 3)
        if has_w:
            c = [d(p, b + 3, b + 4), d(p, b + 7, b + 8)]
            v2 = d(p, b + 4, b + 7)
            line += f"  c01 {c[0]:5.0f}  v2 {v2:5.0f}  c23 {c[1]:5.0f}"
            tot["contract"] += sum(c)
            tot["sync"] += v2
        print(line)
    if has_out:
        print(f"  tile end {d(p, 90, 91):.0f}")
    print("  sums: " + "  ".join(f"{k} {v:.0f}" for k, v in tot.items()) + "   (MFMA-bound: forward layer 2048, contraction of a layer 2048)")
