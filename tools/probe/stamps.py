#!/usr/bin/env python3
"""Where one tile's cycles go in mlp_fwd16_kernel: s_memtime stamps of block 0 (diagnostic build, -DRTXN_STAMPS).
  tools/ablate.sh stamps="-DRTXN_STAMPS"
  RTXN_LIB_PATH=rtx_nerf_amd/librtxn_stamps.so python tools/probe/stamps.py [--weights zero]
Prints, per wave group, the mean cycles of every stage of a tile: wait = barrier at the stage's start, run = its MFMAs."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from rtx_nerf_amd import _lib, api, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--segments", type=int, default=3_000_000)
ap.add_argument("--weights", choices=["xavier", "zero"], default="xavier")
args = ap.parse_args()
P = args.segments
g = torch.Generator(device="cuda").manual_seed(0)
sp = torch.rand((P, 3), device="cuda", generator=g) * 2 - 1
ep = sp + (torch.rand((P, 3), device="cuda", generator=g) - 0.5) * 0.03
sv = torch.rand((P, 2), device="cuda", generator=g) * 3.0
total = torch.tensor([P], dtype=torch.int32, device="cuda")
net = api.Network(n_neurons=128, n_hidden_layers=8)
w = scenes.xavier_params_fp16(128, 8, net.encoded_width())
if args.weights == "zero":
    w = np.zeros_like(w)
net.set_params(torch.from_numpy(w).cuda())
out = torch.empty((P * 32, 4), dtype=torch.float16, device="cuda")
for _ in range(3):
    net.forward_segments_compact(sp, ep, sv, total, P, out)     # the bench frame's variant (segments in, half4 out)
torch.cuda.synchronize()
lib = _lib.lib()
buf = (ctypes.c_uint32 * (8 * 4 * 24))()
fn = lib.rtxn_debug_read_stamps
fn.restype = ctypes.c_int
assert fn(buf) == 0
st = np.frombuffer(buf, dtype=np.uint32).reshape(8, 4, 24).astype(np.int64)
n_layers = 9
names = ["encode"]
for l in range(n_layers):
    names += [f"wait{l}", f"run{l}"]
for grp, rows in (("group A (waves 0-3)", st[:4]), ("group B (waves 4-7)", st[4:])):
    d = (rows[:, :, 1:2 * n_layers + 2] - rows[:, :, 0:2 * n_layers + 1]) & 0xFFFFFFFF     # [wave, tile, stage]
    tile = ((rows[:, 1:, 0] - rows[:, :-1, 0]) & 0xFFFFFFFF).mean()
    print(f"{grp}: tile period {tile:.0f} cycles; top->end {d.sum(axis=2).mean():.0f}")
    m = d.mean(axis=(0, 1))
    print("  " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, m)))
    print(f"  sum wait {m[1::2].sum():.0f}  sum run {m[2::2].sum():.0f}  encode {m[0]:.0f}  (MFMA-bound run of a layer: 2048 alone, 4096 shared)")
