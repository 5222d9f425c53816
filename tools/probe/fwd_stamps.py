#!/usr/bin/env python3
"""Where a tile's cycles go in the training forward (train.hip, mlp_train_fwd_kernel<128, masks>): s_memtime stamps of four
neighbouring blocks in the middle of the grid (diagnostic build, -DRTXN_FWD_STAMPS).
  ABLATE_SRC=train tools/ablate.sh fwdstamps="-DRTXN_FWD_STAMPS -mllvm -amdgpu-mfma-vgpr-form"
  RTXN_LIB_PATH=rtx_nerf_amd/librtxn_fwdstamps.so python tools/probe/fwd_stamps.py [masks|none]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from rtx_nerf_amd import _lib, api, scenes

mode = sys.argv[1] if len(sys.argv) > 1 else "masks"
S = 4_695_827
W, L, E = 128, 8, 112
net = api.Network(n_neurons=W, n_hidden_layers=L)
net.set_params(torch.from_numpy(scenes.xavier_params_fp16(W, L, E, seed=3)).cuda())
Sp = api.padded_samples(S)
g = torch.Generator(device="cuda").manual_seed(5)
encT = (torch.rand((E, Sp), device="cuda", generator=g) * 2 - 1).half()
encT[:, S:] = 0
out = torch.empty((S, 16), dtype=torch.float16, device="cuda")
os.environ["RTXN_TRAIN_FWD16"] = "0"
if mode == "fused":      # sampler + encoder folded in (rtxn_mlp_train_forward_lean_segments): stamps 26 / 27 / 30 = encoded / weights landed / layer 0 multiplied
    P = S // 32
    S = P * 32
    g2 = torch.Generator(device="cuda").manual_seed(1)
    start = torch.rand((P, 3), device="cuda", generator=g2) * 2 - 1
    end = start + (torch.rand((P, 3), device="cuda", generator=g2) - 0.5) * 0.3
    view = torch.rand((P, 2), device="cuda", generator=g2) * 3
    ws = net.train_lean_workspace(S)
    run = lambda: net.train_forward_lean_segments(start, end, view, P, 0, ws, out)
elif mode == "masks":
    ws = net.train_lean_workspace(S)
    run = lambda: net.train_forward_lean(encT, S, ws, out)
else:
    run = lambda: net.forward_outputs(encT, S, out) if hasattr(net, "forward_outputs") else None
for _ in range(3):
    run()
torch.cuda.synchronize()
fn = _lib.lib().rtxn_debug_read_fwd_stamps
fn.restype = ctypes.c_int
buf = (ctypes.c_uint32 * (4 * 4 * 32))()
assert fn(buf) == 0
st = np.frombuffer(buf, dtype=np.uint32).reshape(4, 4, 32).astype(np.int64)
d = lambda a, b: ((st[:, :, b] - st[:, :, a]) & 0xFFFFFFFF)
print(f"mode {mode}: block lifetime {d(0, 2 + 3 * (L - 1) + 1).mean():.0f} cycles (per block: {d(0, 2 + 3 * (L - 1) + 1).mean(axis=1).round()})")
print(f"  layer 0 (fetch + transpose + 7 k-steps) {d(0, 1).mean():.0f}")
if mode == "fused":
    print(f"    entry -> encoding computed (weights in flight) {d(0, 26).mean():.0f}; -> weights landed {d(26, 27).mean():.0f}; 56 MFMAs {d(27, 30).mean():.0f}; pack {d(30, 1).mean():.0f}")
print(f"    entry -> first chunk's loads issued {d(0, 26).mean():.0f}; -> layer 0's weights landed (and that chunk) {d(26, 27).mean():.0f}; chunk 0 {d(27, 28).mean():.0f}; "
      f"chunk 1 {d(28, 29).mean():.0f}; chunk 2 {d(29, 30).mean():.0f}; pack {d(30, 1).mean():.0f}")
print(f"    chunk 0 in parts: scratch written {d(27, 25).mean():.0f}; its 48 two-byte reads back and permuted {d(25, 31).mean():.0f}; 12 fragment reads + 24 MFMAs {d(31, 28).mean():.0f}")
prev = 1
for l in range(1, L):
    b = 2 + 3 * (l - 1)
    print(f"  layer {l}: save(prev) + weights {d(prev, b).mean():5.0f}   mfma {d(b, b + 1).mean():5.0f}   save {d(b + 1, b + 2).mean():5.0f}")
    prev = b + 2
b = 2 + 3 * (L - 1)
print(f"  output layer: weights {d(prev, b).mean():.0f}   mma + stores {d(b, b + 1).mean():.0f}")
print("  starts of the four blocks relative to the first:", (st[:, 0, 0] - st[:, 0, 0].min()))
