#!/usr/bin/env python3
"""Reads the hidden activations of the OUTPUTS-ONLY training forward (which stores none) without touching the kernel: a
2-layer model whose output layer is a one-hot selection of 16 hidden features, run 8 times (W = 128) with the selection moved.
Prints which features of which column-tile parity differ from a torch restatement, and whether the wrong values equal the sum
with one k-step (16 input features) left out or replaced.

    fwd_hidden_dump.py [W [E]]        RTXN_LIB_PATH selects the library under test"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rtx_nerf_amd import api, scenes
W = int(sys.argv[1]) if len(sys.argv) > 1 else 128
E = int(sys.argv[2]) if len(sys.argv) > 2 else 48
L, P = 2, 330
S = P * 32
rng = np.random.default_rng(5)
base = scenes.xavier_params_fp16(W, L, E, seed=9)
Sp = api.padded_samples(S)
enc = rng.uniform(-1, 1, (E, S)).astype(np.float16)
encT = torch.zeros((E, Sp), dtype=torch.float16, device="cuda")
encT[:, :S] = torch.from_numpy(enc).cuda()
net = api.Network(n_neurons=W, n_hidden_layers=L, n_encoded_features=E, output_activation=0)
ws = net.train_workspace(S)
got = {"save": torch.zeros((S, W), device="cuda"), "out": torch.zeros((S, W), device="cuda")}
for g in range(W // 16):
    p = base.copy()
    wo = np.zeros((16, W), np.float16)
    wo[np.arange(16), 16 * g + np.arange(16)] = 1.0
    p[-16 * W:] = wo.reshape(-1)
    net.set_params(torch.from_numpy(p).cuda())
    got["save"][:, 16 * g:16 * g + 16] = net.train_forward(encT, S, ws).float()
    got["out"][:, 16 * g:16 * g + 16] = net.train_forward_outputs(encT, S).float()
torch.cuda.synchronize()
pf = torch.from_numpy(base).cuda().float()
x0 = torch.from_numpy(enc).cuda().float().T
w0 = pf[:W * E].view(W, E)
h0 = torch.relu(x0 @ w0.T).half().float()                      # [S][W]
w1 = pf[W * E:W * E + W * W].view(W, W)
full = h0 @ w1.T
want = torch.relu(full).half().float()
ct = ((torch.arange(S, device="cuda") % 64) // 32)
for name in ("save", "out"):
    d = (got[name] - want).abs()
    for c in (0, 1):
        dc = d[ct == c]
        bad_feat = torch.nonzero((dc > 2e-3).any(dim=0)).flatten().tolist()
        print(f"{name} ct{c}: max err {float(dc.max()):.4f}; features with an error > 2e-3: {len(bad_feat)} {bad_feat[:40]}")
# hypotheses for the wrong features of the outputs-only kernel, column tile 1
d = (got["out"] - want).abs()
sel = ct == 1
bad = torch.nonzero((d[sel] > 2e-3).any(dim=0)).flatten().tolist()
if bad:
    g1 = got["out"][sel][:, bad]
    print("hypotheses over the wrong features (max |got - hypothesis|):")
    for m in range(W // 16):
        part = full - h0[:, 16 * m:16 * m + 16] @ w1[:, 16 * m:16 * m + 16].T
        hyp = torch.relu(part).half().float()[sel][:, bad]
        print(f"  k-step {m} left out: {float((g1 - hyp).abs().max()):.4f}")
    for m in range(W // 16):
        # k-step m's weights replaced by those of another row tile / k-step are too many to enumerate: report the residual's
        # correlation with k-step m's contribution instead
        contrib = (h0[:, 16 * m:16 * m + 16] @ w1[:, 16 * m:16 * m + 16].T)[sel][:, bad]
        resid = (got["out"] - full)[sel][:, bad]
        pos = got["out"][sel][:, bad] > 0
        if pos.any():
            a, b = resid[pos], contrib[pos]
            print(f"  residual . contribution of k-step {m} / |contribution|^2 = {float((a * b).sum() / (b * b).sum()):+.3f}")
