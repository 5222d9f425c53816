#!/usr/bin/env python3
"""Do the saving and the outputs-only instantiation of the training forward (mlp_train_fwd_kernel<W, SAVE>) agree with each
other, with a torch fp32 restatement of the layer stack, and with themselves from launch to launch?

    fwd_flake.py [W [L [iterations [E]]]]       RTXN_LIB_PATH selects the library under test

Reports, per 32-sample column tile parity (ct = (sample % 64) / 32), the largest |kernel - torch| of both instantiations, and
which launches differ from the first saving launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rtx_nerf_amd import api, scenes
W = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
E = int(sys.argv[4]) if len(sys.argv) > 4 else 48
P = 330
S = P * 32
rng = np.random.default_rng(W + L)
net = api.Network(n_neurons=W, n_hidden_layers=L, n_encoded_features=E)
params = scenes.xavier_params_fp16(W, L, E, seed=9)
net.set_params(torch.from_numpy(params).cuda())
Sp = api.padded_samples(S)
enc = rng.uniform(-1, 1, (E, S)).astype(np.float16)
encT = torch.zeros((E, Sp), dtype=torch.float16, device="cuda")
encT[:, :S] = torch.from_numpy(enc).cuda()


def torch_reference():
    """fp32 accumulation, activations rounded to fp16 after every layer (what the kernels keep between layers)."""
    p = torch.from_numpy(params).cuda().float()
    x = torch.from_numpy(enc).cuda().float().T                    # [S][E]
    off = 0
    w = p[off:off + W * E].view(W, E); off += W * E
    x = torch.relu(x @ w.T).half().float()
    for _ in range(L - 1):
        w = p[off:off + W * W].view(W, W); off += W * W
        x = torch.relu(x @ w.T).half().float()
    w = p[off:off + 16 * W].view(16, W)
    z = x @ w.T
    return z, torch.sigmoid(z)


ws = net.train_workspace(S)
ref = net.train_forward(encT, S, ws).clone()
out = net.train_forward_outputs(encT, S).clone()
torch.cuda.synchronize()
z, sg = torch_reference()
want = sg if float((ref.float() - sg).abs().max()) < float((ref.float() - z).abs().max()) else z
ct = (torch.arange(S, device="cuda") % 64) // 32
for name, t in (("save", ref), ("out ", out)):
    d = (t.float() - want).abs().max(dim=1).values
    print(f"W={W} L={L} E={E} {name} vs torch: ct0 max {float(d[ct == 0].max()):.5f}  ct1 max {float(d[ct == 1].max()):.5f}")
d = (out.float() - ref.float()).abs()
print(f"  out vs save: max {float(d.max()):.5f}; output columns that differ: {torch.nonzero(d.max(dim=0).values > 0).flatten().tolist()}")
bad = {"save": 0, "out": 0}
for it in range(iters):
    a = net.train_forward(encT, S, ws)
    b = net.train_forward_outputs(encT, S)
    torch.cuda.synchronize()
    for name, t, first in (("save", a, ref), ("out", b, out)):
        if not torch.equal(t, first):
            bad[name] += 1
print("  launches differing from their first launch:", bad)
