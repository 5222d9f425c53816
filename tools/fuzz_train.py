#!/usr/bin/env python3
"""Fuzz the MLP training kernels (forward with saved activations, backward chain, weight gradients, d(encoding)) against
the oracle: random width (64/128), hidden-layer count 1..9, encoded width (multiples of 16 up to 128), output activation,
batch size around the 256-sample tile and the 1024/2048-sample weight-gradient chunks.
  python tools/fuzz_train.py [--iters 40] [--seed 0]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oracle as O
from rtx_nerf_amd import api, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
torch.cuda.set_device(0)


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


bad = 0
for it in range(a.iters):
    W = int(rng.choice([64, 128]))
    L = int(rng.integers(1, 10))
    E = int(rng.choice([16, 32, 48, 64, 80, 96, 112, 128]))
    act = int(rng.integers(0, 2))
    n = int(rng.choice([1, 255, 256, 257, 1023, 1024, 1025, 2047, 2049, 4100, int(rng.integers(1, 6000))]))
    params = scenes.xavier_params_fp16(W, L, E, seed=it)
    enc = rng.uniform(-1, 1, (n, E)).astype(np.float16)
    Sp = api.padded_samples(n)
    encT = np.zeros((E, Sp), np.float16)
    encT[:, :n] = enc.T
    net = api.Network(n_neurons=W, n_hidden_layers=L, n_encoded_features=E, output_activation=act)
    net.set_params(dev(params))
    encT_d = dev(encT)
    ws = net.train_workspace(n)
    out = net.train_forward(encT_d, n, ws)
    torch.cuda.synchronize()
    errs = []
    o_acts, o_out = O.mlpe_forward(W, L, act, params, enc)
    got = out.cpu().numpy().astype(np.float32)
    tol = 1e-2 if act else 3e-2
    if not (np.isfinite(got).all() and np.abs(got - o_out.astype(np.float32)).max() <= tol):
        errs.append(f"forward output max err {np.abs(got - o_out.astype(np.float32)).max():.4g}")
    acts = ws[:L * W * Sp].reshape(L, W, Sp).cpu().numpy()
    for l in range(L):
        if np.abs(acts[l, :, :n].T.astype(np.float32) - o_acts[l].astype(np.float32)).max() > 2e-2 or np.any(acts[l, :, n:] != 0):
            errs.append(f"activations of layer {l}")
            break
    dout = (rng.standard_normal((n, 4)) * 0.05).astype(np.float16)
    dparams = torch.zeros(net.n_params(), device="cuda")
    dencT = torch.full((E, Sp), 7.0, dtype=torch.float16, device="cuda")
    net.train_backward(encT_d, out, dev(dout), n, ws, dparams, dencT)
    torch.cuda.synchronize()
    acts_sm = np.ascontiguousarray(np.transpose(acts[:, :, :n], (0, 2, 1)))
    want_dp, want_denc = O.mlpe_backward(W, L, act, params, enc, acts_sm, out.cpu().numpy(), dout)
    got_dp = dparams.cpu().numpy()
    scale = np.abs(want_dp).max()
    if not (scale > 0 and np.abs(got_dp - want_dp).max() < 3e-2 * scale and np.linalg.norm(got_dp - want_dp) < 2e-2 * np.linalg.norm(want_dp)):
        errs.append(f"weight gradient: max err {np.abs(got_dp - want_dp).max() / max(scale, 1e-30):.3g} of scale")
    got_denc = dencT.cpu().numpy()[:, :n].T.astype(np.float32)
    if not (np.linalg.norm(got_denc - want_denc) < 2e-2 * np.linalg.norm(want_denc) + 1e-6 and np.all(dencT.cpu().numpy()[:, n:] == 0)):
        errs.append("d(encoding)")
    if errs:
        bad += 1
        print(f"MISMATCH it={it} W={W} L={L} E={E} act={act} n={n}: {errs}", flush=True)
# the lean 8 x 128 path (sign masks + dZ only, weight gradient recomputes the activations): random batch sizes around its 256-sample
# tiles, random spans of zero loss gradient (dead tiles are stepped over), both output activations -- against the oracle's backward
n_lean = max(a.iters // 3, 4)
for it in range(n_lean):
    W, L, E = 128, 8, 112
    act = int(rng.integers(0, 2))
    n = int(rng.choice([1, 31, 255, 256, 257, 511, 513, 1025, int(rng.integers(1, 9000)), int(rng.integers(15000, 22000))]))
    params = scenes.xavier_params_fp16(W, L, E, seed=1000 + it)
    enc = rng.uniform(-1, 1, (n, E)).astype(np.float16)
    Sp = api.padded_samples(n)
    encT = np.zeros((E, Sp), np.float16)
    encT[:, :n] = enc.T
    net = api.Network(n_neurons=W, n_hidden_layers=L, n_encoded_features=E, output_activation=act)
    net.set_params(dev(params))
    assert net.lean_supported()
    encT_d = dev(encT)
    ws = net.train_lean_workspace(n)
    out = torch.empty((n, 16), dtype=torch.float16, device="cuda")
    net.train_forward_lean(encT_d, n, ws, out)
    dout = (rng.standard_normal((n, 4)) * 0.05).astype(np.float16)
    for _ in range(int(rng.integers(0, 4))):                 # dead spans: whole tiles and ragged pieces
        lo = int(rng.integers(0, n)); dout[lo:lo + int(rng.integers(1, 1500))] = 0
    dparams = torch.zeros(net.n_params(), device="cuda")
    net.train_backward_lean(encT_d, out, dev(dout), n, ws, dparams)
    torch.cuda.synchronize()
    o_acts, o_out = O.mlpe_forward(W, L, act, params, enc)
    errs = []
    got = out.cpu().numpy().astype(np.float32)
    if not (np.isfinite(got).all() and np.abs(got[:, :4] - o_out.astype(np.float32)[:, :4]).max() <= (1e-2 if act else 3e-2)):
        errs.append("forward output")
    acts_sm = np.ascontiguousarray(o_acts)
    want_dp, _ = O.mlpe_backward(W, L, act, params, enc, acts_sm, out.cpu().numpy(), dout)
    got_dp = dparams.cpu().numpy()
    scale = np.abs(want_dp).max()
    # (the oracle's backward runs on the ORACLE's activations here -- the lean path stores none to hand over -- so an activation within an
    # fp16 rounding of zero can sit on the other side of its ReLU; with a few hundred samples, part of them dead, that is 2-2.5e-2 of the norm
    # in one case of twenty (4e-2 allowed; the largest single element 8e-2).  The tight statement is the comparison with the
    # saved-activation kernels below, which the first loop holds to the oracle on the GPU's own activations.)
    if scale > 0 and not (np.abs(got_dp - want_dp).max() < 8e-2 * scale and np.linalg.norm(got_dp - want_dp) < 4e-2 * np.linalg.norm(want_dp)):
        errs.append(f"weight gradient: max err {np.abs(got_dp - want_dp).max() / max(scale, 1e-30):.3g} of scale, "
                    f"{np.linalg.norm(got_dp - want_dp) / np.linalg.norm(want_dp):.3g} of the norm")
    if scale == 0 and np.abs(got_dp).max() != 0:
        errs.append("non-zero gradient for an all-zero loss gradient")
    ws_s = net.train_workspace(n)
    out_s = net.train_forward(encT_d, n, ws_s)
    dp_s = torch.zeros(net.n_params(), device="cuda")
    net.train_backward(encT_d, out_s, dev(dout), n, ws_s, dp_s)
    torch.cuda.synchronize()
    ref = dp_s.double().cpu().numpy()
    if np.linalg.norm(ref) > 0 and not np.linalg.norm(got_dp - ref) <= 5e-5 * np.linalg.norm(ref):
        errs.append(f"lean vs saved-activation kernels: {np.linalg.norm(got_dp - ref) / np.linalg.norm(ref):.3g} of the norm")
    del ws_s
    if errs:
        bad += 1
        print(f"MISMATCH lean it={it} act={act} n={n}: {errs}", flush=True)
# ... and the same path with sampler + encoder folded into both kernels (rtxn_mlp_train_forward_lean_segments / _backward_lean_segments)
# against the staged encoder + encT kernels on random segments: outputs and t_vals bit for bit, gradients to the order of the atomics
n_fused = max(a.iters // 6, 3)
for it in range(n_fused):
    P = int(rng.choice([1, 7, 9, 63, int(rng.integers(1, 400)), int(rng.integers(8200, 9000))]))
    stype = int(rng.choice([0, 3]))
    net = api.Network(n_neurons=128, n_hidden_layers=8, output_activation=int(rng.integers(0, 2)))
    net.set_params(dev(scenes.xavier_params_fp16(128, 8, 112, seed=2000 + it)))
    start = rng.uniform(-1, 1, (P, 3)).astype(np.float32)
    end = (start + rng.uniform(-0.3, 0.3, (P, 3))).astype(np.float32)
    view = rng.uniform(0, 3.1, (P, 2)).astype(np.float32)
    n, Sp = P * 32, api.padded_samples(P * 32)
    sd, ed, vd = dev(start), dev(end), dev(view)
    encT = torch.zeros((112, Sp), dtype=torch.float16, device="cuda")
    tv_a, tv_b = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    net.encode_frequency_segments(sd, ed, vd, P, stype, encT, tv_a, 1.7)
    dout = (rng.standard_normal((n, 4)) * 0.05).astype(np.float16)
    dout.reshape(P, 32, 4)[rng.random(P) < 0.3] = 0
    dd = dev(dout)
    got = []
    for fused in (False, True):
        ws = net.train_lean_workspace(n)
        out = torch.zeros((n, 16), dtype=torch.float16, device="cuda")
        dp = torch.zeros(net.n_params(), device="cuda")
        if fused:
            net.train_forward_lean_segments(sd, ed, vd, P, stype, ws, out, t_vals=tv_b, t_scale=1.7)
            net.train_backward_lean_segments(sd, ed, vd, P, stype, out, dd, ws, dp)
        else:
            net.train_forward_lean(encT, n, ws, out)
            net.train_backward_lean(encT, out, dd, n, ws, dp)
        torch.cuda.synchronize()
        got.append((out[:, :4].clone(), dp.double().cpu().numpy()))
    errs = []
    if not torch.equal(got[0][0], got[1][0]):
        errs.append("outputs differ")
    if not torch.equal(tv_a, tv_b):
        errs.append("t_vals differ")
    ga, gb = got[0][1], got[1][1]
    if np.linalg.norm(ga) > 0 and not np.linalg.norm(ga - gb) <= 2e-5 * np.linalg.norm(ga):
        errs.append(f"weight gradient: {np.linalg.norm(ga - gb) / np.linalg.norm(ga):.3g} of the norm")
    if errs:
        bad += 1
        print(f"MISMATCH folded it={it} P={P} sample_type={stype}: {errs}", flush=True)
print(f"fuzz_train: {a.iters} + {n_lean} lean + {n_fused} folded-encoder cases, {bad} mismatches")
sys.exit(1 if bad else 0)
