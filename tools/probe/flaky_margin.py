#!/usr/bin/env python3
"""How close do the order-dependent comparisons of tests/test_gpu_training_loop.py come to their bounds?  Repeats the
captured-vs-eager scenario and prints the largest relative loss difference per repetition."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_training_loop as T
from rtx_nerf_amd import scenes
from rtx_nerf_amd.train import camera_rays
B = 900
focal = scenes.lego_focal_length(True)
for encoding, mode, neurons, layers in [("hash", "nerf", 64, 4), ("freq", "nerf", 128, 2), ("freq", "compat", 64, 2), ("hash", "compat", 128, 2)]:
    worst = []
    for rep in range(4):
        a = T._small_trainer(torch, encoding, mode, neurons, layers)
        b = T._small_trainer(torch, encoding, mode, neurons, layers)
        rng = np.random.default_rng(1)
        b.capture_step(B, launch_segments=B * 30)
        w = 0.0
        for i in range(6):
            o, d = camera_rays(scenes.pose_spherical(40.0 + 50.0 * i, -30.0 + 5.0 * i, origin_scale=10.0), focal, 30, 30)
            t = torch.from_numpy(rng.uniform(0, 1, (B, 3)).astype(np.float32)).cuda()
            la = float(a.step(o, d, t).item())
            b.graph_rays_o.copy_(o); b.graph_rays_d.copy_(d); b.graph_targets.copy_(t)
            lb = float(b.step_captured().item())
            w = max(w, abs(la - lb) / abs(la))
        pa, pb = a.master.cpu().numpy(), b.master.cpu().numpy()
        worst.append((w, float(np.linalg.norm(pa - pb) / np.linalg.norm(pa))))
    print(encoding, mode, neurons, layers, "max rel loss diff per rep:", ["%.1e" % x[0] for x in worst], "(bound 5e-4)  param norm diff:", ["%.1e" % x[1] for x in worst], "(bound 3e-2)")
