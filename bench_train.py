#!/usr/bin/env python3
"""Secondary benchmark (not the driver's): BASELINE.json configs[2] -- training throughput, 4096 rays/batch,
hash-grid encoding (L=16, F=2, T=2^19, base 16, scale 1.5) + Frequency(4) directions + 4x64 MLP, 128^3 procedural
occupancy, K=32, corrected ("nerf") compositor, L2 + Adam.  Prints one JSON line (rays/s of full optimisation steps).
  python bench_train.py [--steps 50] [--warmup 5] [--batch 4096] [--encoding hash|freq]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

from rtx_nerf_amd import scenes
from rtx_nerf_amd.train import Trainer, camera_rays
from train_demo import teacher_field

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--grid", type=int, default=128)
ap.add_argument("--encoding", default="hash")
ap.add_argument("--neurons", type=int, default=64)
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--dir-freqs", type=int, default=4)
a = ap.parse_args()
torch.cuda.set_device(0)
R = a.grid
dense = scenes.lego_standin_density(R, seed=0)
occ = torch.from_numpy(scenes.pack_occupancy(dense).view(np.int32).copy()).cuda()
tr = Trainer(R, occ, encoding=a.encoding, n_neurons=a.neurons, n_hidden_layers=a.layers,
             hashgrid=dict(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.5),
             n_dir_freqs=a.dir_freqs, batch_rays=max(a.batch, 128 * 128), max_segments=max(a.batch, 128 * 128) * 24, lr=1e-2,
             loss_scale=128.0, density_scale=300.0, mode="nerf")
focal = scenes.lego_focal_length(True)
ro, rd, tg = [], [], []
for i in range(8):
    o, d = camera_rays(scenes.pose_spherical(45.0 * i + 15.0, -30.0, origin_scale=10.0), focal, 128, 128)
    ro.append(o); rd.append(d); tg.append(tr.render_rays(o, d, radiance_fn=teacher_field).clone())
ro, rd, tg = torch.cat(ro), torch.cat(rd), torch.cat(tg)
g = torch.Generator(device="cuda").manual_seed(42)
samples = 0


def step():
    global samples
    idx = torch.randint(0, ro.shape[0], (a.batch,), device="cuda", generator=g)
    loss = tr.step(ro[idx].contiguous(), rd[idx].contiguous(), tg[idx].contiguous())
    samples += int(tr.total.item()) * 32
    return loss


for _ in range(a.warmup):
    step()
torch.cuda.synchronize()
samples = 0
t0 = time.perf_counter()
for _ in range(a.steps):
    loss = step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({
    "metric": "training rays/s (full optimisation step)", "value": round(a.batch * a.steps / dt / 1e6, 4), "unit": "Mrays/s",
    "ms_per_step": round(1e3 * dt / a.steps, 3), "samples_per_step": samples // a.steps, "final_loss": float(loss.item()),
    "config": {"workload": f"{a.batch} rays/batch, {a.encoding} encoding + {a.layers}x{a.neurons} MLP, {R}^3 grid ({100 * dense.mean():.1f}% cells), K=32, "
                           "L2 + Adam, teacher = analytic field", "n_gpus": 1}}))
