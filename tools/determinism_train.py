#!/usr/bin/env python3
"""Run-to-run determinism of TRAINING (the companion of tools/determinism_check.py, which covers the inference kernel):
N identical trainers take the same optimisation steps on BASELINE configs[2] (hash grid L=16 F=2 T=2^19 + 4x64, 4096 rays, 128^3)
and the reference's 8x128 model; prints, per mode, how many parameters / table entries differ between the runs.
  determinism_train.py [steps] [--config hash|ref8x128]
Default mode: float atomics (fp32 weight-gradient flushes, packed fp16 hash scatter) -- differences expected.  Deterministic mode
(Trainer(deterministic=True) = rtxn_set_deterministic_workspace): 0 differences, or exit 1."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rtx_nerf_amd import scenes
from rtx_nerf_amd.train import Trainer, camera_rays

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
config = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "hash"
torch.cuda.set_device(0)
R, B = (128, 4096) if config == "hash" else (8, 4096)
occ = None
if config == "hash":
    occ = torch.from_numpy(scenes.pack_occupancy(scenes.lego_standin_density(R, seed=0)).view(np.int32).copy()).cuda()
focal = scenes.lego_focal_length(True)
ro, rd = [], []
for i in range(4):
    o, d = camera_rays(scenes.pose_spherical(45.0 * i + 15.0, -30.0, origin_scale=10.0), focal, 128, 128)
    ro.append(o); rd.append(d)
ro, rd = torch.cat(ro), torch.cat(rd)
g = torch.Generator(device="cuda").manual_seed(7)
tg = torch.rand((ro.shape[0], 3), device="cuda", generator=g)
idx = [torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g) for _ in range(steps)]


def run(det):
    if config == "hash":
        tr = Trainer(R, occ, encoding="hash", n_neurons=64, n_hidden_layers=4, n_dir_freqs=4, batch_rays=B, max_segments=B * 48,
                     hashgrid=dict(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.5),
                     lr=1e-2, loss_scale=128.0, density_scale=300.0, mode="nerf", deterministic=det)
    else:
        tr = Trainer(R, None, encoding="freq", n_neurons=128, n_hidden_layers=8, n_dir_freqs=12, batch_rays=B, max_segments=B * 26,
                     lr=1e-3, loss_scale=1.0, mode="compat", deterministic=det)
    for i in idx:
        tr.step(ro[i].contiguous(), rd[i].contiguous(), tg[i].contiguous())
    torch.cuda.synchronize()
    out = {"mlp": tr.master.clone()}
    if config == "hash":
        out["table"] = tr.table_master.clone()
    return out


bad = 0
for det in (False, True):
    a, b = run(det), run(det)
    for k in a:
        n = int((a[k] != b[k]).sum())
        rel = float((a[k] - b[k]).norm() / a[k].norm())
        print(f"{config:8s} {'deterministic' if det else 'default      '} {k:5s}: {n:8d} of {a[k].numel()} values differ after {steps} steps "
              f"(relative distance {rel:.2e})")
        if det:
            bad += n
print("DETERMINISTIC" if bad == 0 else f"NONDETERMINISTIC ({bad} values in deterministic mode)")
sys.exit(0 if bad == 0 else 1)
