#!/usr/bin/env python3
"""How sparse is one rank's hash-grid gradient?  configs[2] at full size (4096 rays, 128^3 stand-in grid): entries of each
hashed level that one step's scatter touched (non-zero half2), early and late in training -- the input of the data-parallel
exchange's per-level choice between the dense fp16 level and bitmap + packed values."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rtx_nerf_amd import scenes
from rtx_nerf_amd.train import Trainer, camera_rays
R, B = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
occ = torch.from_numpy(scenes.pack_occupancy(scenes.lego_standin_density(R, seed=0)).view(np.int32).copy()).cuda()
hgd = dict(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.5)
tr = Trainer(R, occ, encoding="hash", n_neurons=64, n_hidden_layers=4, hashgrid=hgd, n_dir_freqs=4, batch_rays=128 * 128,
             max_segments=128 * 128 * 32, lr=1e-2, loss_scale=128.0, density_scale=300.0, mode="nerf")
focal = scenes.lego_focal_length(True)
ro, rd, tg = [], [], []
for i in range(8):
    o, d = camera_rays(scenes.pose_spherical(45.0 * i + 15.0, -30.0, origin_scale=10.0), focal, 128, 128)
    ro.append(o); rd.append(d); tg.append(tr.render_rays(o, d, radiance_fn=scenes.teacher_field).clone())
ro, rd, tg = torch.cat(ro), torch.cat(rd), torch.cat(tg)
g = torch.Generator(device="cuda").manual_seed(42)
offs = [tr.hg.level_offset(l) for l in range(17)]          # parameter offset of every level (+ end)
F = 2


def report(tag):
    idx = torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g)
    S = tr.gradients(ro[idx].contiguous(), rd[idx].contiguous(), tg[idx].contiguous())
    live = int(tr.live_ws[0].item()) if tr.live_segments else -1
    print(f"{tag}: {S} samples, {S // 32} segments, {live} live")
    gh = tr.dtable_h.view(-1, F)
    lo = tr.hashed_lo
    for l in range(len(offs) - 1):
        a, b = offs[l], offs[l + 1]
        if a < lo:
            continue
        lev = gh[(a - lo) // F:(b - lo) // F]
        t = int((lev != 0).any(dim=1).sum().item())
        print(f"  level {l:2d}: {lev.shape[0]:7d} entries, touched {t:7d} = {100.0 * t / lev.shape[0]:5.1f} %")


report("step 0")
for _ in range(300):
    idx = torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g)
    tr.step(ro[idx].contiguous(), rd[idx].contiguous(), tg[idx].contiguous())
report("step 300")
for _ in range(1200):
    idx = torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g)
    tr.step(ro[idx].contiguous(), rd[idx].contiguous(), tg[idx].contiguous())
report("step 1500")
