#!/usr/bin/env python3
"""configs[2] step: eager vs captured vs captured with the traversal one batch ahead; and the graph's pieces alone."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rtx_nerf_amd import api, scenes
from rtx_nerf_amd.train import Trainer, camera_rays

R, B = 128, 4096
dense = scenes.lego_standin_density(R, seed=0)
occ = torch.from_numpy(scenes.pack_occupancy(dense).view(np.int32).copy()).cuda()
hgd = dict(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.5)
focal = scenes.lego_focal_length(True)
def make(batch_rays=B):
    return Trainer(R, occ, encoding="hash", n_neurons=64, n_hidden_layers=4, hashgrid=hgd, n_dir_freqs=4, batch_rays=batch_rays,
                   max_segments=batch_rays * 16, lr=1e-2, loss_scale=128.0, density_scale=300.0, mode="nerf")


# targets from the analytic teacher field, as in bench.py: the step's cost depends on how much of the batch carries a gradient
ro, rd, tg = [], [], []
_t = make(128 * 128)
for i in range(4):
    o, d = camera_rays(scenes.pose_spherical(90.0 * i + 15.0, -30.0, origin_scale=10.0), focal, 128, 128)
    ro.append(o); rd.append(d); tg.append(_t.render_rays(o, d, radiance_fn=scenes.teacher_field).clone())
ro, rd, tg = torch.cat(ro), torch.cat(rd), torch.cat(tg)
del _t


def wall(fn, n=40):
    for _ in range(20):          # every variant has trained the same number of steps when it is timed
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


g = torch.Generator(device="cuda").manual_seed(42)
tr = make()
def eager():
    idx = torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g)
    tr.step(ro[idx].contiguous(), rd[idx].contiguous(), tg[idx].contiguous())
print(f"eager step (host segment count)            {wall(eager):8.1f} us")
cap = int(1.5 * int(tr.total.item())) + 1024
for pf in (False, True):
    t2 = make()
    t2.capture_step(B, launch_segments=cap, prefetch=pf)
    def cap_step():
        idx = torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g)
        torch.index_select(ro, 0, idx, out=t2.graph_rays_o); torch.index_select(rd, 0, idx, out=t2.graph_rays_d)
        torch.index_select(tg, 0, idx, out=t2.graph_targets)
        t2.step_captured()
    print(f"captured step, prefetch={pf!s:5}                {wall(cap_step):8.1f} us")
    if pf:
        print(f"  graph pieces alone: traverse-only {wall(lambda: t2._graphs['prime'][0].replay()):8.1f} us   train-only {wall(lambda: t2._graphs['flush'][0].replay()):8.1f} us"
              f"   both (forked) {wall(lambda: t2._graphs['step'][1].replay()):8.1f} us")
def gather_only():
    idx = torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g)
    torch.index_select(ro, 0, idx, out=t2.graph_rays_o); torch.index_select(rd, 0, idx, out=t2.graph_rays_d)
    torch.index_select(tg, 0, idx, out=t2.graph_targets)
print(f"batch gather alone (randint + 3 index_select) {wall(gather_only):8.1f} us")
