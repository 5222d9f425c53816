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

# ---- cost of the exchange's device side on this gradient, and the bytes it would move (every rank assumed to hold this much)
from rtx_nerf_amd import api
idx = torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g)
tr.gradients(ro[idx].contiguous(), rd[idx].contiguous(), tg[idx].contiguous())
vals = tr.dtable_h
T = 1 << 19
nb = vals.numel() // 2 // T
counts, ws = api.half2_count_nonzero(vals, T)
c = counts.cpu().tolist()
cap = sum(c)
pairs = torch.zeros((cap + 1024, 2), dtype=torch.int32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
keep = vals.clone()


def timed(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


t_count = timed(lambda: api.half2_count_nonzero(vals, T, ws))
t_pack = timed(lambda: api.half2_pack_nonzero(vals, T, ws, (1 << nb) - 1, pairs, cnt, clear=False))
api.half2_pack_nonzero(vals, T, ws, (1 << nb) - 1, pairs, cnt, clear=True)
t_add = timed(lambda: api.half2_add_pairs(vals, pairs, cap))
print(f"exchange primitives on {nb} levels x 2^19 entries ({vals.numel() * 2 / 1e6:.1f} MB), {cap} non-zero entries: "
      f"count {t_count:.1f} us, pack {t_pack:.1f} us, add one list {t_add:.1f} us")
dense = 4.0 * nb * T
for N in (2, 4, 8):
    sparse_levels = [b for b in range(nb) if N * c[b] < T]
    lists = (N - 1) * 8.0 * sum(c[b] for b in sparse_levels)
    dn = 2.0 * (N - 1) / N * 4.0 * T * (nb - len(sparse_levels))
    print(f"  N = {N}: {len(sparse_levels)}/{nb} levels as lists: {lists / 1e6:.2f} MB lists + {dn / 1e6:.2f} MB dense levels per rank and step "
          f"(all dense: {2.0 * (N - 1) / N * dense / 1e6:.2f} MB)")
