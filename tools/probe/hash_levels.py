#!/usr/bin/env python3
"""Where does the hash-grid scatter's time go?  Times rtxn_hashgrid_backward(_mixed) and encode on the config-3 batch's own
samples for grids whose 8 levels all have ONE resolution (per_level_scale = 1), from coarse to fine."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rtx_nerf_amd import api, scenes
from rtx_nerf_amd.train import Trainer, camera_rays
torch.cuda.set_device(0)
R = 128
occ = torch.from_numpy(scenes.pack_occupancy(scenes.lego_standin_density(R, seed=0)).view(np.int32).copy()).cuda()
tr = Trainer(R, occ, encoding="hash", n_neurons=64, n_hidden_layers=4, batch_rays=4096, max_segments=4096 * 24, mode="nerf",
             hashgrid=dict(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.5))
focal = scenes.lego_focal_length(True)
ro, rd = [], []
for i in range(8):
    o, d = camera_rays(scenes.pose_spherical(45.0 * i + 15.0, -30.0, origin_scale=10.0), focal, 128, 128)
    ro.append(o); rd.append(d)
ro, rd = torch.cat(ro), torch.cat(rd)
g = torch.Generator(device="cuda").manual_seed(42)
idx = torch.randint(0, ro.shape[0], (4096,), device="cuda", generator=g)
P = tr._segments(ro[idx].contiguous(), rd[idx].contiguous(), 4096)
tr._sample(4096, P)
S = P * 32
samples = tr.samples[:S].contiguous()
print("samples", S)

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

# the real config-3 grid, with the experiment knobs of rtxn_hashgrid_backward
hg = tr.hg
E, Sp = hg.encoded_width(), api.padded_samples(S)
denc = (torch.randn((E, Sp), device="cuda") * 0.01).half()
dt = torch.zeros(hg.n_params(), device="cuda")
dh = torch.zeros(hg.n_params() - hg.hashed_offset(), dtype=torch.float16, device="cuda")
for nl in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16):
    hgn = api.HashGrid(nl, 2, 19, 16, 1.5, n_dir_freqs=4)
    En = hgn.encoded_width()
    dencn = (torch.randn((En, Sp), device="cuda") * 0.01).half()
    dtn = torch.zeros(hgn.n_params(), device="cuda")
    dhn = torch.zeros(max(hgn.n_params() - hgn.hashed_offset(), 2), dtype=torch.float16, device="cuda")
    tb = torch.zeros(hgn.n_params(), dtype=torch.float16, device="cuda")
    encn = torch.empty((En, Sp), dtype=torch.float16, device="cuda")
    print(f"first {nl:2d} levels of the config-3 grid: backward mixed {timeit(lambda: hgn.backward_mixed(samples, dencn, dtn, dhn))*1e3:7.1f} us   encode {timeit(lambda: hgn.encode(tb, samples, encn))*1e3:7.1f} us")
sys.exit(0)
for res in (16, 48, 96, 200, 400, 800, 1600, 3200, 6400):
    hg = api.HashGrid(8, 2, 19, res, 1.0, n_dir_freqs=4)
    E, Sp = hg.encoded_width(), api.padded_samples(S)
    table = (torch.rand(hg.n_params(), device="cuda") - 0.5).half()
    encT = torch.empty((E, Sp), dtype=torch.float16, device="cuda")
    denc = (torch.randn((E, Sp), device="cuda") * 0.01).half()
    dt = torch.zeros(hg.n_params(), device="cuda")
    lo = hg.hashed_offset()
    dh = torch.zeros(max(hg.n_params() - lo, 2), dtype=torch.float16, device="cuda")
    t_enc = timeit(lambda: hg.encode(table, samples, encT))
    t_b32 = timeit(lambda: hg.backward(samples, denc, dt))
    t_mix = timeit(lambda: hg.backward_mixed(samples, denc, dt, dh))
    print(f"res {res:5d} ({'hashed' if lo < hg.n_params() else 'dense '}): encode {t_enc*1e3:7.1f} us  backward fp32 {t_b32*1e3:7.1f} us  mixed {t_mix*1e3:7.1f} us   (8 identical levels)")
