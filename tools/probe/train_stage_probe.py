#!/usr/bin/env python3
"""Timing probes on the configs[2] training batch (GPU): the fused compositor with / without the loss reduction, the two hash
backward launches back to back vs on two streams, traversal beside the compositor."""
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
tr = Trainer(R, occ, encoding="hash", n_neurons=64, n_hidden_layers=4, hashgrid=hgd, n_dir_freqs=4, batch_rays=128 * 128,
             max_segments=128 * 128 * 24, lr=1e-2, loss_scale=128.0, density_scale=300.0, mode="nerf")
focal = scenes.lego_focal_length(True)
ro, rd = [], []
for i in range(4):
    o, d = camera_rays(scenes.pose_spherical(90.0 * i + 15.0, -30.0, origin_scale=10.0), focal, 128, 128)
    ro.append(o); rd.append(d)
ro, rd = torch.cat(ro), torch.cat(rd)
g = torch.Generator(device="cuda").manual_seed(42)
idx = torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g)
o, d = ro[idx].contiguous(), rd[idx].contiguous()
tg = torch.rand((B, 3), device="cuda")
for _ in range(3):
    tr.step(o, d, tg)
torch.cuda.synchronize()
P = int(tr.total.item()); S = P * 32
nh = tr.num_stored[:B].cpu().numpy()
print(f"P={P} S={S} rays with hits {np.count_nonzero(nh)} max segs/ray {nh.max()} mean {nh.mean():.2f}")


def timeit(fn, n=20, name=""):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:60s} {1e3 * ms:8.1f} us")
    return ms


K = 32
timeit(lambda: api.volrender_l2_train(tr.radiance, tr.t_vals, tr.num_stored, tr.indices, B, K, tg, 128.0, tr.pixels[:B], tr.loss_grads[:B], tr.loss, tr.dout), name="volrender_l2_train (loss_sum)")
timeit(lambda: api.volrender_l2_train(tr.radiance, tr.t_vals, tr.num_stored, tr.indices, B, K, tg, 128.0, tr.pixels[:B], tr.loss_grads[:B], None, tr.dout), name="volrender_l2_train (no loss_sum)")
timeit(lambda: api.launch_volrender_cuda(None, tr.radiance, tr.num_stored, tr.indices, tr.t_vals, B, K, tr.pixels[:B], mode=api.VR_NERF), name="volrender fwd NERF")
timeit(lambda: api.launch_volrender_backward_cuda(None, tr.loss_grads, tr.radiance, tr.t_vals, tr.num_stored, tr.indices, B, K, tr.dout, mode=api.VR_NERF), name="volrender bwd NERF")
timeit(lambda: tr.loss.zero_(), name="loss.zero_ (launch floor)")

st = tr._stype()
timeit(lambda: tr.hg.backward_segments(tr.start, tr.end, P, st, tr.dencT, tr.dtable, tr.dtable_h), name="hash backward (2 launches)")
timeit(lambda: tr.hg.encode_segments(tr.table, tr.start, tr.end, tr.seg_view, P, st, tr.encT, tr.t_vals, 300.0), name="hash encode")
timeit(lambda: tr.net.train_forward_outputs(tr.encT, S, tr.out, tr.radiance), name="mlp fwd outputs")
timeit(lambda: tr.net.train_backward_recompute(tr.encT, tr.out, tr.dout, S, tr.dparams, tr.dencT), name="fused64 backward")

# traversal alone, and beside the compositor on a second stream
def trace():
    tr._segments(o, d, B)
timeit(trace, name="traversal (count+scan+write+host sync)")
s2 = torch.cuda.Stream()
def both():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        kw = dict(grid_res=tr.R, rays_o=o, rays_d=d, width=B, height=1, ray_begin=0, ray_count=B, occupancy=tr.occ, occupancy_coarse=tr.coarse,
                  occupancy_bricks=tr.bricks, occupancy_super=tr.super_mip, mode=api.TRACE_DDA, viewing_direction=tr.view_dirs, num_hits=tr.num_hits,
                  sub_rays=tr.sub_rays, sub_hits=tr.sub_hits)
        api.trace_grid(None, **kw)
        api.scan_hits(tr.num_hits[:B], tr.indices[:B], tr.total, tr.scan_ws)
    api.volrender_l2_train(tr.radiance, tr.t_vals, tr.num_stored, tr.indices, B, K, tg, 128.0, tr.pixels[:B], tr.loss_grads[:B], tr.loss, tr.dout)
    torch.cuda.current_stream().wait_stream(s2)
timeit(both, name="compositor || traversal count pass + scan (2 streams)")

# how sparse is the gradient that reaches the hash scatter?
de = tr.dencT[:32, :S]
nz = (de != 0).any(dim=0)
w = nz.view(-1, 64).any(dim=1) if S % 64 == 0 else nz[: S // 64 * 64].view(-1, 64).any(dim=1)
print(f"samples with any non-zero grid gradient: {100.0 * nz.float().mean().item():.1f} %   waves (64 samples) with any: {100.0 * w.float().mean().item():.1f} %")
do = tr.dout[:S].float()
print(f"samples with non-zero dout: {100.0 * (do != 0).any(dim=1).float().mean().item():.1f} %")
