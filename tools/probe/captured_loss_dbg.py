import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from rtx_nerf_amd import scenes
from rtx_nerf_amd.train import Trainer, camera_rays
R, B = 128, 4096
dense = scenes.lego_standin_density(R, seed=0)
occ = torch.from_numpy(scenes.pack_occupancy(dense).view(np.int32).copy()).cuda()
focal = scenes.lego_focal_length(True)
ro, rd = [], []
for i in range(8):
    o, d = camera_rays(scenes.pose_spherical(45.0 * i + 15.0, -30.0, origin_scale=10.0), focal, 256, 256)
    ro.append(o); rd.append(d)
ro, rd = torch.cat(ro), torch.cat(rd)
g = torch.Generator(device="cuda").manual_seed(42)
tg = torch.rand((ro.shape[0], 3), device="cuda", generator=g)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pre = (sys.argv[3] == "1") if len(sys.argv) > 3 else True
MS = int(sys.argv[4]) if len(sys.argv) > 4 else 40000
LS = int(sys.argv[5]) if len(sys.argv) > 5 else 30000
SYNC = (sys.argv[6] == "1") if len(sys.argv) > 6 else True
NSTEP = int(sys.argv[7]) if len(sys.argv) > 7 else 6
if len(sys.argv) > 8:          # poison the caching allocator's blocks: torch.empty() buffers then start as NaN / huge values
    for val in (float("nan"), -4e34):
        junk = [torch.full((1 << 28,), val, device="cuda") for _ in range(24)]
        del junk
tr = Trainer(R, occ, encoding="freq", n_neurons=W, n_hidden_layers=L, n_dir_freqs=12, batch_rays=B, max_segments=MS, lr=1e-3,
             loss_scale=128.0, density_scale=300.0, mode="nerf")
print("two_pass", tr.two_pass, "recompute", tr.recompute)
def batch():
    idx = torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g)
    return ro[idx].contiguous(), rd[idx].contiguous(), tg[idx].contiguous()
for i in range(3):
    print("eager", i, float(tr.step(*batch()).item()), int(tr.total.item()))
tr.capture_step(B, launch_segments=LS, prefetch=pre)
for i in range(NSTEP):
    idx = torch.randint(0, ro.shape[0], (B,), device="cuda", generator=g)
    torch.index_select(ro, 0, idx, out=tr.graph_rays_o)
    torch.index_select(rd, 0, idx, out=tr.graph_rays_d)
    torch.index_select(tg, 0, idx, out=tr.graph_targets)
    l = tr.step_captured()
    if not SYNC and i < NSTEP - 1:
        continue
    torch.cuda.synchronize()
    print("captured", i, None if l is None else float(l.item()), "params finite", bool(torch.isfinite(tr.master).all()),
          "pixels finite", bool(torch.isfinite(tr.pixels[:B]).all()), "radiance finite", bool(torch.isfinite(tr.radiance[:20000*32]).all()))
lf = tr.flush_captured()
print("flush", None if lf is None else float(lf.item()), "truncated", tr.truncated_steps)
print("eager again", float(tr.step(*batch()).item()))
