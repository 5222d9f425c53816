#!/usr/bin/env python3
"""Data-parallel training check: N ranks each train on 1/N of every global batch (gradients all-reduced)
and must end with the same parameters as ONE process training on the whole batches.
  single:  python tools/train_dp_check.py --out ref.npy
  N ranks: RTXN_REHEARSE_ON_ONE_GPU=1 python -m torch.distributed.run --nproc-per-node 2 tools/train_dp_check.py --out dp.npy
(the rehearsal switch puts every rank on cuda:0 with a gloo group; on a multi-GPU node omit it: one GPU per rank, RCCL)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from rtx_nerf_amd import scenes
from rtx_nerf_amd.train import Trainer, camera_rays

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
teacher_field = scenes.teacher_field

ap = argparse.ArgumentParser()
ap.add_argument("--out", required=True)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--encoding", default="hash")
ap.add_argument("--empty-odd", action="store_true", help="every odd ray of the global batch (= all of rank 1's share in a 2-rank run) "
                "points away from the grid: that rank has NO samples and must still take part in every collective and Adam step")
ap.add_argument("--captured", action="store_true", help="Trainer.capture_step / step_captured: gradients and optimizer as two hipGraphs "
                "with the all-reduces between them (the gradients of step 0 are not saved: the captured Adam clears them)")
a = ap.parse_args()
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
rehearse = os.environ.get("RTXN_REHEARSE_ON_ONE_GPU") == "1"
torch.cuda.set_device(0 if rehearse else int(os.environ.get("LOCAL_RANK", "0")))
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo" if rehearse else "nccl")

R, res, B = 32, 32, 2048
occ = torch.from_numpy(scenes.pack_occupancy(scenes.sphere_density(R, 0.72)).view(np.int32).copy()).cuda()
tr = Trainer(R, occ, encoding=a.encoding, n_neurons=64, n_hidden_layers=2,
             hashgrid=dict(n_levels=4, n_features=2, log2_hashmap_size=12, base_resolution=8, per_level_scale=1.5),
             batch_rays=B, max_segments=B * 40, lr=1e-2, loss_scale=128.0, density_scale=150.0, seed=0)
focal = scenes.lego_focal_length(True)
o, d = camera_rays(scenes.pose_spherical(20.0, -30.0, origin_scale=10.0), focal, res, res)
tgt = tr.render_rays(o, d, radiance_fn=teacher_field).clone()
g = torch.Generator().manual_seed(1)
for it in range(a.steps):
    idx = torch.randint(0, o.shape[0], (B,), generator=g).cuda()       # the same global batch on every rank
    mine = idx[rank::world].contiguous()                                # this rank's share
    dm = d[mine].contiguous()
    if a.empty_odd:
        odd = (torch.arange(mine.numel(), device="cuda") * world + rank) % 2 == 1   # position in the GLOBAL batch
        dm[odd] = -dm[odd]
    if a.captured:
        if it == 0:
            tr.capture_step(mine.numel())
        tr.graph_rays_o.copy_(o[mine]); tr.graph_rays_d.copy_(dm); tr.graph_targets.copy_(tgt[mine])
        tr.step_captured()
        if it == 0:
            g0 = np.zeros(tr.dparams.numel() + (tr.dtable.numel() if a.encoding == "hash" else 0), np.float32)
        continue
    tr.step(o[mine].contiguous(), dm, tgt[mine].contiguous())
    if it == 0:   # gradients of the first step (summed over ranks -> mean), before Adam's normalisation amplifies noise
        g0 = torch.cat([tr.dparams, tr.table_grad() if a.encoding == "hash" else tr.dparams[:0]]).cpu().numpy() / world
torch.cuda.synchronize()
assert tr.step_count == a.steps, (rank, tr.step_count)                  # every rank ran Adam in every step
if world > 1:                                                           # ... and they all hold the same parameters
    mine_p = tr.master.clone()
    ref_p = mine_p.clone()
    dist.broadcast(ref_p, src=0)
    assert torch.equal(mine_p, ref_p), f"rank {rank}: parameters diverged from rank 0"
    if a.encoding == "hash":
        ref_t = tr.table_master.clone()
        dist.broadcast(ref_t, src=0)
        assert torch.equal(tr.table_master, ref_t), f"rank {rank}: hash table diverged from rank 0"
        if rank == 0 and tr._dp_table is not None and tr._dp_table.last is not None:
            print("hashed levels, last step's exchange:", {k: v for k, v in tr._dp_table.last.items()}, flush=True)
if rank == 0:
    np.save(a.out, np.stack([g0, np.concatenate([tr.master.cpu().numpy(),
                                                 tr.table_master.cpu().numpy() if a.encoding == "hash" else []])]))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
