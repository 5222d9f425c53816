#!/usr/bin/env python3
"""Time the traversal kernel (count + write passes) on a bench frame with/without the brick occupancy copy.
  python tools/trace_bench.py [--scene lego|llff] [--grid 128]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

from rtx_nerf_amd import api, scenes
from _stages import Stages

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="lego")
ap.add_argument("--grid", type=int, default=128)
ap.add_argument("--width", type=int, default=800)
ap.add_argument("--height", type=int, default=800)
ap.add_argument("--sub-rays", type=int, default=0, help="lanes per ray (rtxn_trace_params.sub_rays)")
ap.add_argument("--shard-of", type=int, default=1, help="trace only rank 0's row shard of an N-rank run")
a = ap.parse_args()
torch.cuda.set_device(0)
R = a.grid
dense = scenes.lego_standin_density(R, 0) if a.scene == "lego" else scenes.llff_standin_density(R, 3)
occ = torch.from_numpy(scenes.pack_occupancy(dense).view(np.int32).copy()).cuda()
net = api.Network(n_neurons=64, n_hidden_layers=2)
net.set_params(torch.from_numpy(scenes.xavier_params_fp16(64, 2, 112)).cuda())
la = scenes.pose_spherical(15.0, -30.0, origin_scale=10.0) if a.scene == "lego" else scenes.pose_forward_facing(0.2, 0.1)
f = scenes.lego_focal_length(True) if a.scene == "lego" else 1.6
from rtx_nerf_amd.shard import RowShard
sh = RowShard(a.width, a.height, 0, a.shard_of)
pipe = Stages(net, R, a.width, a.height, f, occ, max_rays=sh.n_local, window=sh.window, sub_rays=a.sub_rays)
pipe.size_for(la, ray_begin=sh.ray_begin, n=sh.n_local)
n = sh.n_local
sup = pipe.super_mip
for label, bricks, pipe.super_mip in (("3-level", pipe.bricks, sup), ("2-level", pipe.bricks, None), ("3-level", pipe.bricks, sup),
                                      ("2-level", pipe.bricks, None)):
    pipe.bricks = bricks
    ms = {False: [], True: []}
    for _ in range(12):
        for write in (False, True):
            if write:
                pipe.scan(n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pipe.trace(sh.ray_begin, n, write)
            e1.record()
            torch.cuda.synchronize()
            ms[write].append(e0.elapsed_time(e1))
    print(f"rays {n} sub_rays {a.sub_rays} {label:10s} count {np.median(ms[False]):.4f} ms  write {np.median(ms[True]):.4f} ms  segments {int(pipe.total.item())}")
