#!/usr/bin/env python3
"""Which kernels actually run BESIDE a resident MLP kernel?  Launches the MLP kernel on one stream and one other kernel on a
second stream right behind its start, and reports when the other kernel finished.  (A kernel whose waves do not fit next to
the MLP block's 2 x VGPRs per SIMD / 132 KiB LDS only finishes when the MLP kernel ends.)   python tools/co_run.py"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from rtx_nerf_amd import api, scenes
from _stages import Stages
torch.cuda.set_device(0)
R, W, H = 128, 800, 800
occ = torch.from_numpy(scenes.pack_occupancy(scenes.lego_standin_density(R, seed=0)).view(np.int32).copy()).cuda()
net = api.Network(); net.set_params(torch.from_numpy(scenes.xavier_params_fp16(128, 8, net.encoded_width(), seed=1337)).cuda())
la = scenes.pose_spherical(15.0, -30.0, origin_scale=10.0)
pipe = Stages(net, R, W, H, scenes.lego_focal_length(True), occ)
pipe.size_for(la); pipe.geometry(); pipe.shade(); pipe.composite(); torch.cuda.synchronize()
side = torch.cuda.Stream()
x = torch.zeros(1 << 20, device="cuda")
n = W * H
def trial(kind):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e_mlp = torch.cuda.Event(enable_timing=True)
    s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    pipe.shade()
    e_mlp.record()
    with torch.cuda.stream(side):
        side.wait_event(e0)
        s0.record(side)
        if kind == "fill":
            x.fill_(1.0)
        elif kind == "trace_count":
            pipe2.trace(0, n, False)
        elif kind == "scan":
            pipe2.scan(n)
        elif kind == "composite":
            pipe.composite()
        s1.record(side)
    torch.cuda.synchronize()
    return e0.elapsed_time(e_mlp), e0.elapsed_time(s0), e0.elapsed_time(s1)
pipe2 = Stages(net, R, W, H, scenes.lego_focal_length(True), occ)
pipe2.size_for(la); pipe2.geometry(); torch.cuda.synchronize()
for kind in ("fill", "trace_count", "scan", "composite", "fill"):
    for _ in range(2):
        m, a, b = trial(kind)
    print(f"{kind:12s} MLP done at {m:7.3f} ms; side kernel started {a:7.3f} finished {b:7.3f} ms")
