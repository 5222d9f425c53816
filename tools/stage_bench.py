#!/usr/bin/env python3
"""Times every standalone stage of the reference-shaped call sequence (traversal, scan, launchSampler,
launch_volrender_cuda, launch_volrender_backward_cuda, unfused MLP forward) on the bench frame (800x800, 128^3 Lego
stand-in) and prints each one's algorithmic bytes / time = achieved GB/s next to the HBM roofline.
  python tools/stage_bench.py [--iters 10]"""
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
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--width", type=int, default=800)
ap.add_argument("--height", type=int, default=800)
ap.add_argument("--grid", type=int, default=128)
a = ap.parse_args()
torch.cuda.set_device(0)
W, H, R, K = a.width, a.height, a.grid, api.NUM_SAMPLES_PER_SEGMENT
occ = torch.from_numpy(scenes.pack_occupancy(scenes.lego_standin_density(R, seed=0)).view(np.int32).copy()).cuda()
net = api.Network(n_neurons=128, n_hidden_layers=8)
net.set_params(torch.from_numpy(scenes.xavier_params_fp16(128, 8, net.encoded_width(), seed=1337)).cuda())
pose = scenes.pose_spherical(15.0, -30.0, origin_scale=10.0)
pipe = Stages(net, R, W, H, scenes.lego_focal_length(True), occ, compact=False)
P = pipe.size_for(pose)
pipe.geometry()
pipe.shade()
pipe.composite()
torch.cuda.synchronize()
n, S = W * H, P * K
samples = torch.empty((S, 5), device="cuda")
t_vals = torch.empty(S, device="cuda")
pixels = torch.empty((n, 3), device="cuda")
lgrad = (torch.randn((n, 3), device="cuda") * 0.1).half()
rgrad = torch.empty((S, 4), dtype=torch.float16, device="cuda")
rad16 = pipe.radiance[:S].half()


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(a.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms))


h = P / n   # mean segments per ray
stages = [
    ("trace_kernel<DDA> count pass", lambda: pipe.trace(0, n, False), 12 * n),
    ("trace_kernel<DDA> write pass", lambda: pipe.trace(0, n, True), 24 * n + 32 * P),
    ("scan_hits", lambda: pipe.scan(n), 8 * n),
    ("launchSampler REGULAR", lambda: api.launchSampler(pipe.start, pipe.end, pipe.view_dirs, t_vals, samples, n, R, pipe.num_hits_c,
                                                        pipe.indices, api.SAMPLING_REGULAR), 16 * n + 792 * P),
    ("launchSampler JITTER", lambda: api.launchSampler(pipe.start, pipe.end, pipe.view_dirs, t_vals, samples, n, R, pipe.num_hits_c,
                                                       pipe.indices, api.SAMPLING_STRATIFIED_JITTERING), 16 * n + 792 * P),
    ("mlp_forward_radiance (float[N][5] in)", lambda: net.forward_radiance(samples, pipe.radiance), None),
    ("launch_volrender_cuda COMPAT", lambda: api.launch_volrender_cuda(None, pipe.radiance, pipe.num_hits_c, pipe.indices, t_vals, n, K,
                                                                       pixels), 20 * n + 640 * P),
    ("rtxn_volrender_fwd_compact (half4, implicit t)", lambda: api.volrender_compact(rad16, pipe.num_hits_c, pipe.indices, n, K, pixels),
     20 * n + 256 * P),
    ("launch_volrender_cuda NERF", lambda: api.launch_volrender_cuda(None, pipe.radiance, pipe.num_hits_c, pipe.indices, t_vals, n, K,
                                                                     pixels, mode=api.VR_NERF), 20 * n + 640 * P),
    ("launch_volrender_backward_cuda COMPAT", lambda: api.launch_volrender_backward_cuda(None, lgrad, pipe.radiance, t_vals, pipe.num_hits_c,
                                                                                         pipe.indices, n, K, rgrad), 14 * n + 896 * P),
    ("launch_volrender_backward_cuda NERF", lambda: api.launch_volrender_backward_cuda(None, lgrad, pipe.radiance, t_vals, pipe.num_hits_c,
                                                                                       pipe.indices, n, K, rgrad, mode=api.VR_NERF),
     14 * n + 896 * P),
]
print(f"{W}x{H}, {R}^3: {n} rays, {P} segments ({h:.2f}/ray), {S} samples; HBM peak 8000 GB/s")
for name, fn, nbytes in stages:
    ms = timed(fn)
    if nbytes is None:
        fl = net.flops_per_sample() * S
        print(f"{name:44s} {ms:8.3f} ms   {fl / ms / 1e9:8.1f} TFLOP/s  ({fl / ms / 1e9 / 2500:.3f} of 2.5 PF)")
    else:
        print(f"{name:44s} {ms:8.3f} ms   {nbytes / 1e6:9.1f} MB  {nbytes / ms / 1e6:8.1f} GB/s  ({nbytes / ms / 1e6 / 8000:.3f} of HBM peak)")
