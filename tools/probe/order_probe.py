#!/usr/bin/env python3
"""the 4096-ray / 128^3 training record measured AFTER the two large-batch records (70-GB workspaces allocated
and released in the same process) took twice as long, and bench.py's order was swapped.  This probe runs small -> large -> small in
one process and prints, for each small run, the wall-clock step, the per-stage HIP-event times, and torch's allocator state --
kernel time (events) against wall clock tells a slower kernel from gaps between kernels.
  order_probe.py [--saved]     (--saved: RTXN_TRAIN_LEAN=0, the 40/70-GB workspaces of round 3)"""
import os
import sys

if "--saved" in sys.argv:
    os.environ["RTXN_TRAIN_LEAN"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import json
import torch

import bench

torch.cuda.set_device(0)


def small(tag):
    r = bench.train_ref_record(4096, 128, 30, 3, dense_grid=False, mode="nerf")
    st = torch.cuda.memory_stats()
    print(f"{tag}: captured step {r['ms_per_step']:.4f} ms, host-count step {r['ms_per_step_host_count']:.4f} ms, sum of stage events "
          f"{sum(r['stage_ms'].values()):.4f} ms; reserved {st['reserved_bytes.all.current'] / 2**30:.2f} GiB, "
          f"segments {st['segment.all.current']}, cudaMalloc retries {st['num_alloc_retries']}", flush=True)
    print("   stages:", json.dumps(r["stage_ms"]), flush=True)
    return r


a = small("small, fresh process")
big = bench.train_ref_record(256 * 176, 8, 4, 2, captured=False)
print(f"large (45,056 rays): {big['ms_per_step']:.3f} ms, workspace {big['mlp_workspace_gib']} GiB", flush=True)
b = small("small, after the large batch")
torch.cuda.empty_cache()
c = small("small, after empty_cache()")
print(f"ratio after/before: captured {b['ms_per_step'] / a['ms_per_step']:.2f}, host-count {b['ms_per_step_host_count'] / a['ms_per_step_host_count']:.2f}, "
      f"stage events {sum(b['stage_ms'].values()) / sum(a['stage_ms'].values()):.2f}")

# where the eager step's extra time sits after empty_cache(): per-step wall clock of 40 eager steps of a fresh small trainer
import time
import numpy as np
from rtx_nerf_amd import scenes
from rtx_nerf_amd.train import Trainer, camera_rays

torch.cuda.empty_cache()
R, B = 128, 4096
occ = torch.from_numpy(scenes.pack_occupancy(scenes.lego_standin_density(R, seed=0)).view(np.int32).copy()).cuda()
tr = Trainer(R, occ, encoding="freq", n_neurons=128, n_hidden_layers=8, n_dir_freqs=12, batch_rays=B, max_segments=B * 48, lr=1e-3,
             loss_scale=128.0, density_scale=300.0, mode="nerf")
o, d = camera_rays(scenes.pose_spherical(15.0, -30.0, origin_scale=10.0), scenes.lego_focal_length(True), 64, 64)
t = torch.rand((B, 3), device="cuda")
times = []
for i in range(40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(o, d, t)
    torch.cuda.synchronize()
    times.append(1e3 * (time.perf_counter() - t0))
print("eager steps after empty_cache(), ms each:", " ".join(f"{x:.2f}" for x in times))
