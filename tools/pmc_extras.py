#!/usr/bin/env python3
"""The extra bench records as a short standalone command for rocprofv3 passes (tools/collect_profiles.sh):
  python tools/pmc_extras.py config3   -> bench.extra_train_config3 (train_config3 + render_hash4x64), few steps
  python tools/pmc_extras.py ref8x128  -> bench.train_ref_record(22528 rays, dense 8^3) and the 4096-ray NeRF variant
  python tools/pmc_extras.py config5   -> bench.extra_config5"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

torch.cuda.set_device(0)
what = sys.argv[1] if len(sys.argv) > 1 else "config3"
if what == "config3":
    a, b = bench.extra_train_config3(8, 3, kernel_steps=2, train_to=60, frames=4)
    print(json.dumps({"train_config3": a, "render_hash4x64": b}))
elif what == "ref8x128":
    print(json.dumps({"b4096_lego128_nerf": bench.train_ref_record(4096, 128, 8, 2, dense_grid=False, mode="nerf"),
                      "b22528_dense8": bench.train_ref_record(128 * 176, 8, 3, 1, captured=False)}))
else:
    print(json.dumps({"config5": bench.extra_config5(4, 2, 2)}))
