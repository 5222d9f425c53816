#!/usr/bin/env python3
"""A/B of the hash table's optimizer on configs[2]: RTXN_TABLE_ADAM=dense (every entry every step, global step count) against the
default (tiny-cuda-nn's rule: zero-gradient entries skipped, per-entry counts).  Step time and PSNR of held-out views after 1500
steps, from bench.extra_train_config3."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
torch.cuda.set_device(0)
a, b = bench.extra_train_config3(40, 5, kernel_steps=2, train_to=1500, frames=4)
print(json.dumps({"table_adam": os.environ.get("RTXN_TABLE_ADAM", "sparse"), "ms_per_step": a["ms_per_step"], "ms_per_step_host_count": a.get("ms_per_step_host_count"),
                  "adam_ms": a["stage_ms"].get("adam"), "loss": [a.get("loss_first"), a.get("loss_last")],
                  "psnr": {k: v for k, v in b.items() if "psnr" in k or "train" in k}}))
