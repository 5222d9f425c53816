#!/usr/bin/env python3
"""Prints the render_hash4x64 and train_config3 figures of a tools/pmc_extras.py config3 run (one JSON line)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h, t = d["render_hash4x64"], d["train_config3"]
print("hash render", h["mrays_s"], "Mrays/s, kernel", h["roofline"]["kernel_ms"], "ms; config3 step", t["ms_per_step"], "ms, mlp_fwd", t["stage_ms"].get("mlp_fwd"))
