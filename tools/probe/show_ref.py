#!/usr/bin/env python3
"""Prints ms/step and the stage split of the records a tools/pmc_extras.py ref8x128 run wrote (one JSON line)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k, v in d.items():
    print(k, v["ms_per_step"], v.get("ms_per_step_host_count"), v["stage_ms"])
