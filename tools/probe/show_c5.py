#!/usr/bin/env python3
"""Prints the config5 figures of a tools/pmc_extras.py config5 run (one JSON line)."""
import json, sys
c = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["config5"]
print("config5", c["mrays_s"], "Mrays/s, kernel", c["roofline"]["kernel_ms"], "ms, MFMA", c["roofline"]["frac"])
