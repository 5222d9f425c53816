#!/usr/bin/env python3
"""How robust is tools/train_demo.py's convergence?  Runs it for several seeds (same process) and prints the loss trajectory:
   python tools/probe/converge_probe.py [n_seeds] [steps]      (RTXN_TRAIN_FWD16=0: the 32x32x16 outputs-only forward)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import train_demo
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
for seed in range(n):
    p0, p1, losses = train_demo.run(steps=steps, encoding="hash", seed=seed, verbose=False)
    print(f"seed {seed}: psnr {p0:.2f} -> {p1:.2f}  losses {[round(x, 5) for x in losses]}", flush=True)
