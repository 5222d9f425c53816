#!/usr/bin/env python3
"""Bytes per sample of the 8x128 training MLP kernels from the FETCH_SIZE / WRITE_SIZE passes of `tools/pmc_extras.py ref8x128`
(tools/collect_profiles.sh phase pmc4, BEFORE `finish` removes the raw passes):
  python tools/pmc_lean_bytes.py <fetch counter_collection.csv> <write counter_collection.csv> [samples per launch]
Only the launches of the 22,528-ray batch are taken (a kernel's longest launches: within 30 % of its longest).  FETCH_SIZE is
quoted as counted and doubled (gfx950 tallies a wide coalesced read at half its bytes, MI355X_MICROARCH.md)."""
import csv
import json
import sys
from collections import defaultdict

fetch, write = sys.argv[1], sys.argv[2]
S = float(sys.argv[3]) if len(sys.argv) > 3 else 4_695_827.0
KERN = {"forward (outputs + sign masks)": "mlp_train_fwd_kernel<128, 2, 4, 1>", "dgrad chain": "mlp_bwd_kernel<128>",
        "weight gradient, recomputed activations": "wgrad_recompute_all_kernel<7, true>",
        "forward (saved activations)": "mlp_train_fwd_kernel<128, 1,", "weight gradient (saved activations)": "wgrad_lds_kernel"}


def read(path, counter):
    rows = defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            for name, pat in KERN.items():
                if pat in r["Kernel_Name"]:
                    rows[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), float(r["Counter_Value"])))
    out = {}
    for name, v in rows.items():
        longest = max(d for d, _ in v)
        big = [c for d, c in v if d >= 0.7 * longest]
        out[name] = (sum(big) / len(big) * 1024.0, len(big), longest / 1e6)       # KiB -> bytes
    return out


f, w = read(fetch, "FETCH_SIZE"), read(write, "WRITE_SIZE")
res, tot_lo, tot_hi = {}, 0.0, 0.0
for name in KERN:
    if name not in f or name not in w:
        continue
    lo = (f[name][0] + w[name][0]) / S
    hi = (2 * f[name][0] + w[name][0]) / S
    res[name] = {"launches": f[name][1], "kernel_ms_under_counters": round(f[name][2], 3), "fetch_bytes_per_sample": round(f[name][0] / S, 1),
                 "write_bytes_per_sample": round(w[name][0] / S, 1), "bytes_per_sample_as_counted": round(lo, 1),
                 "bytes_per_sample_fetch_doubled": round(hi, 1)}
    if "saved" not in name:
        tot_lo += lo
        tot_hi += hi
res["lean path, three kernels"] = {"bytes_per_sample_as_counted": round(tot_lo, 1), "bytes_per_sample_fetch_doubled": round(tot_hi, 1),
                                   "samples_per_launch_assumed": S}
print(json.dumps(res, indent=1))
