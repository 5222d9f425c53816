#!/usr/bin/env python3
"""Build profiles/rNN/mlp_fwd_pmc.json (what bench.py reads for roofline.traffic) from separate rocprofv3 --pmc passes.
  python tools/make_pmc_json.py <out.json> <samples_per_launch> <source note> <fetch.csv> <write.csv> <sq.csv>
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a wide coalesced read at half its bytes
(MI355X_MICROARCH.md, HBM): the read side is therefore quoted as the interval [1x, 2x] and bench.py reports the high end."""
import csv
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_src_sha16():
    """Same fingerprint bench.py computes: the summary is only quoted for the kernel source it was measured on."""
    h = hashlib.sha256()
    for f in ("rtx_nerf_amd/csrc/mlp.hip", "rtx_nerf_amd/csrc/mlp_internal.h"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def means(path, pat="mlp_fwd16_kernel"):   # the dominant kernel of the bench frame (16x16x32 form)
    acc, dur, name = defaultdict(list), [], None
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if pat in row["Kernel_Name"]:
                name = row["Kernel_Name"]
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    return {k: sum(v) / len(v) for k, v in acc.items()}, sum(dur) / max(1, len(dur)), name


out, samples, note, fetch, write, sq = sys.argv[1:7]
f, _, name = means(fetch)
w, _, _ = means(write)
s, ms, _ = means(sq)
cycles = s["GRBM_GUI_ACTIVE"] / 8
d = {
    "source": note,
    "kernel_src_sha16": kernel_src_sha16(),
    "kernel": name,
    "samples_per_launch": int(samples),
    "FETCH_SIZE_KiB_avg": round(f["FETCH_SIZE"], 2),
    "WRITE_SIZE_KiB_avg": round(w["WRITE_SIZE"], 2),
    **{k + "_avg": round(v, 1) for k, v in s.items()},
    "kernel_ms_under_counters": round(ms, 3),
    "effective_clock_ghz": round(cycles / (ms * 1e-3) / 1e9, 3),
    "mfma_busy_frac": round(s["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cycles), 4),
    "lds_busy_frac": round(s["SQ_LDS_IDX_ACTIVE"] / (256 * cycles), 4),
    "hbm_bytes_per_launch_low": int((f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024),
    "hbm_bytes_per_launch_high": int((2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024),
}
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d, indent=1))
