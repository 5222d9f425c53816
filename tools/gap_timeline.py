#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace CSV of bench.py: what runs between one MLP kernel's end and the next one's start.
  python tools/gap_timeline.py <kernel_trace.csv> [index of the MLP launch to inspect]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ml = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "mlp_fwd" in r["Kernel_Name"])
gaps = [(ml[i][0] - ml[i - 1][1]) / 1e3 for i in range(1, len(ml))]
print("MLP launches", len(ml), "gaps (us):", " ".join(f"{g:.0f}" for g in gaps))
a, b = ml[k - 1][1], ml[k][0]
print(f"between MLP {k-1} end and MLP {k} start ({(b - a) / 1e3:.0f} us):")
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "mlp_fwd" in r["Kernel_Name"] or "pack_kernel" in r["Kernel_Name"]:
        continue
    if e > ml[k - 1][0] and s < b + 50_000:
        name = r["Kernel_Name"].split("(")[0][-44:]
        print(f"  {name:44s} start {(s - a) / 1e3:10.1f} us   end {(e - a) / 1e3:10.1f} us")
