#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per counter, the mean value over the launches of kernels whose
name contains a pattern.   python tools/pmc_summary.py <pattern> <csv> [<csv> ...]"""
import csv
import sys
from collections import defaultdict

pat = sys.argv[1]
for path in sys.argv[2:]:
    acc = defaultdict(list)
    dur = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if pat in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    print(path)
    if dur:
        print(f"  launches {len(dur) // max(1, len(acc))}  mean duration {sum(dur) / len(dur):.3f} ms (under counters)")
    for k, v in sorted(acc.items()):
        print(f"  {k:36s} {sum(v) / len(v):.6g}")
