#!/bin/bash
# one rocprofv3 --pmc pass of a probe, per-kernel averages on stdout:  tools/probe/pmc.sh <tag> "<counters>" <script> [args...]
tag=$1; ctr=$2; shift 2
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d "$repo/gpurun_out/pmc_$tag" -- python3 "$repo/$1" "${@:2}" > "$repo/gpurun_out/pmc_$tag.txt" 2>&1 || echo "pass failed or timed out"
cd "$repo"
f=$(ls gpurun_out/pmc_$tag/*/*counter_collection.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if any(k in n for k in ("wgrad", "mlp_bwd", "mlp_train_fwd", "mlp_fwd16")):
        acc[n[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in acc.items():
    print(n)
    for k, v in c.items():
        print(f"   {k:32s} avg {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
rm -rf gpurun_out/pmc_$tag
