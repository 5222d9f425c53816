#!/bin/bash
# kernel-trace summary of one probe:  tools/probe/kt.sh <tag> <script> [args...]   -> gpurun_out/kt_<tag>.csv (kernel stats), lines of interest on stdout
tag=$1; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/gpurun_out/kt_$tag" -- python3 "$repo/$1" "${@:2}" > "$repo/gpurun_out/kt_$tag.txt" 2>&1
cd "$repo"
f=$(ls gpurun_out/kt_$tag/*/*kernel_stats.csv | head -1)
cp "$f" gpurun_out/kt_$tag.csv
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("wgrad", "mlp_bwd", "mlp_train_fwd", "hashgrid", "adam")):
        print(f'{float(r["AverageNs"]) / 1e6:8.3f} ms x{r["Calls"]:>4}  min {float(r["MinNs"]) / 1e6:.3f}  {n[:110]}')
PY
rm -rf gpurun_out/kt_$tag
