#!/bin/bash
# PMC comparison of the two MFMA shapes of the fused inference kernel (run on the GPU box from the repo root):
#   tools/pmc_shapes.sh <outdir>     -> <outdir>/{sq,grbm_lds}/... counter_collection.csv, summarised by tools/pmc_summary.py
set -e
out=${1:-gpurun_out/pmc_shapes}
mkdir -p "$out"
export TMPDIR=/tmp
repo=$(pwd)
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS \
  --output-format csv -d "$repo/$out/sq" -- python3 "$repo/tools/mfma_shape_ab.py" --rounds 4 > "$repo/$out/sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU \
  --output-format csv -d "$repo/$out/grbm" -- python3 "$repo/tools/mfma_shape_ab.py" --rounds 4 > "$repo/$out/grbm.log" 2>&1
cd "$repo"
python3 - "$out" <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mlp_fwd" not in k:
            continue
        k = "16x16x32" if "fwd16" in k else "32x32x16"
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k in sorted(acc):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    ms = sum(dur[k]) / len(dur[k])
    line = [f"{k}: {ms:.3f} ms under counters"]
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        line.append(f"clock {cyc / (ms * 1e-3) / 1e9:.3f} GHz")
        if "SQ_LDS_IDX_ACTIVE" in m:
            line.append(f"lds_busy {m['SQ_LDS_IDX_ACTIVE'] / (256 * cyc):.3f}")
    print("  ".join(line))
    for c in sorted(m):
        print(f"    {c:34s} {m[c]:.4g}")
PY
