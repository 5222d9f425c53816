#!/bin/bash
# PMC passes over bench_train.py (configs[2]) for the training kernels: tools/pmc_train.sh [outdir]
set -e
out=${1:-gpurun_out/pmc_train}
repo=$(pwd)
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU \
  --output-format csv -d "$repo/$out/p1" -- python3 "$repo/bench_train.py" --steps 20 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA \
  --output-format csv -d "$repo/$out/p2" -- python3 "$repo/bench_train.py" --steps 20 > /dev/null 2>&1 || true
cd "$repo"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2"):
    files = glob.glob(f"{out}/{p}/*/*counter_collection.csv")
    if not files:
        print(p, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if any(t in k for t in ("fused64", "hashgrid", "volrender_l2", "mlp_train_fwd")):
            print(k)
            for c, v in sorted(d.items()):
                print(f"   {c:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
PY
