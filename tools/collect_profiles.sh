#!/bin/bash
# Collects the rocprofv3 evidence quoted in DESIGN.md / profiles/README.md (run on the GPU box from the repo root):
#   tools/collect_profiles.sh [outdir]        default gpurun_out/profiles
# Kernel traces/stats and the three PMC passes are separate rocprofv3 runs (counters never together with other trace domains).
set -e
out=${1:-gpurun_out/profiles}
repo=$(pwd)
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/bench_n1.json" 2> "$out/bench_n1.err"
echo "bench done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/kt_pipelined" -- python3 "$repo/bench.py" --steps 5 --warmup 2 --no-cpu > "$repo/$out/bench_n1_under_rocprof.json" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/kt_serial" -- python3 "$repo/bench.py" --steps 5 --warmup 2 --no-cpu --no-extras --serial > "$repo/$out/serial_bench_n1_under_rocprof.json" 2>/dev/null
echo "kernel traces done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$repo/$out/pmc_$c" -- python3 "$repo/bench.py" --steps 3 --warmup 1 --no-cpu --no-extras --kernel-steps 2 --serial > /dev/null 2>&1
done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
  --output-format csv -d "$repo/$out/pmc_sq" -- python3 "$repo/bench.py" --steps 3 --warmup 1 --no-cpu --no-extras --kernel-steps 2 --serial > /dev/null 2>&1
echo "pmc done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/kt_train3" -- python3 "$repo/bench_train.py" --steps 50 > "$repo/$out/train_config3.json" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/kt_train128" -- python3 "$repo/bench_train.py" --steps 50 --encoding freq --neurons 128 --layers 8 --dir-freqs 12 > "$repo/$out/train_8x128_freq.json" 2>/dev/null
echo "train traces done"
cd "$repo"
python3 tools/stage_bench.py > "$out/stage_bench.txt" 2>&1
python3 tools/probe/train_stage_probe.py > "$out/train_stage_probe.txt" 2>&1 || true
python3 tools/probe/captured_probe.py > "$out/captured_probe.txt" 2>&1 || true
python3 -m pytest tests/test_gpu_parity.py -q -k encoder_error > "$out/encoder_octave_error.txt" 2>&1 || true
# flatten what gets committed
cp $(ls $out/kt_pipelined/*/*kernel_stats.csv | head -1) $out/pipelined_bench_n1_kernel_stats.csv
cp $(ls $out/kt_serial/*/*kernel_stats.csv | head -1) $out/serial_bench_n1_kernel_stats.csv
cp $(ls $out/kt_train3/*/*kernel_stats.csv | head -1) $out/train_config3_kernel_stats.csv
cp $(ls $out/kt_train128/*/*kernel_stats.csv | head -1) $out/train_8x128_freq_kernel_stats.csv
cp $(ls $out/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1) $out/pmc_fetch_size.csv
cp $(ls $out/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1) $out/pmc_write_size.csv
cp $(ls $out/pmc_sq/*/*counter_collection.csv | head -1) $out/pmc_sq.csv
smp=$(python3 -c "import json;print(int(json.load(open('$out/bench_n1.json'))['roofline']['samples_per_launch']))")
python3 tools/make_pmc_json.py $out/mlp_fwd_pmc.json $smp "rocprofv3 --pmc, separate passes (pmc_*.csv), bench.py --steps 3 --warmup 1 --no-cpu --no-extras --kernel-steps 2 --serial" $out/pmc_fetch_size.csv $out/pmc_write_size.csv $out/pmc_sq.csv > /dev/null
rm -rf $out/kt_* $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_sq
ls -la $out
