#!/bin/bash
# Collects the rocprofv3 evidence quoted in DESIGN.md / profiles/README.md (run on the GPU box from the repo root):
#   tools/collect_profiles.sh [outdir] [phases]        default gpurun_out/profiles, phases "bench traces pmc1 pmc2 pmc3 pmc4 train finish"
#   (phase `ref`: the 8x128 reference iteration's counters and trace alone; then `PMC_MERGE=profiles/rNN/pmc_kernels.json ... finish`)
# (the whole set takes longer than one gpurun call allows: run `... bench traces pmc1` and `... pmc2 train finish` as two calls;
# `finish` flattens whatever the earlier phases left under <outdir>)
# Kernel traces/stats and the PMC passes are separate rocprofv3 runs (counters never together with other trace domains; FETCH_SIZE
# and WRITE_SIZE each in a pass of their own).  The program itself follows `--` (never a launcher): python3 <script>.
set -e
out=${1:-gpurun_out/profiles}
phases=${2:-bench traces pmc1 pmc2 pmc3 pmc4 train finish}
has() { case " $phases " in *" $1 "*) return 0;; *) return 1;; esac; }
repo=$(pwd)
mkdir -p "$out"
export TMPDIR=/tmp
if has bench; then
python3 bench.py > "$out/bench_n1.json" 2> "$out/bench_n1.err"
echo "bench done"
fi
cd /tmp
if has traces; then
rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/kt_pipelined" -- python3 "$repo/bench.py" --steps 5 --warmup 2 --no-cpu > "$repo/$out/bench_n1_under_rocprof.json" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/kt_serial" -- python3 "$repo/bench.py" --steps 5 --warmup 2 --no-cpu --no-extras --serial > "$repo/$out/serial_bench_n1_under_rocprof.json" 2>/dev/null
echo "kernel traces done"
fi
SQ="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
TCC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum"
# round 4: what the vector L1 / texture path is asked for (the hash-grid gather's ceiling) and what atomics leave the L2
# (two counters per pass, each pass under its own timeout with a heartbeat: six TCP counters in one pass never came back)
TCP1="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
TCP2="TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"
ATOM="TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum"
extra() {  # extra <tag> <counters...>: one pass of tools/pmc_extras.py config3
  tag=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$repo/$out/pmc_config3_$tag" -- python3 "$repo/tools/pmc_extras.py" config3 > /dev/null 2>&1 || echo "pass $tag failed or timed out"
  echo "pmc config3 $tag done"
}
pmc() {   # pmc <tag> <script> <args...>: four passes of one command
  tag=$1; shift
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$repo/$out/pmc_${tag}_fetch" -- python3 "$@" > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$repo/$out/pmc_${tag}_write" -- python3 "$@" > /dev/null 2>&1
  rocprofv3 --pmc $SQ --output-format csv -d "$repo/$out/pmc_${tag}_sq" -- python3 "$@" > /dev/null 2>&1
  rocprofv3 --pmc $TCC --output-format csv -d "$repo/$out/pmc_${tag}_tcc" -- python3 "$@" > /dev/null 2>&1 || true
  echo "pmc $tag done"
}
if has pmc1; then
pmc headline "$repo/bench.py" --steps 3 --warmup 1 --no-cpu --no-extras --kernel-steps 2 --serial
pmc config5 "$repo/tools/pmc_extras.py" config5
fi
if has pmc2; then
pmc config3 "$repo/tools/pmc_extras.py" config3
fi
if has pmc3; then
extra tcp $TCP1
extra tcp2 $TCP2
extra atom $ATOM
fi
if has pmc4; then
pmc ref8x128 "$repo/tools/pmc_extras.py" ref8x128
fi
if has ref; then   # only the 8x128 reference iteration again (after a change to its kernels): counters + kernel trace
pmc ref8x128 "$repo/tools/pmc_extras.py" ref8x128
rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/kt_ref8x128" -- python3 "$repo/tools/pmc_extras.py" ref8x128 > "$repo/$out/train_ref8x128.json" 2>/dev/null
echo "ref8x128 done"
fi
if has train; then
rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/kt_train3" -- python3 "$repo/bench_train.py" --steps 50 > "$repo/$out/train_config3.json" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/kt_train128" -- python3 "$repo/bench_train.py" --steps 50 --encoding freq --neurons 128 --layers 8 --dir-freqs 12 > "$repo/$out/train_8x128_freq.json" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/kt_ref8x128" -- python3 "$repo/tools/pmc_extras.py" ref8x128 > "$repo/$out/train_ref8x128.json" 2>/dev/null
echo "train traces done"
cd "$repo"
python3 tools/stage_bench.py > "$out/stage_bench.txt" 2>&1 || true
python3 -m pytest tests/test_gpu_parity.py -q -k encoder_error > "$out/encoder_octave_error.txt" 2>&1 || true
fi
cd "$repo"
has finish || exit 0
# flatten what gets committed
flat() { f=$(ls $1 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" "$2" || true; }
flat "$out/kt_pipelined/*/*kernel_stats.csv" $out/pipelined_bench_n1_kernel_stats.csv
flat "$out/kt_serial/*/*kernel_stats.csv" $out/serial_bench_n1_kernel_stats.csv
flat "$out/kt_train3/*/*kernel_stats.csv" $out/train_config3_kernel_stats.csv
flat "$out/kt_train128/*/*kernel_stats.csv" $out/train_8x128_freq_kernel_stats.csv
flat "$out/kt_ref8x128/*/*kernel_stats.csv" $out/train_ref8x128_kernel_stats.csv
flat "$out/pmc_headline_fetch/*/*counter_collection.csv" $out/pmc_fetch_size.csv
flat "$out/pmc_headline_write/*/*counter_collection.csv" $out/pmc_write_size.csv
flat "$out/pmc_headline_sq/*/*counter_collection.csv" $out/pmc_sq.csv
# PMC_MERGE=<earlier pmc_kernels.json>: a partial re-collection (phase `ref`) keeps the earlier entries whose kernels are unchanged
python3 tools/pmc_kernels_json.py ${PMC_MERGE:+--merge $PMC_MERGE} $out/pmc_kernels.json "${PMC_NOTE:-rocprofv3 --pmc, separate passes per counter set (FETCH_SIZE | WRITE_SIZE | SQ group | TCC group) over: bench.py --steps 3 --warmup 1 --no-cpu --no-extras --kernel-steps 2 --serial; tools/pmc_extras.py config5 | config3 | ref8x128}" \
  $(ls $out/pmc_*/*/*counter_collection.csv) > $out/pmc_kernels_summary.txt
flat "$out/pmc_config3_tcp/*/*counter_collection.csv" $out/pmc_config3_tcp.csv
flat "$out/pmc_config3_tcp2/*/*counter_collection.csv" $out/pmc_config3_tcp2.csv
flat "$out/pmc_config3_atom/*/*counter_collection.csv" $out/pmc_config3_atomics.csv
rm -rf $out/kt_* $out/pmc_*_fetch $out/pmc_*_write $out/pmc_*_sq $out/pmc_*_tcc $out/pmc_*_tcp $out/pmc_*_tcp2 $out/pmc_*_atom
ls -la $out
