# A/B of a kernel change on the GPU box: parity subset, then the shipped library against rtx_nerf_amd/librtxn_base.so (built
# beforehand with tools/ablate.sh base="..."), alternating processes, isolated kernel timing (tools/mlp_bench.py)
set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "mlp or fused or determin or encoder or render" > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -2 gpurun_out/ab_tests.log
for i in 1 2 3; do
  echo "new:  $(python tools/mlp_bench.py --iters 12 2>/dev/null | tail -1)"
  echo "base: $(RTXN_LIB_PATH=rtx_nerf_amd/librtxn_base.so python tools/mlp_bench.py --iters 12 2>/dev/null | tail -1)"
done 2>&1 | tee gpurun_out/ab.txt
if [ -f rtx_nerf_amd/librtxn_stamps.so ]; then RTXN_LIB_PATH=rtx_nerf_amd/librtxn_stamps.so python tools/probe/stamps.py 2>/dev/null | tee gpurun_out/stamps.txt; fi
