#!/bin/bash
# Builds timing-only variants of librtxn.so (rtx_nerf_amd/librtxn_<tag>.so) with extra -D flags for mlp.hip:
#   tools/ablate.sh pipe2="-DRTXN_PIPE=2" noskew="-DRTXN_SKEW=0" ...   (value-preserving knobs only, see mlp.hip)
# then on the GPU:  RTXN_LIB_PATH=rtx_nerf_amd/librtxn_<tag>.so python tools/mlp_bench.py
set -e
cd "$(dirname "$0")/.."
make -s -j8 rtx_nerf_amd/librtxn.so
for kv in "$@"; do
  tag="${kv%%=*}"; flags="${kv#*=}"
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Iinclude -Irtx_nerf_amd/csrc $flags \
      -c rtx_nerf_amd/csrc/mlp.hip -o build/mlp_$tag.o 2>/dev/null &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o rtx_nerf_amd/librtxn_$tag.so build/mlp_$tag.o \
      $(ls build/*.o | grep -v "build/mlp") -lz && echo "built $tag" ) &
done
wait
