#!/bin/bash
# Builds timing-only variants of librtxn.so (rtx_nerf_amd/librtxn_<tag>.so) with extra -D flags for mlp.hip:
#   tools/ablate.sh ring3="-DRTXN_PIPE16=3" noshare="-DRTXN_SHARE_DIR=0" ...
# Value-preserving knobs that exist today (mlp.hip, mlp_internal.h): RTXN_PIPE16 (A-fragment ring depth of the 16x16x32 pipeline),
# RTXN_SHARE_DIR (direction encoding shared across a segment), RTXN_STAMPS (diagnostic stamps, tools/probe/stamps.py); RTXN_PIPE is
# the ring depth of the training forward's 32x32x16 pipeline and RTXN_WG_K / RTXN_WG_STAGES / RTXN_WG_CHUNK the staging of
# wgrad_lds_kernel (train.hip: ABLATE_SRC=train tools/ablate.sh ...).  Gone with the legacy
# kernels in round 3: RTXN_SKEW, RTXN_ILV16, RTXN_MFMA_SHAPE.
# then on the GPU:  RTXN_LIB_PATH=rtx_nerf_amd/librtxn_<tag>.so python tools/mlp_bench.py
set -e
cd "$(dirname "$0")/.."
SRC="${ABLATE_SRC:-mlp}"     # which translation unit takes the flags: mlp (default) or train (RTXN_PIPE, RTXN_WG_K / _STAGES / _CHUNK)
make -s -j8 rtx_nerf_amd/librtxn.so
mkdir -p build/variants
for kv in "$@"; do
  tag="${kv%%=*}"; flags="${kv#*=}"
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Iinclude -Irtx_nerf_amd/csrc $flags \
      -c rtx_nerf_amd/csrc/$SRC.hip -o build/variants/${SRC}_$tag.o 2>/dev/null &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o rtx_nerf_amd/librtxn_$tag.so build/variants/${SRC}_$tag.o \
      $(ls build/*.o | grep -v "build/$SRC.o") -lz && echo "built $tag" ) &
done
wait
rm -rf build/variants        # variant objects must not be picked up by a later link of build/*.o
