# usage: bash tools/probe/variants.sh tag1 tag2 ...   (libraries rtx_nerf_amd/librtxn_<tag>.so from tools/ablate.sh; "cur" = the shipped one)
# two alternating rounds of isolated kernel timing per variant, then the stamp profile of every <tag>s.so that exists
for r in 1 2; do
  for t in "$@"; do
    if [ "$t" = cur ]; then L=rtx_nerf_amd/librtxn.so; else L=rtx_nerf_amd/librtxn_$t.so; fi
    echo "$t: $(RTXN_LIB_PATH=$L python tools/mlp_bench.py --iters 12 2>/dev/null | tail -1)"
  done
done
for t in "$@"; do
  S=rtx_nerf_amd/librtxn_${t}s.so; [ "$t" = cur ] && S=rtx_nerf_amd/librtxn_stamps.so
  if [ -f $S ]; then echo "== $t"; RTXN_LIB_PATH=$S python tools/probe/stamps.py 2>/dev/null; fi
done
