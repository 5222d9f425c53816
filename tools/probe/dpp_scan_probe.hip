// Probe: wave-wide inclusive prefix sum with DPP (row_shr 1/2/4/8, row_bcast:15, row_bcast:31) against a serial sum.
// hipcc --offload-arch=gfx950 -O2 tools/probe/dpp_scan_probe.hip -o tools/probe/dpp_scan_probe && tools/probe/dpp_scan_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float dpp_add(float v, float acc_src, int) { return v; }
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_shift(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, true));
}
__device__ __forceinline__ float scan(float v) {
  v += dpp_shift<0x111, 0xf, 0xf>(v);
  v += dpp_shift<0x112, 0xf, 0xf>(v);
  v += dpp_shift<0x114, 0xf, 0xf>(v);
  v += dpp_shift<0x118, 0xf, 0xf>(v);
  v += dpp_shift<0x142, 0xa, 0xf>(v);
  v += dpp_shift<0x143, 0xc, 0xf>(v);
  return v;
}
__global__ void k(const float* in, float* out) { out[threadIdx.x] = scan(in[threadIdx.x]); }
int main() {
  float h[64], r[64], *d, *o;
  for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 11 + 1);
  hipMalloc(&d, 256); hipMalloc(&o, 256);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o);
  hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
  float s = 0; int bad = 0;
  for (int i = 0; i < 64; ++i) { s += h[i]; if (r[i] != s) { ++bad; if (bad < 5) printf("lane %d: %g want %g\n", i, r[i], s); } }
  printf("mismatches: %d\n", bad);
  return bad != 0;
}
