// Probe of ds_read_b64_tr_b16 as the transposing read of the fused training kernel: a [sample][feature] LDS image (row
// stride 136 B) read back as the A/B operand of v_mfma_f32_32x32x16_f16 that contracts over SAMPLES.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
constexpr int STRIDE = 136;   // bytes per sample row: 64 features x 2 B + 8
__global__ void probe(short* out) {   // out[rt][ks][lane][8]
  __shared__ __attribute__((aligned(16))) unsigned char img[64 * STRIDE];
  const int lane = threadIdx.x;
  for (int s = 0; s < 64; ++s) *reinterpret_cast<short*>(img + s * STRIDE + lane * 2) = (short)((s << 6) | lane);   // lane = feature
  __syncthreads();
  const int h = lane >> 5, g16 = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  for (int rt = 0; rt < 2; ++rt)
    for (int ks = 0; ks < 4; ++ks)
      for (int hf = 0; hf < 2; ++hf) {
        const int s = 16 * ks + 8 * h + 4 * hf + q, f = 32 * rt + 16 * (g16 & 1) + 4 * p;
        auto ptr = (__attribute__((address_space(3))) s16x4*)(img + s * STRIDE + f * 2);
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
        for (int e = 0; e < 4; ++e) out[((rt * 4 + ks) * 64 + lane) * 8 + 4 * hf + e] = v[e];
      }
}
int main() {
  short* d;
  hipMalloc(&d, 2 * 4 * 64 * 8 * sizeof(short));
  probe<<<1, 64>>>(d);
  std::vector<short> h(2 * 4 * 64 * 8);
  hipMemcpy(h.data(), d, h.size() * sizeof(short), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int rt = 0; rt < 2; ++rt)
    for (int ks = 0; ks < 4; ++ks)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int r = lane & 31, hh = lane >> 5;
          const short want = (short)(((16 * ks + 8 * hh + j) << 6) | (32 * rt + r));   // M[feature 32rt+r][sample 16ks+8h+j]
          const short got = h[((rt * 4 + ks) * 64 + lane) * 8 + j];
          if (got != want && bad++ < 8) printf("rt %d ks %d lane %d j %d: got s=%d f=%d want s=%d f=%d\n", rt, ks, lane, j, got >> 6, got & 63, want >> 6, want & 63);
        }
  printf("tr probe: %d mismatches\n", bad);
  return bad != 0;
}
