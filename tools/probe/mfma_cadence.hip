// Probe: what one k-step of the 16x16x32 layer pipeline costs in cycles, ingredient by ingredient.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_cadence.hip -o tools/probe/mfma_cadence && tools/probe/mfma_cadence
// A "step" is what mlp_internal.h's PipeStep16 issues: 4 independent v_mfma_f32_16x16x32_f16 sharing one A fragment
// (64 matrix-core cycles), optionally the A fragment through a ring of two ds_read_b128 with a counted lgkmcnt wait,
// optionally two ReLU/convert units (4 VALU).  Every CU runs one block; waves per SIMD = 1 or 2; s_memtime by wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) (void)(x)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 1024 / 4; i += blockDim.x) ((unsigned*)lds)[i] = 0x3c003c00u ^ (i * 2654435761u & 0x03ff03ffu);
  __syncthreads();
  half8 b[4];
  floatx4 acc[4], fin[4];
  for (int c = 0; c < 4; ++c) {
    for (int e = 0; e < 8; ++e) b[c][e] = (_Float16)(0.01f * (lane + c + e));
    for (int e = 0; e < 4; ++e) { acc[c][e] = 0.0f; fin[c][e] = 1.0f + lane + e; }
  }
  half8 ring[2];
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + lane * 16;
  for (int e = 0; e < 8; ++e) ring[0][e] = ring[1][e] = (_Float16)0.5f;
  int r0 = 0, r1 = 0;
  if (V >= 1) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(ring[0]) : "v"(addr));
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(ring[1]) : "v"(addr));
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (V >= 1) __builtin_amdgcn_s_waitcnt(0xC07F | (1 << 8));
      if (V == 4) {          // all four VALU behind the step's last MFMA
        asm volatile(
            "v_mfma_f32_16x16x32_f16 %0, %6, %7, %0\n\tv_mfma_f32_16x16x32_f16 %1, %6, %8, %1\n\t"
            "v_mfma_f32_16x16x32_f16 %2, %6, %9, %2\n\tv_mfma_f32_16x16x32_f16 %3, %6, %10, %3\n\t"
            "v_cvt_pk_f16_f32 %4, %11, %12\n\tv_pk_max_i16 %4, %4, 0\n\tv_cvt_pk_f16_f32 %5, %13, %14\n\tv_pk_max_i16 %5, %5, 0"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(r0), "=&v"(r1)
            : "v"(ring[u & 1]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(fin[0][0]), "v"(fin[0][1]), "v"(fin[1][0]), "v"(fin[1][1]));
      } else if (V == 5) {   // all four VALU behind the step's FIRST MFMA (what hipcc did with builtin MFMAs)
        asm volatile(
            "v_mfma_f32_16x16x32_f16 %0, %6, %7, %0\n\t"
            "v_cvt_pk_f16_f32 %4, %11, %12\n\tv_pk_max_i16 %4, %4, 0\n\tv_cvt_pk_f16_f32 %5, %13, %14\n\tv_pk_max_i16 %5, %5, 0\n\t"
            "v_mfma_f32_16x16x32_f16 %1, %6, %8, %1\n\t"
            "v_mfma_f32_16x16x32_f16 %2, %6, %9, %2\n\tv_mfma_f32_16x16x32_f16 %3, %6, %10, %3"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(r0), "=&v"(r1)
            : "v"(ring[u & 1]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(fin[0][0]), "v"(fin[0][1]), "v"(fin[1][0]), "v"(fin[1][1]));
      } else if (V == 6) {   // interleaved, but four v_mov_b32 instead of the converts (price of ANY VALU in the gap)
        asm volatile(
            "v_mfma_f32_16x16x32_f16 %0, %6, %7, %0\n\tv_mov_b32 %4, %11\n\t"
            "v_mfma_f32_16x16x32_f16 %1, %6, %8, %1\n\tv_mov_b32 %5, %12\n\t"
            "v_mfma_f32_16x16x32_f16 %2, %6, %9, %2\n\tv_mov_b32 %4, %13\n\t"
            "v_mfma_f32_16x16x32_f16 %3, %6, %10, %3\n\tv_mov_b32 %5, %14"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(r0), "=&v"(r1)
            : "v"(ring[u & 1]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(fin[0][0]), "v"(fin[0][1]), "v"(fin[1][0]), "v"(fin[1][1]));
      } else if (V == 7) {   // interleaved, the two converts only (no max)
        asm volatile(
            "v_mfma_f32_16x16x32_f16 %0, %6, %7, %0\n\tv_cvt_pk_f16_f32 %4, %11, %12\n\t"
            "v_mfma_f32_16x16x32_f16 %1, %6, %8, %1\n\t"
            "v_mfma_f32_16x16x32_f16 %2, %6, %9, %2\n\tv_cvt_pk_f16_f32 %5, %13, %14\n\t"
            "v_mfma_f32_16x16x32_f16 %3, %6, %10, %3"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(r0), "=&v"(r1)
            : "v"(ring[u & 1]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(fin[0][0]), "v"(fin[0][1]), "v"(fin[1][0]), "v"(fin[1][1]));
      } else if (V == 8) {   // interleaved, units read the accumulators of the OTHER parity as in the kernel (fin = acc-like registers written by MFMAs)
        asm volatile(
            "v_mfma_f32_16x16x32_f16 %0, %6, %7, %0\n\tv_cvt_pk_f16_f32 %4, %11, %12\n\t"
            "v_mfma_f32_16x16x32_f16 %1, %6, %8, %1\n\tv_pk_max_i16 %4, %4, 0\n\t"
            "v_mfma_f32_16x16x32_f16 %2, %6, %9, %2\n\tv_cvt_pk_f16_f32 %5, %13, %14\n\t"
            "v_mfma_f32_16x16x32_f16 %3, %6, %10, %3\n\tv_pk_max_i16 %5, %5, 0"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(r0), "=&v"(r1)
            : "v"(ring[u & 1]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(fin[u & 3][0]), "v"(fin[u & 3][1]), "v"(fin[(u + 1) & 3][2]), "v"(fin[(u + 1) & 3][3]));
      } else if (V >= 2) {
        asm volatile(
            "v_mfma_f32_16x16x32_f16 %0, %6, %7, %0\n\tv_cvt_pk_f16_f32 %4, %11, %12\n\t"
            "v_mfma_f32_16x16x32_f16 %1, %6, %8, %1\n\tv_pk_max_i16 %4, %4, 0\n\t"
            "v_mfma_f32_16x16x32_f16 %2, %6, %9, %2\n\tv_cvt_pk_f16_f32 %5, %13, %14\n\t"
            "v_mfma_f32_16x16x32_f16 %3, %6, %10, %3\n\tv_pk_max_i16 %5, %5, 0"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(r0), "=&v"(r1)
            : "v"(ring[u & 1]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(fin[0][0]), "v"(fin[0][1]), "v"(fin[1][0]), "v"(fin[1][1]));
      } else {
        asm volatile(
            "v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %6, %1\n\t"
            "v_mfma_f32_16x16x32_f16 %2, %4, %7, %2\n\tv_mfma_f32_16x16x32_f16 %3, %4, %8, %3"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
            : "v"(ring[u & 1]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
      }
      if (V >= 1) {
        if (u & 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[1]) : "v"(addr), "i"(((u + 2) & 31) * 1024));
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[0]) : "v"(addr), "i"(((u + 2) & 31) * 1024));
      }
      if (V >= 3 && u < 1) {   // one LDS-DMA piece per 8 steps (the kernel: 4 per 32)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sink + 4 * threadIdx.x),
                                         (__attribute__((address_space(3))) void*)(lds + 32 * 1024 + (threadIdx.x >> 6) * 1024), 16, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  s += (float)(r0 + r1);
  if (s == 123.456f) sink[threadIdx.x] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (threadIdx.x == blockDim.x - 64) out[256 + blockIdx.x] = t1 - t0;   // the youngest wave: last to finish
}

template <int V>
static void run(const char* name, int threads) {
  unsigned long long* out;
  float* sink;
  CK(hipMalloc(&out, 512 * 8));
  CK(hipMalloc(&sink, 1 << 20));
  CK(hipMemset(sink, 0, 1 << 20));
  const int iters = 2000;
  for (int rep = 0; rep < 3; ++rep) k<V><<<256, threads>>>(out, sink, iters);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int rep = 0; rep < reps; ++rep) k<V><<<256, threads>>>(out, sink, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 256.0 * (threads / 64) * iters * 8.0 * 4.0 * 16384.0 * reps;
  printf("    wall %.3f ms per launch, %.0f TFLOP/s chip-wide (MFMA only)\n", ms / reps, flop / (ms * 1e-3) / 1e12);
  unsigned long long h[512];
  CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  double m = 0, m2 = 0;
  for (int i = 0; i < 256; ++i) { m += (double)h[i]; m2 += (double)h[256 + i]; }
  m /= 256; m2 /= 256;
  const int wps = threads / 256;
  printf("%-44s %d waves/SIMD: oldest wave %.1f, youngest %.1f cycles per step -> matrix core %.0f %% busy, clock %.2f GHz\n", name, wps,
         m / (iters * 8.0), m2 / (iters * 8.0), 100.0 * 64.0 * wps / (m2 / (iters * 8.0)), m2 / (ms / reps * 1e6));
  CK(hipFree(out));
  CK(hipFree(sink));
}

int main() {
  for (int threads : {256, 512}) {
    run<0>("4 MFMA, A in registers", threads);
    run<1>("4 MFMA, A by ds_read_b128 ring of 2", threads);
    run<2>("  + 2 ReLU/convert units, interleaved", threads);
    run<3>("  + one LDS-DMA piece per 8 steps", threads);
    run<4>("ring + 4 VALU behind the LAST MFMA", threads);
    run<5>("ring + 4 VALU behind the FIRST MFMA", threads);
    run<6>("ring + 4 v_mov_b32 interleaved", threads);
    run<7>("ring + the 2 cvt only, interleaved", threads);
    run<8>("ring + units on varying source registers", threads);
  }
  return 0;
}
