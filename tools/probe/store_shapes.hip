// What does the shape of a store instruction cost on this chip when the bytes are the same?
// The 8x128 training kernels write feature-major tensors [128 rows][Sp samples] of fp16 (rows 2*Sp bytes apart) and sit at
// 3.7-4.2 TB/s whatever is done around the stores (profiles/r03/train_store_paths.txt).  This probe writes exactly that
// tensor -- a block owns 256 samples, a wave 64 of them, 8 "layers" of 128 rows -- from registers, with nothing else in the
// kernel, in four shapes that differ only in bytes per lane (and so in lanes per row and rows per instruction):
//   B = 2   lanes 0-31 one row's 32 samples (64 B), lanes 32-63 another row's      2 rows x  64 B per instruction (the kernels' shape)
//   B = 4   16 lanes per 64-byte run                                                4 rows x  64 B
//   B = 8   16 lanes per row: the wave's 64 samples = one whole 128-byte line       4 rows x 128 B
//   B = 16  8 lanes per row                                                         8 rows x 128 B
// and, for reference, a plain linear fill of the same number of bytes (16 B per lane, 1 KiB contiguous per instruction).
//   hipcc -O3 --offload-arch=gfx950 tools/probe/store_shapes.hip -o tools/probe/store_shapes && tools/probe/store_shapes [samples]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(e)                                                                    \
  do {                                                                              \
    hipError_t r_ = (e);                                                            \
    if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 1; } \
  } while (0)

constexpr int kRows = 128, kLayers = 8;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// MFMAS > 0: each "layer" is preceded by MFMAS dependent-free 32x32x16 MFMAs on registers (64 = the matrix work of one 128-wide
// layer for the wave's 64 samples), i.e. the alternation of a matrix phase and a store burst that the training forward has.
template <int B, int MFMAS = 0>
__global__ __launch_bounds__(256, 2) void shaped(unsigned char* base, long Sp, unsigned seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long s0 = (long)blockIdx.x * 256 + wave * 64;       // the wave's 64 samples = 128 bytes of every row
  constexpr int lanes_per_row = 128 / B > 32 ? 32 : 128 / B; // B = 2: 32 lanes (half a line), B = 4: 16 (half a line), 8: 16, 16: 8
  constexpr int bytes_per_row_instr = lanes_per_row * B;     // 64, 64, 128, 128
  constexpr int rows_per_instr = 64 / lanes_per_row;
  constexpr int instr_per_row = 128 / bytes_per_row_instr;   // 2, 2, 1, 1
  const int r_in = lane / lanes_per_row, l_in = lane % lanes_per_row;
  unsigned v0 = seed ^ (unsigned)threadIdx.x, v1 = v0 * 2654435761u, v2 = v1 ^ 0x9e3779b9u, v3 = v2 * 40503u;
  floatx16 acc[4];
  half8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(float)(lane + i); fb[i] = (_Float16)(float)(seed & 3); }
  for (int t = 0; t < 4; ++t)
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
  for (int layer = 0; layer < kLayers; ++layer) {
    if constexpr (MFMAS > 0) {
#pragma unroll 4
      for (int m = 0; m < MFMAS; m += 4)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[t], 0, 0, 0);
      v0 ^= (unsigned)acc[0][0] ^ (unsigned)acc[1][1] ^ (unsigned)acc[2][2] ^ (unsigned)acc[3][3];
    }
    unsigned char* lbase = base + (long)layer * kRows * Sp * 2;
#pragma unroll 4
    for (int r = 0; r < kRows; r += rows_per_instr) {
#pragma unroll
      for (int part = 0; part < instr_per_row; ++part) {
        unsigned char* p = lbase + (long)(r + r_in) * Sp * 2 + s0 * 2 + part * bytes_per_row_instr + l_in * B;
        if constexpr (B == 2) *reinterpret_cast<unsigned short*>(p) = (unsigned short)v0;
        else if constexpr (B == 4) *reinterpret_cast<unsigned*>(p) = v0;
        else if constexpr (B == 8) *reinterpret_cast<uint2*>(p) = make_uint2(v0, v1);
        else *reinterpret_cast<uint4*>(p) = make_uint4(v0, v1, v2, v3);
        v0 += 0x01010101u;
      }
    }
  }
}

// B bytes per lane, LPR lanes per row (a run of LPR * B bytes per row, 64 / LPR rows per instruction): the shapes a
// transposing LDS read (ds_read_b64_tr_b16: 16 columns per 16-lane group) can feed -- 4 lanes per row
template <int B, int LPR, int MFMAS>
__global__ __launch_bounds__(256, 2) void runs(unsigned char* base, long Sp, unsigned seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long s0 = (long)blockIdx.x * 256 + wave * 64;
  constexpr int run = LPR * B, rows_per_instr = 64 / LPR, parts = 128 / run;
  const int r_in = lane % rows_per_instr, l_in = lane / rows_per_instr;     // lane i <-> row i of the group, as the tr read delivers
  unsigned v0 = seed ^ (unsigned)threadIdx.x, v1 = v0 * 2654435761u, v2 = v1 ^ 0x9e3779b9u, v3 = v2 * 40503u;
  floatx16 acc[4];
  half8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(float)(lane + i); fb[i] = (_Float16)(float)(seed & 3); }
  for (int t = 0; t < 4; ++t)
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
  for (int layer = 0; layer < kLayers; ++layer) {
    if constexpr (MFMAS > 0) {
#pragma unroll 4
      for (int m = 0; m < MFMAS; m += 4)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[t], 0, 0, 0);
      v0 ^= (unsigned)acc[0][0] ^ (unsigned)acc[1][1] ^ (unsigned)acc[2][2] ^ (unsigned)acc[3][3];
    }
    unsigned char* lbase = base + (long)layer * kRows * Sp * 2;
#pragma unroll 2
    for (int r = 0; r < kRows; r += rows_per_instr)
#pragma unroll
      for (int part = 0; part < parts; ++part) {
        unsigned char* p = lbase + (long)(r + r_in) * Sp * 2 + s0 * 2 + part * run + l_in * B;
        if constexpr (B == 4) *reinterpret_cast<unsigned*>(p) = v0;
        else if constexpr (B == 8) *reinterpret_cast<uint2*>(p) = make_uint2(v0, v1);
        else *reinterpret_cast<uint4*>(p) = make_uint4(v0, v1, v2, v3);
        v0 += 0x01010101u;
      }
  }
}

__global__ __launch_bounds__(256) void linear_fill(uint4* p, long n16, unsigned seed) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) p[i] = make_uint4(seed, seed + 1, seed + 2, (unsigned)i);
}

int main(int argc, char** argv) {
  const long S = argc > 1 ? atol(argv[1]) : 4700000;
  const long Sp = (S + 255) / 256 * 256;
  const size_t bytes = (size_t)kLayers * kRows * Sp * 2;
  unsigned char* buf;
  CHECK(hipMalloc(&buf, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)(Sp / 256);
  auto time_it = [&](const char* name, auto launch) -> int {
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    printf("%-44s %7.3f ms  %6.2f TB/s\n", name, ms, bytes / ms / 1e9);
    return 0;
  };
  printf("%ld samples x %d rows x %d layers of fp16 = %.2f GB per launch\n", Sp, kRows, kLayers, bytes / 1e9);
  if (time_it("2 B/lane   (2 rows x  64 B per instruction)", [&] { shaped<2><<<grid, 256>>>(buf, Sp, 1u); })) return 1;
  if (time_it("4 B/lane   (4 rows x  64 B per instruction)", [&] { shaped<4><<<grid, 256>>>(buf, Sp, 2u); })) return 1;
  if (time_it("8 B/lane   (4 rows x 128 B per instruction)", [&] { shaped<8><<<grid, 256>>>(buf, Sp, 3u); })) return 1;
  if (time_it("16 B/lane  (8 rows x 128 B per instruction)", [&] { shaped<16><<<grid, 256>>>(buf, Sp, 4u); })) return 1;
  if (time_it("64 MFMAs per layer, no stores to speak of   ", [&] { shaped<16, 64 * 16><<<grid / 16, 256>>>(buf, Sp, 6u); })) return 1;
  printf("  (the line above: 16 x the matrix work on 1/16 of the blocks -- divide its time by 1 to get the matrix phase of a full launch)\n");
  if (time_it("2 B/lane  + 64 MFMAs before every layer     ", [&] { shaped<2, 64><<<grid, 256>>>(buf, Sp, 7u); })) return 1;
  if (time_it("4 B/lane  + 64 MFMAs before every layer     ", [&] { shaped<4, 64><<<grid, 256>>>(buf, Sp, 8u); })) return 1;
  if (time_it("8 B/lane  + 64 MFMAs before every layer     ", [&] { shaped<8, 64><<<grid, 256>>>(buf, Sp, 9u); })) return 1;
  if (time_it("16 B/lane + 64 MFMAs before every layer     ", [&] { shaped<16, 64><<<grid, 256>>>(buf, Sp, 10u); })) return 1;
  if (time_it("4 B/lane, 4 rows x 64 B, lanes strided + MFMAs", [&] { runs<4, 16, 64><<<grid, 256>>>(buf, Sp, 14u); })) return 1;
  if (time_it("8 B/lane, 8 rows x 64 B, lanes strided + MFMAs", [&] { runs<8, 8, 64><<<grid, 256>>>(buf, Sp, 15u); })) return 1;
  if (time_it("8 B/lane, 4 rows x 128 B, lanes strided + MFMAs", [&] { runs<8, 16, 64><<<grid, 256>>>(buf, Sp, 16u); })) return 1;
  if (time_it("8 B/lane, 16 rows x 32 B + 64 MFMAs         ", [&] { runs<8, 4, 64><<<grid, 256>>>(buf, Sp, 11u); })) return 1;
  if (time_it("16 B/lane, 16 rows x 64 B + 64 MFMAs        ", [&] { runs<16, 4, 64><<<grid, 256>>>(buf, Sp, 12u); })) return 1;
  if (time_it("16 B/lane, 8 rows x 128 B + 64 MFMAs (runs) ", [&] { runs<16, 8, 64><<<grid, 256>>>(buf, Sp, 13u); })) return 1;
  if (time_it("linear fill, 16 B/lane (1 KiB per instruction)", [&] { linear_fill<<<256 * 8, 256>>>(reinterpret_cast<uint4*>(buf), (long)(bytes / 16), 5u); })) return 1;
  CHECK(hipFree(buf));
  return 0;
}
