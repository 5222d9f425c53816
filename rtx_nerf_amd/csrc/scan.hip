// CSR compaction: indices = exclusive_scan(num_hits), total = sum(num_hits).
// Replaces thrust::reduce + thrust::exclusive_scan (reference main.cu:631-637)
// with two launches and no host round trip: `total` stays in device memory and
// is read by the consumers (rtxn_mlp_forward_segments) directly.
//
// HBM-bound, 8 B/ray algorithmic (4 read + 4 written).  Each 256-thread block
// owns a 1024-int chunk (int4 per thread, coalesced 16-B accesses).  256 threads,
// not 1024: a block is then one wave per SIMD with ~20 VGPRs and fits on a CU
// beside a resident block of the MLP kernel, so the next frame's compaction runs
// underneath this frame's MLP (render.py, render_async); a 1024-thread block
// (4 waves per SIMD) did not fit and stalled the traversal stream until the MLP
// kernel ended.
#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kItems = 4;
constexpr int kChunk = kBlock * kItems;

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

__device__ __forceinline__ int4 load4(const int* p, long base, int n) {
  int4 v = make_int4(0, 0, 0, 0);
  if (base + 3 < n) {
    v = *reinterpret_cast<const int4*>(p + base);
  } else {
    if (base < n) v.x = p[base];
    if (base + 1 < n) v.y = p[base + 1];
    if (base + 2 < n) v.z = p[base + 2];
  }
  return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix, *total = block sum
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = wave_incl_scan(v, lane);
  if (lane == 63) lds[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    int w = lane < (kBlock / 64) ? lds[lane] : 0;
    int wi = wave_incl_scan(w, lane);
    if (lane < (kBlock / 64)) lds[lane] = wi - w;
    if (lane == (kBlock / 64) - 1) lds[kBlock / 64] = wi;
  }
  __syncthreads();
  int out = lds[wave] + incl - v;
  *total = lds[kBlock / 64];
  __syncthreads();
  return out;
}

__global__ __launch_bounds__(kBlock) void scan_partials(const int* __restrict__ in, int* __restrict__ partials, int n) {
  __shared__ int lds[kBlock / 64 + 1];
  long base = (long)blockIdx.x * kChunk + threadIdx.x * kItems;
  int4 v = load4(in, base, n);
  int s = v.x + v.y + v.z + v.w;
  int tot;
  (void)block_excl_scan(s, &tot, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kBlock) void scan_apply(const int* __restrict__ in, const int* __restrict__ partials,
                                                      int* __restrict__ out, int* __restrict__ total, int n,
                                                      int n_blocks) {
  __shared__ int lds[kBlock / 64 + 1];
  // offset of this chunk = sum of the partials before it (every block recomputes it; <= a few K ints, L2-resident)
  int acc = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += kBlock) acc += partials[i];
  int offset;
  (void)block_excl_scan(acc, &offset, lds);
  long base = (long)blockIdx.x * kChunk + threadIdx.x * kItems;
  int4 v = load4(in, base, n);
  int s = v.x + v.y + v.z + v.w;
  int tot;
  int ex = block_excl_scan(s, &tot, lds) + offset;
  int4 o;
  o.x = ex;
  o.y = ex + v.x;
  o.z = o.y + v.y;
  o.w = o.z + v.z;
  if (base + 3 < n) {
    *reinterpret_cast<int4*>(out + base) = o;
  } else {
    if (base < n) out[base] = o.x;
    if (base + 1 < n) out[base + 1] = o.y;
    if (base + 2 < n) out[base + 2] = o.z;
  }
  if (blockIdx.x == n_blocks - 1 && threadIdx.x == 0) *total = offset + tot;
}

__global__ void scan_empty(int* total) { *total = 0; }

}  // namespace

extern "C" size_t rtxn_scan_workspace_bytes(int n) {
  if (n < 0) n = 0;
  size_t blocks = ((size_t)n + kChunk - 1) / kChunk;
  return (blocks + 1) * sizeof(int);
}

extern "C" int rtxn_scan_hits(const int* num_hits, int* indices, int* total, int n, void* workspace,
                              size_t workspace_bytes, rtxn_stream_t stream) {
  RTXN_REQUIRE(n >= 0, "rtxn_scan_hits: n = %d < 0", n);
  RTXN_REQUIRE(total != nullptr, "rtxn_scan_hits: total is NULL");
  RTXN_DEVICE_OR_FAIL();
  hipStream_t s = rtxn::as_stream(stream);
  if (n == 0) {
    scan_empty<<<1, 1, 0, s>>>(total);
    RTXN_LAUNCH_CHECK("scan_empty");
    return RTXN_OK;
  }
  RTXN_REQUIRE(num_hits && indices, "rtxn_scan_hits: NULL num_hits/indices");
  RTXN_REQUIRE(((uintptr_t)num_hits & 15) == 0 && ((uintptr_t)indices & 15) == 0,
               "rtxn_scan_hits: num_hits/indices must be 16-byte aligned");
  RTXN_REQUIRE(workspace && workspace_bytes >= rtxn_scan_workspace_bytes(n),
               "rtxn_scan_hits: workspace too small (%zu < %zu)", workspace_bytes, rtxn_scan_workspace_bytes(n));
  int blocks = (n + kChunk - 1) / kChunk;
  int* partials = static_cast<int*>(workspace);
  scan_partials<<<blocks, kBlock, 0, s>>>(num_hits, partials, n);
  RTXN_LAUNCH_CHECK("scan_partials");
  scan_apply<<<blocks, kBlock, 0, s>>>(num_hits, partials, indices, total, n, blocks);
  RTXN_LAUNCH_CHECK("scan_apply");
  return RTXN_OK;
}
