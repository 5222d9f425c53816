// Training path: encoders (Composite-Frequency, multiresolution hash grid), MLP
// forward with saved activations, MLP backward (dgrad fused across layers + per-layer
// wgrad), L2 loss and Adam.  Replaces the tiny-cuda-nn calls of the reference's training
// loop: network->forward(..., prepare_input_gradients) (main.cu:721), loss->evaluate
// (:759), network->backward (:781), optimizer->step (:787).  tiny-cuda-nn is un-vendored
// and unpinned; numerics follow oracle/rtxn_oracle.c (orc_mlpe_*, orc_hg_*, orc_l2_loss,
// orc_adam_step).  PARITY UNPINNED.
//
// Layout.  Every per-sample training tensor is FEATURE-MAJOR fp16: X[feature][S_pad],
// S_pad = S rounded up to 256 (one block tile), padding columns written as zeros.  That is
// the layout both consumers want: the fused forward/backward kernels read/write one sample
// per lane (consecutive lanes -> consecutive addresses), and the weight-gradient GEMM
// dW[o][i] = sum_s dZ[o][s] X[i][s] contracts over samples, so each MFMA operand fragment
// (8 consecutive samples of one feature row) is ONE 16-byte load, with no transpose.
//
// Kernels
//   encode_freq_kernel / hashgrid_encode_kernel   [S][5] f32 -> encT[E][S_pad] f16
//   mlp_train_fwd_kernel<W>   encT -> acts[L][W][S_pad] (post-ReLU), out[S][16] (+radiance)
//   mlp_bwd_kernel<W>         dout[S][4] -> dz[L][W][S_pad], dzL[16][S_pad], dencT (optional);
//                             dA = W^T dZ on MFMA with the transposed packed weights, activations
//                             of the backward chain stay in registers exactly as in the forward
//   wgrad_kernel              dW += dZ X^T : 64x64 output super-tile per wave, K = samples,
//                             fp32 atomics into the tcnn-layout gradient buffer
//   hashgrid_backward_kernel  scatter-add of denc into the table gradient (fp32 atomics)
//   l2_loss_kernel, adam_kernel
#include "mlp_internal.h"
#include "hashgrid_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>


namespace {

using rtxn::layer_mma;
using rtxn::out_mma;
using rtxn::pack8;
using rtxn::perm_feature;
using rtxn::stage_rt;
using rtxn::HgLevels;
using rtxn::hg_index;
using rtxn::hg_index_nodiv;
using rtxn::levels_of;
using rtxn::sin_turns;

constexpr int kThreads = 256;
constexpr int kTile = 256;

inline long padded(long S) { return (S + kTile - 1) / kTile * kTile; }

// A training step without a host round trip (rtxn_train_gradients): the kernels take the batch's segment count from the
// scan's device-side total, clamped to what the buffers hold; grids are sized for the capacity and the blocks past the live
// samples leave at once.  total_segments == NULL: the host's count as passed (every per-stage entry point).  The row stride
// S_pad of the feature-major tensors is then that of the CAPACITY, so it does not change from step to step.
struct DevCount {
  const int* total_segments;
  int capacity;
};
__device__ __forceinline__ long live_samples(const DevCount& dc, long S_host) {
  if (!dc.total_segments) return S_host;
  const int t = *dc.total_segments;
  return 32L * (t < dc.capacity ? (t < 0 ? 0 : t) : dc.capacity);
}
__device__ __forceinline__ long padded_dev(long S) { return (S + kTile - 1) / kTile * kTile; }
// the fused training kernels address X[feature][S_pad] with one 32-bit per-lane byte offset (row_elem): (s + 4*S_pad)*2 < 2^32
constexpr long kMaxTrainSamples = (1L << 32) / 10 - kTile;

// Deterministic mode (rtxn_set_deterministic_workspace).  Every gradient sum that crosses workgroups -- the weight-gradient
// kernels' flush of their on-chip accumulators, the hash-grid scatter -- is a float atomic by default, and float addition does not
// associate: two runs of the same step differ in the last bit (fp32) or the last few (packed fp16), which Adam's 1/sqrt(v) turns
// into +-lr wherever a gradient is noise around zero.  With a workspace set, those sums go into a 64-bit FIXED-POINT shadow of the
// gradient buffer instead (value x 2^40, integer atomics: associative, so any arrival order gives the same bits) and are folded
// back -- one rounding per element -- by det_fold_kernel behind the kernels that fed them.  What a workgroup sums on its own
// (registers, LDS, the wave-level run sums of the scatter) has a fixed order already.  q is parallel to the float buffer `base`.
struct DetCtx {
  const float* base;
  long long* q;
};
constexpr float kDetScale = 1099511627776.0f;            // 2^40: 9e-13 resolution, 8e6 range
__device__ __forceinline__ void grad_add(float* addr, float v, const DetCtx& det) {
  if (det.q) atomicAdd(reinterpret_cast<unsigned long long*>(det.q + (addr - det.base)), (unsigned long long)__float2ll_rn(v * kDetScale));
  else atomicAdd(addr, v);
}

// ------------------------------------------------------------------------- encoders
// Where a kernel's samples come from: a materialised float[S][5] batch (the sampler's output, sampler/sampler.cu), or the
// packed segments themselves -- sample (segment g, i) is then formed here exactly as sample_kernel forms it (REGULAR: t = i/32;
// MIDPOINT_WORLD: t = (i + 0.5)/32; position = fma(t, end - start, start); (theta, phi) = the segment's), so the 20-byte
// samples never exist in memory: launchSampler folded into its consumers.
struct SampleSrc {
  const float* in;        // [S][5], or NULL: segments
  const float* start;     // [P][3]
  const float* end;       // [P][3]
  const float* seg_view;  // [P][2]
  int midpoint;           // RTXN_SAMPLING_MIDPOINT_WORLD (1) / RTXN_SAMPLING_REGULAR (0)
};
__device__ __forceinline__ void sample_pos(const SampleSrc& src, long s, bool ok, float (&x)[3]) {
  if (src.in) {
#pragma unroll
    for (int a = 0; a < 3; ++a) x[a] = ok ? src.in[5 * s + a] : 0.0f;
  } else {
    const long g = ok ? (s >> 5) * 3 : 0;
    const float t = ((float)(int)(s & 31) + (src.midpoint ? 0.5f : 0.0f)) * (1.0f / 32);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float og = src.start[g + a];
      x[a] = ok ? fmaf(t, src.end[g + a] - og, og) : 0.0f;
    }
  }
}
// The same for an index the caller has made valid, WITHOUT a branch between the two sources: the addresses are selected, not the
// loaded values (a merge of loaded values is a register copy, and hipcc waits for the loads in front of it); for the explicit
// samples both reads hit the same address and the interpolation weight is 0.  Every lane loads; nothing waits between the loads.
__device__ __forceinline__ void sample_pos_unmasked(const SampleSrc& src, long s, float (&x)[3]) {
  const long g = (s >> 5) * 3;
  const float* p0 = src.in ? src.in + 5 * s : src.start + g;
  const float* p1 = src.in ? p0 : src.end + g;
  const float t = src.in ? 0.0f : ((float)(int)(s & 31) + (src.midpoint ? 0.5f : 0.0f)) * (1.0f / 32);
  float og[3], en[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) og[a] = p0[a], en[a] = p1[a];
#pragma unroll
  for (int a = 0; a < 3; ++a) x[a] = fmaf(t, en[a] - og[a], og[a]);
}
__device__ __forceinline__ void sample_dir(const SampleSrc& src, long s, bool ok, float (&v)[2]) {
  if (src.in) {
    v[0] = ok ? src.in[5 * s + 3] : 0.0f;
    v[1] = ok ? src.in[5 * s + 4] : 0.0f;
  } else {
    const long g = ok ? (s >> 5) * 2 : 0;
    v[0] = ok ? src.seg_view[g] : 0.0f;
    v[1] = ok ? src.seg_view[g + 1] : 0.0f;
  }
}
__device__ __forceinline__ void sample_dir_unmasked(const SampleSrc& src, long s, float (&v)[2]) {
  const float* p = src.in ? src.in + 5 * s + 3 : src.seg_view + (s >> 5) * 2;
  v[0] = p[0];
  v[1] = p[1];
}

// Feature f of the reference's Composite-Frequency(3 dims x 10, 2 dims x 12) encoding in encode_freq_kernel's order
// (dimension-major, frequency, (sin, cos)), padded to 112 with ones: main.cu:35-69.
struct FreqFeat { int dim, freq, ph, pad; };
__host__ __device__ constexpr FreqFeat freq_feat_3_10_2_12(int f) {
  return f < 60 ? FreqFeat{f / 20, (f % 20) / 2, f & 1, 0} : f < 108 ? FreqFeat{3 + (f - 60) / 24, ((f - 60) % 24) / 2, f & 1, 0} : FreqFeat{0, 0, 0, 1};
}
// The encoded sample AS the training kernels' layer-0 B fragments: lane-half h of k-step kk holds the features
// perm_feature(kk, h, j) = b + 4 h, b a compile-time number -- so per value one select between two dimensions (where b and b + 4
// straddle one), one between two frequencies, and encode_freq_kernel's own arithmetic (sin_turns, one rounding to fp16): the
// fragments are bit for bit what the forward reads back out of encT.  x: the sample's (x, y, z, theta, phi); !ok: zeros.
template <int KS0>
__device__ __forceinline__ void encode_freq_fragments_3_10_2_12(const float (&x)[5], int h, bool ok, half8 (&dst)[KS0]) {
  static_assert(KS0 == 7, "112 features");
#pragma unroll
  for (int kk = 0; kk < KS0; ++kk) {
    half8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int b = 16 * kk + 8 * (j >> 2) + (j & 3);
      const FreqFeat f0 = freq_feat_3_10_2_12(b), f1 = freq_feat_3_10_2_12(b + 4);
      const float xv = (f1.pad || f0.dim == f1.dim) ? x[f0.dim] : (h ? x[f1.dim] : x[f0.dim]);
      const int e = f1.pad ? f0.freq : (h ? f1.freq : f0.freq);
#ifdef RTXN_ENC_FAKE      // timing-only (results wrong): what the folded encoder's sines and range reductions cost end to end
      float y = xv * (float)(e + 1) * 0.03125f + 0.25f * (float)f0.ph;
#else
      float y = sin_turns(xv, e, f0.ph);                 // b and b + 4 have the same parity: the same phase
#endif
      if (f1.pad) y = h ? 1.0f : y;
      v[j] = ok ? (_Float16)y : (_Float16)0.0f;
    }
    dst[kk] = v;
  }
}

// the sampler's t_vals of sample s (segments only): REGULAR (i + 1)/32; MIDPOINT_WORLD |end - start|/32, times t_scale
__device__ __forceinline__ float sample_tval(const SampleSrc& src, long s, float t_scale) {
  if (!src.midpoint) return (float)((int)(s & 31) + 1) * (1.0f / 32);
  const long g = (s >> 5) * 3;
  const float dx = src.end[g] - src.start[g], dy = src.end[g + 1] - src.start[g + 1], dz = src.end[g + 2] - src.start[g + 2];
  const float tv = sqrtf(fmaf(dz, dz, fmaf(dx, dx, dy * dy))) * (1.0f / 32);
  return t_scale == 1.0f ? tv : tv * t_scale;
}

__global__ __launch_bounds__(kThreads) void encode_freq_kernel(SampleSrc src, _Float16* __restrict__ encT, float* __restrict__ t_vals,
                                                               float t_scale, long S, long Sp, int PD, int PF, int DD, int DF, int E,
                                                               DevCount dc) {
  const long s = (long)blockIdx.x * kThreads + threadIdx.x;
  S = live_samples(dc, S);
  if (s >= padded_dev(S)) return;
  const bool ok = s < S;
  float x[8];
  if (src.in) {
    for (int c = 0; c < PD + DD; ++c) x[c] = ok ? src.in[(PD + DD) * s + c] : 0.0f;
  } else {   // segments: 3 + 2 dimensions
    float p3[3], v2[2];
    sample_pos(src, s, ok, p3);
    sample_dir(src, s, ok, v2);
    x[0] = p3[0]; x[1] = p3[1]; x[2] = p3[2]; x[3] = v2[0]; x[4] = v2[1];
    if (t_vals && ok) t_vals[s] = sample_tval(src, s, t_scale);
  }
  int j = 0;
  for (int d = 0; d < PD; ++d)
    for (int f = 0; f < PF; ++f)
      for (int ph = 0; ph < 2; ++ph, ++j) encT[(long)j * Sp + s] = ok ? (_Float16)sin_turns(x[d], f, ph) : (_Float16)0.0f;
  for (int d = 0; d < DD; ++d)
    for (int f = 0; f < DF; ++f)
      for (int ph = 0; ph < 2; ++ph, ++j) encT[(long)j * Sp + s] = ok ? (_Float16)sin_turns(x[PD + d], f, ph) : (_Float16)0.0f;
  for (; j < E; ++j) encT[(long)j * Sp + s] = ok ? (_Float16)1.0f : (_Float16)0.0f;
}

// grid.y = level (0..L-1: hash levels; L: direction frequencies + padding)
__global__ __launch_bounds__(kThreads) void hashgrid_encode_kernel(HgLevels lv, int n_dir_freqs, const _Float16* __restrict__ table,
                                                                   SampleSrc src, _Float16* __restrict__ encT, float* __restrict__ t_vals,
                                                                   float t_scale, long S, long Sp, int E, DevCount dc) {
  const long s = (long)blockIdx.x * kThreads + threadIdx.x;
  S = live_samples(dc, S);
  if (s >= padded_dev(S)) return;
  const bool ok = s < S;
  const int l = blockIdx.y;
  const int F = lv.n_features;
  if (l < lv.n_levels) {
    float fr[3], x3[3];
    unsigned g[3];
    sample_pos(src, s, ok, x3);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float x01 = fmaf(x3[a], 0.5f, 0.5f);
      const float p = fmaf(x01, lv.scale[l], 0.5f), fl = floorf(p);
      g[a] = (unsigned)(int)fl;
      fr[a] = p - fl;
    }
    for (int f = 0; f < F; ++f) {
      float acc = 0.0f;
#pragma unroll
      for (int corner = 0; corner < 8; ++corner) {
        float w = 1.0f;
        unsigned p[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const int hi = (corner >> a) & 1;
          w *= hi ? fr[a] : 1.0f - fr[a];
          p[a] = g[a] + (unsigned)hi;
        }
        const unsigned idx = hg_index(p[0], p[1], p[2], lv.res[l], lv.size[l]);
        acc = fmaf(w, (float)table[((size_t)lv.offset[l] + idx) * F + f], acc);
      }
      encT[(long)(l * F + f) * Sp + s] = ok ? (_Float16)acc : (_Float16)0.0f;
    }
  } else {
    int j = lv.n_levels * F;
    float v2[2];
    sample_dir(src, s, ok, v2);
    if (!src.in && t_vals && ok) t_vals[s] = sample_tval(src, s, t_scale);
    for (int d = 0; d < 2; ++d) {
      const float x = v2[d];
      for (int f = 0; f < n_dir_freqs; ++f)
        for (int ph = 0; ph < 2; ++ph, ++j) encT[(long)j * Sp + s] = ok ? (_Float16)sin_turns(x, f, ph) : (_Float16)0.0f;
    }
    for (; j < E; ++j) encT[(long)j * Sp + s] = ok ? (_Float16)1.0f : (_Float16)0.0f;
  }
}

// F == 2 (tcnn's default and BASELINE configs[2]): ONE thread per sample walks all levels.  The per-(sample, level) launch
// above spent most of its time in vector ALU work that does not depend on the level -- forming the sample from its segment,
// 64-bit addresses, an integer division per corner index (rocprofv3: 259 VALU instructions per wave and level) -- so here
// the position is formed once, level constants are scalar, indices need no division (hg_index_nodiv), the loads take a
// 32-bit offset from a scalar level base, and (x y) is shared by the two z corners (same products, same rounding).  The
// gathers: both features of a corner are one 4-byte entry, and the two corners that differ in x share an aligned 8-byte pair
// on the hashed levels whenever x is even -- the hash is x ^ (y p1) ^ (z p2) modulo a power of two, so x + 1 = x ^ 1 flips
// only the lowest index bit -- and then one 8-byte load serves both; odd x (and the densely stored levels) take a second
// 4-byte load.  Bit-identical to the general kernel (tests/test_gpu_train.py).
__global__ __launch_bounds__(kThreads) void hashgrid_encode_f2_kernel(HgLevels lv, int n_dir_freqs, const _Float16* __restrict__ table,
                                                                      SampleSrc src, _Float16* __restrict__ encT, float* __restrict__ t_vals,
                                                                      float t_scale, long S, long Sp, int E, DevCount dc) {
  const long s = (long)blockIdx.x * kThreads + threadIdx.x;
  S = live_samples(dc, S);
  if (s >= padded_dev(S)) return;
  const bool ok = s < S;
  float x3[3], x01[3];
  sample_pos(src, s, ok, x3);
#pragma unroll
  for (int a = 0; a < 3; ++a) x01[a] = fmaf(x3[a], 0.5f, 0.5f);
  _Float16* out = encT + s;
  for (int l = 0; l < lv.n_levels; ++l) {
    const unsigned res = lv.res[l], size = lv.size[l];
    const float scale = lv.scale[l];
    const bool hashed = (unsigned long long)res * res * res > size;
    const uint8_t* base = reinterpret_cast<const uint8_t*>(table) + (size_t)lv.offset[l] * 4;     // scalar
    float fr[3];
    unsigned g[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float p = fmaf(x01[a], scale, 0.5f), fl = floorf(p);
      g[a] = (unsigned)(int)fl;
      fr[a] = p - fl;
    }
    const bool shared = hashed && (g[0] & 1u) == 0;
    // All of a level's gathers are issued before the first one is used: written corner by corner (load, select, load) hipcc
    // waited for every load where it stood -- eight serial L2 round trips per level and thread (107 -> 86 us on the configs[2]
    // batch, bit-identical).  Keeping the NEXT level's gathers in flight as well (software-pipelined levels, counted waits)
    // measured no better (91 us): past this point the kernel is bound by the gather rate, not by the round trips.
    unsigned e[8], i_lo[4], i_hi[4];
    uint2 pr[4];
    rtxn::hg_corner_indices(g, res, size, hashed, i_lo, i_hi);
#pragma unroll
    for (int yz = 0; yz < 4; ++yz) pr[yz] = *reinterpret_cast<const uint2*>(base + ((i_lo[yz] & ~1u) << 2));
    unsigned hi[4] = {0u, 0u, 0u, 0u};
    if (!shared) {
#pragma unroll
      for (int yz = 0; yz < 4; ++yz) hi[yz] = *reinterpret_cast<const unsigned*>(base + (i_hi[yz] << 2));
    }
#pragma unroll
    for (int yz = 0; yz < 4; ++yz) {
      e[2 * yz] = (i_lo[yz] & 1u) ? pr[yz].y : pr[yz].x;
      e[2 * yz + 1] = shared ? ((i_lo[yz] & 1u) ? pr[yz].x : pr[yz].y) : hi[yz];
    }
    float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {        // weights ((x) y) z and corner order of the general kernel
      float w = 1.0f;
#pragma unroll
      for (int a = 0; a < 3; ++a) w *= ((corner >> a) & 1) ? fr[a] : 1.0f - fr[a];
      const half2v v = __builtin_bit_cast(half2v, e[corner]);
      acc0 = fmaf(w, (float)v[0], acc0);
      acc1 = fmaf(w, (float)v[1], acc1);
    }
    // rounded to fp32 and THEN to fp16, as the oracle does: without the barrier hipcc folds the last multiply-add and the
    // conversion into v_fma_mixlo_f16 -- one rounding, 2 of 96,000 values an fp16 ulp apart
    asm volatile("" : "+v"(acc0), "+v"(acc1));
    out[(long)(2 * l) * Sp] = ok ? (_Float16)acc0 : (_Float16)0.0f;
    out[(long)(2 * l + 1) * Sp] = ok ? (_Float16)acc1 : (_Float16)0.0f;
  }
  int j = lv.n_levels * 2;
  float v2[2];
  sample_dir(src, s, ok, v2);
  if (!src.in && t_vals && ok) t_vals[s] = sample_tval(src, s, t_scale);
  for (int d = 0; d < 2; ++d) {
    const float x = v2[d];
    for (int f = 0; f < n_dir_freqs; ++f)
      for (int ph = 0; ph < 2; ++ph, ++j) out[(long)j * Sp] = ok ? (_Float16)sin_turns(x, f, ph) : (_Float16)0.0f;
  }
  for (; j < E; ++j) out[(long)j * Sp] = ok ? (_Float16)1.0f : (_Float16)0.0f;
}

// Scatter-add of dL/d(encoding) into the table gradient: one atomic per (sample, corner) and feature (fp32), or per
// (sample, corner) for both features at once (PK, below).  What bounds it is the L2's atomic throughput on SCATTERED addresses
// (MI355X_MICROARCH.md, Global float atomics: lanes in different rows run an order of magnitude below the streaming rate), so
// the lever is the number of atomics that leave the wave.  Consecutive lanes are consecutive samples of a segment: on every
// level but the finest, runs of lanes share a grid cell and with it all eight corner addresses (whole waves on the coarse
// levels).  A segmented inclusive scan keyed on the cell leaves each run's sum in its last lane, which alone issues the
// atomic; the finest levels, where no two lanes share a cell, skip the scans on a wave-uniform branch.
// (Round 1 first reduced the coarsest levels in an LDS copy of the level, flushed once per block.  Measured inside the
// config-3 step against sending those levels through the run aggregation as well: LDS copy 0.37-0.47 ms for the whole scatter,
// none 0.28 ms -- the 64/128-KiB copy leaves one block per CU and its zero/flush passes cost more than the few atomics the
// aggregation leaves.  The LDS path is gone.)
//
// PK (hashed levels, two features per entry): both features of a corner go out as ONE global_atomic_pk_add_f16 into an fp16
// gradient table -- half the atomics -- which is also what tiny-cuda-nn does (its grid gradient is __half2).  Entries of the
// hashed levels receive ~10 contributions each, so the fp16 accumulation costs ~1e-3 relative; the densely stored coarse
// levels, which receive thousands per entry, stay fp32.
// Wave-level run sums without the LDS crossbar.  A run = consecutive lanes whose samples sit in the same grid cell; all of a
// sample's 8 x F corner contributions share the run structure, so the per-step "may I add my neighbour" decisions are
// worked out once as 0 / 1 multipliers and every value then takes seven DPP multiply-adds: a segmented Hillis-Steele scan
// inside each row of 16 lanes (row_shr 1, 2, 4, 8; lanes before the row read 0) and a carry handed from row to row through
// lane 15 (row_bcast:15 into rows 1, 2, 3 in turn).  The run's LAST lane ends up with the run's sum.  The __shfl_up form
// this replaces cost 12 dependent ds_bpermute per corner: with the atomics switched off the kernel took the same time, i.e.
// it was bound by exactly those shuffles (tools/probe: 222.7 vs 219.7 us on the configs[2] batch).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true);
}
struct RunSteps { float m1, m2, m4, m8, c1, c2, c3; };
__device__ __forceinline__ RunSteps run_steps(int lane, int run_start) {
  const int row = lane >> 4;
  RunSteps k;
  k.m1 = lane - 1 >= run_start ? 1.0f : 0.0f;
  k.m2 = lane - 2 >= run_start ? 1.0f : 0.0f;
  k.m4 = lane - 4 >= run_start ? 1.0f : 0.0f;
  k.m8 = lane - 8 >= run_start ? 1.0f : 0.0f;
  k.c1 = row == 1 && run_start < 16 ? 1.0f : 0.0f;     // the run began in an earlier row: take that row's carry
  k.c2 = row == 2 && run_start < 32 ? 1.0f : 0.0f;
  k.c3 = row == 3 && run_start < 48 ? 1.0f : 0.0f;
  return k;
}
__device__ __forceinline__ float run_sum(float v, const RunSteps& k) {
  v = fmaf(dpp_f<0x111, 0xf>(v), k.m1, v);
  v = fmaf(dpp_f<0x112, 0xf>(v), k.m2, v);
  v = fmaf(dpp_f<0x114, 0xf>(v), k.m4, v);
  v = fmaf(dpp_f<0x118, 0xf>(v), k.m8, v);
  v = fmaf(dpp_f<0x142, 0x2>(v), k.c1, v);
  v = fmaf(dpp_f<0x142, 0x4>(v), k.c2, v);
  v = fmaf(dpp_f<0x142, 0x8>(v), k.c3, v);
  return v;
}

// DET: every level through the fp32 form into the fixed-point shadow of dtable (det.q), see DetCtx
template <bool PK, bool DET = false>
__global__ __launch_bounds__(kThreads) void hashgrid_backward_kernel(HgLevels lv, int level0, SampleSrc src,
                                                                     const _Float16* __restrict__ dencT, long S, long Sp,
                                                                     float* __restrict__ dtable, _Float16* __restrict__ dtable_h,
                                                                     long hashed_lo, DevCount dc, const int* __restrict__ live_list,
                                                                     const int* __restrict__ live_count, DetCtx det = DetCtx{nullptr, nullptr}) {
  static_assert(!(PK && DET), "the deterministic scatter is the fp32-shaped one");
  S = live_samples(dc, S);
  if (live_list) S = 32L * *live_count;                         // the launch walks the listed segments only
  if ((long)blockIdx.x * kThreads >= S) return;                 // whole block past the live samples (block-uniform)
  const int l = level0 + blockIdx.y;
  const int F = lv.n_features;
  const bool hashed_level = (unsigned long long)lv.res[l] * lv.res[l] * lv.res[l] > lv.size[l];
  float* gdst = dtable + (size_t)lv.offset[l] * F;
  half2v* gdst_h = reinterpret_cast<half2v*>(dtable_h + ((size_t)lv.offset[l] * F - (size_t)hashed_lo));   // PK: F == 2
  const int lane = threadIdx.x & 63;
  long s = (long)blockIdx.x * kThreads + threadIdx.x;           // every lane stays: the aggregation reads across the wave
  const bool ok = s < S;
  if (live_list) s = ok ? (long)live_list[s >> 5] * 32 + (s & 31) : 0;
  // Loads are unconditional on a clamped index and masked afterwards: a load under `ok ? ... : 0` comes out of hipcc as a branch
  // with `s_waitcnt vmcnt(0)` behind it -- this kernel's start was five memory round trips in a row (list entry, one per
  // feature, two for the position); now the list entry, then everything else at once.
  const long sc = ok ? s : 0;
  // the position is fetched beside the gradient, its loads first (most waves leave right below: 12-24 bytes per lane read in vain,
  // but no round trip of their own)
  float x3[3];
  sample_pos_unmasked(src, sc, x3);
  float d[8];
  bool any_grad = false;
  if (F == 2) {
    const float d0 = (float)dencT[(long)(l * 2) * Sp + sc], d1 = (float)dencT[(long)(l * 2 + 1) * Sp + sc];
    d[0] = ok ? d0 : 0.0f;
    d[1] = ok ? d1 : 0.0f;
    any_grad = d[0] != 0.0f || d[1] != 0.0f;
  } else {
    for (int f = 0; f < F && f < 8; ++f) {
      d[f] = ok ? (float)dencT[(long)(l * F + f) * Sp + s] : 0.0f;
      any_grad |= d[f] != 0.0f;
    }
  }

  // A wave whose 64 samples all have a zero gradient at this level adds nothing: most of a NeRF batch (samples behind the
  // surface: transmittance 0) -- on the configs[2] batch 71 % of the waves leave here.
  if (__ballot(any_grad) == 0) return;
  float fr[3];
  unsigned g[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (!ok) x3[a] = 0.0f;
    const float x01 = fmaf(x3[a], 0.5f, 0.5f);
    const float p = fmaf(x01, lv.scale[l], 0.5f), fl = floorf(p);
    g[a] = (unsigned)(int)fl;
    fr[a] = p - fl;
  }
  unsigned i_lo[4], i_hi[4];
  rtxn::hg_corner_indices(g, lv.res[l], lv.size[l], hashed_level, i_lo, i_hi);
  int run_start = lane;
  bool run_last = true;
  const unsigned k0 = ok ? (g[0] | (g[1] << 16)) : 0xffffffffu, k1 = ok ? g[2] : (unsigned)lane;
  const unsigned p0 = (unsigned)dpp_i<0x138, 0xf>((int)k0), p1 = (unsigned)dpp_i<0x138, 0xf>((int)k1);   // wave_shr:1: lane - 1's keys
  const bool head = lane == 0 || p0 != k0 || p1 != k1;
  const bool aggregate = __ballot(head) != ~0ull;   // wave-uniform: finest levels have no runs and skip the scans
  RunSteps steps = {};
  if (aggregate) {
    // run_start = the nearest head at or below the lane: an inclusive max scan of (head ? lane : 0)
    int rs = head ? lane : 0;
    rs = max(rs, dpp_i<0x111, 0xf>(rs));
    rs = max(rs, dpp_i<0x112, 0xf>(rs));
    rs = max(rs, dpp_i<0x114, 0xf>(rs));
    rs = max(rs, dpp_i<0x118, 0xf>(rs));
    rs = max(rs, dpp_i<0x142, 0xa>(rs));
    rs = max(rs, dpp_i<0x143, 0xc>(rs));
    run_start = rs;
    const int next_head = dpp_i<0x130, 0xf>((int)head);      // wave_shl:1: lane + 1's flag
    run_last = lane == 63 || next_head != 0;
    steps = run_steps(lane, run_start);
  }
#pragma unroll
  for (int corner = 0; corner < 8; ++corner) {
    float w = 1.0f;
    unsigned p[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int hi = (corner >> a) & 1;
      w *= hi ? fr[a] : 1.0f - fr[a];
      p[a] = g[a] + (unsigned)hi;
    }
    const unsigned idx = (corner & 1) ? i_hi[corner >> 1] : i_lo[corner >> 1];
    if (PK) {
      float v0 = w * d[0], v1 = w * d[1];
      if (aggregate) {
        v0 = run_sum(v0, steps);
        v1 = run_sum(v1, steps);
      }
      if (ok && run_last && (v0 != 0.0f || v1 != 0.0f)) {
        const half2v hv = {(_Float16)v0, (_Float16)v1};
        __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) half2v*)(gdst_h + idx), hv);
      }
    } else {
      for (int f = 0; f < F && f < 8; ++f) {
        float v = w * d[f];
        if (aggregate) v = run_sum(v, steps);
        if (ok && run_last && v != 0.0f) {
          if (DET) grad_add(&gdst[(size_t)idx * F + f], v, det);
          else atomicAdd(&gdst[(size_t)idx * F + f], v);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------- MLP forward (training)
struct TrainArgs {
  const uint8_t* packed;    // packed_train (fwd) or packed_t (bwd)
  int n_hidden, out_act, E;
  long S, Sp;
  DevCount dc;              // S from the device (see DevCount); Sp stays the stride
  const _Float16* encT;     // [E][Sp]
  _Float16* acts;           // [L][W][Sp]
  unsigned long long* masks; // [L][Sp][2]: bit 8*kk + j of lane-half h's word = (post-ReLU activation != 0) of feature
                            // perm_feature(kk, h, j): all the backward chain needs of the activations (16 B instead of 2*W B)
  _Float16* out_half;       // [S][16]
  float4* radiance;         // [S] or NULL
  // backward
  const _Float16* dout;     // [S][4]
  _Float16* dz;             // [L][W][Sp]
  _Float16* dzL;            // [16][Sp]
  _Float16* dencT;          // [E][Sp] or NULL
  uint8_t* live_tiles;      // [Sp / 256]: backward: 1 where the tile carries a non-zero loss gradient (weight-gradient kernels skip the others)
  const int* live_list;     // segments that carry a loss gradient (rtxn_live_segments), or NULL.  Backward: the chain then visits
  const int* live_count;    // only those, and dz / dzL are written COMPACT (slot * 32 + sample) for the weight-gradient kernels
  SampleSrc src;            // forward with the encoder fused (ENC = 1): where the samples come from; encT is not read
  float* t_vals;            //   ... and the sampler's t_vals, if asked for (as rtxn_encode_frequency_segments writes them)
  float t_scale;
  int skip_last_dz;         // backward chain: dZ of the LAST hidden layer is not stored (the folded weight-gradient kernel forms it itself)
};

// Element (feature row f0 + 4h, sample s) of a feature-major tensor X[feature][Sp]: the address is split into a wave-uniform
// part (X + f0*Sp: scalar registers) and ONE 32-bit per-lane byte offset ((s + 4h*Sp)*2, < 2^32 for Sp < 2^28) that is the
// same for every access of the kernel -- the global instruction takes both (saddr + voffset).  Forming a 64-bit VGPR address
// per access instead made hipcc keep hundreds of them live across the layer loop.
__device__ __forceinline__ _Float16* row_elem(_Float16* X, int f0, long Sp, unsigned lane_off) {
  return reinterpret_cast<_Float16*>(reinterpret_cast<char*>(X + (long)f0 * Sp) + lane_off);
}
__device__ __forceinline__ const _Float16* row_elem(const _Float16* X, int f0, long Sp, unsigned lane_off) {
  return reinterpret_cast<const _Float16*>(reinterpret_cast<const char*>(X + (long)f0 * Sp) + lane_off);
}

// One B fragment (k-step kk: features perm_feature(kk, 0, 0..7), +4h folded into lane_off) of BOTH of a lane's samples to the
// rows of a feature-major tensor.  In mlp_bwd_kernel lane `col` of a wave owns the neighbouring samples 2 col and 2 col + 1 of
// the wave's 64 (column tile ct = the parity), so its two values of a feature are one aligned dword, the 32 lanes of a
// lane-half cover a whole 128-byte line, and a layer is 128 store instructions per wave instead of 256 (a wave holds at most 63
// outstanding memory instructions; tools/probe/store_shapes.hip, profiles/r03/train_store_paths.txt items 10-11).  The saving
// forward keeps its halves-of-the-tile ownership and two-byte stores: the same change left its time where it was.
// Each store is ONE instruction, written as asm: saddr = the row (a running scalar pointer, two scalar adds per store) +
// voffset = lane_off.  Left to itself hipcc re-associates the address into (X + lane_off) + row, a 64-bit per-lane base, pays a
// 64-bit VALU add per store, and forms every row's product with Sp ahead of the layer loop, spilling scalar registers to hold them.
__device__ __forceinline__ void store_fragment_rows_pair(_Float16* X, int kk, long Sp, unsigned lane_off, const half8& even, const half8& odd) {
  const rtxn::int4v w0 = __builtin_bit_cast(rtxn::int4v, even), w1 = __builtin_bit_cast(rtxn::int4v, odd);
  const char* row = reinterpret_cast<const char*>(X + 16L * kk * Sp);
  const long step = 2 * Sp;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // v_perm_b32: bytes 0-3 index the second operand, 4-7 the first: (even.lo | odd.lo << 16) or (even.hi | odd.hi << 16)
    const unsigned d = __builtin_amdgcn_perm((unsigned)w1[j >> 1], (unsigned)w0[j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
    asm volatile("" : "+s"(row));
    asm volatile("global_store_dword %0, %1, %2" ::"v"(lane_off), "v"(d), "s"(row) : "memory");
    row += j == 3 ? 5 * step : step;
  }
}

// SAVE = kSaveNone: outputs only (no activations, no masks): the forward half of the recompute path, whose backward
// (mlp_bwd_fused64_kernel) rebuilds the activations in registers from the encoded input.  kSaveMasks: outputs + the 16-byte
// sign masks per sample and layer and nothing else -- the forward of the LEAN 128-wide path (below: the dgrad chain needs
// only the masks, the weight-gradient kernel recomputes the activations): 160 instead of 2,208 bytes written per sample.
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
constexpr int kSaveNone = 0, kSaveAll = 1, kSaveMasks = 2;
// Diagnostic build only (-DRTXN_FWD_STAMPS, tools/probe/fwd_stamps.py; never in the shipped library): blocks kFwdStampBlock.. + 3 record
// s_memtime per wave: 0 entry, 1 layer 0 done (encoding fetched, transposed, multiplied), then per hidden layer l = 1..L-1 at 2 + 3 (l - 1):
// weights ready, MFMAs done, activations / masks saved; 2 + 3 (L - 1): output layer's weights ready, + 1: outputs stored.
#ifdef RTXN_FWD_STAMPS
constexpr int kFwdStampBlock = 6000, kFwdStampSlots = 32;
__device__ unsigned g_fwd_stamps[4 * 4 * kFwdStampSlots];
#define RTXN_FWD_STAMP(k)                                                                                               \
  do {                                                                                                                  \
    if (blockIdx.x >= kFwdStampBlock && blockIdx.x < kFwdStampBlock + 4 && NW == 4) {                                   \
      const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime();                                                       \
      if (lane == 0) g_fwd_stamps[((blockIdx.x - kFwdStampBlock) * 4 + wave) * kFwdStampSlots + (k)] = t_;              \
    }                                                                                                                   \
  } while (0)
#else
#define RTXN_FWD_STAMP(k)
#endif
constexpr int kEncScratch = 6 * 1024;                    // per wave: 48 feature rows x 64 samples of the encoding (layer 0's operand)
// NW: waves per block = 64-sample column pairs per block tile.  4 (256 samples, two blocks per CU, one weight buffer): the form of
// rounds 1-3.  8 (512 samples, ONE block per CU, the weights double-buffered, the next layer's fetched under this layer's MFMAs;
// 128 wide only): a layer's 32 KiB of A fragments then feed twice the samples -- an experiment (RTXN_TRAIN_FWD_WAVES=8): half the
// LDS-DMA weight stream per sample bought nothing (train_forward_impl has the A/B), so this kernel is not paced by it.
// ENC = 1 (lean forward of the reference's model): the Composite-Frequency(3 x 10, 2 x 12) encoding is computed in the kernel, straight
// into layer 0's fragments (encode_freq_fragments_3_10_2_12): no encT read, no LDS transposition -- and the standalone encoder, whose
// output only the weight-gradient kernel still needs, can run beside this kernel instead of in front of it.
template <int W, int SAVE = kSaveAll, int NW = 4, int ENC = 0>
__global__ __launch_bounds__(64 * NW, 2) void mlp_train_fwd_kernel(TrainArgs a) {
  constexpr int RT = W / 32, KS = W / 16, TILE = 64 * NW;
  constexpr bool DB = NW == 8;                             // double-buffered weights
  static_assert(NW == 4 || (NW == 8 && W == 128), "8-wave blocks are built for the 128-wide model");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
  a.S = live_samples(a.dc, a.S);
  // Live list (SAVE only): the saving pass of a step whose outputs were already computed for every sample by the outputs-only
  // kernel -- it visits only the segments that carry a loss gradient (a column tile is one segment, a block tile any eight),
  // and leaves their activations where the full pass would have (the backward kernels and the weight-gradient GEMM read them
  // there); slots past the list neither store nor count.
  const int live_n = a.live_list ? *a.live_count : 0;
  if (a.live_list ? (int)blockIdx.x * (2 * NW) >= live_n : (long)blockIdx.x * TILE >= a.S) return;
  const long tile0 = (long)blockIdx.x * TILE + wave * 64;
  const int KS0 = a.E / 16;
  const int L = a.n_hidden;
  long off = 0;
  RTXN_FWD_STAMP(0);
  const int WB = (KS0 > KS ? KS0 : KS) * RT * 1024;        // one weight buffer
  // every wave of the block fetches its share of a layer's fragments (1 KiB per wave instruction)
  auto stage_w = [&](const uint8_t* g, uint8_t* lds_buf, int bytes) {
    for (int o = wave * 1024; o < bytes; o += NW * 1024)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + o + lane * 16),
                                       (__attribute__((address_space(3))) void*)(lds_buf + o), 16, 0, 0);
  };
  auto wbuf = [&](int l) -> uint8_t* { return smem + (DB ? (l & 1) * WB : 0); };
  auto layer_bytes = [&](int l) -> int { return (l == 0 ? KS0 * RT : (l < L ? KS * RT : KS)) * 1024; };
  // Layer l's fragments in LDS, ready.  Single buffer: fetched here (everyone is done with layer l-1 first).  Double buffer: they
  // were fetched a layer ago; wait for them, and the buffer layer l-1 used is free for layer l+1 (prefetch(l) issues that).
  auto weights_ready = [&](int l) -> const uint8_t* {
    if (!DB) {
      if (l > 0) __syncthreads();
      stage_w(a.packed + off, wbuf(l), layer_bytes(l));
    }
    rtxn::staged_barrier();
    return wbuf(l);
  };
  auto prefetch = [&](int l) {                             // layer l + 1 into the other buffer (no output layer in the live pass)
    if (DB && l + 1 <= L && !(a.live_list && l + 1 == L)) stage_w(a.packed + off + layer_bytes(l), wbuf(l + 1), layer_bytes(l + 1));
  };

  // per-lane byte offset of this lane's sample within a feature row, lane-half row shift (4h rows) folded in
  unsigned lane_off[2];
  bool ok_s[2], store_s[2];
  long samp[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    long s = tile0 + ct * 32 + col;
    ok_s[ct] = s < a.S;
    store_s[ct] = true;                                   // padding columns of the last tile are written as zeros
    if (a.live_list) {
      const int slot = (int)blockIdx.x * (2 * NW) + wave * 2 + ct;
      ok_s[ct] = store_s[ct] = slot < live_n;
      s = (long)(ok_s[ct] ? a.live_list[slot] : 0) * 32 + col;     // slots past the list read segment 0 and store nothing
    } else if (tile0 >= a.Sp) {                             // 512-sample tiles over a row stride that is a multiple of 256: waves past it
      ok_s[ct] = store_s[ct] = false;                       // read the first columns and store nothing
      s = ct * 32 + col;
    }
    samp[ct] = s;
    lane_off[ct] = (unsigned)((s + 4L * h * a.Sp) * 2);
  }
  // ---- layer 0: B fragments from encT ----
  // A lane owns ONE sample and needs half of its features: read directly that is 8 two-byte loads per k-step and column tile,
  // 112 wave instructions of 128 useful bytes for the 112-feature encoding, and the texture path takes an instruction's 64
  // addresses at the same pace whatever their width -- 14,000 cycles per tile and CU, 40 % of this kernel's time when it saves
  // nothing (round 4).  Instead each wave fetches its 64 samples of 48 feature rows at a time as six 16-byte loads per lane (8
  // rows x 128 B per instruction, whole lines), drops them row-major into a 6-KiB scratch of its own behind the weights, and
  // picks its fragments out with two-byte LDS reads.  The two 64-byte halves of a row are swapped in rows 4-7 of every eight so
  // that lane-half h = 1 (rows + 4) reads the other half of the banks than h = 0.
  half8 bf[KS][2], bg[KS][2];
  if constexpr (ENC == 1) {
    static_assert(W == 128 && NW == 4, "the fused encoder is built into the four-wave 128-wide forward");
    constexpr int KSE = 7;
    stage_w(a.packed, wbuf(0), KSE * RT * 1024);          // layer 0's weights travel while the encoding is computed
    half8 b[2][KSE];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const long sc = ok_s[ct] ? samp[ct] : 0;
      float p3[3], v2[2];
      sample_pos_unmasked(a.src, sc, p3);
      sample_dir_unmasked(a.src, sc, v2);
      const float x[5] = {p3[0], p3[1], p3[2], v2[0], v2[1]};
      encode_freq_fragments_3_10_2_12<KSE>(x, h, ok_s[ct], b[ct]);
      if (a.t_vals && ok_s[ct] && h == 0) a.t_vals[samp[ct]] = sample_tval(a.src, samp[ct], a.t_scale);
    }
    RTXN_FWD_STAMP(26);
    rtxn::staged_barrier();
    RTXN_FWD_STAMP(27);
    const uint8_t* w0 = wbuf(0);
    floatx16 acc[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[rt][ct][e] = 0.0f;
#pragma unroll
    for (int kk = 0; kk < KSE; ++kk)                       // k-step outer, as the chunked form: the same sums in the same order
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const half8 af = *reinterpret_cast<const half8*>(w0 + ((rt * KSE + kk) * 64 + lane) * 16);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, b[ct][kk], acc[rt][ct], 0, 0, 0);
      }
    RTXN_FWD_STAMP(30);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) bf[2 * rt + s2][ct] = pack8<true>(acc[rt][ct], s2);
    off += (long)KSE * RT * 1024;
  } else {
    if (DB) stage_w(a.packed, wbuf(0), layer_bytes(0));
    uint8_t* scratch = smem + (DB ? 2 : 1) * WB + wave * kEncScratch;
    const int r8 = lane >> 3, jg = lane & 7;              // row of a piece; 16-byte sample group of the wave's 64 samples
    // (column of sample group jg: the wave's samples are two runs of 32 -- consecutive, or with the live list two listed segments)
    const long gcol = samp[jg >> 2] - col + 8 * (jg & 3);
    const _Float16* src = a.encT + gcol + (long)r8 * a.Sp;
    const int wr_off = r8 * 128 + ((jg * 16) ^ (((r8 >> 2) & 1) * 64));
    const int rd_off = (4 * h) * 128 + ((col * 2) ^ (h * 64));   // + feature row (of the chunk, rows & 7 < 4) * 128 + (ct * 64, same swap)
    // a chunk's six loads are issued a chunk ahead -- the first in front of the wait for layer 0's weights -- so that the kernel
    // stands still for ONE memory round trip per tile, not one per chunk plus the weights' (19,000 of the tile's 73,000 cycles)
    auto load_chunk = [&](int c0, rtxn::int4v (&pc)[6]) {
#ifdef RTXN_FWD_NO_ENC_FETCH      // timing-only (results wrong): what layer 0 costs when its input is free
      for (int p = 0; p < 6; ++p) pc[p] = rtxn::int4v{0x3c003c00, 0x3c003c00, 0x3c003c00, 0x3c003c00};
#else
#pragma unroll
      for (int p = 0; p < 6; ++p)
        // rows past the encoding are clamped, not skipped: their k-steps are never multiplied, and a conditional load here came out
        // of hipcc as six branches with `s_waitcnt vmcnt(0)` behind every single load -- six memory round trips in a row per chunk
        // (6,900 cycles from the block's entry to its first chunk's last load: phase stamps)
        pc[p] = *reinterpret_cast<const rtxn::int4v*>(src + (long)min(c0 + 8 * p, a.E - 8) * a.Sp);
#endif
    };
    rtxn::int4v piece[6], ahead[6];                      // two chunks in flight, taking turns (no copies: a copy would wait for the load)
    load_chunk(0, piece);
    __builtin_amdgcn_sched_barrier(0);
    RTXN_FWD_STAMP(26);
    const uint8_t* w0 = weights_ready(0);
    RTXN_FWD_STAMP(27);
    floatx16 acc[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[rt][ct][e] = 0.0f;
    auto chunk = [&](int c0, const rtxn::int4v (&pc)[6]) {
#ifndef RTXN_FWD_NO_ENC_FETCH
#pragma unroll
      for (int p = 0; p < 6; ++p) *reinterpret_cast<rtxn::int4v*>(scratch + p * 1024 + wr_off) = pc[p];
#endif
#ifdef RTXN_FWD_STAMPS
      if (c0 == 0) RTXN_FWD_STAMP(25);
#endif
      // all of the chunk's fragment reads first, then its MFMAs: k-step by k-step (reads, wait, permute, multiply) every k-step
      // was two or three LDS round trips in a row with the partner block's layer on the same pipe (1,700-2,000 cycles each: stamps)
      half8 b[3][2];
#ifdef RTXN_FWD_NO_ENC_FETCH
#pragma unroll
      for (int k3 = 0; k3 < 3; ++k3)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) b[k3][ct] = __builtin_bit_cast(half8, pc[(k3 + ct) % 6]);
#else
#pragma unroll
      for (int k3 = 0; k3 < 3; ++k3)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int j = 0; j < 8; ++j)
            b[k3][ct][j] = *reinterpret_cast<const _Float16*>(scratch + (perm_feature(k3, 0, j)) * 128 + (rd_off ^ (ct * 64)));
#endif
#ifdef RTXN_FWD_STAMPS
      if (c0 == 0) {
        asm volatile("" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]));
        RTXN_FWD_STAMP(31);
      }
#endif
#pragma unroll
      for (int k3 = 0; k3 < 3; ++k3) {
        const int kk = c0 / 16 + k3;
        if (kk >= KS0) break;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const half8 af = *reinterpret_cast<const half8*>(w0 + ((rt * KS0 + kk) * 64 + lane) * 16);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, b[k3][ct], acc[rt][ct], 0, 0, 0);
        }
      }
    };
    if (a.E == 112) {
      // the reference's encoding, straight-line: in the loop below hipcc waits `vmcnt(0)` in front of every chunk's LDS writes --
      // for the chunk it has just asked for as well (the registers are loop-carried) -- so a chunk's fetch never ran under the
      // previous chunk's work (3,400 cycles of round trip per chunk: stamps); without the back edge its counts are exact
      // (sched_barrier: left alone, hipcc's scheduler sinks each chunk's loads down to their first use -- behind the previous
      // chunk's MFMAs -- to save registers)
      load_chunk(48, ahead);
      __builtin_amdgcn_sched_barrier(0);
      chunk(0, piece);
      RTXN_FWD_STAMP(28);
      load_chunk(96, piece);
      __builtin_amdgcn_sched_barrier(0);
      chunk(48, ahead);
      RTXN_FWD_STAMP(29);
      chunk(96, piece);
      RTXN_FWD_STAMP(30);
    } else {
      for (int c0 = 0; c0 < a.E; c0 += 96) {             // E is a multiple of 16: a chunk is 1-3 whole k-steps
        if (c0 + 48 < a.E) load_chunk(c0 + 48, ahead);
        chunk(c0, piece);
        if (c0 + 48 < a.E) {
          if (c0 + 96 < a.E) load_chunk(c0 + 96, piece);
          chunk(c0 + 48, ahead);
        }
      }
    }
    // (layer 0 reads LDS with compiler-generated loads, in front of which hipcc waits for every LDS-DMA in flight: layer 1's
    // fragments are fetched behind them, not underneath)
    prefetch(0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) bf[2 * rt + s2][ct] = pack8<true>(acc[rt][ct], s2);
    off += (long)KS0 * RT * 1024;
  }
  // (the stores stay the compiler's: k-step outer, column tile inner, so the two 64-byte halves of a row's 128-byte line leave
  // the wave close together -- column-tile outer cost 0.5 ms per 4.7 M samples -- and in this kernel neither dropping the
  // per-store select, nor one-instruction stores, nor store_fragment_rows_pair's 4-byte form changed the time:
  // profiles/r03/train_store_paths.txt)
  auto save_acts = [&](int l, const half8 (&v)[KS][2]) {
    if constexpr (SAVE == kSaveNone) return;
    if constexpr (SAVE == kSaveAll) {
      _Float16* dst = a.acts + (long)l * W * a.Sp;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (store_s[ct]) *row_elem(dst, perm_feature(kk, 0, j), a.Sp, lane_off[ct]) = ok_s[ct] ? v[kk][ct][j] : (_Float16)0.0f;
    }
    // sign masks for the backward chain (mlp_bwd_kernel, see there): values are post-ReLU (>= 0), so "> 0" is "the half is
    // not +0".  Two instructions per dword of activations: v_pk_min_u16 with (1, 1) turns both halves into their bit, and
    // v_dot2_u32_u16 with the weights (1 << s, 2 << s) drops the pair at bit s of a running 16-bit field (two k-steps per field:
    // the weights are 16-bit).  Spelled as compares, shifts and ors this was ~550 VALU instructions per layer and wave -- 2,200 of
    // a layer's 5,800 cycles in the forward that stores nothing else (phase stamps, profiles/r04/fwd_stamps.txt).
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      unsigned field[KS / 2];
#pragma unroll
      for (int kp = 0; kp < KS / 2; ++kp) {
        unsigned f = 0;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const rtxn::int4v w = __builtin_bit_cast(rtxn::int4v, v[2 * kp + k2][ct]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int s = 8 * k2 + 2 * e;
            unsigned nz;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz) : "v"(w[e]), "s"(0x00010001u));
            f = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, nz), ushort2v{(unsigned short)(1u << s), (unsigned short)(2u << s)}, f, false);
          }
        }
        field[kp] = f;
      }
      unsigned long long mk = 0;
#pragma unroll
      for (int kp = 0; kp < KS / 2; ++kp) mk |= (unsigned long long)field[kp] << (16 * kp);
      if (store_s[ct]) a.masks[((long)l * a.Sp + samp[ct]) * 2 + h] = mk;
    }
  };
  RTXN_FWD_STAMP(1);
  save_acts(0, bf);
  // hidden layer on the hand-scheduled pipeline of the inference kernel (accumulators double-buffered by row tile: the
  // compiler-scheduled loop kept all RT tiles live and spilled at W = 128); nothing is staged underneath it here, and
  // the row tile it leaves pending is converted at once because the activations are stored after every layer
  auto hidden_layer = [&](int l, half8 (&in)[KS][2], half8 (&out)[KS][2]) {
    const uint8_t* w = weights_ready(l);
    prefetch(l);                                          // under this layer's MFMAs (pipe_layer reads LDS in asm: no compiler wait)
    RTXN_FWD_STAMP(2 + 3 * (l - 1));
    floatx16 acc2[2][2];
    rtxn::pipe_layer<RT, KS, KS>(w, in, out, acc2, lane);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) out[2 * (RT - 1) + s2][ct] = rtxn::relu_pack(acc2[1][ct], s2);
    RTXN_FWD_STAMP(2 + 3 * (l - 1) + 1);
  };
  // ---- hidden layers 1..L-1 (ping-pong bf <-> bg) ----
  int l = 1;
  for (; l + 1 < L; l += 2) {
    hidden_layer(l, bf, bg);
    save_acts(l, bg);
    RTXN_FWD_STAMP(2 + 3 * (l - 1) + 2);
    off += (long)KS * RT * 1024;
    hidden_layer(l + 1, bg, bf);
    save_acts(l + 1, bf);
    RTXN_FWD_STAMP(2 + 3 * l + 2);
    off += (long)KS * RT * 1024;
  }
  if (l < L) {
    hidden_layer(l, bf, bg);
    save_acts(l, bg);
    RTXN_FWD_STAMP(2 + 3 * (l - 1) + 2);
    off += (long)KS * RT * 1024;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) bf[kk][ct] = bg[kk][ct];
  }
  if (a.live_list) return;                                // the live pass saves activations only: the outputs exist already
  // ---- output layer ----
  const uint8_t* wL = weights_ready(L);
  RTXN_FWD_STAMP(2 + 3 * (L - 1));
  floatx16 acc[2];
  out_mma<KS, KS>(wL, bf, acc, lane);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const long s = tile0 + ct * 32 + col;
    if (s >= a.S) continue;
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float z = acc[ct][e];
      y[e] = a.out_act == RTXN_ACT_SIGMOID ? rtxn::sigmoidf_fast(z) : z;
    }
    half4v lo, hi;
#pragma unroll
    for (int e = 0; e < 4; ++e) { lo[e] = (_Float16)y[e]; hi[e] = (_Float16)y[4 + e]; }
    _Float16* o = a.out_half + s * 16;
    *reinterpret_cast<half4v*>(o + 4 * h) = lo;
    *reinterpret_cast<half4v*>(o + 8 + 4 * h) = hi;
    if (a.radiance && h == 0) a.radiance[s] = make_float4((float)lo[0], (float)lo[1], (float)lo[2], (float)lo[3]);
  }
  RTXN_FWD_STAMP(2 + 3 * (L - 1) + 1);
}
#ifdef RTXN_FWD_STAMPS
extern "C" int rtxn_debug_read_fwd_stamps(unsigned* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fwd_stamps), sizeof(unsigned) * 4 * 4 * kFwdStampSlots) == hipSuccess ? 0 : 1;
}
#endif

// ------------------------------------------------------------------------- MLP backward (dgrad)
template <int W>
__global__ __launch_bounds__(kThreads, 2) void mlp_bwd_kernel(TrainArgs a) {
  constexpr int RT = W / 32, KS = W / 16;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
  a.S = live_samples(a.dc, a.S);
  const int live_n = a.live_list ? *a.live_count : 0;
  if (a.live_list ? (int)blockIdx.x * 8 >= live_n : (long)blockIdx.x * kTile >= a.S) return;
  const long tile0 = (long)blockIdx.x * kTile + wave * 64;      // also the COMPACT position of the wave's 64 samples (live list)
  const int L = a.n_hidden;
  long off = 0;
  // lane_off: where the lane's sample sits in the tensors the forward wrote (encT, acts, masks; d(encoding) goes there too);
  // lane_dst: where its dZ goes -- the same place, or with the live list the compact position (slot * 32 + sample)
  unsigned lane_off[2], lane_dst[2];   // see row_elem
  long samp[2];
  bool ok_s[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    // lane col has the NEIGHBOURING samples 2 col and 2 col + 1 of the wave's 64 (column tile = parity; an MFMA column is a
    // sample either way, and nothing ties this kernel's ownership to the forward's): see store_fragment_rows_pair.  With the
    // live list the 64 samples are two listed segments: lanes 0-15 of a lane-half hold the first, 16-31 the second
    const int q = 2 * col + ct;
    long sidx = tile0 + q;
    ok_s[ct] = sidx < a.S;
    lane_dst[ct] = (unsigned)((sidx + 4L * h * a.Sp) * 2);
    if (a.live_list) {
      const int slot = (int)blockIdx.x * 8 + wave * 2 + (q >> 5);
      ok_s[ct] = slot < live_n;
      sidx = (long)(ok_s[ct] ? a.live_list[slot] : 0) * 32 + (q & 31);      // slots past the list read segment 0 and add zeros
    }
    samp[ct] = sidx;
    lane_off[ct] = (unsigned)((sidx + 4L * h * a.Sp) * 2);
  }

  // ---- output layer: dZ_out = dout (*) act'(out), one k-step (16 rows) ----
  half8 bo[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const long s = samp[ct];
    const bool ok = ok_s[ct];
    half8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.0f;
    if (ok && h == 0) {
      const half4v g = *reinterpret_cast<const half4v*>(a.dout + s * 4);
      const half4v y = *reinterpret_cast<const half4v*>(a.out_half + s * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gg = (float)g[j];
        if (a.out_act == RTXN_ACT_SIGMOID) { const float yy = (float)y[j]; gg = gg * yy * (1.0f - yy); }
        v[j] = (_Float16)gg;
      }
    }
    bo[ct] = v;
  }
  // A tile whose loss gradients are all zero (most of a NeRF batch: samples behind the surface) contributes nothing to any
  // gradient: it is marked dead for the weight-gradient kernels, its d(encoding) is written as zeros, and the block leaves.
  {
    bool any_grad = false;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int j = 0; j < 4; ++j) any_grad |= bo[ct][j] != (_Float16)0.0f;
    const bool live = __syncthreads_or(any_grad);
    if (a.live_tiles && threadIdx.x == 0) a.live_tiles[blockIdx.x] = live ? 1 : 0;
    if (!live && a.live_tiles) {
      // d(encoding) of the tile: zeros.  With the live list this still matters: a LISTED segment (non-zero dL/d(radiance)) can
      // have dZ_out == 0 in every sample -- a saturated sigmoid (y (1 - y) == 0 in fp16) or a product that underflows -- and the
      // hash scatter walks the same list, so its columns must not be left as they were (uninitialised memory: NaNs there once
      // poisoned a whole training run, 14 steps in).
      if (a.dencT) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          if (ok_s[ct] || !a.live_list)
            for (int f = 0; f < a.E; f += 8)
#pragma unroll
              for (int j = 0; j < 4; ++j) *row_elem(a.dencT, f + j, a.Sp, lane_off[ct]) = (_Float16)0.0f;
      }
      return;
    }
  }
  store_fragment_rows_pair(a.dzL, 0, a.Sp, lane_dst[0], bo[0], bo[1]);
  // One accumulator pair at a time: row tile rt of dA_{l} = W^T dZ is masked with relu'(act_l), rounded to fp16 and packed
  // straight into the B fragments of the next (earlier) layer's MFMAs -- the backward chain stays in registers exactly as the
  // forward does, and no full-layer fp32 dA is ever held (the first version kept one: 128 VGPRs at W = 128, 378 spills).
  // relu'(act_l) comes from the forward's sign masks: ONE 8-byte load per column tile and layer instead of W/2 two-byte
  // activation loads (bit 16*rt + e of the lane's word belongs to accumulator element e of row tile rt)
  // (rounds 1-2: at W = 128 the mask form made hipcc spill 165 VGPRs, so the 128-wide kernel kept reading the activations
  // themselves.  The spills were 64-bit row ADDRESSES of the dZ stores, which LICM hoisted out of the layer loop once the
  // activation loads that shared them were gone; with the row stride laundered through an empty asm per call
  // (mask_pack_store) they are formed where they are used: 168 VGPRs, no scratch, and the kernel no longer reads 2 KB of
  // activations per sample.  At W = 64 the masks were worth 5 % of the step.  The dZ stores themselves are
  // store_fragment_rows_pair's one-instruction, two-samples-per-lane form: 2.6 -> 2.25 -> 1.85 ms per 4.7 M samples at W = 128.)
  unsigned mk[2][2] = {{0, 0}, {0, 0}};   // low / high word of the lane's mask (row tiles 0-1 / 2-3)
  auto load_masks = [&](int l) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const uint2 w = *reinterpret_cast<const uint2*>(a.masks + ((long)l * a.Sp + samp[ct]) * 2 + h);
      mk[ct][0] = w.x;
      mk[ct][1] = w.y;
    }
  };
  auto mask_pack_store = [&](int l, int rt, const floatx16 (&acc)[2], half8 (&dst)[KS][2]) {
    long Sp_l = a.Sp;
    asm volatile("" : "+s"(Sp_l));          // not loop-invariant to the compiler: row addresses are formed where they are used
    _Float16* dzl = a.dz + (long)l * W * Sp_l;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      floatx16 m;
      const unsigned bits = (mk[ct][rt >> 1] >> (16 * (rt & 1))) & 0xffffu;
#pragma unroll
      for (int e = 0; e < 16; ++e) m[e] = (bits >> e) & 1u ? acc[ct][e] : 0.0f;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) dst[2 * rt + s2][ct] = pack8<false>(m, s2);
    }
    if (a.skip_last_dz && l == L - 1) return;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) store_fragment_rows_pair(dzl, 2 * rt + s2, Sp_l, lane_dst[0], dst[2 * rt + s2][0], dst[2 * rt + s2][1]);
  };

  half8 bz[KS][2], bn[KS][2];
  stage_rt(a.packed, smem, RT * 1024, tid);
  rtxn::staged_barrier();
  load_masks(L - 1);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {   // dZ_{L-1}: the output layer's single k-step
    const half8 af = *reinterpret_cast<const half8*>(smem + (rt * 64 + lane) * 16);
    floatx16 acc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      floatx16 z;
#pragma unroll
      for (int e = 0; e < 16; ++e) z[e] = 0.0f;
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bo[ct], z, 0, 0, 0);
    }
    mask_pack_store(L - 1, rt, acc, bz);
  }
  off += (long)RT * 1024;

  for (int l = L - 1; l >= 1; --l) {   // bz = dZ_l  ->  dZ_{l-1} = relu'(act_{l-1}) (*) W_l^T dZ_l
    __syncthreads();
    stage_rt(a.packed + off, smem, RT * KS * 1024, tid);
    rtxn::staged_barrier();
    off += (long)RT * KS * 1024;
    load_masks(l - 1);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      floatx16 acc[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ct][e] = 0.0f;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const half8 af = *reinterpret_cast<const half8*>(smem + ((rt * KS + kk) * 64 + lane) * 16);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bz[kk][ct], acc[ct], 0, 0, 0);
      }
      mask_pack_store(l - 1, rt, acc, bn);
      __builtin_amdgcn_sched_barrier(0);   // keep the tiles apart: hoisted activation loads of all tiles do not fit
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) bz[kk][ct] = bn[kk][ct];
  }
  if (!a.dencT) return;
  // ---- d(encoding) = W_0^T dZ_0 (hash-grid models) ----
  const int rows_t = (a.E + 31) / 32;
  __syncthreads();
  stage_rt(a.packed + off, smem, rows_t * KS * 1024, tid);
  rtxn::staged_barrier();
  for (int rt = 0; rt < rows_t; ++rt) {
    floatx16 acc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ct][e] = 0.0f;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const half8 af = *reinterpret_cast<const half8*>(smem + ((rt * KS + kk) * 64 + lane) * 16);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bz[kk][ct], acc[ct], 0, 0, 0);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const bool ok = ok_s[ct];
      if (a.live_list && !ok) continue;                  // a slot past the list has no column of its own to write
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int feat0 = 32 * rt + (e & 3) + 8 * (e >> 2);
        if (feat0 + 4 * h < a.E) *row_elem(a.dencT, feat0, a.Sp, lane_off[ct]) = ok ? (_Float16)acc[ct][e] : (_Float16)0.0f;
      }
    }
  }
}

// ------------------------------------------------------------------------- fused backward, 64-wide models
// network->backward for models whose whole gradient fits on the chip (64 wide, <= 4 hidden layers, encoded width <= 64:
// BASELINE configs[2], hash grid + 4x64).  The separate kernels above move every layer's activations and dZ through HBM
// three times (forward writes the activations, the dgrad chain reads them and writes dZ, the weight-gradient GEMM reads
// both): 2 x 2 B x W per sample and layer, which is what bounded them.  Here NOTHING per-sample and per-layer touches
// memory:
//   * the forward is RECOMPUTED from the encoded input (48-64 features per sample, read once) with the activations of all
//     layers kept in registers as the B fragments they already are -- the MFMA work is a few percent of the old kernels'
//     memory time;
//   * the dgrad chain runs in registers as before (accumulator -> next B operand);
//   * the weight gradient dW_l = dZ_l X_l^T contracts over SAMPLES, the lane index of those fragments, so each wave drops its
//     dZ_l / X_l tile into a [sample][feature] LDS image (136-byte rows) and the operands come back through
//     ds_read_b64_tr_b16, the hardware-transposing LDS read (probe: tools/probe/tr_probe.hip).  Wave w of the block owns
//     quadrant (w >> 1, w & 1) of every layer's 64x64 gradient and accumulates it in registers over ALL the block's samples
//     and over all its tiles (persistent grid); one pass of fp32 atomics per wave at the very end;
//   * all weights, forward-packed and transposed, are staged in LDS once per block (64-68 KiB) and stay.
// One wave per SIMD (the register file holds 4 layers of activations, 5 gradient quadrants and the chain's fragments:
// <= 512 VGPRs), 4 waves per block, one block per CU.
constexpr int kImgStride = 136;                 // bytes per sample row of an image: 64 features x 2 B + 8 (ds_write banks)
constexpr int kImgBytes = 64 * kImgStride;
constexpr int kFusedMaxL = 4;

struct FusedArgs {
  const uint8_t* packed_fwd;   // packed_train: layers 0 .. L-1 (the output layer is not recomputed)
  const uint8_t* packed_t;     // transposed layers in backward order: out, L-1, ..., 1, 0
  int L, KS0, out_act, E;
  long S, Sp;
  DevCount dc;                 // S and n_tiles from the device (see DevCount)
  int n_tiles;                 // 256-sample block tiles
  const _Float16* encT;        // [E][Sp]
  const _Float16* out_half;    // [S][16]
  const _Float16* dout;        // [S][4]
  _Float16* dencT;             // [E][Sp] or NULL
  float* dparams;              // tcnn layout, accumulated into
  const int* live_list;        // segments (32 samples) that carry a loss gradient, ascending, or NULL: all samples in order
  const int* live_count;       // device count of live_list
  DetCtx det;                  // deterministic mode: the final flush goes to the fixed-point shadow of dparams
};

typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));

// A/B operand of v_mfma_f32_32x32x16_f16 for a product that sums over SAMPLES: lane (r = l & 31, hh = l >> 5) gets
// image[sample 16 ks + 8 hh + j][feature 32 t + r], j = 0..7, as two transposing reads of 4 samples x 16 features per
// 16-lane group.  lane_tr = this lane's fixed part of the address (see the kernel).
__device__ __forceinline__ half8 tr_operand(const uint8_t* image, unsigned lane_tr, int t, int ks) {
  const uint8_t* p = image + lane_tr + (16 * ks) * kImgStride + 64 * t;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * kImgStride));
  const half4v l4 = __builtin_bit_cast(half4v, lo), h4 = __builtin_bit_cast(half4v, hi);
  return __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
}

// L hidden layers and KS0 = encoded width / 16 are template parameters: with them in registers the body was full of
// `if (l < L)` exec-mask branches, every accumulator was zeroed by 16 v_mov (now the first MFMA takes C = 0) and scalar
// registers spilled into VGPR lanes.
template <int L, int KS0>
__global__ __launch_bounds__(kThreads, 1) void mlp_bwd_fused64_kernel(FusedArgs a) {
  constexpr int W = 64, RT = 2, KS = 4;
  static_assert(L >= 1 && L <= kFusedMaxL && KS0 >= 1 && KS0 <= KS, "model outside the fused kernel's range");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (a.dc.total_segments) {
    a.S = live_samples(a.dc, a.S);
    a.n_tiles = (int)(padded_dev(a.S) / kTile);
  }
  // Live list: a column tile of 32 samples is exactly one segment, so a block tile is any eight segments -- with the list the
  // kernel visits only segments that carry a loss gradient (rtxn_live_segments), wherever they lie in the batch.
  const int live_n = a.live_list ? *a.live_count : 0;
  if (a.live_list) a.n_tiles = (live_n + 7) / 8;
  if ((int)blockIdx.x >= a.n_tiles) return;
  const int rt_e = (a.E + 31) / 32;
  const int fwd_bytes = (KS0 * RT + (L - 1) * KS * RT) * 1024;
  const int t_bytes = (RT + (L - 1) * RT * KS + rt_e * KS) * 1024;
  uint8_t* wF = smem;
  uint8_t* wT = smem + fwd_bytes;
  uint8_t* img = wT + t_bytes;                         // [wave][0: dZ | 1: X][kImgBytes]
  stage_rt(a.packed_fwd, wF, fwd_bytes, tid);
  stage_rt(a.packed_t, wT, t_bytes, tid);
  rtxn::staged_barrier();
  uint8_t* my_dz = img + (wave * 2) * kImgBytes;
  uint8_t* my_x = my_dz + kImgBytes;
  const int rt_w = wave >> 1, ct_w = wave & 1;         // this wave's quadrant of every layer's gradient
  // transposing-read address, fixed part: within a 16-lane group lane 4q+p supplies row q (a sample), columns 4p..4p+3
  // (features); groups 0/1 take features 0-15 / 16-31 of the 32-wide tile, lane half h the upper 8 samples of the k-step
  const unsigned lane_tr = (unsigned)((8 * h + ((lane & 15) >> 2)) * kImgStride + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
  // image store address, fixed part: sample (ct 32 + col) row, features 4h.. of a 16-feature k-step group
  const unsigned lane_wr = (unsigned)(col * kImgStride + 8 * h);

  floatx16 acc[kFusedMaxL], accL;
#pragma unroll
  for (int e = 0; e < 16; ++e) accL[e] = 0.0f;
#pragma unroll
  for (int l = 0; l < kFusedMaxL; ++l)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[l][e] = 0.0f;

  auto write_frag = [&](uint8_t* image, int kk, int ct, const half8& v) {
    uint8_t* p = image + lane_wr + ct * 32 * kImgStride + kk * 32;
    *reinterpret_cast<half4v*>(p) = __builtin_shufflevector(v, v, 0, 1, 2, 3);            // features 16kk + 4h + 0..3
    *reinterpret_cast<half4v*>(p + 16) = __builtin_shufflevector(v, v, 4, 5, 6, 7);       // features 16kk + 8 + 4h + 0..3
  };
  auto wgrad = [&](floatx16& q, int a_tile, int b_tile) {   // q += dZ[a_tile rows] X^T[b_tile cols] over the block's 256 samples
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const half8 af = tr_operand(img + (v * 2) * kImgBytes, lane_tr, a_tile, ks);
        const half8 bf = tr_operand(img + (v * 2 + 1) * kImgBytes, lane_tr, b_tile, ks);
        // the gradient quadrants live in AGPRs for the whole kernel (only these MFMAs touch them): said explicitly, so that the
        // 256 architectural VGPRs are left to the chain and the compiler has no accumulator to shuttle
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(q) : "v"(af), "v"(bf));
        if (ks == 3) __builtin_amdgcn_sched_barrier(0);   // one image's eight operand reads in flight at a time (all 32 hoisted: spills)
      }
  };

  for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const long tile0 = (long)tile * kTile + wave * 64;
    unsigned lane_off[2];
    bool ok_s[2];
    long sidx_ct[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      long sidx = tile0 + ct * 32 + col;
      ok_s[ct] = sidx < a.S;
      if (a.live_list) {
        const int slot = tile * 8 + wave * 2 + ct;
        ok_s[ct] = slot < live_n;
        sidx = (long)(ok_s[ct] ? a.live_list[slot] : 0) * 32 + col;     // slots past the list read segment 0 and add zeros
      }
      sidx_ct[ct] = sidx;
      lane_off[ct] = (unsigned)((sidx + 4L * h * a.Sp) * 2);
    }
    // ---- output layer: dZ_out = dout (*) act'(out) -- first, because a tile whose loss gradients are ALL zero contributes
    // exactly nothing to any gradient and is skipped whole (no recompute, no chain, no weight-gradient pass): in NeRF
    // training most samples lie behind the surface, where the transmittance and with it dL/d(radiance) is 0 (configs[2]
    // batch: 12 % of the samples carry a gradient).  Block-uniform decision; the tile's d(encoding) is written as zeros.
    half8 bo[2];
    bool any_grad = false;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      half8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.0f;
      const long sidx = sidx_ct[ct];
      if (ok_s[ct] && h == 0) {
        const half4v g = *reinterpret_cast<const half4v*>(a.dout + sidx * 4);
        const half4v y = *reinterpret_cast<const half4v*>(a.out_half + sidx * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float gg = (float)g[j];
          if (a.out_act == RTXN_ACT_SIGMOID) { const float yy = (float)y[j]; gg = gg * yy * (1.0f - yy); }
          v[j] = (_Float16)gg;
          any_grad |= v[j] != (_Float16)0.0f;
        }
      }
      bo[ct] = v;
    }
    if (!__syncthreads_or(any_grad)) {
      if (a.dencT && !a.live_list) {      // the tile's 256 columns of every row: 512 contiguous bytes per row, 16 bytes per thread and store
        _Float16* base = a.dencT + (long)tile * kTile;
        for (int i = tid; i < a.E * (kTile / 8); i += kThreads)
          *reinterpret_cast<uint4*>(base + (long)(i / (kTile / 8)) * a.Sp + (i % (kTile / 8)) * 8) = make_uint4(0u, 0u, 0u, 0u);
      } else if (a.dencT) {
        // Live list: the tile's segments ARE listed (non-zero dL/d(radiance)), yet every dZ_out is zero -- a saturated sigmoid
        // (y (1 - y) == 0 in fp16) or an underflowing product.  The hash scatter walks the same list, so their d(encoding)
        // columns must read zero, not what the buffer held before (uninitialised memory: NaNs there once poisoned a whole
        // training run 14 steps in).  lane (col, h) of column tile ct owns rows f + 4h + 0..3 of every 8.
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          if (ok_s[ct])
            for (int f = 0; f < a.E; f += 8)
#pragma unroll
              for (int j = 0; j < 4; ++j) *row_elem(a.dencT, f + j, a.Sp, lane_off[ct]) = (_Float16)0.0f;
      }
      continue;
    }
    // ---- encoded input as layer-0 B fragments (X_0); read again for layer 0's weight gradient (a cache hit) rather than
    // held in 32 VGPRs through the whole backward chain ----
    auto load_x0 = [&](half8 (&x)[KS][2]) {
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          half8 v;
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.0f;
          if (kk < KS0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *row_elem(a.encT, perm_feature(kk, 0, j), a.Sp, lane_off[ct]);
          }
          x[kk][ct] = v;
        }
    };
    // ---- forward, recomputed: post-ReLU activations of every hidden layer stay in registers ----
    half8 act[kFusedMaxL][KS][2];
    {
      half8 x0[KS][2];
      load_x0(x0);
      floatx16 f[RT][2];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int e = 0; e < 16; ++e) f[rt][ct][e] = 0.0f;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
        if (kk < KS0) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const half8 af = *reinterpret_cast<const half8*>(wF + ((rt * KS0 + kk) * 64 + lane) * 16);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) f[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, x0[kk][ct], f[rt][ct], 0, 0, 0);
          }
        }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) act[0][2 * rt + s2][ct] = pack8<true>(f[rt][ct], s2);
    }
    int foff = KS0 * RT * 1024;
#pragma unroll
    for (int l = 1; l < kFusedMaxL; ++l)
      if (l < L) {
        layer_mma<RT, KS, KS, 2>(wF + foff, act[l - 1], act[l], lane);
        foff += KS * RT * 1024;
      }
    half8 zero8;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero8[j] = (_Float16)0.0f;

    // dZ_l (accumulator tile rt of W_{l+1}^T dZ_{l+1}) masked with relu'(act) and packed as the next chain fragment
    // relu' on the PACKED halves: the activation is post-ReLU fp16 (bits 0 or a positive number), so min(bits, 1) is 0 / 1 per
    // half and an integer multiply of the gradient's bits by it keeps or clears the half -- two packed ops per two values
    // (v_pk_min_u16, v_pk_mul_lo_u16) where convert-compare-select on fp32 took seven.
    // (spelled in asm: written as vector min / multiply, hipcc turns it back into per-half compares, selects and v_perm.)
    const int ones16 = 0x00010001;
    auto mask_pack = [&](const floatx16 (&d)[2], int rt, const half8 (&actl)[KS][2], half8 (&dst)[KS][2]) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const rtxn::int4v g = __builtin_bit_cast(rtxn::int4v, pack8<false>(d[ct], s2));
          const rtxn::int4v x = __builtin_bit_cast(rtxn::int4v, actl[2 * rt + s2][ct]);
          rtxn::int4v r;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            int keep, out;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(keep) : "v"(x[q]), "s"(ones16));
            asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(out) : "v"(g[q]), "v"(keep));
            r[q] = out;
          }
          dst[2 * rt + s2][ct] = __builtin_bit_cast(half8, r);
        }
    };

    // ---- output layer: weight gradient dW_L[16 x 64] = dZ_out act_{L-1}^T, then dZ_{L-1} ----
    half8 bz[KS][2], bn[KS][2];
    __syncthreads();                                   // the previous tile's last weight-gradient pass is done with the images
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      write_frag(my_dz, 0, ct, bo[ct]);                // rows 0..15 (4..15 are zero)
      write_frag(my_dz, 1, ct, zero8);                 // rows 16..31 of the 32-row operand tile
    }
#pragma unroll
    for (int l = 0; l < kFusedMaxL; ++l)
      if (l == L - 1) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) write_frag(my_x, kk, ct, act[l][kk][ct]);
      }
    __syncthreads();
    if (rt_w == 0) wgrad(accL, 0, ct_w);
#pragma unroll
    for (int l = 0; l < kFusedMaxL; ++l)
      if (l == L - 1) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const half8 af = *reinterpret_cast<const half8*>(wT + (rt * 64 + lane) * 16);
          floatx16 d[2];
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            floatx16 z;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] = 0.0f;
            d[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bo[ct], z, 0, 0, 0);
          }
          mask_pack(d, rt, act[l], bz);
        }
      }
    int toff = RT * 1024;
    // ---- hidden layers L-1 .. 0: bz = dZ_l ----
#pragma unroll
    for (int li = 0; li < kFusedMaxL; ++li) {
      const int l = kFusedMaxL - 1 - li;               // compile-time: 3, 2, 1, 0
      if (l < L) {
        half8 x0[KS][2];
        if (l == 0) load_x0(x0);                       // issued before the barrier: in flight while the others catch up
        __syncthreads();                               // everyone is done reading the images of the layer above
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            write_frag(my_dz, kk, ct, bz[kk][ct]);
            if (l == 0) write_frag(my_x, kk, ct, x0[kk][ct]);
            else write_frag(my_x, kk, ct, act[l > 0 ? l - 1 : 0][kk][ct]);
          }
        __syncthreads();
        wgrad(acc[l], rt_w, ct_w);
        if (l > 0) {                                   // dZ_{l-1} = relu'(act_{l-1}) (*) W_l^T dZ_l
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            floatx16 d[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
              for (int e = 0; e < 16; ++e) d[ct][e] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
              const half8 af = *reinterpret_cast<const half8*>(wT + toff + ((rt * KS + kk) * 64 + lane) * 16);
#pragma unroll
              for (int ct = 0; ct < 2; ++ct) d[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bz[kk][ct], d[ct], 0, 0, 0);
            }
            mask_pack(d, rt, act[l > 0 ? l - 1 : 0], bn);
          }
#pragma unroll
          for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) bz[kk][ct] = bn[kk][ct];
          toff += RT * KS * 1024;
        } else if (a.dencT) {                          // d(encoding) = W_0^T dZ_0 (hash-grid models)
          for (int rt = 0; rt < rt_e; ++rt) {
            floatx16 d[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
              for (int e = 0; e < 16; ++e) d[ct][e] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
              const half8 af = *reinterpret_cast<const half8*>(wT + toff + ((rt * KS + kk) * 64 + lane) * 16);
#pragma unroll
              for (int ct = 0; ct < 2; ++ct) d[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bz[kk][ct], d[ct], 0, 0, 0);
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const int feat0 = 32 * rt + (e & 3) + 8 * (e >> 2);
                if (feat0 + 4 * h < a.E && (ok_s[ct] || !a.live_list)) *row_elem(a.dencT, feat0, a.Sp, lane_off[ct]) = ok_s[ct] ? (_Float16)d[ct][e] : (_Float16)0.0f;
              }
          }
        }
      }
    }
  }
  // ---- one pass of atomics per wave: its quadrant of every layer ----
  float* dW = a.dparams;
#pragma unroll
  for (int l = 0; l < kFusedMaxL; ++l)
    if (l < L) {
      const int N = l == 0 ? a.E : W;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int o = 32 * rt_w + (e & 3) + 8 * (e >> 2) + 4 * h, c = 32 * ct_w + col;
        if (c < N && acc[l][e] != 0.0f) grad_add(&dW[(long)o * N + c], acc[l][e], a.det);
      }
      dW += (long)W * N;
    }
  if (rt_w == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int o = (e & 3) + 8 * (e >> 2) + 4 * h, c = 32 * ct_w + col;
      if (o < 16 && accL[e] != 0.0f) grad_add(&dW[(long)o * W + c], accL[e], a.det);
    }
  }
}

// ------------------------------------------------------------------------- weight gradient
// dW_l[o][i] += sum_s dZ_l[o][s] * X_l[i][s] for every layer l in ONE launch.  One wave = one 64x64 output super-tile
// of one layer over a chunk of samples; grid = (super-tiles of the widest layer, sample chunks / 4, layers); block = 4
// waves on 4 consecutive chunks.  The contraction runs over samples, so each MFMA operand fragment (8 consecutive
// samples of one feature row of the feature-major tensors) is one 16-byte load; the loop is unrolled 4 k-steps deep with
// all 16 loads issued before the first MFMA (a wave is latency-bound otherwise: ~2 us per dependent load round).
struct WgradLayer {
  const _Float16* dZ;
  const _Float16* X;
  float* dW;
  int M, N, tiles_n, n_tiles;
};
struct WgradArgs {
  WgradLayer layer[17];
  long Sp, chunk;
  int lds_path;   // layers with >= 2 tiles are left to wgrad_lds_kernel
  DevCount dc;    // contraction length from the device (see DevCount); Sp stays the row stride
  const uint8_t* live_tiles;   // [Sp / 256] from mlp_bwd_kernel, or NULL: 256-sample tiles with dZ == 0 are not read
  const int* live_list;        // live segments: dZ is COMPACT (slot * 32 + sample), X sits where the forward wrote it; the
  const int* live_count;       // contraction runs over 32 * count samples
  DetCtx det;
};

__global__ __launch_bounds__(kThreads) void wgrad_kernel(WgradArgs a) {
  const WgradLayer& L = a.layer[blockIdx.z];
  if ((int)blockIdx.x >= L.n_tiles || (a.lds_path && L.n_tiles >= 2)) return;   // multi-tile layers: wgrad_lds_kernel
  const _Float16* __restrict__ dZ = L.dZ;
  const _Float16* __restrict__ X = L.X;
  const int M = L.M, N = L.N;
  const long Sp = a.Sp;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int tm = blockIdx.x / L.tiles_n, tn = blockIdx.x % L.tiles_n;
  long Send = a.dc.total_segments ? padded_dev(live_samples(a.dc, 0)) : Sp;
  if (a.live_list) Send = padded_dev(32L * *a.live_count);        // dZ beyond 32 * count in the last tile: zero rows of mlp_bwd_kernel
  const long s_begin = ((long)blockIdx.y * 4 + wave) * a.chunk;
  const long s_end = s_begin + a.chunk < Send ? s_begin + a.chunk : Send;
  if (s_begin >= Send) return;
  const int live_n = a.live_list ? *a.live_count : 0;
  // X position of compact sample position s (a multiple of 16): segments are 32 samples, a 16-sample k-step never straddles one
  auto x_of = [&](long s) -> long {
    if (!a.live_list) return s;
    const int slot = (int)(s >> 5);
    return (long)(slot < live_n ? a.live_list[slot] : 0) * 32 + (s & 31);
  };
  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  const int row0 = 64 * tm + r, row1 = row0 + 32, col0 = 64 * tn + r, col1 = col0 + 32;
  // rows beyond the layer read row 0 (valid memory) and are masked at the store
  const _Float16* pa0 = dZ + (long)(row0 < M ? row0 : 0) * Sp + 8 * h;
  const _Float16* pa1 = dZ + (long)(row1 < M ? row1 : 0) * Sp + 8 * h;
  const _Float16* pb0 = X + (long)(col0 < N ? col0 : 0) * Sp + 8 * h;
  const _Float16* pb1 = X + (long)(col1 < N ? col1 : 0) * Sp + 8 * h;
  constexpr int U = 4;   // Sp and the chunk are multiples of 256: whole groups of U k-steps
  for (long s = s_begin; s < s_end; s += 16 * U) {
    if (a.live_tiles && (s & (kTile - 1)) == 0 && !a.live_tiles[s / kTile]) {   // wave-uniform: a dead tile (dZ == 0) adds nothing
      s += kTile - 16 * U;
      continue;
    }
    half8 a0[U], a1[U], b0[U], b1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a0[u] = *reinterpret_cast<const half8*>(pa0 + s + 16 * u);
      a1[u] = *reinterpret_cast<const half8*>(pa1 + s + 16 * u);
      const long xs = x_of(s + 16 * u);
      b0[u] = *reinterpret_cast<const half8*>(pb0 + xs);
      b1[u] = *reinterpret_cast<const half8*>(pb1 + xs);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[u], b0[u], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[u], b1[u], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[u], b0[u], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[u], b1[u], acc[1][1], 0, 0, 0);
    }
  }
  float* __restrict__ dW = L.dW;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int o = 64 * tm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int c = 64 * tn + 32 * j + r;
        if (o < M && c < N) grad_add(&dW[(long)o * N + c], acc[i][j][e], a.det);
      }
}

// Layers with more than one 64x64 output tile (M or N up to 128): one BLOCK per sample chunk computes the whole (up to)
// 128x128 gradient, wave w its 64x64 quadrant (w >> 1, w & 1), from operands staged ONCE in LDS.  The per-wave kernel
// above reads every dZ / X row from global memory once per tile that uses it -- twice for a 128x128 layer, and it is
// bound by exactly that traffic (32 FLOP per byte).
// A stage is 64 samples of all 256 rows (128 dZ rows, then 128 X rows): one full 128-byte line per row, 32 KiB.  It is
// filled by LDS-DMA in pieces of 8 rows x 128 B (a wave instruction reads 8 whole lines; the fragment-shaped fill of
// rounds 2-3, 32 rows x 32 B per instruction, asked the texture path for four times the lines).  LDS-DMA writes lane i at
// piece + 16 i, so the image is row-major [256][128 B] and the bank swizzle sits on the SOURCE side: lane (row, slot')
// fetches the row's 16-byte sample group slot' ^ ((row >> 1) & 7), and the fragment read of lane (h, r) for k-step ks --
// samples 16 ks + 8 h .. + 7 of row 32 q + r -- is one ds_read_b128 at row * 128 + 16 * ((2 ks + h) ^ ((r >> 1) & 7)):
// conflict-free over ds_read_b128's four 16-lane groups (rows distinct mod 16 in each).
// The ring holds kWgStages stages with kWgStages - 1 outstanding per block, retired in order by counted
// `s_waitcnt vmcnt(N)` (LDS-DMA completes in issue order); nothing else inside the loop touches vmcnt -- the chunk's
// live-tile flags are gathered into a 64-bit mask up front and live-list entries come through the scalar cache.
#ifndef RTXN_WG_STAGES
#define RTXN_WG_STAGES 2
#endif
#ifndef RTXN_WG_CHUNK
#define RTXN_WG_CHUNK 0                                  // 0: chosen per launch (wgrad_chunk)
#endif
constexpr int kWgK = 4;                                  // k-steps (of 16 samples) per stage: one 128-byte line per row
constexpr int kWgStages = RTXN_WG_STAGES;                // ring depth; kWgStages - 1 stages in flight
constexpr int kWgStage = 256 * 128;                      // bytes per stage
constexpr int kWgLoads = 8;                              // LDS-DMA pieces per wave per stage
constexpr long kWgMaxChunk = 64L * kTile;                // samples per block: one 64-bit mask of 256-sample tiles
static_assert(kWgStages >= 2 && (kWgStages - 2) * kWgLoads <= 63, "vmcnt is a 6-bit count");
static_assert(kWgStages * kWgStage <= 160 * 1024, "LDS");
static_assert(RTXN_WG_CHUNK % kTile == 0 && RTXN_WG_CHUNK <= kWgMaxChunk, "chunk: whole tiles, at most 64");

// samples per block: long chunks keep the 64 KiB of atomic adds a block ends with rare (4.7 M samples x 9 layers at 2048 per
// block: 1.4 GB of them), short ones keep a small batch spread over the chip
static long wgrad_chunk(long Sp, int layers, bool live_list) {
  if (RTXN_WG_CHUNK) return RTXN_WG_CHUNK;
  if (live_list) return 2048;                            // the contraction length is on the device and usually short
  for (long c = 8192; c > 2048; c /= 2)
    if ((Sp + c - 1) / c * layers >= 4096) return c;
  return 2048;
}

template <int N>
__device__ __forceinline__ void wg_wait_then_barrier() {
  // own DMA pieces of the oldest stage landed (N newer ones may still fly), own LDS reads of the previous stage done
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
template <int D>
__device__ __forceinline__ void wg_retire_oldest(int newer) {   // newer: block-uniform number of stages issued after the oldest
  if constexpr (D == 0) wg_wait_then_barrier<0>();
  else {
    if (newer >= D) wg_wait_then_barrier<D * kWgLoads>();
    else wg_retire_oldest<D - 1>(newer);
  }
}

__global__ __launch_bounds__(kThreads) void wgrad_lds_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t wsm[];   // kWgStages stages
  const WgradLayer& L = a.layer[blockIdx.y];
  if (L.n_tiles < 2) return;                             // single-tile layers: wgrad_kernel
  const int M = L.M, N = L.N;
  const long Sp = a.Sp;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tm = wave >> 1, tn = wave & 1;
  long Send = a.dc.total_segments ? padded_dev(live_samples(a.dc, 0)) : Sp;
  const int live_n = a.live_list ? *a.live_count : 0;
  if (a.live_list) Send = padded_dev(32L * live_n);
  const long s_begin = (long)blockIdx.x * a.chunk;
  const long s_end = s_begin + a.chunk < Send ? s_begin + a.chunk : Send;
  if (s_begin >= Send) return;
  // 256-sample tiles of this chunk that carry a gradient (dZ != 0: mlp_bwd_kernel's live_tiles), one bit each
  unsigned long long live_mask = ~0ull;
  if (a.live_tiles) {
    const long t = s_begin / kTile + lane;
    live_mask = __ballot(t * kTile < s_end && a.live_tiles[t] != 0);
  }
  // stages of dead tiles are stepped over: block-uniform, registers only
  auto live_from = [&](long s) -> long {
    if (s >= s_end || (s & (kTile - 1)) != 0) return s;
    const unsigned long long rest = live_mask >> ((s - s_begin) / kTile);
    return rest ? s + (long)kTile * __builtin_ctzll(rest) : s_end;
  };
  // this wave's pieces: image rows 64 wave + 8 i + (lane >> 3), i = 0..7 -- waves 0, 1 fill the dZ rows, waves 2, 3 the X rows
  const bool x_rows = wave >= 2;
  const _Float16* src[kWgLoads];
#pragma unroll
  for (int i = 0; i < kWgLoads; ++i) {
    const int row = 64 * (wave & 1) + 8 * i + (lane >> 3), lim = x_rows ? N : M;
    const int j = (lane & 7) ^ ((row >> 1) & 7);         // the sample group this lane's LDS slot holds
    src[i] = (x_rows ? L.X : L.dZ) + (long)(row < lim ? row : 0) * Sp + 8 * j;   // rows beyond the layer: row 0, masked at the store
  }
  // j >> 2 = ((lane >> 2) & 1) ^ (i & 1): which of the stage's two 32-sample segments the lane reads in pieces of parity i & 1
  const bool second_even = ((lane >> 2) & 1) != 0;
  auto stage = [&](int buf, long s0) {
    // live list: dZ rows are compact, X rows sit where the forward wrote them -- the stage's 64 compact samples are the
    // segments list[s0 / 32] and list[s0 / 32 + 1]
    long xa = s0, xb = s0;
    if (a.live_list && x_rows) {
      const int slot = (int)(s0 >> 5);
      int seg0 = 0, seg1 = 0;
      const int* p = a.live_list + slot;
      if (slot < live_n) asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(seg0) : "s"(p) : "memory");
      if (slot + 1 < live_n) asm volatile("s_load_dword %0, %1, 0x4\n\ts_waitcnt lgkmcnt(0)" : "=s"(seg1) : "s"(p) : "memory");
      xa = (long)seg0 * 32;
      xb = (long)seg1 * 32 - 32;
    }
    const long off_even = second_even ? xb : xa, off_odd = second_even ? xa : xb;
#pragma unroll
    for (int i = 0; i < kWgLoads; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + ((i & 1) ? off_odd : off_even)),
                                       (__attribute__((address_space(3))) void*)(wsm + buf * kWgStage + (wave * kWgLoads + i) * 1024), 16, 0, 0);
  };
  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  long s_iss = live_from(s_begin);
  if (s_iss >= s_end) return;
  int in_flight = 0, ib = 0, cb = 0;                     // stages issued and not yet consumed; ring slots to fill / to read
#pragma unroll
  for (int d = 0; d < kWgStages - 1; ++d)
    if (s_iss < s_end) {
      stage(ib, s_iss);
      ib = ib + 1 == kWgStages ? 0 : ib + 1;
      ++in_flight;
      s_iss = live_from(s_iss + 16 * kWgK);
    }
  // fragment (k-step ks, 32-row group q) of lane (h, r): row 32 q + r, 16-byte slot (2 ks + h) ^ ((r >> 1) & 7)
  int frag_off[kWgK];
#pragma unroll
  for (int ks = 0; ks < kWgK; ++ks) frag_off[ks] = r * 128 + 16 * ((2 * ks + h) ^ ((r >> 1) & 7));
  while (in_flight > 0) {
    wg_retire_oldest<kWgStages - 2>(in_flight - 1);      // the oldest stage landed; everyone is done with the slot filled next
    if (s_iss < s_end) {
      stage(ib, s_iss);
      ib = ib + 1 == kWgStages ? 0 : ib + 1;
      s_iss = live_from(s_iss + 16 * kWgK);
    } else {
      --in_flight;
    }
    const uint8_t* st = wsm + cb * kWgStage;
#pragma unroll
    for (int ks = 0; ks < kWgK; ++ks) {
      const uint8_t* f = st + frag_off[ks];
      const half8 a0 = *reinterpret_cast<const half8*>(f + (2 * tm) * 4096);
      const half8 a1 = *reinterpret_cast<const half8*>(f + (2 * tm + 1) * 4096);
      const half8 b0 = *reinterpret_cast<const half8*>(f + (4 + 2 * tn) * 4096);
      const half8 b1 = *reinterpret_cast<const half8*>(f + (4 + 2 * tn + 1) * 4096);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
    }
    cb = cb + 1 == kWgStages ? 0 : cb + 1;
  }
  float* __restrict__ dW = L.dW;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int o = 64 * tm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int c = 64 * tn + 32 * j + r;
        if (o < M && c < N) grad_add(&dW[(long)o * N + c], acc[i][j][e], a.det);
      }
}

// ------------------------------------------------------------------------- lean path (128 wide): weight gradient, activations recomputed
// The saved-activation path above moves 8.8 KB per sample through HBM (forward: 2 KB of activations out; dgrad: 2 KB of dZ
// out; weight gradient: both back in) and its three kernels run at the streaming rate of those bytes.  The lean
// path keeps dZ and drops the activations: the forward writes outputs + sign masks only (mlp_train_fwd_kernel<128, kSaveMasks>),
// the dgrad chain is mlp_bwd_kernel as it stands (it reads the masks, writes dZ), and THIS kernel forms
//   dW_l += dZ_l A_{l-1}^T
// from dZ streamed out of HBM once and A_{l-1} RECOMPUTED from the encoded input, layer by layer in forward order, so that no
// activation is ever stored: 5.4 KB per sample instead of 8.8 (the encoding is read once per pass), a workspace of 2.2 instead
// of 4.3 KB per sample.
//   * One persistent block of four waves per CU (one wave per SIMD, 512 registers); a block tile is 256 samples, a wave owns 64
//     of them through the forward chain exactly as in the other training kernels (two 32-column tiles, activations in
//     registers as the next layer's B fragments).
//   * dW_l = 128 x 128 fp32 = 64 KiB; wave w owns quadrant (w >> 1, w & 1) of every layer of the pass in AGPRs (4 tiles x 16
//     registers per layer).  The model's gradient therefore takes several passes of this kernel, each recomputing the forward as
//     far as its layers need: as shipped THREE -- layers 0-2 (forward through layer 1), 3-5 (through 4), 6-7 + the output
//     layer (through 7): 15 layer-forwards = 1.9 forward passes of recompute, 192 AGPRs.  Four layers per pass (two passes, 1.4
//     forward passes) fill all 256 AGPRs and hipcc then spills accumulator tiles around the tile loop (tried, gone).
//     The output layer's 16 x 128 gradient costs no image at all: the last hidden layer is computed with the MFMA operands
//     exchanged, which leaves its activations as an A operand over samples (see the last layer's forward), 64 AGPRs.
//   * The contraction runs over samples -- the LANE index of the chain's fragments -- so each wave drops A_{l-1} into a
//     [sample][feature] LDS image (rows of 328 bytes: conflict-free ds_write_b64, the transposing ds_read_b64_tr_b16 two-way on 3
//     of 32 lanes) and the B operands come back through the transposing read (as mlp_bwd_fused64_kernel); only two images fit, so
//     a layer's contraction runs in two halves of 128 samples (waves 0, 1 write, all contract; waves 2, 3 write, all contract).
//   * dZ_l arrives by LDS-DMA in stages of 64 samples x 128 rows (whole 128-byte lines, source-side bank swizzle: the layout of
//     wgrad_lds_kernel) through a ring of four 16-KiB slots = one layer of one tile; a layer is contracted as two pairs of stages,
//     and a pair is retired by a counted `s_waitcnt vmcnt(K)` + barrier.  The order of every wave's vector-memory operations is
//     static, so each K is a constant: the operations issued after the awaited pair -- the other pair (8), the next layer's
//     weight fetch (8; 7 for layer 0), the look-ahead to the next layer's first pair (8).  Where more has been issued than K
//     assumes, the wait is only longer.  The first gradient layer's four stages are issued at the top of the tile, behind the
//     wait for the encoding; the last gradient layer looks ahead to nothing and touches the next tile's encoding instead.
//   * Tiles whose loss gradients are all zero (mlp_bwd_kernel's live_tiles: it writes no dZ for them) are stepped over.
constexpr int kLnStr = 328;                   // bytes per sample row of an X image: 256 + 72 (72 = 8 x 9: rows land 9 bank pairs apart)
constexpr int kLnImg = 64 * kLnStr;           // one wave's 64 samples
constexpr int kLnStage = 16 * 1024;           // 128 rows x 128 B
constexpr int kLnOffW = 0;                    // one layer's forward weights (32 KiB)
constexpr int kLnOffRing = 32 * 1024;
constexpr int kLnOffX = kLnOffRing + 4 * kLnStage;
constexpr int kLnOffOL = kLnOffX + 2 * kLnImg;   // dZ of the output layer, 16 rows x 256 samples
constexpr int kLnEncScratch = 112 * 128;           // one wave's encoded tile, [feature][64 samples]: 14 KiB, from kLnOffX on (the images, the
constexpr int kLnOffJunk = kLnOffX + 4 * kLnEncScratch;   // output layer's dZ and 7 KiB more are all free at the top of a tile); 256 B nobody reads
static_assert(kLnOffJunk >= kLnOffOL + 8192, "the scratch ends behind the output layer's dZ");
constexpr int kLnOffWoT = kLnOffOL + 8192;         // ENC passes (no encoding scratch): the output layer's W^T, four 1-KiB fragments, resident for the kernel
static_assert(kLnOffWoT + 4096 <= kLnOffJunk, "W_out^T sits in the tail of the (then unused) encoding scratch");
constexpr int kLnLds = kLnOffJunk + 256;
#ifdef RTXN_LN_STAMPS
constexpr int kLnLdsLaunch = kLnLds + 4096;   // the stamps
#else
constexpr int kLnLdsLaunch = kLnLds;
#endif
static_assert(kLnLdsLaunch <= 160 * 1024, "LDS");
static_assert(kLnStr % 8 == 0 && kLnOffX % 16 == 0 && kLnOffOL % 16 == 0, "alignment of the transposing and 16-byte reads");

struct LeanArgs {
  const uint8_t* packed_fwd;   // packed_train
  int out_act;
  long S, Sp;
  DevCount dc;
  int n_tiles;
  const _Float16* encT;        // [E][Sp]
  const _Float16* dz;          // [L][128][Sp]   (COMPACT with the live list, as mlp_bwd_kernel writes it)
  const _Float16* dzL;         // [16][Sp]
  float* dparams;              // tcnn layout, accumulated into
  const uint8_t* live_tiles;   // [tiles]: mlp_bwd_kernel's; a tile with 0 has no dZ
  const int* live_list;
  const int* live_count;
  DetCtx det;
  SampleSrc src;               // ENC passes: the packed segments the encoding is recomputed from (encT is not read)
  const uint8_t* packed_bwd;   // ENC passes: packed_t, whose first four fragments are the output layer's W^T (dZ of the last hidden layer is formed here)
};

// (timing-only ablations, results wrong: -DRTXN_LN_NO_WAIT no wait for the dZ stages / weights, -DRTXN_LN_NO_CONTRACT no contraction,
// -DRTXN_LN_NO_FWD no recomputed forward: tools/ablate.sh with ABLATE_SRC=train)
template <int N>
__device__ __forceinline__ void ln_wait_vm() {
#ifndef RTXN_LN_NO_WAIT
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
// The workgroup barrier of this kernel, as a bare instruction.  __syncthreads() is a release fence + s_barrier, and with LDS-DMA
// in flight hipcc makes the fence `s_waitcnt vmcnt(0)`: every barrier would drain the dZ ring (first build: 25 % MFMA busy).
// What a wave must have finished before it arrives is said explicitly: its LDS writes (lgkmcnt) -- the DMA pieces by the
// counted ln_wait_vm in front of it.
__device__ __forceinline__ void ln_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// LDS reads of the contraction as asm: hipcc treats the transposing-read builtin as aliasing every LDS-DMA in flight and puts
// `s_waitcnt vmcnt(0)` in front of the first one of each stage (same drain).  The waits for their results are then ours too
// (ln_operands_ready takes the operands THROUGH the wait, see rtxn::mfma_results_settle for why).
template <int OFF>
__device__ __forceinline__ void ln_read_b128(half8& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
__device__ __forceinline__ void ln_read_tr(half8& dst, unsigned addr, int off_lo, int off_hi) {
  s16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr + (unsigned)off_lo));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(addr + (unsigned)off_hi));
  const half4v l4 = __builtin_bit_cast(half4v, lo), h4 = __builtin_bit_cast(half4v, hi);
  dst = __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <int N>
__device__ __forceinline__ void ln_operands_ready(half8& a0, half8& a1, half8& b0, half8& b1) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1) : "n"(N));
}

// Diagnostic build only (-DRTXN_LN_STAMPS, tools/probe/lean_stamps.py; never in the shipped library): the first block of each
// pass records s_memtime at the phase boundaries of its tiles 20 and 21, every wave its own, into the LDS left over behind
// kLnLds, and copies them out when it is done.  Slot 0 top of the tile, 1 encoding in registers; layer l at 2 + 11 l:
// +0 step begins, +1 past [W+T], +2 forward done, +3 past [v0], +4 first pair contracted, +7 past [v2] (second images written),
// +8 second pair contracted; 91 tile done.
#ifdef RTXN_LN_STAMPS
constexpr int kLnStampTiles = 2, kLnStampSlots = 92, kLnStampFirst = 20;   // tiles 20, 21 of the block: steady state
__device__ unsigned g_ln_stamps[3 * 4 * kLnStampTiles * kLnStampSlots];
__device__ unsigned long long g_ln_clock[3 * 8];   // per pass, first block: s_memtime / s_memrealtime (100 MHz) at its begin and end, its tiles
#define RTXN_LN_STAMP(k)                                                                                          \
  do {                                                                                                            \
    if (sub_block == 0 && tile_it >= kLnStampFirst && tile_it < kLnStampFirst + kLnStampTiles) {                      \
      const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime();                                                 \
      if (lane == 0) stamp_lds[(wave * kLnStampTiles + (tile_it - kLnStampFirst)) * kLnStampSlots + (k)] = t_;    \
    }                                                                                                             \
  } while (0)
#else
#define RTXN_LN_STAMP(k)
#endif

// L0 <= l < L1: the layers whose gradient this pass accumulates (at most 4); OUT: also the output layer (then L1 == LTOT)
// sub_block of sub_grid: this block's place among the blocks that run THIS pass (see wgrad_recompute_kernel)
// ENC: the tile's encoded input is COMPUTED (the reference's Composite-Frequency(3 x 10, 2 x 12), encode_freq_fragments_3_10_2_12)
// from the packed segments instead of fetched: a column tile is one segment, so its start / end / view direction come through the
// scalar cache and the tile's dZ stages are in flight while the vector ALU encodes.
template <int KS0, int L0, int L1, bool OUT, int LTOT, bool ENC = false>
__device__ __forceinline__ void wgrad_recompute_pass(LeanArgs a, const int sub_block, const int sub_grid) {
  constexpr int W = 128, RT = 4, KS = 8, NL = L1 - L0;
  constexpr int FWD_END = OUT ? LTOT : L1 - 1;            // forward layers 0 .. FWD_END-1 are recomputed
  static_assert(NL >= 1 && NL <= 4 && L0 >= 0 && L1 <= LTOT && (!OUT || L1 == LTOT), "pass layout");
  static_assert(!OUT || (NL >= 2 && NL <= 3), "the output layer's dZ is fetched in the step before the last (a gradient layer); its accumulators take 64 AGPRs");
  static_assert(KS0 >= 1 && KS0 <= KS, "encoded width");
  constexpr int W0_OPS = (KS0 * RT + 3) / 4;              // stage_rt's LDS-DMA instructions per wave for layer 0; 8 for a hidden layer
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tm = wave >> 1, tn = wave & 1;
  if (a.dc.total_segments) {
    a.S = live_samples(a.dc, a.S);
    a.n_tiles = (int)(padded_dev(a.S) / kTile);
  }
  const int live_n = a.live_list ? *a.live_count : 0;
  if (a.live_list) a.n_tiles = (live_n + 7) / 8;
  // tiles sub_block, + sub_grid, ...; dead ones stepped over (scalar loads: nothing of this may enter the vector-memory queue)
  auto next_live = [&](int t) -> int {
    while (t < a.n_tiles) {
      int word;
      const uint8_t* p = a.live_tiles + (t & ~3);
      asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(word) : "s"(p) : "memory");
      if ((word >> (8 * (t & 3))) & 0xff) break;
      t += sub_grid;
    }
    return t;
  };
  int tile = next_live(sub_block);
  if (tile >= a.n_tiles) return;
  static_assert(KS0 == 7, "the encoding's scratch and its 14 fetches per wave are spelled for 112 features");

  uint8_t* const ring = smem + kLnOffRing;
  uint8_t* const ximg = smem + kLnOffX;
#ifdef RTXN_LN_STAMPS
  unsigned* const stamp_lds = reinterpret_cast<unsigned*>(smem + kLnLds);
  for (int i = tid; i < 4 * kLnStampTiles * kLnStampSlots; i += kThreads) stamp_lds[i] = 0;
  int tile_it = 0;
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- fixed per-lane address parts ----
  // dZ stage fill: this wave's pieces are image rows 32 wave + 8 i + (lane >> 3), i = 0..3; lane slot (lane & 7) of a row holds
  // the 16-byte sample group (lane & 7) ^ ((row >> 1) & 7)
  // The address of a piece is split into a wave-uniform part (layer, tile, the piece's first row: scalar registers) and ONE
  // 32-bit per-lane byte offset per piece parity (row (lane >> 3) of the piece, sample group): the saddr + voffset form.  Formed
  // as 64-bit per-lane pointers they are loop-invariant per (layer, piece), and hipcc hoisted a hundred of them out of the tile
  // loop (see mlp_bwd_kernel's mask_pack_store): the row stride is laundered through an empty asm per call.
  const unsigned fill_off[2] = {(unsigned)(((long)(lane >> 3) * a.Sp + 8 * ((lane & 7) ^ (lane >> 4))) * 2),
                                (unsigned)(((long)(lane >> 3) * a.Sp + 8 * ((lane & 7) ^ (4 + (lane >> 4)))) * 2)};
  auto issue_stage = [&](int slot, int layer, long s0) {
    long Sp_l = a.Sp;
    asm volatile("" : "+s"(Sp_l));
    const _Float16* dz_l = a.dz;
    asm volatile("" : "+s"(dz_l));
    const char* base = reinterpret_cast<const char*>(dz_l + ((long)layer * W + 32 * wave) * Sp_l + s0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + fill_off[i & 1]),
                                       (__attribute__((address_space(3))) void*)(ring + slot * kLnStage + (wave * 4 + i) * 1024), 16, 0, 0);
      base += 16 * Sp_l;                                  // 8 rows on
    }
  };
  // output layer's dZ: wave w fetches its own 64 samples, rows 0-7 and 8-15, into sub-image w (2 KiB, same row-major swizzle)
  auto issue_ol = [&](long s0) {
    long Sp_l = a.Sp;
    asm volatile("" : "+s"(Sp_l));
    const _Float16* dz_l = a.dzL;
    asm volatile("" : "+s"(dz_l));
    const char* base = reinterpret_cast<const char*>(dz_l + s0 + 64 * wave);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + fill_off[i]),
                                       (__attribute__((address_space(3))) void*)(smem + kLnOffOL + wave * 2048 + i * 1024), 16, 0, 0);
      base += 16 * Sp_l;
    }
  };
  // Every 128-byte line of tile t's encoding for this wave, touched once (4 bytes per lane into 256 B of LDS nobody reads): by the
  // time the tile's real fetch asks for them they sit in the L2.  Four instructions: rows lane and 64 + lane of both 32-sample segments.
  auto prefetch_enc = [&](int t) {
    long Sp_l = a.Sp;
    asm volatile("" : "+s"(Sp_l));
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      long c0 = (long)t * kTile + wave * 64 + ct * 32;
      if (a.live_list) {
        const int slot = t * 8 + wave * 2 + ct;
        int seg = 0;
        const int* p = a.live_list + slot;
        if (slot < live_n) asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(seg) : "s"(p) : "memory");
        c0 = (long)seg * 32;
      }
#pragma unroll
      for (int rh = 0; rh < 2; ++rh) {
        const int row = lane + 64 * rh < 112 ? lane + 64 * rh : 111;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.encT + c0 + (long)row * Sp_l),
                                         (__attribute__((address_space(3))) void*)(smem + kLnOffJunk), 4, 0, 0);
      }
    }
  };
  // fragment (k-step ks) of lane (h, r = col): row 32 q + r, 16-byte slot (2 ks + h) ^ ((r >> 1) & 7); row group q at + q * 4096
  int frag_off[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) frag_off[ks] = col * 128 + 16 * ((2 * ks + h) ^ ((col >> 1) & 7));
  // X images: see mlp_bwd_fused64_kernel (tr_operand / write_frag), row stride kLnStr
  const unsigned lane_tr = (unsigned)((8 * h + ((lane & 15) >> 2)) * kLnStr + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
  const unsigned lane_wr = (unsigned)(col * kLnStr + 8 * h);
  auto write_image = [&](uint8_t* image, const half8 (&v)[KS][2]) {
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        uint8_t* p = image + lane_wr + ct * 32 * kLnStr + kk * 32;
        *reinterpret_cast<half4v*>(p) = __builtin_shufflevector(v[kk][ct], v[kk][ct], 0, 1, 2, 3);
        *reinterpret_cast<half4v*>(p + 16) = __builtin_shufflevector(v[kk][ct], v[kk][ct], 4, 5, 6, 7);
      }
  };
  // the gradient quadrants: AGPRs for the whole kernel, touched only by the asm MFMAs below and the final atomics
  floatx16 acc[NL][4];
#pragma unroll
  for (int i = 0; i < NL; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][t][e] = 0.0f;
  // the output layer's gradient, TRANSPOSED and for this wave's samples only: oq[rt] lane (o = col < 16, h), register e = feature
  // 32 rt + (e & 3) + 8 (e >> 2) + 4 h (64 more AGPRs of the pass with two hidden layers; see the last layer's forward below)
  floatx16 oq[4];
  if constexpr (OUT) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) oq[t][e] = 0.0f;
  }

  // 128 samples of the contraction: dZ stages in ring slots s0, s0 + 1 against X images 0, 1 -- eight k-steps in one software
  // pipeline.  Operands double-buffered by k-step: the six reads of k-step ks + 1 are in flight while the four MFMAs of ks run (one
  // wave per SIMD: nothing else hides them); run as two separate 64-sample contractions (first build) every stage paid the
  // pipeline's ramp and a barrier of its own.
  const unsigned ring_addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)ring;
  const unsigned ximg_addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)ximg;
  auto contract_pair = [&](floatx16 (&q)[4], int s0) {
#ifdef RTXN_LN_NO_CONTRACT
    return;
#endif
    // (laundered: as loop invariants of the tile loop hipcc computed every stage's sixteen operand addresses ahead of it and
    // parked them in AGPRs -- 52 of them, and then spilled an accumulator tile to scratch)
    unsigned st = ring_addr + s0 * kLnStage + (2 * tm) * 4096;
    unsigned xi = ximg_addr + lane_tr + 64 * (2 * tn);
    asm volatile("" : "+s"(st), "+v"(xi));
    half8 a0[2], a1[2], b0[2], b1[2];
    auto fetch = [&](int k8, int s) {                    // k-step k8 of the pair: stage s0 + (k8 >> 2), image k8 >> 2, k-step k8 & 3 of it
      const unsigned sa = st + (k8 >> 2) * kLnStage + frag_off[k8 & 3], xa = xi + (k8 >> 2) * kLnImg;
      const int ks = k8 & 3;
      ln_read_b128<0>(a0[s], sa);
      ln_read_b128<4096>(a1[s], sa);
      ln_read_tr(b0[s], xa, (16 * ks) * kLnStr, (16 * ks + 4) * kLnStr);
      ln_read_tr(b1[s], xa, (16 * ks) * kLnStr + 64, (16 * ks + 4) * kLnStr + 64);
    };
    fetch(0, 0);
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) {
      const int s = k8 & 1;
      if (k8 + 1 < 8) {
        fetch(k8 + 1, s ^ 1);
        ln_operands_ready<6>(a0[s], a1[s], b0[s], b1[s]);
      } else {
        ln_operands_ready<0>(a0[s], a1[s], b0[s], b1[s]);
      }
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(q[0]) : "v"(a0[s]), "v"(b0[s]));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(q[1]) : "v"(a0[s]), "v"(b1[s]));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(q[2]) : "v"(a1[s]), "v"(b0[s]));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(q[3]) : "v"(a1[s]), "v"(b1[s]));
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- prologue: layer 0's weights (from then on every forward fetches the next one's) and the first three stages of the first tile ----
  if (FWD_END > 0) stage_rt(a.packed_fwd, smem + kLnOffW, KS0 * RT * 1024, tid);
  constexpr bool DZ_LAST = ENC && OUT;                    // dZ of the last hidden layer is computed here, not fetched (see the last layer's step)
  if constexpr (DZ_LAST) stage_rt(a.packed_bwd, smem + kLnOffWoT, RT * 1024, tid);   // lands with layer 0's weights, in front of the first [W+T] barrier

  while (tile < a.n_tiles) {
    const int nxt_tile = next_live(tile + sub_grid);
    const int look = nxt_tile < a.n_tiles ? nxt_tile : tile;       // no further tile: the encoding prefetch touches this one again (harmless; the wait counts stay static)
    RTXN_LN_STAMP(0);
    const long tile0 = (long)tile * kTile + wave * 64;
    // where the wave's two 32-sample column tiles sit in encT: consecutive, or with the live list two listed segments
    long col0[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      col0[ct] = tile0 + ct * 32;
      if (a.live_list) {
        // the list entry through the scalar cache (the slot is wave-uniform): a vector load here would enter the vector-memory
        // queue, and hipcc's wait for it would drain the dZ ring
        const int slot = tile * 8 + wave * 2 + ct;
        int seg = 0;
        const int* p = a.live_list + slot;
        if (slot < live_n) asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(seg) : "s"(p) : "memory");
        col0[ct] = (long)seg * 32;                        // slots past the list read segment 0; their dZ is zero
      }
    }
    // ---- encoded input as B fragments (k-steps >= KS0: zeros, the padding columns of dW_0's operand) ----
    // A lane of the chain owns ONE sample and needs 56 of its 112 features: as two-byte loads that is 112 wave instructions of 128
    // useful bytes each, and the texture path takes an instruction's 64 addresses at the same pace whatever their width (first
    // build: 14,000 cycles per tile).  Each wave fetches its 64 samples x 112 rows as 14 LDS-DMA instructions (8 rows x 128 B each,
    // whole lines) into a 14-KiB scratch of its own -- everything from the X images on is free between tiles -- and picks its
    // fragments out with two-byte LDS reads.  The wait for those 14 is the one place where this kernel stands still for a memory
    // round trip (phase stamps, profiles/r04/lean_stamps.txt: 5,000-10,000 cycles per tile with the lines coming from HBM), so the
    // previous tile's last step has touched every line once (prefetch_enc below: they come from the L2 now), and the tile's four
    // dZ stages are issued behind the wait, not in front of it, so that it does not wait for them as well.
    half8 act[2][KS][2];                                // ping-pong: layer l reads act[l & 1], its forward writes act[(l + 1) & 1]
    if constexpr (ENC) {
      ln_barrier();                                     // everyone has left the previous tile's last contraction: the ring is free
      // the two segments' constants through the scalar cache (col0 is wave-uniform): nothing of this enters the vector-memory queue
      float seg[2][8];
      const long n_seg = (a.S + 31) >> 5;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        long g = col0[ct] >> 5;
        g = g < n_seg ? g : n_seg - 1;                  // tiles past the batch (the last one's padding): any real segment, their dZ is zero
        const float* ps = a.src.start + 3 * g;
        const float* pe = a.src.end + 3 * g;
        const float* pv = a.src.seg_view + 2 * g;
        unsigned long long s01, e01, v01;
        unsigned s2, e2;
        asm volatile("s_nop 4\n\ts_load_dwordx2 %0, %5, 0x0\n\ts_load_dword %1, %5, 0x8\n\ts_load_dwordx2 %2, %6, 0x0\n\ts_load_dword %3, %6, 0x8\n\t"
                     "s_load_dwordx2 %4, %7, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(s01), "=&s"(s2), "=&s"(e01), "=&s"(e2), "=&s"(v01) : "s"(ps), "s"(pe), "s"(pv) : "memory");
        seg[ct][0] = __builtin_bit_cast(float, (unsigned)s01);
        seg[ct][1] = __builtin_bit_cast(float, (unsigned)(s01 >> 32));
        seg[ct][2] = __builtin_bit_cast(float, s2);
        seg[ct][3] = __builtin_bit_cast(float, (unsigned)e01);
        seg[ct][4] = __builtin_bit_cast(float, (unsigned)(e01 >> 32));
        seg[ct][5] = __builtin_bit_cast(float, e2);
        seg[ct][6] = __builtin_bit_cast(float, (unsigned)v01);
        seg[ct][7] = __builtin_bit_cast(float, (unsigned)(v01 >> 32));
      }
      // the four dZ stages of the tile's first gradient layer: the whole ring is free, and they travel while the encoding is computed
#pragma unroll
      for (int v = 0; v < 4; ++v) issue_stage(v, L0, (long)tile * kTile + 64 * v);
      const float tpar = ((float)col + (a.src.midpoint ? 0.5f : 0.0f)) * (1.0f / 32);      // as sample_pos: sample i of its segment
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float x[5];
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = fmaf(tpar, seg[ct][3 + c] - seg[ct][c], seg[ct][c]);
        x[3] = seg[ct][6];
        x[4] = seg[ct][7];
        half8 frag[KS0];
        encode_freq_fragments_3_10_2_12<KS0>(x, h, true, frag);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) act[0][kk][ct] = kk < KS0 ? frag[kk] : half8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    } else {
    long Sp_t = a.Sp;
    asm volatile("" : "+s"(Sp_t));
    {
      ln_barrier();                                     // everyone has left the previous tile's last contraction: the X images are free
      uint8_t* scratch = smem + kLnOffX + wave * kLnEncScratch;
      const int r8 = lane >> 3, jg = lane & 7;          // row of a piece, 16-byte sample group of the wave's 64 samples
      const _Float16* src = a.encT + (jg < 4 ? col0[0] : col0[1]) + 8 * (jg & 3) + (long)r8 * Sp_t;
#pragma unroll
      for (int p = 0; p < 14; ++p)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)(8 * p) * Sp_t),
                                         (__attribute__((address_space(3))) void*)(scratch + p * 1024), 16, 0, 0);
      ln_wait_vm<0>();                                  // own pieces, own reads: no barrier
      rtxn::int4v w[KS][2];
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) w[kk][ct] = rtxn::int4v{0, 0, 0, 0};
      const uint8_t* mine = scratch + (4 * h) * 128 + col * 2;     // + feature row * 128 + ct * 64
#pragma unroll
      for (int kk = 0; kk < KS0; ++kk)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int f0 = perm_feature(kk, 0, j);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            const unsigned v = *reinterpret_cast<const unsigned short*>(mine + f0 * 128 + ct * 64);
            w[kk][ct][j >> 1] |= (int)(v << (16 * (j & 1)));
          }
        }
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) act[0][kk][ct] = __builtin_bit_cast(half8, w[kk][ct]);
      // (the first barrier of step 0 stands between these reads and the first image written over the scratch)
      // the four dZ stages of the tile's first gradient layer: the whole ring is free
#pragma unroll
      for (int v = 0; v < 4; ++v) issue_stage(v, L0, (long)tile * kTile + 64 * v);
    }
    }
    RTXN_LN_STAMP(1);
    // One layer, l a compile-time constant: [forward: nxt = relu(W_l cur)] THEN [weight gradient of layer l from A_{l-1} = cur],
    // then cur = nxt.  Forward first, so that the four dZ stages of the layer -- issued one by one as the previous layer's
    // contraction freed their slots -- have the forward's ~3,000 cycles to arrive: with the contraction first, the fourth stage
    // was issued when the layer began and every layer paid one memory latency (first build: 14.5 us per tile of layers 0-2).
    // Vector-memory order of a wave, which the counted waits below rely on (all static):
    //   [W+T] wait W_l, barrier (also: everyone is done with the previous layer's X images and ring slots 2, 3) -> issue stages
    //         (l, 2), (l, 3) if the previous step was a gradient layer of this tile (it looked ahead to (l, 0), (l, 1))
    //   forward_l; waves 0, 1 write X
    //   [v0]  wait stages (l, 0), (l, 1): behind them (l, 2), (l, 3) = 8; barrier (also: everyone is done with W_l) -> issue W of the
    //         next forward layer (the next tile's layer 0 after the last one) [+ the output layer's dZ]; contract the first pair
    //   [v2]  barrier -> look ahead: stages (l + 1, 0), (l + 1, 1), or the prefetch of the next tile's encoding (4); waves 2, 3 write
    //         X; wait stages (l, 2), (l, 3): behind them the weights and the look-ahead; barrier; contract the second pair
    // W_l therefore has behind it the look-ahead of the previous step (8) if that step was a gradient layer.
    auto layer_step = [&](auto LC) {
      constexpr int l = decltype(LC)::value;
      constexpr bool has_w = l >= L0 && l < L1, has_f = l < FWD_END;
      half8 (&cur)[KS][2] = act[l & 1];
      half8 (&nxt)[KS][2] = act[(l + 1) & 1];
      if constexpr (has_w || has_f) {
        // the step before this one in program order (cyclically: the last step of the previous tile) and whether it contracted
        constexpr int first_step = 0, last_step = OUT ? LTOT - 1 : L1 - 1;        // steps are layers 0 .. last_step (all of them forward, contract or both)
        constexpr int prev = l == first_step ? last_step : l - 1;
        constexpr bool prev_w = prev >= L0 && prev < L1;
        // the next layer with a forward, cyclically, and its LDS-DMA instructions per wave
        constexpr int next_f = l + 1 < FWD_END ? l + 1 : 0;
        constexpr int w_next = has_f ? (next_f == 0 ? W0_OPS : 8) : 0;           // issued by this step only if it runs a forward (see [v0])
        constexpr int next_w = l + 1 < L1 && l + 1 >= L0 ? l + 1 : L0;            // the next layer with a gradient (cyclically)
        constexpr bool next_w_same_tile = has_w && l + 1 < L1;
        RTXN_LN_STAMP(2 + 11 * l);
        // ---- [W+T] ----
        constexpr bool dz_here = DZ_LAST && l == LTOT - 1;                       // this layer's dZ is formed in the kernel: nothing of it is in flight
        if constexpr (has_f) ln_wait_vm<(dz_here ? 0 : prev_w ? 8 : 0)>();        // (dz_here: behind the weights only the output layer's dZ, needed as well)
        ln_barrier();
        if constexpr (has_w && l > L0 && !dz_here) {
          // slots 2, 3 (the previous step's second pair has just left them): stages 2, 3 of this layer, whose stages 0, 1 the
          // previous step looked ahead to
          issue_stage(2, l, (long)tile * kTile + 64 * 2);
          issue_stage(3, l, (long)tile * kTile + 64 * 3);
        }
        RTXN_LN_STAMP(2 + 11 * l + 1);
        // ---- forward ----
#ifdef RTXN_LN_NO_FWD
        if constexpr (false) {
#else
        if constexpr (has_f) {
#endif
          const uint8_t* wl = smem + kLnOffW;
          if constexpr (l == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
              floatx16 f[2];
#pragma unroll
              for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int e = 0; e < 16; ++e) f[ct][e] = 0.0f;
#pragma unroll
              for (int kk = 0; kk < KS0; ++kk) {
                const half8 af = *reinterpret_cast<const half8*>(wl + ((rt * KS0 + kk) * 64 + lane) * 16);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) f[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, cur[kk][ct], f[ct], 0, 0, 0);
              }
#pragma unroll
              for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) nxt[2 * rt + s2][ct] = pack8<true>(f[ct], s2);
            }
          } else if constexpr (OUT && l == LTOT - 1) {
            // The last hidden layer's activations feed nothing but the output layer's gradient dW_L[o][f] = sum_s dZ_L[o][s] A[f][s]:
            // a contraction over SAMPLES, the lane index of the chain's fragments.  Instead of a round through LDS images (first
            // build: two image halves, four barriers, 6,400 of the tile's 66,000 cycles) the layer is computed with the MFMA
            // operands exchanged: the pipeline then leaves in nxt[2 rt + s][ct] feature 32 rt + col at eight samples per lane
            // (rtxn::pipe_layer, SWAP) -- already an A operand with k = samples.  The B operand is dZ_L at the same eight samples
            // (two 8-byte reads from the wave's own 2-KiB sub-image: no barrier), the product dW_L^T stays in AGPRs for the kernel.
            floatx16 acc2[2][2];
            rtxn::pipe_layer<RT, KS, KS, true>(wl, cur, nxt, acc2, lane);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
              for (int ct = 0; ct < 2; ++ct) nxt[2 * (RT - 1) + s2][ct] = rtxn::relu_pack(acc2[1][ct], s2);
            // row o = col & 15 of the sub-image (lanes col >= 16 repeat rows: columns 16-31 of the product, never stored); sample
            // group g sits in 16-byte slot g ^ ((row >> 1) & 7)
            const unsigned ol_row = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)(smem + kLnOffOL) + wave * 2048 + (col & 15) * 128 + 8 * h;
            const unsigned swz = ((col & 15) >> 1) & 7;
            s16x4 dl[2][2][2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
              for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                  const unsigned ad = ol_row + (((unsigned)(4 * ct + 2 * s + q) ^ swz) << 4);
                  asm volatile("ds_read_b64 %0, %1" : "=v"(dl[ct][s][q]) : "v"(ad));
                }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dl[0][0][0]), "+v"(dl[0][0][1]), "+v"(dl[0][1][0]), "+v"(dl[0][1][1]), "+v"(dl[1][0][0]), "+v"(dl[1][0][1]), "+v"(dl[1][1][0]), "+v"(dl[1][1][1]));
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
              for (int s = 0; s < 2; ++s) {
                const half8 bz = __builtin_shufflevector(__builtin_bit_cast(half4v, dl[ct][s][0]), __builtin_bit_cast(half4v, dl[ct][s][1]), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)         // four different accumulators in a row: a dependent MFMA is four behind
                  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(oq[rt]) : "v"(nxt[2 * rt + s][ct]), "v"(bz));
              }
            if constexpr (DZ_LAST) {
              // dZ of THIS layer, formed here instead of fetched (the dgrad chain does not store it: 256 B per sample less written and
              // read): dZ^T[f][s] = relu'(A[f][s]) sum_o W_L[o][f] dZ_L[o][s] with the operands exchanged as above -- A operand dZ_L with
              // the sample on the lane (eight two-byte reads per column tile from the wave's own sub-image), B operand the output
              // layer's W^T fragment -- so the product comes out feature-on-lane exactly like nxt, whose zeros ARE the mask, and a lane
              // holds four consecutive samples per accumulator quad: one ds_write_b64 each into the ring slot of the wave's 64 samples, in
              // the layout the DMA stages have (row f, 16-byte sample group g in slot g ^ ((f >> 1) & 7)).  The same products, one
              // rounding to fp16: the values the chain kernel would have stored.
              const uint8_t* olw = smem + kLnOffOL + wave * 2048;
              uint8_t* stage = ring + wave * kLnStage;
#pragma unroll
              for (int ct = 0; ct < 2; ++ct) {
                half8 dza;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  const int R = 8 * h + j, sl = 32 * ct + col;
                  dza[j] = *reinterpret_cast<const _Float16*>(olw + R * 128 + ((((sl >> 3) ^ ((R >> 1) & 7))) << 4) + (sl & 7) * 2);
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                  const half8 wt = *reinterpret_cast<const half8*>(smem + kLnOffWoT + (rt * 64 + lane) * 16);
                  floatx16 z;
#pragma unroll
                  for (int e = 0; e < 16; ++e) z[e] = 0.0f;
                  const floatx16 gz = __builtin_amdgcn_mfma_f32_32x32x16_f16(dza, wt, z, 0, 0, 0);
                  const int f = 32 * rt + col;
#pragma unroll
                  for (int q = 0; q < 4; ++q) {
                    const half8& tq = nxt[2 * rt + (q >> 1)][ct];
                    half4v o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = tq[4 * (q & 1) + i] != (_Float16)0.0f ? (_Float16)gz[4 * q + i] : (_Float16)0.0f;
                    *reinterpret_cast<half4v*>(stage + f * 128 + (((4 * ct + q) ^ ((f >> 1) & 7)) << 4) + 8 * h) = o;
                  }
                }
              }
            }
          } else {
            floatx16 acc2[2][2];
            rtxn::pipe_layer<RT, KS, KS>(wl, cur, nxt, acc2, lane);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
              for (int ct = 0; ct < 2; ++ct) nxt[2 * (RT - 1) + s2][ct] = rtxn::relu_pack(acc2[1][ct], s2);
          }
        }
        RTXN_LN_STAMP(2 + 11 * l + 2);
        auto issue_next_weights = [&]() {
          if constexpr (has_f) {
            constexpr int wbytes = (next_f == 0 ? KS0 : KS) * RT * 1024;
            constexpr long woff = next_f == 0 ? 0 : (long)(KS0 + (next_f - 1) * KS) * RT * 1024;
            const uint8_t* pf = a.packed_fwd;
            asm volatile("" : "+s"(pf));
            stage_rt(pf + woff, smem + kLnOffW, wbytes, tid);
          }
        };
        if constexpr (has_w) {
          constexpr int li = l - L0;
          // the output layer's dZ of this tile is fetched one step ahead of its use (the last layer's forward): two more
          // instructions behind this step's stages 1-3
          constexpr bool ol_here = OUT && l == LTOT - 2;
          constexpr int w_behind = w_next + (ol_here ? 2 : 0);
          // Issued when the first pair has left slots 0, 1: the next gradient layer's stages 0, 1.  The tile's last gradient layer
          // looks ahead to nothing (the next tile issues its four stages itself, behind its wait for the encoding): it warms the L2
          // with the next tile's encoding instead.
          constexpr bool next_dz_here = DZ_LAST && l + 1 == LTOT - 1;            // the next layer's dZ will be formed in the kernel: no look-ahead
          auto look_ahead = [&]() {
            if constexpr (next_w_same_tile && !next_dz_here) {
              issue_stage(0, next_w, (long)tile * kTile);
              issue_stage(1, next_w, (long)tile * kTile + 64);
            } else if constexpr (!ENC) {
              prefetch_enc(look);
            }
          };
          constexpr int ahead_ops = next_w_same_tile && !next_dz_here ? 8 : (ENC ? 0 : 4);
          if (wave < 2) write_image(ximg + wave * kLnImg, cur);
          ln_wait_vm<8>();                                // stages 0, 1: behind them stages 2, 3
          ln_barrier();                                   // + the images of waves 0, 1; everyone has left the forward
          issue_next_weights();
          if constexpr (ol_here) issue_ol((long)tile * kTile);
          RTXN_LN_STAMP(2 + 11 * l + 3);
          contract_pair(acc[li], 0);
          RTXN_LN_STAMP(2 + 11 * l + 4);
          ln_barrier();                                   // both images and slots 0, 1 have been read by everyone
          look_ahead();
          if (wave >= 2) write_image(ximg + (wave - 2) * kLnImg, cur);
          ln_wait_vm<w_behind + ahead_ops>();             // stages 2, 3: behind them the weights (+ the output layer's dZ) and the look-ahead
          ln_barrier();
          RTXN_LN_STAMP(2 + 11 * l + 7);
          contract_pair(acc[li], 2);
          RTXN_LN_STAMP(2 + 11 * l + 8);
        } else {
          ln_barrier();                                   // everyone has left the forward: its weights may be overwritten
          issue_next_weights();
          RTXN_LN_STAMP(2 + 11 * l + 3);
        }
      }
    };
    layer_step(std::integral_constant<int, 0>{});
    layer_step(std::integral_constant<int, 1>{});
    layer_step(std::integral_constant<int, 2>{});
    layer_step(std::integral_constant<int, 3>{});
    layer_step(std::integral_constant<int, 4>{});
    layer_step(std::integral_constant<int, 5>{});
    layer_step(std::integral_constant<int, 6>{});
    layer_step(std::integral_constant<int, 7>{});
    static_assert(LTOT <= 8, "layer_step is spelled out for eight layers");
    RTXN_LN_STAMP(91);
#ifdef RTXN_LN_STAMPS
    ++tile_it;
#endif
    tile = nxt_tile;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last prefetch instructions land in LDS this block still owns
  __syncthreads();
  // The accumulators were last written by asm MFMAs, which hipcc's hazard recogniser does not see: the MFMA-result -> read wait
  // states are spent here, with the accumulators going THROUGH the statement (rtxn::mfma_results_settle, for AGPRs;
  // tools/check_asm_mfma_reads.py counted 18 of the 19 between the last contraction and the first v_accvgpr_read without it).
#pragma unroll
  for (int i = 0; i < NL; ++i) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]));
  if constexpr (OUT) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(oq[0]), "+a"(oq[1]), "+a"(oq[2]), "+a"(oq[3]));
  // ---- one pass of atomics per wave: its quadrant of every layer of the pass ----
  const int E = KS0 * 16;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int l = L0 + i;
    const int N = l == 0 ? E : W;
    float* dW = a.dparams + (l == 0 ? 0L : (long)W * E + (long)(l - 1) * W * W);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int orow = 64 * tm + 32 * (t >> 1) + (e & 3) + 8 * (e >> 2) + 4 * h, c = 64 * tn + 32 * (t & 1) + col;
        if (c < N && acc[i][t][e] != 0.0f) grad_add(&dW[(long)orow * N + c], acc[i][t][e], a.det);
      }
  }
  if constexpr (OUT) {
    float* dW = a.dparams + (long)W * E + (long)(LTOT - 1) * W * W;     // [16][W]; every wave adds its samples' share
    if (col < 16) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int f = 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (oq[rt][e] != 0.0f) grad_add(&dW[col * W + f], oq[rt][e], a.det);
        }
    }
  }
#ifdef RTXN_LN_STAMPS
  if (sub_block == 0) {
    for (int i = tid; i < 4 * kLnStampTiles * kLnStampSlots; i += kThreads) g_ln_stamps[(L0 / 3) * 4 * kLnStampTiles * kLnStampSlots + i] = stamp_lds[i];
    if (tid == 0) {
      unsigned long long* c = g_ln_clock + (L0 / 3) * 8;
      c[0] = clk0, c[1] = __builtin_amdgcn_s_memtime(), c[2] = rt0, c[3] = __builtin_amdgcn_s_memrealtime(), c[4] = (unsigned long long)tile_it;
    }
  }
#endif
}

// One pass as a launch of its own (small batches: fewer tiles than CUs).
template <int KS0, int L0, int L1, bool OUT, int LTOT, bool ENC = false>
__global__ __launch_bounds__(kThreads, 1) void wgrad_recompute_kernel(LeanArgs a) {
  wgrad_recompute_pass<KS0, L0, L1, OUT, LTOT, ENC>(a, (int)blockIdx.x, (int)gridDim.x);
}

// All three passes of the 8-layer model in ONE launch, side by side on disjoint CUs.  Run one after the other they alternate
// between two bounds: a layer's contraction streams 256 B of dZ per sample (HBM: every block of the chip at once), the
// recomputed forward layers in front of it stream nothing (matrix core) -- pass 0-2 is mostly the first, pass 6-7 + output
// mostly the second, and measured back to back they took 1.04 + 1.37 + 1.78 ms per 4.7 M samples for 2.0 ms of HBM time.
// Side by side the chip's memory pipe serves fewer blocks at a time and is not idle while others recompute.
// Workgroups are dealt round-robin to the 8 XCDs, so block b sits in slot b / 8 of XCD b % 8: slots [0, s1) run layers 0-2,
// [s1, s2) layers 3-5, the rest 6-7 + output -- every XCD gets the same mix.  gridDim.x is a multiple of 8.
template <int KS0, bool ENC = false>
__global__ __launch_bounds__(kThreads, 1) void wgrad_recompute_all_kernel(LeanArgs a, int s1, int s2) {
  const int b = (int)blockIdx.x, slot = b >> 3, xcd = b & 7, slots = (int)gridDim.x >> 3;
  if (slot < s1) wgrad_recompute_pass<KS0, 0, 3, false, 8, ENC>(a, slot * 8 + xcd, s1 * 8);
  else if (slot < s2) wgrad_recompute_pass<KS0, 3, 6, false, 8, ENC>(a, (slot - s1) * 8 + xcd, (s2 - s1) * 8);
  else wgrad_recompute_pass<KS0, 6, 8, true, 8, ENC>(a, (slot - s2) * 8 + xcd, (slots - s2) * 8);
}

// ------------------------------------------------------------------------- segments that carry a loss gradient
// In NeRF training most samples lie behind the first surface, where the transmittance and with it dL/d(radiance) is exactly
// zero (configs[2] batch: 12 % of the samples, 30 % of the 64-sample waves carry a gradient).  The backward kernels do not
// have to visit the others: live_flags_kernel marks every 32-sample segment with a non-zero radiance gradient, live_compact_
// kernel (one block) writes their indices in ascending order and their count, and mlp_bwd_fused64_kernel /
// hashgrid_backward_kernel walk that list (a column tile of the fused kernel is exactly one segment).
__global__ __launch_bounds__(kThreads) void live_flags_kernel(const uint2* __restrict__ dout_half4, long S, DevCount dc, uint8_t* __restrict__ flags) {
  S = live_samples(dc, S);
  const long s = (long)blockIdx.x * kThreads + threadIdx.x;
  if ((long)blockIdx.x * kThreads >= S) return;
  bool nz = false;
  if (s < S) {
    const uint2 g = dout_half4[s];
    nz = ((g.x | g.y) & 0x7fff7fffu) != 0;            // four halves; -0.0 is zero
  }
  const unsigned long long b = __ballot(nz);
  const int lane = threadIdx.x & 63;
  if ((lane & 31) == 0 && s < S) flags[s >> 5] = ((b >> lane) & 0xffffffffull) != 0 ? 1 : 0;
}
constexpr int kCompactThreads = 1024;
// one block: every thread takes FOUR consecutive flags per round (one 4-byte load), so a round covers 4096 segments
__global__ __launch_bounds__(kCompactThreads) void live_compact_kernel(const uint8_t* __restrict__ flags, long S, DevCount dc,
                                                                       int* __restrict__ list, int* __restrict__ count) {
  __shared__ int wave_tot[kCompactThreads / 64];
  S = live_samples(dc, S);
  const int P = (int)(S >> 5);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int base = 0;                                       // every thread keeps the running total: no broadcast round
  for (int p0 = 0; p0 < P; p0 += 4 * kCompactThreads) {
    const int p = p0 + 4 * tid;
    unsigned f = 0;
    if (p + 3 < P) f = *reinterpret_cast<const unsigned*>(flags + p);          // flags is 16-byte aligned, p a multiple of 4
    else
      for (int k = 0; k < 4; ++k)
        if (p + k < P) f |= (unsigned)flags[p + k] << (8 * k);
    const int mine = (f & 1u) + ((f >> 8) & 1u) + ((f >> 16) & 1u) + ((f >> 24) & 1u);
    // exclusive prefix of `mine` inside the wave (DPP adds), then across the block's 16 waves through LDS
    int incl = mine;
    incl += dpp_i<0x111, 0xf>(incl);
    incl += dpp_i<0x112, 0xf>(incl);
    incl += dpp_i<0x114, 0xf>(incl);
    incl += dpp_i<0x118, 0xf>(incl);
    incl += dpp_i<0x142, 0xa>(incl);
    incl += dpp_i<0x143, 0xc>(incl);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int off = base, total = 0;
#pragma unroll
    for (int w = 0; w < kCompactThreads / 64; ++w) {
      const int t = wave_tot[w];
      if (w < wave) off += t;
      total += t;
    }
    off += incl - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((f >> (8 * k)) & 1u) list[off++] = p + k;
    base += total;
    __syncthreads();                                  // wave_tot is rewritten in the next round
  }
  if (tid == 0) *count = base;
}

// ------------------------------------------------------------------------- loss, optimizer
__global__ __launch_bounds__(kThreads) void l2_loss_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                           long n, float scale, float* __restrict__ values,
                                                           __half* __restrict__ grads, float* __restrict__ loss_sum) {
  __shared__ float red[kThreads / 64];
  float local = 0.0f;
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
    const float d = pred[i] - target[i];
    const float v = d * d / (float)n;
    const float g = scale * 2.0f * d / (float)n;
    if (values) values[i] = v;
    if (grads) grads[i] = __float2half(g);
    local += v;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0 && loss_sum) {
    float t = 0.0f;
    for (int w = 0; w < kThreads / 64; ++w) t += red[w];
    atomicAdd(loss_sum, t);
  }
}

// One Adam update (tcnn "Adam": no weight decay, bias correction folded into lr_eff on the host / lr_dev).
__device__ __forceinline__ void adam_one(float g, float& mi, float& vi, float& w, float lr_eff, float beta1, float beta2, float eps) {
  mi = beta1 * mi + (1.0f - beta1) * g;
  vi = beta2 * vi + (1.0f - beta2) * g * g;
  w = w - lr_eff * mi / (sqrtf(vi) + eps);
}
// HBM-bound (22-28 B per parameter): four parameters per thread, 16-byte accesses
// ZERO: the gradient is cleared as it is consumed (the next step accumulates into zeros without a separate fill pass)
template <bool HALF_GRADS, bool ZERO>
__global__ __launch_bounds__(kThreads) void adam_kernel(long n, float* __restrict__ master, __half* __restrict__ params,
                                                        void* __restrict__ grads_v, float* __restrict__ m,
                                                        float* __restrict__ v, float lr_eff, float beta1, float beta2,
                                                        float eps, float inv_loss_scale, const float* __restrict__ lr_dev) {
  if (lr_dev) lr_eff = *lr_dev;      // captured steps: the bias-corrected rate changes every replay, the graph does not
  float* gf = static_cast<float*>(grads_v);
  __half* gh = static_cast<__half*>(grads_v);
  const bool vec = (((uintptr_t)master | (uintptr_t)m | (uintptr_t)v | (uintptr_t)grads_v) & 15) == 0 && ((uintptr_t)params & 7) == 0;
  const long n4 = vec ? n / 4 : 0;
  for (long q = (long)blockIdx.x * kThreads + threadIdx.x; q < n4; q += (long)gridDim.x * kThreads) {
    float4 w4 = reinterpret_cast<float4*>(master)[q], m4 = reinterpret_cast<float4*>(m)[q], v4 = reinterpret_cast<float4*>(v)[q];
    float g[4];
    if (HALF_GRADS) {
      const half4v h = reinterpret_cast<const half4v*>(gh)[q];
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = (float)h[e] * inv_loss_scale;
      if (ZERO) reinterpret_cast<uint2*>(gh)[q] = make_uint2(0u, 0u);
    } else {
      const float4 f = reinterpret_cast<const float4*>(gf)[q];
      g[0] = f.x * inv_loss_scale; g[1] = f.y * inv_loss_scale; g[2] = f.z * inv_loss_scale; g[3] = f.w * inv_loss_scale;
      if (ZERO) reinterpret_cast<float4*>(gf)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    adam_one(g[0], m4.x, v4.x, w4.x, lr_eff, beta1, beta2, eps);
    adam_one(g[1], m4.y, v4.y, w4.y, lr_eff, beta1, beta2, eps);
    adam_one(g[2], m4.z, v4.z, w4.z, lr_eff, beta1, beta2, eps);
    adam_one(g[3], m4.w, v4.w, w4.w, lr_eff, beta1, beta2, eps);
    reinterpret_cast<float4*>(master)[q] = w4;
    reinterpret_cast<float4*>(m)[q] = m4;
    reinterpret_cast<float4*>(v)[q] = v4;
    const half4v o = {(_Float16)w4.x, (_Float16)w4.y, (_Float16)w4.z, (_Float16)w4.w};
    reinterpret_cast<half4v*>(params)[q] = o;
  }
  for (long i = 4 * n4 + (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
    const float g = (HALF_GRADS ? __half2float(gh[i]) : gf[i]) * inv_loss_scale;
    if (ZERO) { if (HALF_GRADS) gh[i] = __float2half(0.0f); else gf[i] = 0.0f; }
    float mi = m[i], vi = v[i], w = master[i];
    adam_one(g, mi, vi, w, lr_eff, beta1, beta2, eps);
    m[i] = mi;
    v[i] = vi;
    master[i] = w;
    params[i] = __float2half(w);
  }
}

// tiny-cuda-nn's Adam as it treats "non-matrix" parameters -- the hash table (optimizers/adam.h, adam_step): an entry whose
// gradient is exactly zero is SKIPPED (neither moment decays, the weight stays), and the bias correction uses the entry's OWN
// count of updates (param_steps), "since some parameters might see fewer steps than others".  A step's batch touches 0.2 .. 25 %
// of a hashed level, so this is also the cheap form: the gradient is read everywhere (2 or 4 B per parameter), the 14 B of
// state only where it is non-zero.  Four parameters per thread; ZERO clears the gradient as it is consumed.
template <bool HALF_GRADS, bool ZERO>
__global__ __launch_bounds__(kThreads) void adam_sparse_kernel(long n, float* __restrict__ master, __half* __restrict__ params,
                                                               void* __restrict__ grads_v, float* __restrict__ m, float* __restrict__ v,
                                                               unsigned* __restrict__ steps, float lr, float beta1, float beta2,
                                                               float eps, float inv_loss_scale, float log2_beta1, float log2_beta2) {
  float* gf = static_cast<float*>(grads_v);
  __half* gh = static_cast<__half*>(grads_v);
  auto one = [&](float g, float& mi, float& vi, float& w, unsigned& st) {
    if (g == 0.0f) return;
    st += 1u;
    const float t = (float)st;
    // beta^t = 2^(t log2 beta): one v_exp_f32 each (a libm powf here made the kernel slower than the dense one while the
    // gradient is still dense, early in training); relative error ~1e-6 of a factor that multiplies lr
    const float lr_eff = lr * sqrtf(1.0f - __builtin_amdgcn_exp2f(t * log2_beta2)) / (1.0f - __builtin_amdgcn_exp2f(t * log2_beta1));
    adam_one(g, mi, vi, w, lr_eff, beta1, beta2, eps);
  };
  const bool vec = (((uintptr_t)master | (uintptr_t)m | (uintptr_t)v | (uintptr_t)steps | (uintptr_t)grads_v) & 15) == 0 && ((uintptr_t)params & 7) == 0;
  const long n4 = vec ? n / 4 : 0;
  for (long q = (long)blockIdx.x * kThreads + threadIdx.x; q < n4; q += (long)gridDim.x * kThreads) {
    float g[4];
    if (HALF_GRADS) {
      const uint2 raw = reinterpret_cast<const uint2*>(gh)[q];
      if (((raw.x | raw.y) & 0x7fff7fffu) == 0u) continue;
      const half4v h = __builtin_bit_cast(half4v, raw);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = (float)h[e] * inv_loss_scale;
      if (ZERO) reinterpret_cast<uint2*>(gh)[q] = make_uint2(0u, 0u);
    } else {
      const float4 f = reinterpret_cast<const float4*>(gf)[q];
      if (f.x == 0.0f && f.y == 0.0f && f.z == 0.0f && f.w == 0.0f) continue;
      g[0] = f.x * inv_loss_scale; g[1] = f.y * inv_loss_scale; g[2] = f.z * inv_loss_scale; g[3] = f.w * inv_loss_scale;
      if (ZERO) reinterpret_cast<float4*>(gf)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 w4 = reinterpret_cast<float4*>(master)[q], m4 = reinterpret_cast<float4*>(m)[q], v4 = reinterpret_cast<float4*>(v)[q];
    uint4 s4 = reinterpret_cast<uint4*>(steps)[q];
    one(g[0], m4.x, v4.x, w4.x, s4.x);
    one(g[1], m4.y, v4.y, w4.y, s4.y);
    one(g[2], m4.z, v4.z, w4.z, s4.z);
    one(g[3], m4.w, v4.w, w4.w, s4.w);
    reinterpret_cast<float4*>(master)[q] = w4;
    reinterpret_cast<float4*>(m)[q] = m4;
    reinterpret_cast<float4*>(v)[q] = v4;
    reinterpret_cast<uint4*>(steps)[q] = s4;
    const half4v o = {(_Float16)w4.x, (_Float16)w4.y, (_Float16)w4.z, (_Float16)w4.w};
    reinterpret_cast<half4v*>(params)[q] = o;
  }
  for (long i = 4 * n4 + (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
    const float g = (HALF_GRADS ? __half2float(gh[i]) : gf[i]) * inv_loss_scale;
    if (g == 0.0f) continue;
    if (ZERO) { if (HALF_GRADS) gh[i] = __float2half(0.0f); else gf[i] = 0.0f; }
    float mi = m[i], vi = v[i], w = master[i];
    unsigned st = steps[i];
    one(g, mi, vi, w, st);
    m[i] = mi;
    v[i] = vi;
    master[i] = w;
    steps[i] = st;
    params[i] = __float2half(w);
  }
}

// fp32 <-> fp16 copies of a gradient block (8 elements per thread): the data-parallel exchange of the hashed levels' gradient
// travels in fp16 (tcnn keeps that gradient in fp16 to begin with), see rtx_nerf_amd/train.py
// deterministic mode: fold the fixed-point sums into the gradient buffers they shadow (accumulate semantics, one rounding per
// element) and clear them.  dst_h != NULL: elements from hashed_lo on live in the fp16 buffer (the hashed levels' table gradient).
__global__ __launch_bounds__(kThreads) void det_fold_kernel(long long* __restrict__ q, long n, float* __restrict__ dst, _Float16* __restrict__ dst_h,
                                                            long hashed_lo) {
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
    const long long v = q[i];
    if (v == 0) continue;
    q[i] = 0;
    const float g = (float)((double)v * (1.0 / (double)kDetScale));
    if (dst_h && i >= hashed_lo) dst_h[i - hashed_lo] = (_Float16)((float)dst_h[i - hashed_lo] + g);
    else dst[i] += g;
  }
}

__global__ __launch_bounds__(kThreads) void f32_to_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long n) {
  const long i = ((long)blockIdx.x * kThreads + threadIdx.x) * 8;
  if (i + 8 <= n && ((uintptr_t)(src + i) & 15) == 0 && ((uintptr_t)(dst + i) & 15) == 0) {
    const float4 a = *reinterpret_cast<const float4*>(src + i), b = *reinterpret_cast<const float4*>(src + i + 4);
    half8 o = {(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w, (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
    *reinterpret_cast<half8*>(dst + i) = o;
  } else {
    for (long k = i; k < n && k < i + 8; ++k) dst[k] = (_Float16)src[k];
  }
}
__global__ __launch_bounds__(kThreads) void f16_to_f32_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, long n) {
  const long i = ((long)blockIdx.x * kThreads + threadIdx.x) * 8;
  if (i + 8 <= n && ((uintptr_t)(src + i) & 15) == 0 && ((uintptr_t)(dst + i) & 15) == 0) {
    const half8 v = *reinterpret_cast<const half8*>(src + i);
    *reinterpret_cast<float4*>(dst + i) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    *reinterpret_cast<float4*>(dst + i + 4) = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
  } else {
    for (long k = i; k < n && k < i + 8; ++k) dst[k] = (float)src[k];
  }
}

int check_train(const rtxn_mlp* m, const char* who) {
  if (!m) { rtxn::set_error("%s: NULL model", who); return RTXN_ERR_INVALID; }
  if (m->cfg.n_neurons != 64 && m->cfg.n_neurons != 128) {
    // the 256-wide model (config 5) has an inference kernel only; running the 128-wide training kernels over its packing
    // would silently produce wrong activations and gradients
    rtxn::set_error("%s: no training kernels for n_neurons = %d (built: 64, 128)", who, m->cfg.n_neurons);
    return RTXN_ERR_UNSUPPORTED;
  }
  if (!m->packed_train || !m->packed_t) { rtxn::set_error("%s: rtxn_mlp_set_params has not been called", who); return RTXN_ERR_INVALID; }
  return RTXN_OK;
}

// deterministic mode: the fixed-point shadows the caller registered (rtxn_set_deterministic_workspace), or NULL
std::atomic<long long*> g_det_mlp{nullptr}, g_det_table{nullptr};

DetCtx det_ctx(const float* base, std::atomic<long long*>& slot) {
  long long* q = slot.load(std::memory_order_relaxed);
  return DetCtx{q ? base : nullptr, q};
}

int det_fold(long long* q, long n, float* dst, void* dst_half, long hashed_lo, hipStream_t s) {
  if (!q || n <= 0) return RTXN_OK;
  const long blocks = (n + kThreads - 1) / kThreads;
  det_fold_kernel<<<(unsigned)(blocks > 4096 ? 4096 : blocks), kThreads, 0, s>>>(q, n, dst, static_cast<_Float16*>(dst_half), hashed_lo);
  RTXN_LAUNCH_CHECK("det_fold_kernel");
  return RTXN_OK;
}

}  // namespace

// ============================================================================ C ABI
extern "C" size_t rtxn_deterministic_workspace_bytes(long n_params) { return n_params < 0 ? 0 : (size_t)n_params * sizeof(long long); }

extern "C" int rtxn_set_deterministic_workspace(void* mlp_shadow, void* table_shadow) {
  RTXN_REQUIRE((((uintptr_t)mlp_shadow | (uintptr_t)table_shadow) & 7) == 0, "rtxn_set_deterministic_workspace: shadows must be 8-byte aligned");
  g_det_mlp.store(static_cast<long long*>(mlp_shadow), std::memory_order_relaxed);
  g_det_table.store(static_cast<long long*>(table_shadow), std::memory_order_relaxed);
  return RTXN_OK;
}

extern "C" long rtxn_padded_samples(long n_samples) { return n_samples < 0 ? -1 : padded(n_samples); }

extern "C" size_t rtxn_mlp_train_workspace_bytes(const rtxn_mlp* m, long n_samples) {
  if (!m || n_samples < 0) return 0;
  const long Sp = padded(n_samples), W = m->cfg.n_neurons, L = m->cfg.n_hidden_layers;
  // acts | dz | dzL | sign masks (16 B per sample and layer) | one live flag per 256-sample tile
  return (size_t)((2 * L * W + 16 + 8 * L) * Sp) * sizeof(_Float16) + (size_t)((Sp / kTile + 15) / 16 * 16);
}

// In every *_impl below: dc.total_segments == NULL: n_samples is the batch's; otherwise n_samples is the CAPACITY (grids, row
// stride) and the kernels take the live count from the device.
static int encode_frequency_impl(const rtxn_mlp* m, const SampleSrc& src, void* encT, float* t_vals, float t_scale, long n_samples,
                                 DevCount dc, rtxn_stream_t stream) {
  const long Sp = padded(n_samples);
  encode_freq_kernel<<<(unsigned)(Sp / kThreads), kThreads, 0, rtxn::as_stream(stream)>>>(
      src, static_cast<_Float16*>(encT), t_vals, t_scale, n_samples, Sp, m->cfg.n_pos_dims, m->cfg.n_pos_freqs, m->cfg.n_dir_dims,
      m->cfg.n_dir_freqs, m->enc_padded, dc);
  RTXN_LAUNCH_CHECK("encode_freq_kernel");
  return RTXN_OK;
}

extern "C" int rtxn_encode_frequency(const rtxn_mlp* m, const float* input, void* encT, long n_samples,
                                     rtxn_stream_t stream) {
  RTXN_REQUIRE(m && m->cfg.encoding == RTXN_ENC_FREQUENCY, "rtxn_encode_frequency: model has no frequency encoding");
  RTXN_REQUIRE(n_samples >= 0, "rtxn_encode_frequency: n_samples = %ld < 0", n_samples);
  RTXN_DEVICE_OR_FAIL();
  if (n_samples == 0) return RTXN_OK;
  RTXN_REQUIRE(input && encT, "rtxn_encode_frequency: NULL buffer");
  const SampleSrc src{input, nullptr, nullptr, nullptr, 0};
  return encode_frequency_impl(m, src, encT, nullptr, 1.0f, n_samples, DevCount{nullptr, 0}, stream);
}

static int check_segments(const char* who, const float* start_points, const float* end_points, const float* seg_view,
                          long n_segments, int sample_type) {
  RTXN_REQUIRE(n_segments >= 0 && n_segments <= kMaxTrainSamples / 32, "%s: n_segments = %ld", who, n_segments);
  RTXN_REQUIRE(sample_type == RTXN_SAMPLING_REGULAR || sample_type == RTXN_SAMPLING_MIDPOINT_WORLD,
               "%s: sample_type %d (the deterministic modes only: REGULAR, MIDPOINT_WORLD)", who, sample_type);
  RTXN_REQUIRE(n_segments == 0 || (start_points && end_points && seg_view), "%s: NULL segment buffer", who);
  return RTXN_OK;
}

extern "C" int rtxn_encode_frequency_segments(const rtxn_mlp* m, const float* start_points, const float* end_points,
                                              const float* seg_view, long n_segments, int sample_type, float t_scale,
                                              void* encT, float* t_vals, rtxn_stream_t stream) {
  RTXN_REQUIRE(m && m->cfg.encoding == RTXN_ENC_FREQUENCY, "rtxn_encode_frequency_segments: model has no frequency encoding");
  RTXN_REQUIRE(m->cfg.n_pos_dims == 3 && m->cfg.n_dir_dims == 2, "rtxn_encode_frequency_segments: needs 3 + 2 input dimensions");
  int rc = check_segments("rtxn_encode_frequency_segments", start_points, end_points, seg_view, n_segments, sample_type);
  if (rc != RTXN_OK) return rc;
  RTXN_DEVICE_OR_FAIL();
  if (n_segments == 0) return RTXN_OK;
  RTXN_REQUIRE(encT, "rtxn_encode_frequency_segments: NULL buffer");
  const SampleSrc src{nullptr, start_points, end_points, seg_view, sample_type == RTXN_SAMPLING_MIDPOINT_WORLD};
  return encode_frequency_impl(m, src, encT, t_vals, t_scale, n_segments * 32, DevCount{nullptr, 0}, stream);
}

// live-segment workspace (rtxn_live_segments): [int count | 12 B pad | int list[capacity] | uint8 flags[capacity]]
static size_t live_ws_bytes(long capacity) { return (size_t)(16 + 4 * capacity + ((capacity + 15) / 16) * 16); }
static const int* live_count_of(const void* ws) { return static_cast<const int*>(ws); }
static const int* live_list_of(const void* ws) { return reinterpret_cast<const int*>(static_cast<const uint8_t*>(ws) + 16); }

// hipFuncSetAttribute once per (device, kernel): not repeated in front of every launch (and never inside a stream capture
// after the first, un-captured, call)
static hipError_t set_lds_once(const void* fn, int bytes) {
  struct Seen { int dev; const void* fn; int bytes; };
  static std::mutex mu;
  static std::vector<Seen> seen;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  for (Seen& q : seen)
    if (q.dev == dev && q.fn == fn) {
      if (q.bytes >= bytes) return hipSuccess;
      e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      if (e == hipSuccess) q.bytes = bytes;
      return e;
    }
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) seen.push_back(Seen{dev, fn, bytes});
  return e;
}

// workspace == NULL: outputs only (the forward half of the recompute path)
// lean: `workspace` is the lean workspace (rtxn_mlp_train_lean_workspace_bytes): outputs + sign masks, no activations
// src != NULL (lean only): the encoder fused into the kernel (the reference's Composite-Frequency(3 x 10, 2 x 12)); encT is not read
static int train_forward_impl(const rtxn_mlp* m, const void* encT, long n_samples, void* workspace, void* output_half,
                              float* radiance, DevCount dc, rtxn_stream_t stream, const int* live_list = nullptr,
                              const int* live_count = nullptr, bool lean = false, const SampleSrc* src = nullptr, float* t_vals = nullptr,
                              float t_scale = 1.0f) {
  const int W = m->cfg.n_neurons;
  const long Sp = padded(n_samples);
  // outputs only, 64 wide: the all-asm 16x16x32 kernel with the weights resident in LDS (hashmlp.hip) -- the same layer stack
  // the fused hash-grid inference kernel runs.  RTXN_TRAIN_FWD16=0: the 32x32x16 kernel below (A/B).
  if (!workspace && rtxn::enc_forward16_supported(m)) {
    static const bool use16 = !(getenv("RTXN_TRAIN_FWD16") && atoi(getenv("RTXN_TRAIN_FWD16")) == 0);
    if (use16) return rtxn::launch_enc_forward16(m, encT, n_samples, Sp, dc.total_segments, dc.capacity, output_half, radiance, rtxn::as_stream(stream));
  }
  TrainArgs a;
  memset(&a, 0, sizeof(a));
  a.packed = static_cast<const uint8_t*>(m->packed_train);
  a.n_hidden = m->cfg.n_hidden_layers;
  a.out_act = m->cfg.output_activation;
  a.E = m->enc_padded;
  a.S = n_samples;
  a.Sp = Sp;
  a.dc = dc;
  a.encT = static_cast<const _Float16*>(encT);
  if (workspace && lean) {
    a.masks = reinterpret_cast<unsigned long long*>(static_cast<_Float16*>(workspace) + ((long)m->cfg.n_hidden_layers * W + 16) * Sp);
  } else if (workspace) {
    a.acts = static_cast<_Float16*>(workspace);
    a.masks = reinterpret_cast<unsigned long long*>(static_cast<_Float16*>(workspace) + (2L * m->cfg.n_hidden_layers * W + 16) * Sp);
  }
  a.out_half = static_cast<_Float16*>(output_half);
  a.radiance = reinterpret_cast<float4*>(radiance);
  a.live_list = live_list;
  a.live_count = live_count;
  if (src) {
    a.src = *src;
    a.t_vals = t_vals;
    a.t_scale = t_scale;
  }
  const int RT = W / 32, KS = W / 16, KS0 = a.E / 16;
  // 128 wide, RTXN_TRAIN_FWD_WAVES=8: 8-wave blocks of 512 samples with double-buffered weights.  Built on the guess that the kernel
  // is paced by its weight stream; the same-box A/B says it is not (outputs only 1.050 against 1.068 ms per 4.7 M samples, with the
  // sign masks 1.43 against 1.35: the double buffer's vmcnt(0) also waits for the mask stores), so the 4-wave form stays the default.
  static const bool waves8 = getenv("RTXN_TRAIN_FWD_WAVES") && atoi(getenv("RTXN_TRAIN_FWD_WAVES")) == 8;
  const int NW = W == 128 && waves8 && !src ? 8 : 4;
  // (RTXN_TRAIN_FWD_ALONE=1, diagnostic: 40 KiB of LDS nobody uses, so that only ONE block fits a CU -- what a block's phases cost
  // without a partner on its SIMDs: profiles/r04/fwd_stamps.txt)
  static const bool alone = getenv("RTXN_TRAIN_FWD_ALONE") && atoi(getenv("RTXN_TRAIN_FWD_ALONE")) == 1;
  const size_t lds = (size_t)(KS0 > KS ? KS0 : KS) * RT * 1024 * (NW == 8 ? 2 : 1) + NW * kEncScratch + (alone ? 40 * 1024 : 0);
  hipStream_t s = rtxn::as_stream(stream);
  const dim3 grid((unsigned)((Sp + 64 * NW - 1) / (64 * NW))), block(64 * NW);
#define RTXN_FWD_LAUNCH(WW, SAVE, NWV)                                                                        \
  do {                                                                                                        \
    RTXN_HIP(set_lds_once(reinterpret_cast<const void*>(mlp_train_fwd_kernel<WW, SAVE, NWV>), (int)lds));      \
    hipLaunchKernelGGL((mlp_train_fwd_kernel<WW, SAVE, NWV>), grid, block, lds, s, a);                         \
  } while (0)
  if (src) {
    RTXN_HIP(set_lds_once(reinterpret_cast<const void*>(mlp_train_fwd_kernel<128, kSaveMasks, 4, 1>), (int)lds));
    hipLaunchKernelGGL((mlp_train_fwd_kernel<128, kSaveMasks, 4, 1>), grid, block, lds, s, a);
  } else if (W == 64) { if (workspace) RTXN_FWD_LAUNCH(64, kSaveAll, 4); else RTXN_FWD_LAUNCH(64, kSaveNone, 4); }
  else if (NW == 8) {
    if (lean) RTXN_FWD_LAUNCH(128, kSaveMasks, 8);
    else if (workspace) RTXN_FWD_LAUNCH(128, kSaveAll, 8);
    else RTXN_FWD_LAUNCH(128, kSaveNone, 8);
  } else if (lean) RTXN_FWD_LAUNCH(128, kSaveMasks, 4);
  else         { if (workspace) RTXN_FWD_LAUNCH(128, kSaveAll, 4); else RTXN_FWD_LAUNCH(128, kSaveNone, 4); }
#undef RTXN_FWD_LAUNCH
  RTXN_LAUNCH_CHECK(workspace ? "mlp_train_fwd_kernel" : "mlp_train_fwd_kernel<outputs only>");
  return RTXN_OK;
}

extern "C" int rtxn_mlp_train_forward(const rtxn_mlp* m, const void* encT, long n_samples, void* workspace,
                                      void* output_half, float* radiance, rtxn_stream_t stream) {
  int rc = check_train(m, "rtxn_mlp_train_forward");
  if (rc != RTXN_OK) return rc;
  RTXN_REQUIRE(n_samples >= 0 && n_samples <= kMaxTrainSamples, "rtxn_mlp_train_forward: n_samples = %ld out of [0, %ld]", n_samples, kMaxTrainSamples);
  RTXN_DEVICE_OR_FAIL();
  if (n_samples == 0) return RTXN_OK;
  RTXN_REQUIRE(encT && workspace && output_half, "rtxn_mlp_train_forward: NULL buffer");
  return train_forward_impl(m, encT, n_samples, workspace, output_half, radiance, DevCount{nullptr, 0}, stream);
}

extern "C" int rtxn_mlp_train_forward_live(const rtxn_mlp* m, const void* encT, long n_samples, void* workspace, const void* live_ws,
                                           rtxn_stream_t stream) {
  int rc = check_train(m, "rtxn_mlp_train_forward_live");
  if (rc != RTXN_OK) return rc;
  RTXN_REQUIRE(n_samples >= 0 && n_samples <= kMaxTrainSamples && n_samples % 32 == 0,
               "rtxn_mlp_train_forward_live: n_samples = %ld must be whole segments in [0, %ld]", n_samples, kMaxTrainSamples);
  RTXN_DEVICE_OR_FAIL();
  if (n_samples == 0) return RTXN_OK;
  RTXN_REQUIRE(encT && workspace && live_ws, "rtxn_mlp_train_forward_live: NULL buffer");
  return train_forward_impl(m, encT, n_samples, workspace, nullptr, nullptr, DevCount{nullptr, 0}, stream, live_list_of(live_ws),
                            live_count_of(live_ws));
}

static int train_backward_impl(const rtxn_mlp* m, const void* encT, const void* output_half, const void* dout_half4,
                               long n_samples, void* workspace, float* dparams, void* dencT, DevCount dc, rtxn_stream_t stream,
                               const int* live_list = nullptr, const int* live_count = nullptr) {
  const int W = m->cfg.n_neurons, L = m->cfg.n_hidden_layers, E = m->enc_padded;
  const long Sp = padded(n_samples);
  _Float16* ws = static_cast<_Float16*>(workspace);
  TrainArgs a;
  memset(&a, 0, sizeof(a));
  a.packed = static_cast<const uint8_t*>(m->packed_t);
  a.n_hidden = L;
  a.out_act = m->cfg.output_activation;
  a.E = E;
  a.S = n_samples;
  a.Sp = Sp;
  a.dc = dc;
  a.encT = static_cast<const _Float16*>(encT);
  a.acts = ws;
  a.dz = ws + (long)L * W * Sp;
  a.dzL = ws + 2L * L * W * Sp;
  a.masks = reinterpret_cast<unsigned long long*>(ws + (2L * L * W + 16) * Sp);
  a.live_tiles = reinterpret_cast<uint8_t*>(ws + (2L * L * W + 16 + 8L * L) * Sp);
  a.live_list = live_list;
  a.live_count = live_count;
  a.out_half = const_cast<_Float16*>(static_cast<const _Float16*>(output_half));
  a.dout = static_cast<const _Float16*>(dout_half4);
  a.dencT = static_cast<_Float16*>(dencT);
  const int RT = W / 32, KS = W / 16, RTE = (E + 31) / 32;
  const size_t lds = (size_t)(RTE > RT ? RTE : RT) * KS * 1024;
  hipStream_t s = rtxn::as_stream(stream);
  if (W == 64) {
    RTXN_HIP(set_lds_once(reinterpret_cast<const void*>(mlp_bwd_kernel<64>), (int)lds));
    hipLaunchKernelGGL(mlp_bwd_kernel<64>, dim3((unsigned)(Sp / kTile)), dim3(kThreads), lds, s, a);
  } else {
    RTXN_HIP(set_lds_once(reinterpret_cast<const void*>(mlp_bwd_kernel<128>), (int)lds));
    hipLaunchKernelGGL(mlp_bwd_kernel<128>, dim3((unsigned)(Sp / kTile)), dim3(kThreads), lds, s, a);
  }
  RTXN_LAUNCH_CHECK("mlp_bwd_kernel");
  // weight gradients of all layers in one launch: dW_l += dZ_l X_l^T, X_0 = enc, X_l = acts[l-1], X_L = acts[L-1]
  RTXN_REQUIRE(L + 1 <= 17, "rtxn_mlp_train_backward: %d layers exceed the weight-gradient launch table", L + 1);
  WgradArgs wa;
  wa.det = det_ctx(dparams, g_det_mlp);
  wa.Sp = Sp;
  wa.dc = dc;
  wa.live_tiles = a.live_tiles;
  wa.live_list = live_list;
  wa.live_count = live_count;
  wa.chunk = 1024;
  const unsigned kblocks = (unsigned)((Sp + 4 * wa.chunk - 1) / (4 * wa.chunk));
  long poff = 0;
  int max_tiles = 0;
  for (int l = 0; l <= L; ++l) {
    const int M = l == L ? 16 : W, N = l == 0 ? E : W;
    WgradLayer& wl = wa.layer[l];
    wl.dZ = l == L ? a.dzL : a.dz + (long)l * W * Sp;
    wl.X = l == 0 ? a.encT : a.acts + (long)(l - 1) * W * Sp;
    wl.dW = dparams + poff;
    wl.M = M, wl.N = N;
    wl.tiles_n = (N + 63) / 64;
    wl.n_tiles = ((M + 63) / 64) * wl.tiles_n;
    if (wl.n_tiles > max_tiles) max_tiles = wl.n_tiles;
    poff += (long)M * N;
  }
  int multi = 0, single = 0;
  for (int l = 0; l <= L; ++l) (wa.layer[l].n_tiles >= 2 ? multi : single)++;
  wa.lds_path = multi > 0 && W <= 128 && E <= 128;
  if (single > 0 || !wa.lds_path) {
    wgrad_kernel<<<dim3((unsigned)(wa.lds_path ? 1 : max_tiles), kblocks, (unsigned)(L + 1)), kThreads, 0, s>>>(wa);
    RTXN_LAUNCH_CHECK("wgrad_kernel");
  }
  if (wa.lds_path) {
    WgradArgs wl = wa;
    wl.chunk = wgrad_chunk(Sp, L + 1, live_list != nullptr);
    RTXN_HIP(set_lds_once(reinterpret_cast<const void*>(wgrad_lds_kernel), kWgStages * kWgStage));
    wgrad_lds_kernel<<<dim3((unsigned)((Sp + wl.chunk - 1) / wl.chunk), (unsigned)(L + 1)), kThreads, kWgStages * kWgStage, s>>>(wl);
    RTXN_LAUNCH_CHECK("wgrad_lds_kernel");
  }
  return det_fold(wa.det.q, m->n_params, dparams, nullptr, 0, s);
}

extern "C" int rtxn_mlp_train_backward(const rtxn_mlp* m, const void* encT, const void* output_half,
                                       const void* dout_half4, long n_samples, void* workspace, float* dparams,
                                       void* dencT, rtxn_stream_t stream) {
  int rc = check_train(m, "rtxn_mlp_train_backward");
  if (rc != RTXN_OK) return rc;
  RTXN_REQUIRE(n_samples >= 0 && n_samples <= kMaxTrainSamples, "rtxn_mlp_train_backward: n_samples = %ld out of [0, %ld]", n_samples, kMaxTrainSamples);
  RTXN_DEVICE_OR_FAIL();
  if (n_samples == 0) return RTXN_OK;
  RTXN_REQUIRE(encT && output_half && dout_half4 && workspace && dparams, "rtxn_mlp_train_backward: NULL buffer");
  return train_backward_impl(m, encT, output_half, dout_half4, n_samples, workspace, dparams, dencT, DevCount{nullptr, 0}, stream);
}

// ---- lean path (128-wide models): forward with sign masks only, dgrad chain, weight gradient with recomputed activations ----
// Workspace (halfs): dz [L][128][Sp] | dzL [16][Sp] | sign masks [L][Sp] x 16 B | one live flag per 256-sample tile.
extern "C" int rtxn_mlp_train_lean_supported(const rtxn_mlp* m) {
  if (!m) return 0;
  return m->cfg.n_neurons == 128 && m->cfg.n_hidden_layers == 8 && m->enc_padded == 112;
}

extern "C" size_t rtxn_mlp_train_lean_workspace_bytes(const rtxn_mlp* m, long n_samples) {
  if (!rtxn_mlp_train_lean_supported(m) || n_samples < 0) return 0;
  const long Sp = padded(n_samples), W = m->cfg.n_neurons, L = m->cfg.n_hidden_layers;
  return (size_t)((L * W + 16 + 8 * L) * Sp) * sizeof(_Float16) + (size_t)((Sp / kTile + 15) / 16 * 16);
}

// src != NULL: the weight gradient recomputes the ENCODING too (the reference's Composite-Frequency model); encT is not read
static int train_backward_lean_impl(const rtxn_mlp* m, const void* encT, const void* output_half, const void* dout_half4,
                                    long n_samples, void* workspace, float* dparams, DevCount dc, rtxn_stream_t stream,
                                    const int* live_list = nullptr, const int* live_count = nullptr, const SampleSrc* src = nullptr) {
  const int W = 128, L = m->cfg.n_hidden_layers, E = m->enc_padded;
  const long Sp = padded(n_samples);
  _Float16* ws = static_cast<_Float16*>(workspace);
  TrainArgs a;
  memset(&a, 0, sizeof(a));
  a.packed = static_cast<const uint8_t*>(m->packed_t);
  a.n_hidden = L;
  a.out_act = m->cfg.output_activation;
  a.E = E;
  a.S = n_samples;
  a.Sp = Sp;
  a.dc = dc;
  a.encT = static_cast<const _Float16*>(encT);
  a.dz = ws;
  a.dzL = ws + (long)L * W * Sp;
  a.masks = reinterpret_cast<unsigned long long*>(ws + ((long)L * W + 16) * Sp);
  a.live_tiles = reinterpret_cast<uint8_t*>(ws + ((long)L * W + 16 + 8L * L) * Sp);
  a.skip_last_dz = src ? 1 : 0;     // the folded weight-gradient kernel forms the last hidden layer's dZ itself
  a.live_list = live_list;
  a.live_count = live_count;
  a.out_half = const_cast<_Float16*>(static_cast<const _Float16*>(output_half));
  a.dout = static_cast<const _Float16*>(dout_half4);
  const int RT = W / 32, KS = W / 16;
  const size_t lds = (size_t)RT * KS * 1024;
  hipStream_t s = rtxn::as_stream(stream);
  RTXN_HIP(set_lds_once(reinterpret_cast<const void*>(mlp_bwd_kernel<128>), (int)lds));
  hipLaunchKernelGGL(mlp_bwd_kernel<128>, dim3((unsigned)(Sp / kTile)), dim3(kThreads), lds, s, a);
  RTXN_LAUNCH_CHECK("mlp_bwd_kernel");
  LeanArgs la;
  memset(&la, 0, sizeof(la));
  la.packed_fwd = static_cast<const uint8_t*>(m->packed_train);
  la.out_act = a.out_act;
  la.S = n_samples;
  la.Sp = Sp;
  la.dc = dc;
  la.n_tiles = (int)(Sp / kTile);
  la.encT = a.encT;
  la.dz = a.dz;
  la.dzL = a.dzL;
  la.dparams = dparams;
  la.live_tiles = a.live_tiles;
  la.live_list = live_list;
  la.live_count = live_count;
  la.det = det_ctx(dparams, g_det_mlp);
  if (src) la.src = *src;
  la.packed_bwd = static_cast<const uint8_t*>(m->packed_t);
  int dev = 0, n_cu = 0;
  RTXN_HIP(hipGetDevice(&dev));
  RTXN_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  if (n_cu <= 0) n_cu = 256;
  const int grid = la.n_tiles < n_cu ? la.n_tiles : n_cu;   // persistent: one block per CU (registers and LDS)
  typedef void (*lean_fn)(LeanArgs);
  // Three passes: layers 0-2 | 3-5 | 6-7 + output -- 192 accumulator registers per wave, 1.9 forward passes of recompute.  (Two
  // passes, 0-3 | 4-7 + output: 1.4 passes of recompute, need all 256 AGPRs for the hidden layers' accumulators alone; tried in
  // round 4 -- hipcc spilled accumulator tiles to scratch around the tile loop's back edge -- and gone since the output layer's
  // gradient lives in AGPRs too.)
  constexpr int n_pass = 3;
  static const lean_fn pass_enc[2][3] = {{wgrad_recompute_kernel<7, 0, 3, false, 8>, wgrad_recompute_kernel<7, 3, 6, false, 8>,
                                          wgrad_recompute_kernel<7, 6, 8, true, 8>},
                                         {wgrad_recompute_kernel<7, 0, 3, false, 8, true>, wgrad_recompute_kernel<7, 3, 6, false, 8, true>,
                                          wgrad_recompute_kernel<7, 6, 8, true, 8, true>}};
  const lean_fn* pass = pass_enc[src ? 1 : 0];
  // large batches: the three passes side by side in one launch (wgrad_recompute_all_kernel); the CU split follows the passes'
  // measured cost (RTXN_LEAN_SPLIT="s1,s2" of 32 slots per XCD for experiments; RTXN_LEAN_SPLIT=0: one launch per pass)
  int split[2] = {src ? 8 : 9, src ? 19 : 20};   // 9:11:12 reading encT, 8:11:13 with the encoder folded in (tools/probe/lean_split.py, same-process sweeps)
  if (const char* e = getenv("RTXN_LEAN_SPLIT")) {
    if (sscanf(e, "%d,%d", &split[0], &split[1]) != 2) split[0] = split[1] = 0;
  }
  const int slots = n_cu / 8;
  if (split[0] > 0 && split[0] < split[1] && split[1] < slots && la.n_tiles >= 4 * n_cu) {
    if (src) {
      RTXN_HIP(set_lds_once(reinterpret_cast<const void*>(wgrad_recompute_all_kernel<7, true>), kLnLdsLaunch));
      hipLaunchKernelGGL((wgrad_recompute_all_kernel<7, true>), dim3((unsigned)(slots * 8)), dim3(kThreads), kLnLdsLaunch, s, la, split[0] * slots / 32, split[1] * slots / 32);
    } else {
      RTXN_HIP(set_lds_once(reinterpret_cast<const void*>(wgrad_recompute_all_kernel<7>), kLnLdsLaunch));
      hipLaunchKernelGGL(wgrad_recompute_all_kernel<7>, dim3((unsigned)(slots * 8)), dim3(kThreads), kLnLdsLaunch, s, la, split[0] * slots / 32, split[1] * slots / 32);
    }
    RTXN_LAUNCH_CHECK("wgrad_recompute_all_kernel");
    return det_fold(la.det.q, m->n_params, dparams, nullptr, 0, s);
  }
  for (int i = 0; i < n_pass; ++i) {
    RTXN_HIP(set_lds_once(reinterpret_cast<const void*>(pass[i]), kLnLdsLaunch));
    hipLaunchKernelGGL(pass[i], dim3((unsigned)grid), dim3(kThreads), kLnLdsLaunch, s, la);
    RTXN_LAUNCH_CHECK("wgrad_recompute_kernel");
  }
  return det_fold(la.det.q, m->n_params, dparams, nullptr, 0, s);
}

static int check_lean(const rtxn_mlp* m, const char* who, long n_samples, bool whole_segments) {
  int rc = check_train(m, who);
  if (rc != RTXN_OK) return rc;
  if (!rtxn_mlp_train_lean_supported(m)) {
    rtxn::set_error("%s: the lean path is built for the reference's model (128 wide, 8 hidden layers, 112 encoded features); this model: "
                    "%d wide, %d layers, %d features -- use rtxn_mlp_train_forward + rtxn_mlp_train_backward", who, m->cfg.n_neurons,
                    m->cfg.n_hidden_layers, m->enc_padded);
    return RTXN_ERR_UNSUPPORTED;
  }
  RTXN_REQUIRE(n_samples >= 0 && n_samples <= kMaxTrainSamples && (!whole_segments || n_samples % 32 == 0),
               "%s: n_samples = %ld out of [0, %ld]%s", who, n_samples, kMaxTrainSamples, whole_segments ? " or not whole segments" : "");
  return RTXN_OK;
}

extern "C" int rtxn_mlp_train_forward_lean(const rtxn_mlp* m, const void* encT, long n_samples, void* workspace_lean,
                                           void* output_half, float* radiance, rtxn_stream_t stream) {
  int rc = check_lean(m, "rtxn_mlp_train_forward_lean", n_samples, false);
  if (rc != RTXN_OK) return rc;
  RTXN_DEVICE_OR_FAIL();
  if (n_samples == 0) return RTXN_OK;
  RTXN_REQUIRE(encT && workspace_lean && output_half, "rtxn_mlp_train_forward_lean: NULL buffer");
  return train_forward_impl(m, encT, n_samples, workspace_lean, output_half, radiance, DevCount{nullptr, 0}, stream, nullptr, nullptr, true);
}

extern "C" int rtxn_mlp_train_forward_lean_fused_supported(const rtxn_mlp* m) {
  return m && rtxn_mlp_train_lean_supported(m) && m->cfg.encoding == RTXN_ENC_FREQUENCY && m->cfg.n_pos_dims == 3 && m->cfg.n_pos_freqs == 10 &&
         m->cfg.n_dir_dims == 2 && m->cfg.n_dir_freqs == 12;
}

extern "C" int rtxn_mlp_train_forward_lean_segments(const rtxn_mlp* m, const float* start_points, const float* end_points,
                                                    const float* seg_view, long n_segments, int sample_type, float t_scale, float* t_vals,
                                                    void* workspace_lean, void* output_half, float* radiance, rtxn_stream_t stream) {
  int rc = check_lean(m, "rtxn_mlp_train_forward_lean_segments", n_segments * 32, true);
  if (rc != RTXN_OK) return rc;
  if (!rtxn_mlp_train_forward_lean_fused_supported(m)) {
    rtxn::set_error("rtxn_mlp_train_forward_lean_segments: the fused encoder is the reference's Composite-Frequency(3 x 10, 2 x 12); this model: "
                    "%d x %d, %d x %d -- use rtxn_encode_frequency_segments + rtxn_mlp_train_forward_lean", m->cfg.n_pos_dims, m->cfg.n_pos_freqs,
                    m->cfg.n_dir_dims, m->cfg.n_dir_freqs);
    return RTXN_ERR_UNSUPPORTED;
  }
  rc = check_segments("rtxn_mlp_train_forward_lean_segments", start_points, end_points, seg_view, n_segments, sample_type);
  if (rc != RTXN_OK) return rc;
  RTXN_DEVICE_OR_FAIL();
  if (n_segments == 0) return RTXN_OK;
  RTXN_REQUIRE(workspace_lean && output_half, "rtxn_mlp_train_forward_lean_segments: NULL buffer");
  const SampleSrc src{nullptr, start_points, end_points, seg_view, sample_type == RTXN_SAMPLING_MIDPOINT_WORLD};
  return train_forward_impl(m, nullptr, n_segments * 32, workspace_lean, output_half, radiance, DevCount{nullptr, 0}, stream, nullptr, nullptr, true, &src,
                            t_vals, t_scale);
}

#ifdef RTXN_LN_STAMPS
// diagnostic builds only: the stamps of the last lean weight-gradient launch (3 passes x 4 waves x 2 tiles x 92 slots)
extern "C" int rtxn_debug_read_lean_stamps(unsigned* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ln_stamps), sizeof(unsigned) * 3 * 4 * kLnStampTiles * kLnStampSlots) == hipSuccess ? 0 : 1;
}
extern "C" int rtxn_debug_read_lean_clock(unsigned long long* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ln_clock), sizeof(unsigned long long) * 3 * 8) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int rtxn_mlp_train_backward_lean(const rtxn_mlp* m, const void* encT, const void* output_half, const void* dout_half4,
                                            long n_samples, void* workspace_lean, const void* live_ws, float* dparams,
                                            rtxn_stream_t stream) {
  int rc = check_lean(m, "rtxn_mlp_train_backward_lean", n_samples, live_ws != nullptr);
  if (rc != RTXN_OK) return rc;
  RTXN_DEVICE_OR_FAIL();
  if (n_samples == 0) return RTXN_OK;
  RTXN_REQUIRE(encT && output_half && dout_half4 && workspace_lean && dparams, "rtxn_mlp_train_backward_lean: NULL buffer");
  return train_backward_lean_impl(m, encT, output_half, dout_half4, n_samples, workspace_lean, dparams, DevCount{nullptr, 0}, stream,
                                  live_ws ? live_list_of(live_ws) : nullptr, live_ws ? live_count_of(live_ws) : nullptr);
}

extern "C" int rtxn_mlp_train_backward_lean_segments(const rtxn_mlp* m, const float* start_points, const float* end_points,
                                                     const float* seg_view, long n_segments, int sample_type, const void* output_half,
                                                     const void* dout_half4, void* workspace_lean, const void* live_ws, float* dparams,
                                                     rtxn_stream_t stream) {
  int rc = check_lean(m, "rtxn_mlp_train_backward_lean_segments", n_segments * 32, true);
  if (rc != RTXN_OK) return rc;
  if (!rtxn_mlp_train_forward_lean_fused_supported(m)) {
    rtxn::set_error("rtxn_mlp_train_backward_lean_segments: the fused encoder is the reference's Composite-Frequency(3 x 10, 2 x 12); use "
                    "rtxn_mlp_train_backward_lean with encT");
    return RTXN_ERR_UNSUPPORTED;
  }
  rc = check_segments("rtxn_mlp_train_backward_lean_segments", start_points, end_points, seg_view, n_segments, sample_type);
  if (rc != RTXN_OK) return rc;
  RTXN_DEVICE_OR_FAIL();
  if (n_segments == 0) return RTXN_OK;
  RTXN_REQUIRE(output_half && dout_half4 && workspace_lean && dparams, "rtxn_mlp_train_backward_lean_segments: NULL buffer");
  const SampleSrc src{nullptr, start_points, end_points, seg_view, sample_type == RTXN_SAMPLING_MIDPOINT_WORLD};
  return train_backward_lean_impl(m, nullptr, output_half, dout_half4, n_segments * 32, workspace_lean, dparams, DevCount{nullptr, 0}, stream,
                                  live_ws ? live_list_of(live_ws) : nullptr, live_ws ? live_count_of(live_ws) : nullptr, &src);
}

// ---- recompute path (64-wide models): forward without saved activations + fused backward ----
extern "C" int rtxn_mlp_train_recompute_supported(const rtxn_mlp* m) {
  if (!m) return 0;
  return m->cfg.n_neurons == 64 && m->cfg.n_hidden_layers <= kFusedMaxL && m->enc_padded <= 64 && m->enc_padded % 16 == 0;
}

extern "C" int rtxn_mlp_train_forward_outputs(const rtxn_mlp* m, const void* encT, long n_samples, void* output_half,
                                              float* radiance, rtxn_stream_t stream) {
  int rc = check_train(m, "rtxn_mlp_train_forward_outputs");
  if (rc != RTXN_OK) return rc;
  RTXN_REQUIRE(n_samples >= 0 && n_samples <= kMaxTrainSamples, "rtxn_mlp_train_forward_outputs: n_samples = %ld out of [0, %ld]", n_samples, kMaxTrainSamples);
  RTXN_DEVICE_OR_FAIL();
  if (n_samples == 0) return RTXN_OK;
  RTXN_REQUIRE(encT && output_half, "rtxn_mlp_train_forward_outputs: NULL buffer");
  return train_forward_impl(m, encT, n_samples, nullptr, output_half, radiance, DevCount{nullptr, 0}, stream);
}

static int train_backward_recompute_impl(const rtxn_mlp* m, const void* encT, const void* output_half, const void* dout_half4,
                                         long n_samples, float* dparams, void* dencT, DevCount dc, rtxn_stream_t stream,
                                         const int* live_list = nullptr, const int* live_count = nullptr) {
  const int L = m->cfg.n_hidden_layers, E = m->enc_padded;
  FusedArgs a;
  memset(&a, 0, sizeof(a));
  a.packed_fwd = static_cast<const uint8_t*>(m->packed_train);
  a.packed_t = static_cast<const uint8_t*>(m->packed_t);
  a.L = L;
  a.KS0 = E / 16;
  a.out_act = m->cfg.output_activation;
  a.E = E;
  a.S = n_samples;
  a.Sp = padded(n_samples);
  a.dc = dc;
  a.n_tiles = (int)(a.Sp / kTile);
  a.encT = static_cast<const _Float16*>(encT);
  a.out_half = static_cast<const _Float16*>(output_half);
  a.dout = static_cast<const _Float16*>(dout_half4);
  a.dencT = static_cast<_Float16*>(dencT);
  a.dparams = dparams;
  a.live_list = live_list;
  a.live_count = live_count;
  a.det = det_ctx(dparams, g_det_mlp);
  const int RT = 2, KS = 4;
  const size_t lds = (size_t)(a.KS0 * RT + (L - 1) * KS * RT) * 1024 + (size_t)(RT + (L - 1) * RT * KS + ((E + 31) / 32) * KS) * 1024 +
                     8 * (size_t)kImgBytes;
  int dev = 0, n_cu = 0;
  RTXN_HIP(hipGetDevice(&dev));
  RTXN_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  if (n_cu <= 0) n_cu = 256;
  const int grid = a.n_tiles < n_cu ? a.n_tiles : n_cu;   // persistent: one block per CU (LDS- and register-bound)
  typedef void (*fused_fn)(FusedArgs);
  static const fused_fn table[kFusedMaxL][4] = {
      {mlp_bwd_fused64_kernel<1, 1>, mlp_bwd_fused64_kernel<1, 2>, mlp_bwd_fused64_kernel<1, 3>, mlp_bwd_fused64_kernel<1, 4>},
      {mlp_bwd_fused64_kernel<2, 1>, mlp_bwd_fused64_kernel<2, 2>, mlp_bwd_fused64_kernel<2, 3>, mlp_bwd_fused64_kernel<2, 4>},
      {mlp_bwd_fused64_kernel<3, 1>, mlp_bwd_fused64_kernel<3, 2>, mlp_bwd_fused64_kernel<3, 3>, mlp_bwd_fused64_kernel<3, 4>},
      {mlp_bwd_fused64_kernel<4, 1>, mlp_bwd_fused64_kernel<4, 2>, mlp_bwd_fused64_kernel<4, 3>, mlp_bwd_fused64_kernel<4, 4>}};
  const fused_fn fn = table[L - 1][a.KS0 - 1];
  RTXN_HIP(set_lds_once(reinterpret_cast<const void*>(fn), (int)lds));
  hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kThreads), lds, rtxn::as_stream(stream), a);
  RTXN_LAUNCH_CHECK("mlp_bwd_fused64_kernel");
  return det_fold(a.det.q, m->n_params, dparams, nullptr, 0, rtxn::as_stream(stream));
}

extern "C" int rtxn_mlp_train_backward_recompute(const rtxn_mlp* m, const void* encT, const void* output_half,
                                                 const void* dout_half4, long n_samples, float* dparams, void* dencT,
                                                 rtxn_stream_t stream) {
  int rc = check_train(m, "rtxn_mlp_train_backward_recompute");
  if (rc != RTXN_OK) return rc;
  if (!rtxn_mlp_train_recompute_supported(m)) {
    rtxn::set_error("rtxn_mlp_train_backward_recompute: built for 64-wide models with <= %d hidden layers and an encoded width <= 64 "
                    "(this model: %d wide, %d layers, %d features); use rtxn_mlp_train_forward + rtxn_mlp_train_backward",
                    kFusedMaxL, m->cfg.n_neurons, m->cfg.n_hidden_layers, m->enc_padded);
    return RTXN_ERR_UNSUPPORTED;
  }
  RTXN_REQUIRE(n_samples >= 0 && n_samples <= kMaxTrainSamples, "rtxn_mlp_train_backward_recompute: n_samples = %ld out of [0, %ld]", n_samples, kMaxTrainSamples);
  RTXN_DEVICE_OR_FAIL();
  if (n_samples == 0) return RTXN_OK;
  RTXN_REQUIRE(encT && output_half && dout_half4 && dparams, "rtxn_mlp_train_backward_recompute: NULL buffer");
  return train_backward_recompute_impl(m, encT, output_half, dout_half4, n_samples, dparams, dencT, DevCount{nullptr, 0}, stream);
}

extern "C" int rtxn_hashgrid_create(const rtxn_hashgrid_config* cfg, rtxn_hashgrid** out) {
  RTXN_REQUIRE(cfg && out, "rtxn_hashgrid_create: NULL argument");
  RTXN_REQUIRE(cfg->n_levels >= 1 && cfg->n_levels <= 16, "rtxn_hashgrid_create: n_levels = %d not in [1,16]", cfg->n_levels);
  RTXN_REQUIRE(cfg->n_features == 1 || cfg->n_features == 2 || cfg->n_features == 4 || cfg->n_features == 8,
               "rtxn_hashgrid_create: n_features = %d not in {1,2,4,8}", cfg->n_features);
  RTXN_REQUIRE(cfg->log2_hashmap_size >= 4 && cfg->log2_hashmap_size <= 24, "rtxn_hashgrid_create: log2_hashmap_size = %d",
               cfg->log2_hashmap_size);
  RTXN_REQUIRE(cfg->base_resolution >= 1 && cfg->per_level_scale >= 1.0f, "rtxn_hashgrid_create: bad resolution/scale");
  rtxn_hashgrid* g = new rtxn_hashgrid();
  g->cfg = *cfg;
  unsigned off = 0;
  for (int l = 0; l < cfg->n_levels; ++l) {
    const float s = exp2f((float)l * log2f(cfg->per_level_scale)) * (float)cfg->base_resolution - 1.0f;
    const unsigned r = (unsigned)ceilf(s) + 1u;
    unsigned long long dense = (unsigned long long)r * r * r;
    dense = (dense + 7ull) / 8ull * 8ull;
    const unsigned long long cap = 1ull << cfg->log2_hashmap_size;
    const unsigned sz = (unsigned)(dense < cap ? dense : cap);
    g->scale[l] = s; g->res[l] = r; g->size[l] = sz; g->offset[l] = off;
    off += sz;
  }
  g->n_params = (long)off * cfg->n_features;
  *out = g;
  return RTXN_OK;
}

extern "C" int rtxn_hashgrid_destroy(rtxn_hashgrid* g) { delete g; return RTXN_OK; }

extern "C" long rtxn_hashgrid_level_offset(const rtxn_hashgrid* g, int level) {
  if (!g || level < 0 || level > g->cfg.n_levels) return -1;
  if (level == g->cfg.n_levels) return g->n_params;
  return (long)g->offset[level] * g->cfg.n_features;
}

extern "C" int rtxn_hashgrid_level_is_hashed(const rtxn_hashgrid* g, int level) {
  if (!g || level < 0 || level >= g->cfg.n_levels) return -1;
  const unsigned long long r = g->res[level];
  return r * r * r > (unsigned long long)g->size[level] ? 1 : 0;
}

extern "C" int rtxn_convert_f32_to_f16(const float* src, void* dst_half, long n, rtxn_stream_t stream) {
  RTXN_REQUIRE(n >= 0, "rtxn_convert_f32_to_f16: n = %ld < 0", n);
  RTXN_DEVICE_OR_FAIL();
  if (n == 0) return RTXN_OK;
  RTXN_REQUIRE(src && dst_half, "rtxn_convert_f32_to_f16: NULL buffer");
  f32_to_f16_kernel<<<(unsigned)((n + 8L * kThreads - 1) / (8L * kThreads)), kThreads, 0, rtxn::as_stream(stream)>>>(
      src, static_cast<_Float16*>(dst_half), n);
  RTXN_LAUNCH_CHECK("f32_to_f16_kernel");
  return RTXN_OK;
}

extern "C" int rtxn_convert_f16_to_f32(const void* src_half, float* dst, long n, rtxn_stream_t stream) {
  RTXN_REQUIRE(n >= 0, "rtxn_convert_f16_to_f32: n = %ld < 0", n);
  RTXN_DEVICE_OR_FAIL();
  if (n == 0) return RTXN_OK;
  RTXN_REQUIRE(src_half && dst, "rtxn_convert_f16_to_f32: NULL buffer");
  f16_to_f32_kernel<<<(unsigned)((n + 8L * kThreads - 1) / (8L * kThreads)), kThreads, 0, rtxn::as_stream(stream)>>>(
      static_cast<const _Float16*>(src_half), dst, n);
  RTXN_LAUNCH_CHECK("f16_to_f32_kernel");
  return RTXN_OK;
}
// ------------------------------------------------------------------------- sparse view of a half2 gradient
// The data-parallel exchange of the hashed levels' gradient.  One rank's batch touches 0.2 % (late in training) to
// 25 % (first steps, finest level) of a 2^19-entry level (tools/probe/hash_grad_density.py), so a level travels as a list of
// (entry index, half2 bits) pairs wherever that is smaller than the level itself.  An entry is one half2 (both features of a
// grid corner, as the scatter adds them); "non-zero" ignores the sign of a zero.
__device__ __forceinline__ bool half2_nonzero(unsigned bits) { return (bits & 0x7fff7fffu) != 0u; }

// Geometry shared by the count and the pack kernel: a block (level) is cut into kHalf2Waves chunks of whole uint4s, one wave
// each (blockIdx.y = block, 4 waves per workgroup).  The count pass leaves every wave's count in the workspace, so the pack
// pass knows where each wave's entries go WITHOUT a global atomic: 18 k appends to one counter cost 177 us (same-address
// atomics retire one at a time, ~10 ns each), the two passes below read their 25 MB in ~10 us each -- and the list comes out
// sorted by entry index, the same on every run.
constexpr int kHalf2Waves = 128;
struct Half2Geom {
  long chunk;        // entries per wave, a multiple of 4
  int waves;         // waves per block that have entries (<= kHalf2Waves)
};
__host__ __device__ __forceinline__ Half2Geom half2_geom(long block_entries) {
  long chunk = (block_entries + kHalf2Waves - 1) / kHalf2Waves;
  chunk = (chunk + 3) / 4 * 4;
  if (chunk < 256) chunk = 256;                    // one iteration of a wave
  return Half2Geom{chunk, (int)((block_entries + chunk - 1) / chunk)};
}
// the calling wave's entries [lo, hi) and its index within the block; false: nothing to do
__device__ __forceinline__ bool half2_wave_range(long n, long block_entries, long& lo, long& hi, int& wl) {
  const Half2Geom g = half2_geom(block_entries);
  wl = (int)blockIdx.x * (kThreads / 64) + (int)(threadIdx.x >> 6);
  const long b_lo = (long)blockIdx.y * block_entries, b_hi = b_lo + block_entries < n ? b_lo + block_entries : n;
  lo = b_lo + (long)wl * g.chunk;
  hi = lo + g.chunk < b_hi ? lo + g.chunk : b_hi;
  return wl < g.waves && lo < hi;
}

// ws: int counts[n_blocks] | int wave_counts[n_blocks][kHalf2Waves]
__global__ __launch_bounds__(kThreads) void half2_count_kernel(const unsigned* __restrict__ v, long n, long block_entries,
                                                               int n_blocks, int* __restrict__ ws) {
  long lo, hi;
  int wl;
  const bool any = half2_wave_range(n, block_entries, lo, hi, wl);
  const int lane = threadIdx.x & 63;
  int c = 0;
  if (any) {
    const bool vec = (lo & 3) == 0 && ((uintptr_t)v & 15) == 0;
    const long n4 = vec ? (hi - lo) / 4 : 0;
    const uint4* v4 = reinterpret_cast<const uint4*>(v + lo);
    for (long q = lane; q < n4; q += 64) {
      const uint4 w = v4[q];
      c += (int)half2_nonzero(w.x) + (int)half2_nonzero(w.y) + (int)half2_nonzero(w.z) + (int)half2_nonzero(w.w);
    }
    for (long i = lo + 4 * n4 + lane; i < hi; i += 64) c += (int)half2_nonzero(v[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (lane == 0 && wl < kHalf2Waves) {
    ws[n_blocks + (int)blockIdx.y * kHalf2Waves + wl] = c;
    if (c) atomicAdd(&ws[blockIdx.y], c);          // <= 128 per address
  }
}

// Writes the non-zero entries of the selected blocks (bit b of block_mask) to `pairs`, in ascending entry index, at the
// positions the counts of the count pass (same values!) give; *count = the entries NEEDED -- entries past `capacity` are
// counted, not written.  CLEAR: every entry of a selected block is left zero (the exchange adds all ranks' lists, the rank's own
// included, back into it).
template <bool CLEAR>
__global__ __launch_bounds__(kThreads) void half2_pack_kernel(unsigned* __restrict__ v, long n, long block_entries, int n_blocks,
                                                              unsigned long long block_mask, long capacity,
                                                              uint2* __restrict__ pairs, int* __restrict__ count,
                                                              const int* __restrict__ ws) {
  const int lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    int need = 0;
    for (int b = 0; b < n_blocks; ++b) need += ((block_mask >> b) & 1ull) ? ws[b] : 0;
    *count = need;
  }
  if (!((block_mask >> blockIdx.y) & 1ull)) return;
  long lo, hi;
  int wl;
  if (!half2_wave_range(n, block_entries, lo, hi, wl)) return;
  // where this wave's entries start: the selected blocks before this one, then the waves before this one
  int before = 0;
  for (int b = lane; b < (int)blockIdx.y; b += 64) before += ((block_mask >> b) & 1ull) ? ws[b] : 0;
  const int* wc = ws + n_blocks + (int)blockIdx.y * kHalf2Waves;
  for (int w = lane; w < wl; w += 64) before += wc[w];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
  if (wc[wl] == 0 && !CLEAR) return;               // wave-uniform
  long at = before;
  const unsigned long long below = (1ull << lane) - 1ull;
  const bool vec = (lo & 3) == 0 && ((uintptr_t)v & 15) == 0;
  const long n4 = vec ? (hi - lo) / 4 : 0;
  uint4* v4 = reinterpret_cast<uint4*>(v + lo);
  for (long q0 = 0; q0 < n4; q0 += 64) {           // whole waves iterate together: the ballots need every lane
    const long q = q0 + lane;
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    if (q < n4) w = v4[q];
    const unsigned e[4] = {w.x, w.y, w.z, w.w};
    if (CLEAR && (w.x | w.y | w.z | w.w) != 0u) v4[q] = make_uint4(0u, 0u, 0u, 0u);      // the non-zero entries and any -0
    unsigned long long m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = __ballot(half2_nonzero(e[j]));
    if ((m[0] | m[1] | m[2] | m[3]) == 0ull) continue;      // wave-uniform
    // ascending index = lane-major, then j: the entries of the lanes below, then this lane's earlier ones
    long mine = at + __popcll(m[0] & below) + __popcll(m[1] & below) + __popcll(m[2] & below) + __popcll(m[3] & below);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (half2_nonzero(e[j])) {
        if (mine < capacity) pairs[mine] = make_uint2((unsigned)(lo + 4 * q + j), e[j]);
        ++mine;
      }
    at += __popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]);
  }
  for (long i0 = lo + 4 * n4; i0 < hi; i0 += 64) {
    const long i = i0 + lane;
    unsigned bits = 0u;
    if (i < hi) bits = v[i];
    if (CLEAR && bits != 0u) v[i] = 0u;
    const bool nz = half2_nonzero(bits);
    const unsigned long long m = __ballot(nz);
    if (m == 0ull) continue;
    const long mine = at + __popcll(m & below);
    if (nz && mine < capacity) pairs[mine] = make_uint2((unsigned)i, bits);
    at += __popcll(m);
  }
}

// values[index] += value for every pair of ONE list (unique indices: plain read-modify-write).  fp16 adds, as the scatter's
// packed atomics do; every rank applies the ranks' lists in rank order to zeroed entries, so all ranks end bit-identical.
__global__ __launch_bounds__(kThreads) void half2_add_pairs_kernel(unsigned* __restrict__ v, long n, const uint2* __restrict__ pairs,
                                                                   long count) {
  for (long k = (long)blockIdx.x * kThreads + threadIdx.x; k < count; k += (long)gridDim.x * kThreads) {
    const uint2 e = pairs[k];
    if (e.x >= (unsigned long long)n) continue;   // a foreign list is data, not a contract
    const half2v a = __builtin_bit_cast(half2v, v[e.x]), b = __builtin_bit_cast(half2v, e.y);
    const half2v r = {(_Float16)(a[0] + b[0]), (_Float16)(a[1] + b[1])};
    v[e.x] = __builtin_bit_cast(unsigned, r);
  }
}

static unsigned half2_grid(long n) {
  const long blocks = (n + kThreads - 1) / kThreads;
  return (unsigned)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks));
}
static dim3 half2_grid2(long n_entries, long block_entries) {
  const long nb = (n_entries + block_entries - 1) / block_entries;
  const Half2Geom g = half2_geom(block_entries);
  return dim3((unsigned)((g.waves + kThreads / 64 - 1) / (kThreads / 64)), (unsigned)nb);
}

extern "C" size_t rtxn_half2_workspace_bytes(long n_entries, long block_entries) {
  if (n_entries < 0 || block_entries <= 0) return 0;
  const long nb = (n_entries + block_entries - 1) / block_entries;
  return sizeof(int) * (size_t)(nb > 0 ? nb : 1) * (1 + kHalf2Waves);
}

extern "C" int rtxn_half2_count_nonzero(const void* values, long n_entries, long block_entries, void* workspace, rtxn_stream_t stream) {
  RTXN_REQUIRE(n_entries >= 0 && block_entries > 0 && n_entries < (1L << 32), "rtxn_half2_count_nonzero: n_entries = %ld, block_entries = %ld",
               n_entries, block_entries);
  RTXN_DEVICE_OR_FAIL();
  if (n_entries == 0) return RTXN_OK;
  RTXN_REQUIRE(values && workspace, "rtxn_half2_count_nonzero: NULL buffer");
  const long n_blocks = (n_entries + block_entries - 1) / block_entries;
  RTXN_REQUIRE(n_blocks <= 65535, "rtxn_half2_count_nonzero: %ld blocks", n_blocks);
  RTXN_HIP(rtxn::zero_words(workspace, (size_t)n_blocks, rtxn::as_stream(stream)));
  half2_count_kernel<<<half2_grid2(n_entries, block_entries), kThreads, 0, rtxn::as_stream(stream)>>>(
      static_cast<const unsigned*>(values), n_entries, block_entries, (int)n_blocks, static_cast<int*>(workspace));
  RTXN_LAUNCH_CHECK("half2_count_kernel");
  return RTXN_OK;
}

extern "C" int rtxn_half2_pack_nonzero(void* values, long n_entries, long block_entries, const void* workspace,
                                       unsigned long long block_mask, long capacity, void* pairs, int* count, int clear,
                                       rtxn_stream_t stream) {
  RTXN_REQUIRE(n_entries >= 0 && block_entries > 0 && n_entries < (1L << 32) && capacity >= 0,
               "rtxn_half2_pack_nonzero: n_entries = %ld, block_entries = %ld, capacity = %ld", n_entries, block_entries, capacity);
  const long n_blocks = (n_entries + block_entries - 1) / block_entries;
  RTXN_REQUIRE(n_blocks <= 64, "rtxn_half2_pack_nonzero: %ld blocks do not fit the 64-bit block_mask", n_blocks);
  RTXN_DEVICE_OR_FAIL();
  RTXN_REQUIRE(count, "rtxn_half2_pack_nonzero: NULL count");
  if (n_entries == 0 || block_mask == 0ull) {
    RTXN_HIP(rtxn::zero_words(count, 1, rtxn::as_stream(stream)));
    return RTXN_OK;
  }
  RTXN_REQUIRE(values && workspace && (pairs || capacity == 0), "rtxn_half2_pack_nonzero: NULL buffer");
  if (clear)
    half2_pack_kernel<true><<<half2_grid2(n_entries, block_entries), kThreads, 0, rtxn::as_stream(stream)>>>(
        static_cast<unsigned*>(values), n_entries, block_entries, (int)n_blocks, block_mask, capacity, static_cast<uint2*>(pairs), count,
        static_cast<const int*>(workspace));
  else
    half2_pack_kernel<false><<<half2_grid2(n_entries, block_entries), kThreads, 0, rtxn::as_stream(stream)>>>(
        static_cast<unsigned*>(values), n_entries, block_entries, (int)n_blocks, block_mask, capacity, static_cast<uint2*>(pairs), count,
        static_cast<const int*>(workspace));
  RTXN_LAUNCH_CHECK("half2_pack_kernel");
  return RTXN_OK;
}

extern "C" int rtxn_half2_add_pairs(void* values, long n_entries, const void* pairs, long count, rtxn_stream_t stream) {
  RTXN_REQUIRE(n_entries >= 0 && count >= 0 && n_entries < (1L << 32), "rtxn_half2_add_pairs: n_entries = %ld, count = %ld", n_entries, count);
  RTXN_DEVICE_OR_FAIL();
  if (count == 0) return RTXN_OK;
  RTXN_REQUIRE(values && pairs, "rtxn_half2_add_pairs: NULL buffer");
  half2_add_pairs_kernel<<<half2_grid(count), kThreads, 0, rtxn::as_stream(stream)>>>(static_cast<unsigned*>(values), n_entries,
                                                                                  static_cast<const uint2*>(pairs), count);
  RTXN_LAUNCH_CHECK("half2_add_pairs_kernel");
  return RTXN_OK;
}

extern "C" long rtxn_hashgrid_n_params(const rtxn_hashgrid* g) { return g ? g->n_params : -1; }
extern "C" int rtxn_hashgrid_encoded_width(const rtxn_hashgrid* g, int n_dir_freqs) {
  if (!g || n_dir_freqs < 0) return -1;
  return (g->cfg.n_levels * g->cfg.n_features + 4 * n_dir_freqs + 15) / 16 * 16;
}

static int hashgrid_encode_impl(const rtxn_hashgrid* g, int n_dir_freqs, const void* table_fp16, const SampleSrc& src, void* encT,
                                float* t_vals, float t_scale, long n_samples, DevCount dc, rtxn_stream_t stream) {
  const long Sp = padded(n_samples);
  const int E = rtxn_hashgrid_encoded_width(g, n_dir_freqs);
  if (g->cfg.n_features == 2 && ((uintptr_t)table_fp16 & 7) == 0)
    hashgrid_encode_f2_kernel<<<dim3((unsigned)(Sp / kThreads)), kThreads, 0, rtxn::as_stream(stream)>>>(
        levels_of(g), n_dir_freqs, static_cast<const _Float16*>(table_fp16), src, static_cast<_Float16*>(encT), t_vals, t_scale,
        n_samples, Sp, E, dc);
  else
    hashgrid_encode_kernel<<<dim3((unsigned)(Sp / kThreads), (unsigned)(g->cfg.n_levels + 1)), kThreads, 0, rtxn::as_stream(stream)>>>(
        levels_of(g), n_dir_freqs, static_cast<const _Float16*>(table_fp16), src, static_cast<_Float16*>(encT), t_vals, t_scale,
        n_samples, Sp, E, dc);
  RTXN_LAUNCH_CHECK("hashgrid_encode_kernel");
  return RTXN_OK;
}

extern "C" int rtxn_hashgrid_encode(const rtxn_hashgrid* g, int n_dir_freqs, const void* table_fp16, const float* input,
                                    void* encT, long n_samples, rtxn_stream_t stream) {
  RTXN_REQUIRE(g && n_dir_freqs >= 0 && n_dir_freqs <= 16, "rtxn_hashgrid_encode: bad argument");
  RTXN_REQUIRE(n_samples >= 0, "rtxn_hashgrid_encode: n_samples = %ld < 0", n_samples);
  RTXN_DEVICE_OR_FAIL();
  if (n_samples == 0) return RTXN_OK;
  RTXN_REQUIRE(table_fp16 && input && encT, "rtxn_hashgrid_encode: NULL buffer");
  const SampleSrc src{input, nullptr, nullptr, nullptr, 0};
  return hashgrid_encode_impl(g, n_dir_freqs, table_fp16, src, encT, nullptr, 1.0f, n_samples, DevCount{nullptr, 0}, stream);
}

extern "C" int rtxn_hashgrid_encode_segments(const rtxn_hashgrid* g, int n_dir_freqs, const void* table_fp16,
                                             const float* start_points, const float* end_points, const float* seg_view,
                                             long n_segments, int sample_type, float t_scale, void* encT, float* t_vals,
                                             rtxn_stream_t stream) {
  RTXN_REQUIRE(g && n_dir_freqs >= 0 && n_dir_freqs <= 16, "rtxn_hashgrid_encode_segments: bad argument");
  int rc = check_segments("rtxn_hashgrid_encode_segments", start_points, end_points, seg_view, n_segments, sample_type);
  if (rc != RTXN_OK) return rc;
  RTXN_DEVICE_OR_FAIL();
  if (n_segments == 0) return RTXN_OK;
  RTXN_REQUIRE(table_fp16 && encT, "rtxn_hashgrid_encode_segments: NULL buffer");
  const SampleSrc src{nullptr, start_points, end_points, seg_view, sample_type == RTXN_SAMPLING_MIDPOINT_WORLD};
  return hashgrid_encode_impl(g, n_dir_freqs, table_fp16, src, encT, t_vals, t_scale, n_segments * 32, DevCount{nullptr, 0}, stream);
}

// dtable_hashed_half == NULL: every level into the fp32 table.  Otherwise (n_features == 2): the hashed levels go to the fp16
// buffer, which holds the parameters from the first hashed level on.
static int hashgrid_backward_impl(const rtxn_hashgrid* g, const SampleSrc& input, const void* dencT, long n_samples,
                                  float* dtable, void* dtable_hashed_half, DevCount dc, rtxn_stream_t stream,
                                  const int* live_list = nullptr, const int* live_count = nullptr) {
  const long Sp = padded(n_samples);
  const int NL = g->cfg.n_levels, F = g->cfg.n_features;
  int first_hashed = NL;
  for (int l = NL - 1; l >= 0; --l)
    if ((unsigned long long)g->res[l] * g->res[l] * g->res[l] > g->size[l]) first_hashed = l;
  const long hashed_lo = first_hashed < NL ? (long)g->offset[first_hashed] * F : g->n_params;
  if (!dtable_hashed_half) first_hashed = NL;          // everything fp32
  hipStream_t st = rtxn::as_stream(stream);
  const HgLevels lv = levels_of(g);
  const _Float16* de = static_cast<const _Float16*>(dencT);
  const unsigned sblocks = (unsigned)((n_samples + kThreads - 1) / kThreads);
  const DetCtx det = det_ctx(dtable, g_det_table);
  if (det.q) {
    // deterministic: every level as fp32-shaped contributions into the fixed-point shadow, then one fold into dtable / dtable_h
    hashgrid_backward_kernel<false, true><<<dim3(sblocks, (unsigned)NL), kThreads, 0, st>>>(
        lv, 0, input, de, n_samples, Sp, dtable, nullptr, 0, dc, live_list, live_count, det);
    RTXN_LAUNCH_CHECK("hashgrid_backward_kernel<deterministic>");
    return det_fold(det.q, g->n_params, dtable, dtable_hashed_half, hashed_lo, st);
  }
  if (first_hashed > 0) {
    hashgrid_backward_kernel<false><<<dim3(sblocks, (unsigned)first_hashed), kThreads, 0, st>>>(
        lv, 0, input, de, n_samples, Sp, dtable, nullptr, 0, dc, live_list, live_count);
    RTXN_LAUNCH_CHECK("hashgrid_backward_kernel");
  }
  if (first_hashed < NL) {
    hashgrid_backward_kernel<true><<<dim3(sblocks, (unsigned)(NL - first_hashed)), kThreads, 0, st>>>(
        lv, first_hashed, input, de, n_samples, Sp, dtable, static_cast<_Float16*>(dtable_hashed_half), hashed_lo, dc, live_list, live_count);
    RTXN_LAUNCH_CHECK("hashgrid_backward_kernel<pk_f16>");
  }
  return RTXN_OK;
}

extern "C" int rtxn_hashgrid_backward(const rtxn_hashgrid* g, const float* input, const void* dencT, long n_samples,
                                      float* dtable, rtxn_stream_t stream) {
  RTXN_REQUIRE(g, "rtxn_hashgrid_backward: NULL grid");
  RTXN_REQUIRE(n_samples >= 0, "rtxn_hashgrid_backward: n_samples = %ld < 0", n_samples);
  RTXN_DEVICE_OR_FAIL();
  if (n_samples == 0) return RTXN_OK;
  RTXN_REQUIRE(input && dencT && dtable, "rtxn_hashgrid_backward: NULL buffer");
  return hashgrid_backward_impl(g, SampleSrc{input, nullptr, nullptr, nullptr, 0}, dencT, n_samples, dtable, nullptr, DevCount{nullptr, 0}, stream);
}

extern "C" int rtxn_hashgrid_backward_mixed(const rtxn_hashgrid* g, const float* input, const void* dencT, long n_samples,
                                            float* dtable, void* dtable_hashed_half, rtxn_stream_t stream) {
  RTXN_REQUIRE(g, "rtxn_hashgrid_backward_mixed: NULL grid");
  RTXN_REQUIRE(n_samples >= 0, "rtxn_hashgrid_backward_mixed: n_samples = %ld < 0", n_samples);
  RTXN_REQUIRE(g->cfg.n_features == 2, "rtxn_hashgrid_backward_mixed: packed fp16 atomics need n_features == 2 (got %d)", g->cfg.n_features);
  RTXN_DEVICE_OR_FAIL();
  if (n_samples == 0) return RTXN_OK;
  RTXN_REQUIRE(input && dencT && dtable && dtable_hashed_half, "rtxn_hashgrid_backward_mixed: NULL buffer");
  RTXN_REQUIRE(((uintptr_t)dtable_hashed_half & 3) == 0, "rtxn_hashgrid_backward_mixed: fp16 table must be 4-byte aligned");
  return hashgrid_backward_impl(g, SampleSrc{input, nullptr, nullptr, nullptr, 0}, dencT, n_samples, dtable, dtable_hashed_half, DevCount{nullptr, 0}, stream);
}

extern "C" int rtxn_hashgrid_backward_segments(const rtxn_hashgrid* g, const float* start_points, const float* end_points,
                                               long n_segments, int sample_type, const void* dencT, float* dtable,
                                               void* dtable_hashed_half, rtxn_stream_t stream) {
  RTXN_REQUIRE(g, "rtxn_hashgrid_backward_segments: NULL grid");
  int rc = check_segments("rtxn_hashgrid_backward_segments", start_points, end_points, start_points, n_segments, sample_type);
  if (rc != RTXN_OK) return rc;
  RTXN_REQUIRE(!dtable_hashed_half || g->cfg.n_features == 2, "rtxn_hashgrid_backward_segments: packed fp16 atomics need n_features == 2 (got %d)", g->cfg.n_features);
  RTXN_DEVICE_OR_FAIL();
  if (n_segments == 0) return RTXN_OK;
  RTXN_REQUIRE(dencT && dtable, "rtxn_hashgrid_backward_segments: NULL buffer");
  const SampleSrc src{nullptr, start_points, end_points, nullptr, sample_type == RTXN_SAMPLING_MIDPOINT_WORLD};
  return hashgrid_backward_impl(g, src, dencT, n_segments * 32, dtable, dtable_hashed_half, DevCount{nullptr, 0}, stream);
}

extern "C" int rtxn_l2_loss(const float* pred, const float* target, long n, float loss_scale, float* values,
                            void* grads_half, float* loss_sum, rtxn_stream_t stream) {
  RTXN_REQUIRE(n >= 0, "rtxn_l2_loss: n = %ld < 0", n);
  RTXN_DEVICE_OR_FAIL();
  hipStream_t s = rtxn::as_stream(stream);
  if (loss_sum) RTXN_HIP(rtxn::zero_words(loss_sum, 1, s));
  if (n == 0) return RTXN_OK;
  RTXN_REQUIRE(pred && target, "rtxn_l2_loss: NULL buffer");
  const unsigned blocks = (unsigned)((n + kThreads - 1) / kThreads < 1024 ? (n + kThreads - 1) / kThreads : 1024);
  l2_loss_kernel<<<blocks, kThreads, 0, s>>>(pred, target, n, loss_scale, values, static_cast<__half*>(grads_half), loss_sum);
  RTXN_LAUNCH_CHECK("l2_loss_kernel");
  return RTXN_OK;
}

static float adam_lr_eff(float lr, float beta1, float beta2, int step) {
  return lr * sqrtf(1.0f - powf(beta2, (float)step)) / (1.0f - powf(beta1, (float)step));
}
static int adam_impl(const char* who, long n, float* master, void* params_fp16, void* grads, int grads_fp16, bool zero, float* m, float* v,
                     float lr_eff, const float* lr_dev, float beta1, float beta2, float eps, float loss_scale, rtxn_stream_t stream) {
  RTXN_REQUIRE(loss_scale != 0.0f, "%s: loss_scale = 0", who);
  RTXN_DEVICE_OR_FAIL();
  if (n == 0) return RTXN_OK;
  RTXN_REQUIRE(master && params_fp16 && grads && m && v, "%s: NULL buffer", who);
  const long work = (n + 3) / 4;
  const unsigned blocks = (unsigned)((work + kThreads - 1) / kThreads < 4096 ? (work + kThreads - 1) / kThreads : 4096);
  hipStream_t st = rtxn::as_stream(stream);
  __half* p16 = static_cast<__half*>(params_fp16);
  const float ils = 1.0f / loss_scale;
  if (grads_fp16) {
    if (zero) adam_kernel<true, true><<<blocks, kThreads, 0, st>>>(n, master, p16, grads, m, v, lr_eff, beta1, beta2, eps, ils, lr_dev);
    else adam_kernel<true, false><<<blocks, kThreads, 0, st>>>(n, master, p16, grads, m, v, lr_eff, beta1, beta2, eps, ils, lr_dev);
  } else {
    if (zero) adam_kernel<false, true><<<blocks, kThreads, 0, st>>>(n, master, p16, grads, m, v, lr_eff, beta1, beta2, eps, ils, lr_dev);
    else adam_kernel<false, false><<<blocks, kThreads, 0, st>>>(n, master, p16, grads, m, v, lr_eff, beta1, beta2, eps, ils, lr_dev);
  }
  RTXN_LAUNCH_CHECK("adam_kernel");
  return RTXN_OK;
}

extern "C" int rtxn_adam_step(long n, float* master, void* params_fp16, const float* grads, float* m, float* v, int step,
                              float lr, float beta1, float beta2, float eps, float loss_scale, rtxn_stream_t stream) {
  RTXN_REQUIRE(n >= 0 && step >= 1, "rtxn_adam_step: n = %ld, step = %d", n, step);
  return adam_impl("rtxn_adam_step", n, master, params_fp16, const_cast<float*>(grads), 0, false, m, v, adam_lr_eff(lr, beta1, beta2, step),
                   nullptr, beta1, beta2, eps, loss_scale, stream);
}

extern "C" int rtxn_adam_step_half_grads(long n, float* master, void* params_fp16, const void* grads_fp16, float* m, float* v,
                                         int step, float lr, float beta1, float beta2, float eps, float loss_scale,
                                         rtxn_stream_t stream) {
  RTXN_REQUIRE(n >= 0 && step >= 1, "rtxn_adam_step_half_grads: n = %ld, step = %d", n, step);
  return adam_impl("rtxn_adam_step_half_grads", n, master, params_fp16, const_cast<void*>(grads_fp16), 1, false, m, v,
                   adam_lr_eff(lr, beta1, beta2, step), nullptr, beta1, beta2, eps, loss_scale, stream);
}

extern "C" float rtxn_adam_effective_lr(float lr, float beta1, float beta2, int step) {
  return step >= 1 ? adam_lr_eff(lr, beta1, beta2, step) : 0.0f;
}

extern "C" int rtxn_adam_step_captured(long n, float* master, void* params_fp16, void* grads, int grad_flags, float* m,
                                       float* v, const float* effective_lr, float beta1, float beta2, float eps, float loss_scale,
                                       rtxn_stream_t stream) {
  RTXN_REQUIRE(n >= 0 && effective_lr, "rtxn_adam_step_captured: n = %ld, effective_lr = %p", n, (const void*)effective_lr);
  RTXN_REQUIRE((grad_flags & ~3) == 0, "rtxn_adam_step_captured: grad_flags = %d (RTXN_ADAM_GRADS_FP16 | RTXN_ADAM_ZERO_GRADS)", grad_flags);
  return adam_impl("rtxn_adam_step_captured", n, master, params_fp16, grads, grad_flags & RTXN_ADAM_GRADS_FP16,
                   (grad_flags & RTXN_ADAM_ZERO_GRADS) != 0, m, v, 0.0f, effective_lr, beta1, beta2, eps, loss_scale, stream);
}

extern "C" int rtxn_adam_step_sparse(long n, float* master, void* params_fp16, void* grads, int grad_flags, float* m, float* v,
                                     unsigned* param_steps, float lr, float beta1, float beta2, float eps, float loss_scale,
                                     rtxn_stream_t stream) {
  RTXN_REQUIRE(n >= 0, "rtxn_adam_step_sparse: n = %ld", n);
  RTXN_REQUIRE((grad_flags & ~3) == 0, "rtxn_adam_step_sparse: grad_flags = %d (RTXN_ADAM_GRADS_FP16 | RTXN_ADAM_ZERO_GRADS)", grad_flags);
  RTXN_REQUIRE(loss_scale != 0.0f, "rtxn_adam_step_sparse: loss_scale = 0");
  RTXN_DEVICE_OR_FAIL();
  if (n == 0) return RTXN_OK;
  RTXN_REQUIRE(master && params_fp16 && grads && m && v && param_steps, "rtxn_adam_step_sparse: NULL buffer");
  const long blocks = (n / 4 + kThreads - 1) / kThreads;
  const unsigned gridx = (unsigned)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));
  hipStream_t st = rtxn::as_stream(stream);
  const float ils = 1.0f / loss_scale;
  RTXN_REQUIRE(beta1 > 0.0f && beta1 < 1.0f && beta2 > 0.0f && beta2 < 1.0f, "rtxn_adam_step_sparse: beta1 = %g, beta2 = %g outside (0, 1)", beta1, beta2);
  const float l2b1 = (float)log2((double)beta1), l2b2 = (float)log2((double)beta2);
  __half* p16 = static_cast<__half*>(params_fp16);
  const bool half = grad_flags & RTXN_ADAM_GRADS_FP16, zero = grad_flags & RTXN_ADAM_ZERO_GRADS;
  if (half && zero) adam_sparse_kernel<true, true><<<gridx, kThreads, 0, st>>>(n, master, p16, grads, m, v, param_steps, lr, beta1, beta2, eps, ils, l2b1, l2b2);
  else if (half) adam_sparse_kernel<true, false><<<gridx, kThreads, 0, st>>>(n, master, p16, grads, m, v, param_steps, lr, beta1, beta2, eps, ils, l2b1, l2b2);
  else if (zero) adam_sparse_kernel<false, true><<<gridx, kThreads, 0, st>>>(n, master, p16, grads, m, v, param_steps, lr, beta1, beta2, eps, ils, l2b1, l2b2);
  else adam_sparse_kernel<false, false><<<gridx, kThreads, 0, st>>>(n, master, p16, grads, m, v, param_steps, lr, beta1, beta2, eps, ils, l2b1, l2b2);
  RTXN_LAUNCH_CHECK("adam_sparse_kernel");
  return RTXN_OK;
}

// ------------------------------------------------------------------------- live segments

extern "C" size_t rtxn_live_segments_workspace_bytes(long segment_capacity) {
  return segment_capacity < 0 ? 0 : live_ws_bytes(segment_capacity);
}

static int live_segments_impl(const void* dout_half4, long n_segments, long capacity, void* live_ws, DevCount dc, rtxn_stream_t stream) {
  uint8_t* ws = static_cast<uint8_t*>(live_ws);
  uint8_t* flags = ws + 16 + 4 * capacity;
  const long S = n_segments * 32;
  hipStream_t st = rtxn::as_stream(stream);
  live_flags_kernel<<<(unsigned)((S + kThreads - 1) / kThreads), kThreads, 0, st>>>(static_cast<const uint2*>(dout_half4), S, dc, flags);
  RTXN_LAUNCH_CHECK("live_flags_kernel");
  live_compact_kernel<<<1, kCompactThreads, 0, st>>>(flags, S, dc, reinterpret_cast<int*>(ws + 16), reinterpret_cast<int*>(ws));
  RTXN_LAUNCH_CHECK("live_compact_kernel");
  return RTXN_OK;
}

extern "C" int rtxn_live_segments(const void* radiance_gradients_half4, long n_segments, long segment_capacity, void* live_ws,
                                  rtxn_stream_t stream) {
  RTXN_REQUIRE(n_segments >= 0 && segment_capacity >= n_segments && segment_capacity <= kMaxTrainSamples / 32,
               "rtxn_live_segments: n_segments = %ld, segment_capacity = %ld", n_segments, segment_capacity);
  RTXN_REQUIRE(live_ws && ((uintptr_t)live_ws & 15) == 0, "rtxn_live_segments: workspace NULL or not 16-byte aligned");
  RTXN_DEVICE_OR_FAIL();
  if (n_segments == 0) {
    RTXN_HIP(rtxn::zero_words(live_ws, 4, rtxn::as_stream(stream)));
    return RTXN_OK;
  }
  RTXN_REQUIRE(radiance_gradients_half4 && ((uintptr_t)radiance_gradients_half4 & 7) == 0, "rtxn_live_segments: gradients NULL or not 8-byte aligned");
  return live_segments_impl(radiance_gradients_half4, n_segments, segment_capacity, live_ws, DevCount{nullptr, 0}, stream);
}

extern "C" int rtxn_mlp_train_backward_recompute_live(const rtxn_mlp* m, const void* encT, const void* output_half,
                                                      const void* dout_half4, long n_samples, const void* live_ws, float* dparams,
                                                      void* dencT, rtxn_stream_t stream) {
  int rc = check_train(m, "rtxn_mlp_train_backward_recompute_live");
  if (rc != RTXN_OK) return rc;
  RTXN_REQUIRE(rtxn_mlp_train_recompute_supported(m), "rtxn_mlp_train_backward_recompute_live: this model has no recompute path");
  RTXN_REQUIRE(n_samples >= 0 && n_samples <= kMaxTrainSamples && n_samples % 32 == 0,
               "rtxn_mlp_train_backward_recompute_live: n_samples = %ld must be whole segments in [0, %ld]", n_samples, kMaxTrainSamples);
  RTXN_DEVICE_OR_FAIL();
  if (n_samples == 0) return RTXN_OK;
  RTXN_REQUIRE(encT && output_half && dout_half4 && dparams && live_ws, "rtxn_mlp_train_backward_recompute_live: NULL buffer");
  return train_backward_recompute_impl(m, encT, output_half, dout_half4, n_samples, dparams, dencT, DevCount{nullptr, 0}, stream,
                                       live_list_of(live_ws), live_count_of(live_ws));
}

extern "C" int rtxn_mlp_train_backward_live(const rtxn_mlp* m, const void* encT, const void* output_half, const void* dout_half4,
                                            long n_samples, void* workspace, const void* live_ws, float* dparams, void* dencT,
                                            rtxn_stream_t stream) {
  int rc = check_train(m, "rtxn_mlp_train_backward_live");
  if (rc != RTXN_OK) return rc;
  RTXN_REQUIRE(n_samples >= 0 && n_samples <= kMaxTrainSamples && n_samples % 32 == 0,
               "rtxn_mlp_train_backward_live: n_samples = %ld must be whole segments in [0, %ld]", n_samples, kMaxTrainSamples);
  RTXN_DEVICE_OR_FAIL();
  if (n_samples == 0) return RTXN_OK;
  RTXN_REQUIRE(encT && output_half && dout_half4 && workspace && dparams && live_ws, "rtxn_mlp_train_backward_live: NULL buffer");
  return train_backward_impl(m, encT, output_half, dout_half4, n_samples, workspace, dparams, dencT, DevCount{nullptr, 0}, stream,
                             live_list_of(live_ws), live_count_of(live_ws));
}

extern "C" int rtxn_hashgrid_backward_segments_live(const rtxn_hashgrid* g, const float* start_points, const float* end_points,
                                                    long n_segments, int sample_type, const void* dencT, const void* live_ws,
                                                    float* dtable, void* dtable_hashed_half, rtxn_stream_t stream) {
  RTXN_REQUIRE(g, "rtxn_hashgrid_backward_segments_live: NULL grid");
  int rc = check_segments("rtxn_hashgrid_backward_segments_live", start_points, end_points, start_points, n_segments, sample_type);
  if (rc != RTXN_OK) return rc;
  RTXN_REQUIRE(!dtable_hashed_half || g->cfg.n_features == 2, "rtxn_hashgrid_backward_segments_live: packed fp16 atomics need n_features == 2 (got %d)", g->cfg.n_features);
  RTXN_DEVICE_OR_FAIL();
  if (n_segments == 0) return RTXN_OK;
  RTXN_REQUIRE(dencT && dtable && live_ws, "rtxn_hashgrid_backward_segments_live: NULL buffer");
  const SampleSrc src{nullptr, start_points, end_points, nullptr, sample_type == RTXN_SAMPLING_MIDPOINT_WORLD};
  return hashgrid_backward_impl(g, src, dencT, n_segments * 32, dtable, dtable_hashed_half, DevCount{nullptr, 0}, stream,
                                live_list_of(live_ws), live_count_of(live_ws));
}

// ------------------------------------------------------------------------- a whole batch, segment count on the device
extern "C" int rtxn_train_gradients(const rtxn_train_batch* b, rtxn_stream_t stream) {
  RTXN_REQUIRE(b && b->mlp, "rtxn_train_gradients: NULL batch or model");
  const rtxn_mlp* m = b->mlp;
  int rc = check_train(m, "rtxn_train_gradients");
  if (rc != RTXN_OK) return rc;
  RTXN_REQUIRE(b->n_rays > 0, "rtxn_train_gradients: n_rays = %d", b->n_rays);
  RTXN_REQUIRE(b->segment_capacity > 0 && b->segment_capacity <= kMaxTrainSamples / 32 && b->segment_capacity <= 0x7fffffffL,
               "rtxn_train_gradients: segment_capacity = %ld", b->segment_capacity);
  rc = check_segments("rtxn_train_gradients", b->start_points, b->end_points, b->seg_view, b->segment_capacity, b->sample_type);
  if (rc != RTXN_OK) return rc;
  RTXN_REQUIRE(b->total_segments && b->num_stored && b->indices, "rtxn_train_gradients: NULL total_segments / num_stored / indices");
  RTXN_REQUIRE(b->vr_mode == RTXN_VR_COMPAT || b->vr_mode == RTXN_VR_NERF, "rtxn_train_gradients: vr_mode %d", b->vr_mode);
  RTXN_REQUIRE(b->encT && b->output_half && b->radiance && b->t_vals && b->radiance_gradients && b->pixels && b->loss_gradients_half &&
               b->targets && b->dparams, "rtxn_train_gradients: NULL buffer");
  const bool hash = b->grid != nullptr;
  if (hash) {
    RTXN_REQUIRE(b->table_fp16 && b->dencT && b->dtable, "rtxn_train_gradients: hash grid without table / dencT / dtable");
    RTXN_REQUIRE(!b->dtable_hashed_half || b->grid->cfg.n_features == 2, "rtxn_train_gradients: packed fp16 atomics need n_features == 2");
    RTXN_REQUIRE(rtxn_hashgrid_encoded_width(b->grid, b->n_dir_freqs) == m->enc_padded,
                 "rtxn_train_gradients: the grid encodes %d features, the model takes %d", rtxn_hashgrid_encoded_width(b->grid, b->n_dir_freqs), m->enc_padded);
  } else {
    RTXN_REQUIRE(m->cfg.encoding == RTXN_ENC_FREQUENCY && m->cfg.n_pos_dims == 3 && m->cfg.n_dir_dims == 2,
                 "rtxn_train_gradients: without a grid the model must carry the 3 + 2 frequency encoding");
  }
  const bool lean = b->workspace_lean != 0;
  if (lean) {
    RTXN_REQUIRE(b->workspace && !hash && rtxn_mlp_train_lean_supported(m), "rtxn_train_gradients: workspace_lean needs a lean workspace and a "
                 "model with the lean path (rtxn_mlp_train_lean_supported; frequency encoding)");
  }
  const bool recompute = b->workspace == nullptr;
  if (recompute)
    RTXN_REQUIRE(rtxn_mlp_train_recompute_supported(m), "rtxn_train_gradients: workspace == NULL selects the recompute path, which this model "
                 "(%d wide, %d layers, %d features) does not have", m->cfg.n_neurons, m->cfg.n_hidden_layers, m->enc_padded);
  RTXN_DEVICE_OR_FAIL();
  const DevCount dc{b->total_segments, (int)b->segment_capacity};
  const long cap_samples = b->segment_capacity * 32;
  const SampleSrc src{nullptr, b->start_points, b->end_points, b->seg_view, b->sample_type == RTXN_SAMPLING_MIDPOINT_WORLD};
  // The reference's own model on the lean path: sampler AND encoder folded into the forward and into the weight gradient (encT is
  // never written; RTXN_TRAIN_LEAN_FUSED=0: the staged encoder, for the A/B -- the same values bit for bit)
  const char* fused_env = getenv("RTXN_TRAIN_LEAN_FUSED");       // (read per call: a test switches it between trainers)
  const bool fused_off = fused_env && atoi(fused_env) == 0;
  const bool fused = lean && !fused_off && rtxn_mlp_train_forward_lean_fused_supported(m);
  // launchSampler + encoding (main.cu:703,721)
  if (!fused) {
    rc = hash ? hashgrid_encode_impl(b->grid, b->n_dir_freqs, b->table_fp16, src, b->encT, b->t_vals, b->t_scale, cap_samples, dc, stream)
              : encode_frequency_impl(m, src, b->encT, b->t_vals, b->t_scale, cap_samples, dc, stream);
    if (rc != RTXN_OK) return rc;
  }
  // network->forward (main.cu:721).  Saved-activation models with a live list and the NeRF compositor (whose gradient vanishes
  // behind the first surface): outputs only here, the activations of the live segments are saved after the compositor.
  // (lean: the forward saves 16 bytes of sign masks per sample and layer for EVERY sample -- cheap enough that no second pass is needed)
  const bool two_pass = !recompute && !lean && b->live_ws != nullptr && b->vr_mode == RTXN_VR_NERF;
  rc = fused ? train_forward_impl(m, nullptr, cap_samples, b->workspace, b->output_half, b->radiance, dc, stream, nullptr, nullptr, true, &src, b->t_vals, b->t_scale)
             : train_forward_impl(m, b->encT, cap_samples, two_pass ? nullptr : b->workspace, b->output_half, b->radiance, dc, stream, nullptr, nullptr, lean);
  if (rc != RTXN_OK) return rc;
  // launch_volrender_cuda, loss->evaluate, launch_volrender_backward_cuda (main.cu:737-767): per ray, no sample count needed
  if (b->vr_mode == RTXN_VR_NERF) {
    rc = rtxn_volrender_l2_train(b->radiance, b->t_vals, b->num_stored, b->indices, b->n_rays, 32, b->targets, b->loss_scale, b->pixels,
                                 b->loss_gradients_half, b->loss_sum, b->radiance_gradients, stream);
    if (rc != RTXN_OK) return rc;
  } else {
    rc = rtxn_volrender_fwd(nullptr, b->radiance, b->num_stored, b->indices, b->t_vals, b->n_rays, 32, b->pixels, b->vr_mode, stream);
    if (rc != RTXN_OK) return rc;
    rc = rtxn_l2_loss(b->pixels, b->targets, 3L * b->n_rays, b->loss_scale, nullptr, b->loss_gradients_half, b->loss_sum, stream);
    if (rc != RTXN_OK) return rc;
    rc = rtxn_volrender_bwd(nullptr, b->loss_gradients_half, b->radiance, b->t_vals, b->num_stored, b->indices, b->n_rays, 32,
                            b->radiance_gradients, b->vr_mode, stream);
    if (rc != RTXN_OK) return rc;
  }
  // network->backward (main.cu:781); with live_ws only over the segments that carry a loss gradient
  const bool use_live = b->live_ws != nullptr;
  if (use_live) {
    RTXN_REQUIRE(((uintptr_t)b->live_ws & 15) == 0, "rtxn_train_gradients: live_ws not 16-byte aligned");
    rc = live_segments_impl(b->radiance_gradients, b->segment_capacity, b->segment_capacity, b->live_ws, dc, stream);
    if (rc != RTXN_OK) return rc;
  }
  const int* ll = use_live ? live_list_of(b->live_ws) : nullptr;
  const int* lc = use_live ? live_count_of(b->live_ws) : nullptr;
  if (two_pass) {
    rc = train_forward_impl(m, b->encT, cap_samples, b->workspace, nullptr, nullptr, dc, stream, ll, lc);
    if (rc != RTXN_OK) return rc;
  }
  rc = lean      ? train_backward_lean_impl(m, b->encT, b->output_half, b->radiance_gradients, cap_samples, b->workspace, b->dparams, dc, stream, ll, lc,
                                            fused ? &src : nullptr)
     : recompute ? train_backward_recompute_impl(m, b->encT, b->output_half, b->radiance_gradients, cap_samples, b->dparams, b->dencT, dc, stream, ll, lc)
                 : train_backward_impl(m, b->encT, b->output_half, b->radiance_gradients, cap_samples, b->workspace, b->dparams,
                                       hash ? b->dencT : nullptr, dc, stream, ll, lc);
  if (rc != RTXN_OK) return rc;
  if (hash && !b->skip_table_backward) {
    const SampleSrc bsrc{nullptr, b->start_points, b->end_points, nullptr, b->sample_type == RTXN_SAMPLING_MIDPOINT_WORLD};
    rc = hashgrid_backward_impl(b->grid, bsrc, b->dencT, cap_samples, b->dtable, b->dtable_hashed_half, dc, stream, ll, lc);
  }
  return rc;
}
