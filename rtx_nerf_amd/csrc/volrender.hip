// Alpha-compositing volume renderer, forward and backward.
// Replaces volrender_cuda / volrender_backward_cuda and their launchers
// (reference vol_render/vol_render.cu:19-190).
//
// The reference walks a ray's samples serially in one thread.  Here one
// 64-lane wavefront owns a ray and consumes 64 consecutive samples per step
// (a ray's samples are contiguous: [indices[r]*K, (indices[r]+num_hits[r])*K)):
// coalesced 16-B radiance loads (1 KiB per wave instruction), the optical depth
// running sum as a wavefront-wide inclusive prefix scan carried across steps,
// and a final cross-lane reduction of the weighted colour.
//
// HBM-bound.  Algorithmic bytes: forward 20 B/sample (16 radiance + 4 t) + 20
// B/ray; backward 20 B/sample read + 8 B/sample written + 14 B/ray.
//
// RTXN_VR_COMPAT reproduces the reference arithmetic:
//   delta_i = |t_i - t_{i-1}|, t_{-1} = 0, NOT reset at segment boundaries;
//   T_i = sum_{k<=i} delta_k sigma_k (inclusive); w_i = exp(-T_i)(1-exp(-delta_i sigma_i)).
// The scan sums in a different order than the reference's serial loop, hence
// the 1e-5 absolute tolerance on pixels stated in tests/.
// RTXN_VR_NERF is the canonical quadrature (exclusive transmittance, ray_hit
// holds each sample's world-space step length) with its exact gradient.
#include "common.h"

namespace {

// Inclusive prefix sum over the 64 lanes in six DPP adds (row_shr 1/2/4/8 inside each row of 16, then row_bcast:15 into rows
// 1 and 3 and row_bcast:31 into rows 2 and 3): no LDS crossbar round trips -- the __shfl_up form (six dependent
// ds_bpermute_b32) was most of a compositor step's latency (tools/probe/dpp_scan_probe.hip checks the lane pattern).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_term(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_incl_scan_f(float v, int /*lane*/) {
  v += dpp_term<0x111, 0xf>(v);
  v += dpp_term<0x112, 0xf>(v);
  v += dpp_term<0x114, 0xf>(v);
  v += dpp_term<0x118, 0xf>(v);
  v += dpp_term<0x142, 0xa>(v);
  v += dpp_term<0x143, 0xc>(v);
  return v;
}
__device__ __forceinline__ float lane63(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Sum over the wave, returned in every lane: the DPP inclusive scan above leaves the total in lane 63, read back as a scalar
// (six VALU adds + v_readlane; the __shfl_xor butterfly was six dependent ds_swizzle / ds_bpermute round trips, three times
// per ray in the compositors).
__device__ __forceinline__ float wave_sum(float v) { return lane63(wave_incl_scan_f(v, 0)); }
// lane - 1's value (lane 0: 0): DPP wave_shr:1
__device__ __forceinline__ float lane_below(float v) { return dpp_term<0x138, 0xf>(v); }

struct alignas(8) half4 {
  __half x, y, z, w;
};

// COMPACT: radiance is half[N][4] (the network's own output, 8 B/sample instead of 16) and per-sample t_vals are not read at
// all.  RTXN_VR_COMPAT: REGULAR sampling makes them the function (i + 1) / K of the sample index (sampler.cu:52-66);
// RTXN_VR_NERF: every sample of a segment has the same world-space step, so `ray_hit` is one float per SEGMENT (4 B per K
// samples).  Same arithmetic from there on, so the pixels are bit-identical to the float4 + t_vals form at 40 % of its bytes.
template <int MODE, bool COMPACT = false>
__global__ __launch_bounds__(256) void volrender_fwd_kernel(const float4* __restrict__ radiance,
                                                            const int* __restrict__ num_hits,
                                                            const int* __restrict__ indices,
                                                            const float* __restrict__ ray_hit, int batch_size, int K,
                                                            float* __restrict__ pixels) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= batch_size) return;
  const long base = (long)indices[ray] * K;
  const long n = (long)num_hits[ray] * K;
  float T_carry = 0.0f, t_carry = 0.0f;
  float ar = 0.0f, ag = 0.0f, ab = 0.0f;
  for (long s0 = 0; s0 < n; s0 += 64) {
    const bool act = s0 + lane < n;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    float t = 0.0f;
    if (act) {
      if (COMPACT) {
        const half4 c16 = reinterpret_cast<const half4*>(radiance)[base + s0 + lane];
        c = make_float4(__half2float(c16.x), __half2float(c16.y), __half2float(c16.z), __half2float(c16.w));
        if (MODE == RTXN_VR_COMPAT) t = (float)((int)((s0 + lane) % K) + 1) * (1.0f / (float)K);
        else t = ray_hit[(base + s0 + lane) / K];       // compact NERF: one step length per SEGMENT
      } else {
        c = radiance[base + s0 + lane];
        t = ray_hit[base + s0 + lane];
      }
    }
    float x, w;
    if (MODE == RTXN_VR_COMPAT) {
      float tp = lane_below(t);
      if (lane == 0) tp = t_carry;
      const float delta = fabsf(t - tp);
      x = act ? delta * c.w : 0.0f;
      const float T = T_carry + wave_incl_scan_f(x, lane);
      w = act ? expf(-T) * (1.0f - expf(-x)) : 0.0f;
      T_carry = lane63(T);
      // last ACTIVE lane's t carries over; inactive lanes only occur in the final step
      t_carry = lane63(t);
    } else {
      x = act ? t * c.w : 0.0f;  // ray_hit = step length
      const float incl = wave_incl_scan_f(x, lane);
      const float T_excl = T_carry + incl - x;
      w = act ? expf(-T_excl) * (1.0f - expf(-x)) : 0.0f;
      T_carry += lane63(incl);
    }
    ar = fmaf(w, c.x, ar);
    ag = fmaf(w, c.y, ag);
    ab = fmaf(w, c.z, ab);
  }
  ar = wave_sum(ar);
  ag = wave_sum(ag);
  ab = wave_sum(ab);
  if (lane == 0) {
    pixels[3 * (long)ray] = ar;
    pixels[3 * (long)ray + 1] = ag;
    pixels[3 * (long)ray + 2] = ab;
  }
}

// Two samples per lane, 128 per step, and the next step's loads in flight while this step is scanned.  The one-sample form
// above has ONE 8/16-byte load outstanding per lane at a time and a serial dependence from step to step (the carried optical
// depth): measured 2.5 TB/s on the compact half4 radiance (0.31 of HBM).  Here a wave instruction moves 1 KiB (compact) or
// 2 x 1 KiB + 512 B (float4 + t) and two steps' worth is outstanding.  Needs an even K (pairs never straddle a ray's end and
// the 16-byte loads stay aligned); the arithmetic is the same for the compact and the float4 form, so their pixels remain
// bit-identical to each other.
// A wave takes kRaysPerWave CONSECUTIVE rays one after the other: 70 % of the bench frame's rays cross no occupied cell, and
// with a wave per ray the launch was bound by wave turnover (640 k waves, most of them exiting at once), not by HBM.
constexpr int kRaysPerWave = 4;

template <int MODE, bool COMPACT>
__global__ __launch_bounds__(256) void volrender_fwd_pair_kernel(const float4* __restrict__ radiance,
                                                                 const int* __restrict__ num_hits,
                                                                 const int* __restrict__ indices,
                                                                 const float* __restrict__ ray_hit, int batch_size, int K,
                                                                 float* __restrict__ pixels) {
  const int lane = threadIdx.x & 63;
  const int ray0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * kRaysPerWave;
  if (ray0 >= batch_size) return;
  const float rK = 1.0f / (float)K;
  struct Pair { float4 c0, c1; float t0, t1; };
  // per-ray CSR records of the wave's rays, fetched together up front (lanes 0..kRaysPerWave-1)
  const int my = ray0 + lane < batch_size && lane < kRaysPerWave ? ray0 + lane : ray0;
  const int idx_l = indices[my], nh_l = num_hits[my];
  float out = 0.0f;                                   // lane 3 r + ch of the wave holds pixel channel ch of ray r
  // sample counts of one ray fit 32 bits (<= 3R segments x K); only the ray's base offset is 64-bit.  K is a power of two in
  // every configuration of the reference (32): the index within the segment is then a mask, not a division.
  const int kmask = (K & (K - 1)) == 0 ? K - 1 : 0;
  const int kshift = kmask ? __builtin_ctz((unsigned)K) : 0;
  auto load = [&](long base, long seg0, int n, int s0, Pair& p) {
    const int i0 = s0 + 2 * lane;
    p.c0 = p.c1 = make_float4(0.f, 0.f, 0.f, 0.f);
    p.t0 = p.t1 = 0.0f;
    if (i0 < n) {
      if (COMPACT) {
        const uint4 raw = *reinterpret_cast<const uint4*>(reinterpret_cast<const half4*>(radiance) + base + i0);   // two half4
        const __half2 a = *reinterpret_cast<const __half2*>(&raw.x), b = *reinterpret_cast<const __half2*>(&raw.y);
        const __half2 c = *reinterpret_cast<const __half2*>(&raw.z), d = *reinterpret_cast<const __half2*>(&raw.w);
        p.c0 = make_float4(__low2float(a), __high2float(a), __low2float(b), __high2float(b));
        p.c1 = make_float4(__low2float(c), __high2float(c), __low2float(d), __high2float(d));
        if (MODE == RTXN_VR_COMPAT) {
          // REGULAR t_vals (sampler.cu:52-66): (i + 1) / K of the index in the segment; i0 is even and K even: no wrap inside a pair
          const int k0 = kmask ? (i0 & kmask) : i0 % K;
          p.t0 = (float)(k0 + 1) * rK;
          p.t1 = (float)(k0 + 2) * rK;
        } else {
          p.t0 = p.t1 = ray_hit[seg0 + (kmask ? i0 >> kshift : i0 / K)];   // the segment's step length (a pair never straddles two)
        }
      } else {
        p.c0 = radiance[base + i0];
        p.c1 = radiance[base + i0 + 1];
        const float2 tt = *reinterpret_cast<const float2*>(ray_hit + base + i0);
        p.t0 = tt.x;
        p.t1 = tt.y;
      }
    }
  };
  // The frame has few samples per ray (70 % of the bench frame's rays have none, the rest ~500), so a wave's time is its
  // chain of dependent memory round trips: the first step of ALL its rays is requested before any ray is composited.
  long base_r[kRaysPerWave], seg_r[kRaysPerWave];
  int n_r[kRaysPerWave];
  Pair first[kRaysPerWave];
#pragma unroll
  for (int r = 0; r < kRaysPerWave; ++r) {
    seg_r[r] = (long)__shfl(idx_l, r, 64);
    base_r[r] = seg_r[r] * K;
    n_r[r] = ray0 + r < batch_size ? __shfl(nh_l, r, 64) * K : 0;     // even
    if (n_r[r] > 0) load(base_r[r], seg_r[r], n_r[r], 0, first[r]);
  }
#pragma unroll
  for (int r = 0; r < kRaysPerWave; ++r) {
    if (ray0 + r >= batch_size) break;
    const long base = base_r[r];
    const int n = n_r[r];
    float T_carry = 0.0f, t_carry = 0.0f;
    float ar = 0.0f, ag = 0.0f, ab = 0.0f;
    Pair cur = first[r], nxt;
    for (int s0 = 0; s0 < n; s0 += 128) {
      if (s0 + 128 < n) load(base, seg_r[r], n, s0 + 128, nxt);          // in flight under this step's scan
      const bool act = s0 + 2 * lane < n;
      float x0, x1, w0, w1;
      if (MODE == RTXN_VR_COMPAT) {
        float tp = lane_below(cur.t1);          // previous sample's t: the neighbour lane's second sample
        if (lane == 0) tp = t_carry;
        x0 = act ? fabsf(cur.t0 - tp) * cur.c0.w : 0.0f;
        x1 = act ? fabsf(cur.t1 - cur.t0) * cur.c1.w : 0.0f;
        const float pr = x0 + x1;
        const float T0 = (T_carry + (wave_incl_scan_f(pr, lane) - pr)) + x0;   // inclusive optical depth at sample 0 of the pair
        const float T1 = T0 + x1;
        // three hardware exponentials per pair (v_exp_f32 on x log2 e; exp(-T1) = exp(-T0) exp(-x1)) instead of four libm
        // expf of ~12 instructions each: the compact form of this kernel is bound by its instruction count, not by HBM
        const float e0 = __expf(-T0), ex0 = __expf(-x0), ex1 = __expf(-x1);
        w0 = act ? e0 * (1.0f - ex0) : 0.0f;
        w1 = act ? (e0 * ex1) * (1.0f - ex1) : 0.0f;
        T_carry = lane63(T1);
        t_carry = lane63(cur.t1);             // inactive lanes only occur in the final step
      } else {
        x0 = act ? cur.t0 * cur.c0.w : 0.0f;          // ray_hit = step length
        x1 = act ? cur.t1 * cur.c1.w : 0.0f;
        const float pr = x0 + x1;
        const float incl = wave_incl_scan_f(pr, lane);
        const float T0 = T_carry + (incl - pr);       // exclusive transmittance exponent of sample 0
        const float e0 = __expf(-T0), ex0 = __expf(-x0), ex1 = __expf(-x1);
        w0 = act ? e0 * (1.0f - ex0) : 0.0f;
        w1 = act ? (e0 * ex0) * (1.0f - ex1) : 0.0f;
        T_carry += lane63(incl);
      }
      ar = fmaf(w1, cur.c1.x, fmaf(w0, cur.c0.x, ar));
      ag = fmaf(w1, cur.c1.y, fmaf(w0, cur.c0.y, ag));
      ab = fmaf(w1, cur.c1.z, fmaf(w0, cur.c0.z, ab));
      cur = nxt;
    }
    if (n > 0) {                                      // wave-uniform
      ar = wave_sum(ar);
      ag = wave_sum(ag);
      ab = wave_sum(ab);
    }
    if (lane == 3 * r) out = ar;
    if (lane == 3 * r + 1) out = ag;
    if (lane == 3 * r + 2) out = ab;
  }
  // the wave's pixels are 3 * kRaysPerWave consecutive floats: one store
  if (lane < 3 * kRaysPerWave && ray0 + lane / 3 < batch_size) pixels[3 * (long)ray0 + lane] = out;
}

// COMPAT backward: per-sample, reference vol_render.cu:75-143 (not the analytic
// gradient of the forward).
__global__ __launch_bounds__(256) void volrender_bwd_compat_kernel(const __half* __restrict__ loss_gradients,
                                                                   const float4* __restrict__ radiance,
                                                                   const float* __restrict__ t_hit,
                                                                   const int* __restrict__ num_hits,
                                                                   const int* __restrict__ indices, int batch_size,
                                                                   int K, half4* __restrict__ grads) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= batch_size) return;
  const long base = (long)indices[ray] * K;
  const long n = (long)num_hits[ray] * K;
  const float g0 = __half2float(loss_gradients[3 * (long)ray]);
  const float g1 = __half2float(loss_gradients[3 * (long)ray + 1]);
  const float g2 = __half2float(loss_gradients[3 * (long)ray + 2]);
  float t_carry = 0.0f;
  for (long s0 = 0; s0 < n; s0 += 64) {
    const bool act = s0 + lane < n;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    float t = 0.0f;
    if (act) {
      c = radiance[base + s0 + lane];
      t = t_hit[base + s0 + lane];
    }
    float tp = lane_below(t);
    if (lane == 0) tp = t_carry;
    t_carry = lane63(t);
    if (act) {
      const float delta = fabsf(t - tp);
      const float sigma = c.w;
      const float tr = delta * sigma;
      const float e = expf(-sigma * delta);
      float dg = 0.0f;
      dg += g0 * tr * c.x * delta * e;
      dg += g1 * tr * c.y * delta * e;
      dg += g2 * tr * c.z * delta * e;
      const float om = 1.0f - expf(-delta * sigma);
      half4 o;
      o.x = __float2half(g0 * tr * om);
      o.y = __float2half(g1 * tr * om);
      o.z = __float2half(g2 * tr * om);
      o.w = __float2half(dg);
      grads[base + s0 + lane] = o;
    }
  }
}

// NERF backward: exact gradient of the RTXN_VR_NERF forward.
//   C = sum_i T_i a_i c_i,  a_i = 1 - exp(-x_i), x_i = d_i sigma_i, T_i = exp(-sum_{k<i} x_k)
//   dC/dc_i = T_i a_i ;  dC/dsigma_i = d_i ( T_i exp(-x_i) c_i  -  sum_{k>i} T_k a_k c_k )   (per channel, dotted with g)
// Two sweeps per ray: forward sweep accumulates the total S = sum_k w_k (g.c_k); the
// per-sample suffix is S - inclusive_prefix.
__global__ __launch_bounds__(256) void volrender_bwd_nerf_kernel(const __half* __restrict__ loss_gradients,
                                                                 const float4* __restrict__ radiance,
                                                                 const float* __restrict__ step_len,
                                                                 const int* __restrict__ num_hits,
                                                                 const int* __restrict__ indices, int batch_size,
                                                                 int K, half4* __restrict__ grads) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= batch_size) return;
  const long base = (long)indices[ray] * K;
  const long n = (long)num_hits[ray] * K;
  const float g0 = __half2float(loss_gradients[3 * (long)ray]);
  const float g1 = __half2float(loss_gradients[3 * (long)ray + 1]);
  const float g2 = __half2float(loss_gradients[3 * (long)ray + 2]);
  // sweep 1: S = sum_k w_k (g . c_k)
  float T_carry = 0.0f, S = 0.0f;
  for (long s0 = 0; s0 < n; s0 += 64) {
    const bool act = s0 + lane < n;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    float d = 0.0f;
    if (act) {
      c = radiance[base + s0 + lane];
      d = step_len[base + s0 + lane];
    }
    const float x = d * c.w;
    const float incl = wave_incl_scan_f(x, lane);
    const float w = expf(-(T_carry + incl - x)) * (1.0f - expf(-x));
    S += w * (g0 * c.x + g1 * c.y + g2 * c.z);
    T_carry += lane63(incl);
  }
  S = wave_sum(S);
  // sweep 2: per-sample gradients
  T_carry = 0.0f;
  float P_carry = 0.0f;  // inclusive prefix of w_k (g.c_k)
  for (long s0 = 0; s0 < n; s0 += 64) {
    const bool act = s0 + lane < n;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    float d = 0.0f;
    if (act) {
      c = radiance[base + s0 + lane];
      d = step_len[base + s0 + lane];
    }
    const float x = d * c.w;
    const float incl = wave_incl_scan_f(x, lane);
    const float Ti = expf(-(T_carry + incl - x));
    const float ex = expf(-x);
    const float a = 1.0f - ex;
    const float gc = g0 * c.x + g1 * c.y + g2 * c.z;
    const float wgc = Ti * a * gc;
    const float pincl = P_carry + wave_incl_scan_f(wgc, lane);
    if (act) {
      const float suffix = S - pincl;
      half4 o;
      o.x = __float2half(g0 * Ti * a);
      o.y = __float2half(g1 * Ti * a);
      o.z = __float2half(g2 * Ti * a);
      o.w = __float2half(d * (Ti * ex * gc - suffix));
      grads[base + s0 + lane] = o;
    }
    T_carry += lane63(incl);
    P_carry = lane63(pincl);
  }
}

// launch_volrender_cuda + loss->evaluate (L2) + launch_volrender_backward_cuda of one training batch in ONE launch
// (RTXN_VR_NERF).  A ray's loss gradient depends on its own pixel only, and the first sweep of the exact backward --
// S = sum_k w_k (g . c_k) -- is the forward's pixel dotted with g, so the forward sweep serves both: sweep 1 composites the
// pixel, the wave forms d = pixel - target, the L2 value and g = half(loss_scale 2 d / n) (the fp16 rounding
// network->backward sees, main.cu:759), sweep 2 writes the per-sample gradients.  Three launches and one pass over the
// radiance fewer than the separate entry points; same arithmetic per sample (S differs by fp32 rounding: g . pixel
// instead of the per-sample sum).
__global__ __launch_bounds__(256) void volrender_l2_fused_kernel(const float4* __restrict__ radiance, const float* __restrict__ step_len,
                                                                 const int* __restrict__ num_hits, const int* __restrict__ indices,
                                                                 int batch_size, int K, const float* __restrict__ target,
                                                                 float loss_scale, float* __restrict__ pixels,
                                                                 __half* __restrict__ loss_gradients, float* __restrict__ loss_sum,
                                                                 half4* __restrict__ grads) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= batch_size) return;
  const long base = (long)indices[ray] * K;
  const long n = (long)num_hits[ray] * K;
  const float inv_n = 1.0f / (float)(3L * batch_size);
  // sweep 1: the pixel
  float T_carry = 0.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f;
  for (long s0 = 0; s0 < n; s0 += 64) {
    const bool act = s0 + lane < n;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    float d = 0.0f;
    if (act) {
      c = radiance[base + s0 + lane];
      d = step_len[base + s0 + lane];
    }
    const float x = d * c.w;
    const float incl = wave_incl_scan_f(x, lane);
    const float w = act ? expf(-(T_carry + incl - x)) * (1.0f - expf(-x)) : 0.0f;
    ar = fmaf(w, c.x, ar);
    ag = fmaf(w, c.y, ag);
    ab = fmaf(w, c.z, ab);
    T_carry += lane63(incl);
  }
  ar = wave_sum(ar);
  ag = wave_sum(ag);
  ab = wave_sum(ab);
  // L2 (tcnn "L2": values = d^2 / n, gradients = loss_scale * 2 d / n, rounded to fp16)
  const float d0 = ar - target[3 * (long)ray], d1 = ag - target[3 * (long)ray + 1], d2 = ab - target[3 * (long)ray + 2];
  const __half h0 = __float2half(loss_scale * 2.0f * d0 * inv_n), h1 = __float2half(loss_scale * 2.0f * d1 * inv_n),
               h2 = __float2half(loss_scale * 2.0f * d2 * inv_n);
  const float g0 = __half2float(h0), g1 = __half2float(h1), g2 = __half2float(h2);
  if (lane == 0) {
    pixels[3 * (long)ray] = ar;
    pixels[3 * (long)ray + 1] = ag;
    pixels[3 * (long)ray + 2] = ab;
    if (loss_gradients) {
      loss_gradients[3 * (long)ray] = h0;
      loss_gradients[3 * (long)ray + 1] = h1;
      loss_gradients[3 * (long)ray + 2] = h2;
    }
    if (loss_sum) atomicAdd(loss_sum, (d0 * d0 + d1 * d1 + d2 * d2) * inv_n);
  }
  const float S = g0 * ar + g1 * ag + g2 * ab;     // = sum_k w_k (g . c_k)
  // sweep 2: per-sample gradients (volrender_bwd_nerf_kernel's second sweep)
  T_carry = 0.0f;
  float P_carry = 0.0f;
  for (long s0 = 0; s0 < n; s0 += 64) {
    const bool act = s0 + lane < n;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    float d = 0.0f;
    if (act) {
      c = radiance[base + s0 + lane];
      d = step_len[base + s0 + lane];
    }
    const float x = d * c.w;
    const float incl = wave_incl_scan_f(x, lane);
    const float Ti = expf(-(T_carry + incl - x));
    const float ex = expf(-x);
    const float a = 1.0f - ex;
    const float gc = g0 * c.x + g1 * c.y + g2 * c.z;
    const float wgc = Ti * a * gc;
    const float pincl = P_carry + wave_incl_scan_f(wgc, lane);
    if (act) {
      const float suffix = S - pincl;
      half4 o;
      o.x = __float2half(g0 * Ti * a);
      o.y = __float2half(g1 * Ti * a);
      o.z = __float2half(g2 * Ti * a);
      o.w = __float2half(d * (Ti * ex * gc - suffix));
      grads[base + s0 + lane] = o;
    }
    T_carry += lane63(incl);
    P_carry = lane63(pincl);
  }
}

// The training compositor as it runs by default: U blocks of 128 samples (two adjacent samples per lane) per step, their scans
// independent and interleaved, the block offsets chained through scalar registers.  A training batch has few, long rays (the
// configs[2] batch: 12 % of the 4096 rays hit anything, those cross 44 occupied cells on average and up to 128 = 4096
// samples), so the launch lasts as long as the LONGEST ray's chain of dependent steps: 512 samples per step make that 8
// steps per sweep instead of 32.  The loss is reduced per block and added ONCE per block at the very end: a same-address
// atomic per ray sat in front of the second sweep's loads in every wave's in-order memory counter (+45 us of 91).
template <int U>
__global__ __launch_bounds__(256) void volrender_l2_fused_multi_kernel(const float4* __restrict__ radiance, const float* __restrict__ step_len,
                                                                       const int* __restrict__ num_hits, const int* __restrict__ indices,
                                                                       int batch_size, int K, const float* __restrict__ target,
                                                                       float loss_scale, float* __restrict__ pixels,
                                                                       __half* __restrict__ loss_gradients, float* __restrict__ loss_sum,
                                                                       half4* __restrict__ grads) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ray = blockIdx.x * 4 + wave;
  float loss_part = 0.0f;
  if (ray < batch_size) {
    const long base = (long)indices[ray] * K;
    const long n = (long)num_hits[ray] * K;          // even
    const float inv_n = 1.0f / (float)(3L * batch_size);
    constexpr long STEP = 128L * U;
    struct Pair { float4 c0, c1; float d0, d1; };
    auto load = [&](long s0, Pair (&p)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long i0 = s0 + 128 * u + 2 * lane;
        p[u].c0 = p[u].c1 = make_float4(0.f, 0.f, 0.f, 0.f);
        p[u].d0 = p[u].d1 = 0.0f;
        if (i0 < n) {
          p[u].c0 = radiance[base + i0];
          p[u].c1 = radiance[base + i0 + 1];
          const float2 dd = *reinterpret_cast<const float2*>(step_len + base + i0);
          p[u].d0 = dd.x;
          p[u].d1 = dd.y;
        }
      }
    };
    // sweep 1: the pixel
    float T_carry = 0.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f;
    Pair cur[U], nxt[U];
    if (n > 0) load(0, cur);
    for (long s0 = 0; s0 < n; s0 += STEP) {
      if (s0 + STEP < n) load(s0 + STEP, nxt);
      float x0[U], x1[U], pr[U], incl[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        x0[u] = cur[u].d0 * cur[u].c0.w;                 // inactive lanes hold zeros: x = 0, w = 0
        x1[u] = cur[u].d1 * cur[u].c1.w;
        pr[u] = x0[u] + x1[u];
        incl[u] = wave_incl_scan_f(pr[u], lane);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float T0 = T_carry + (incl[u] - pr[u]);
        const float w0 = expf(-T0) * (1.0f - expf(-x0[u])), w1 = expf(-(T0 + x0[u])) * (1.0f - expf(-x1[u]));
        ar = fmaf(w1, cur[u].c1.x, fmaf(w0, cur[u].c0.x, ar));
        ag = fmaf(w1, cur[u].c1.y, fmaf(w0, cur[u].c0.y, ag));
        ab = fmaf(w1, cur[u].c1.z, fmaf(w0, cur[u].c0.z, ab));
        T_carry += lane63(incl[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
    ar = wave_sum(ar);
    ag = wave_sum(ag);
    ab = wave_sum(ab);
    float g0, g1, g2;
    if (target) {
      const float e0 = ar - target[3 * (long)ray], e1 = ag - target[3 * (long)ray + 1], e2 = ab - target[3 * (long)ray + 2];
      const __half h0 = __float2half(loss_scale * 2.0f * e0 * inv_n), h1 = __float2half(loss_scale * 2.0f * e1 * inv_n),
                   h2 = __float2half(loss_scale * 2.0f * e2 * inv_n);
      g0 = __half2float(h0); g1 = __half2float(h1); g2 = __half2float(h2);
      if (lane == 0) {
        pixels[3 * (long)ray] = ar;
        pixels[3 * (long)ray + 1] = ag;
        pixels[3 * (long)ray + 2] = ab;
        if (loss_gradients) {
          loss_gradients[3 * (long)ray] = h0;
          loss_gradients[3 * (long)ray + 1] = h1;
          loss_gradients[3 * (long)ray + 2] = h2;
        }
      }
      loss_part = (e0 * e0 + e1 * e1 + e2 * e2) * inv_n;
    } else {      // rtxn_volrender_bwd: the pixel gradients are given (launch_volrender_backward_cuda's loss_gradients)
      g0 = __half2float(loss_gradients[3 * (long)ray]);
      g1 = __half2float(loss_gradients[3 * (long)ray + 1]);
      g2 = __half2float(loss_gradients[3 * (long)ray + 2]);
    }
    const float S = g0 * ar + g1 * ag + g2 * ab;
    // sweep 2: per-sample gradients (the radiance is re-read: cache hits)
    T_carry = 0.0f;
    float P_carry = 0.0f;
    if (n > 0) load(0, cur);
    for (long s0 = 0; s0 < n; s0 += STEP) {
      if (s0 + STEP < n) load(s0 + STEP, nxt);
      float x0[U], x1[U], pr[U], incl[U], T0[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        x0[u] = cur[u].d0 * cur[u].c0.w;
        x1[u] = cur[u].d1 * cur[u].c1.w;
        pr[u] = x0[u] + x1[u];
        incl[u] = wave_incl_scan_f(pr[u], lane);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        T0[u] = T_carry + (incl[u] - pr[u]);
        T_carry += lane63(incl[u]);
      }
      float Ti0[U], Ti1[U], ex0[U], ex1[U], gc0[U], gc1[U], wgc1[U], pin[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        Ti0[u] = expf(-T0[u]);
        Ti1[u] = expf(-(T0[u] + x0[u]));
        ex0[u] = expf(-x0[u]);
        ex1[u] = expf(-x1[u]);
        gc0[u] = g0 * cur[u].c0.x + g1 * cur[u].c0.y + g2 * cur[u].c0.z;
        gc1[u] = g0 * cur[u].c1.x + g1 * cur[u].c1.y + g2 * cur[u].c1.z;
        const float wgc0 = Ti0[u] * (1.0f - ex0[u]) * gc0[u];
        wgc1[u] = Ti1[u] * (1.0f - ex1[u]) * gc1[u];
        pin[u] = wave_incl_scan_f(wgc0 + wgc1[u], lane);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long i0 = s0 + 128 * u + 2 * lane;
        const float pincl1 = P_carry + pin[u];          // inclusive prefix at the pair's second sample
        const float pincl0 = pincl1 - wgc1[u];
        P_carry += lane63(pin[u]);
        if (i0 < n) {
          const float a0 = 1.0f - ex0[u], a1 = 1.0f - ex1[u];
          half4 o0, o1;
          o0.x = __float2half(g0 * Ti0[u] * a0);
          o0.y = __float2half(g1 * Ti0[u] * a0);
          o0.z = __float2half(g2 * Ti0[u] * a0);
          o0.w = __float2half(cur[u].d0 * (Ti0[u] * ex0[u] * gc0[u] - (S - pincl0)));
          o1.x = __float2half(g0 * Ti1[u] * a1);
          o1.y = __float2half(g1 * Ti1[u] * a1);
          o1.z = __float2half(g2 * Ti1[u] * a1);
          o1.w = __float2half(cur[u].d1 * (Ti1[u] * ex1[u] * gc1[u] - (S - pincl1)));
          uint4 packed;
          packed.x = *reinterpret_cast<const unsigned*>(&o0.x);
          packed.y = *reinterpret_cast<const unsigned*>(&o0.z);
          packed.z = *reinterpret_cast<const unsigned*>(&o1.x);
          packed.w = *reinterpret_cast<const unsigned*>(&o1.z);
          *reinterpret_cast<uint4*>(grads + base + i0) = packed;      // two half4: one 16-byte store
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
  }
  if (loss_sum) {
    if (lane == 0) red[wave] = loss_part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, (red[0] + red[1]) + (red[2] + red[3]));
  }
}

}  // namespace

extern "C" int rtxn_volrender_fwd(const float* network_inputs, const float* network_outputs, const int* num_hits,
                                  const int* indices, const float* ray_hit, int batch_size,
                                  int num_samples_per_hit, float* pixels, int mode, rtxn_stream_t stream) {
  (void)network_inputs;  // unused by the reference as well (vol_render.cu:19-73)
  RTXN_REQUIRE(batch_size >= 0, "rtxn_volrender_fwd: batch_size = %d < 0", batch_size);
  RTXN_REQUIRE(num_samples_per_hit > 0, "rtxn_volrender_fwd: num_samples_per_hit = %d", num_samples_per_hit);
  RTXN_REQUIRE(mode == RTXN_VR_COMPAT || mode == RTXN_VR_NERF, "rtxn_volrender_fwd: unknown mode %d", mode);
  RTXN_DEVICE_OR_FAIL();
  if (batch_size == 0) return RTXN_OK;
  RTXN_REQUIRE(network_outputs && num_hits && indices && ray_hit && pixels, "rtxn_volrender_fwd: NULL buffer");
  RTXN_REQUIRE(((uintptr_t)network_outputs & 15) == 0, "rtxn_volrender_fwd: network_outputs must be 16-byte aligned");
  hipStream_t s = rtxn::as_stream(stream);
  dim3 grid((batch_size + 3) / 4), block(256);
  const float4* rad = reinterpret_cast<const float4*>(network_outputs);
  const bool pairs = num_samples_per_hit % 2 == 0 && ((uintptr_t)ray_hit & 7) == 0;   // two samples per lane (see the kernel)
  dim3 pgrid((batch_size + 4 * kRaysPerWave - 1) / (4 * kRaysPerWave));
  if (mode == RTXN_VR_COMPAT) {
    if (pairs) volrender_fwd_pair_kernel<RTXN_VR_COMPAT, false><<<pgrid, block, 0, s>>>(rad, num_hits, indices, ray_hit, batch_size, num_samples_per_hit, pixels);
    else volrender_fwd_kernel<RTXN_VR_COMPAT><<<grid, block, 0, s>>>(rad, num_hits, indices, ray_hit, batch_size, num_samples_per_hit, pixels);
  } else {
    if (pairs) volrender_fwd_pair_kernel<RTXN_VR_NERF, false><<<pgrid, block, 0, s>>>(rad, num_hits, indices, ray_hit, batch_size, num_samples_per_hit, pixels);
    else volrender_fwd_kernel<RTXN_VR_NERF><<<grid, block, 0, s>>>(rad, num_hits, indices, ray_hit, batch_size, num_samples_per_hit, pixels);
  }
  RTXN_LAUNCH_CHECK("volrender_fwd_kernel");
  return RTXN_OK;
}

extern "C" int rtxn_volrender_fwd_compact(const void* radiance_half4, const int* num_hits, const int* indices, int batch_size,
                                          int num_samples_per_hit, float* pixels, rtxn_stream_t stream) {
  RTXN_REQUIRE(batch_size >= 0, "rtxn_volrender_fwd_compact: batch_size = %d < 0", batch_size);
  RTXN_REQUIRE(num_samples_per_hit > 0, "rtxn_volrender_fwd_compact: num_samples_per_hit = %d", num_samples_per_hit);
  RTXN_DEVICE_OR_FAIL();
  if (batch_size == 0) return RTXN_OK;
  RTXN_REQUIRE(radiance_half4 && num_hits && indices && pixels, "rtxn_volrender_fwd_compact: NULL buffer");
  RTXN_REQUIRE(((uintptr_t)radiance_half4 & 7) == 0, "rtxn_volrender_fwd_compact: radiance must be 8-byte aligned");
  dim3 grid((batch_size + 3) / 4), block(256);
  if (num_samples_per_hit % 2 == 0 && ((uintptr_t)radiance_half4 & 15) == 0)
    volrender_fwd_pair_kernel<RTXN_VR_COMPAT, true><<<dim3((batch_size + 4 * kRaysPerWave - 1) / (4 * kRaysPerWave)), block, 0, rtxn::as_stream(stream)>>>(
        static_cast<const float4*>(radiance_half4), num_hits, indices, nullptr, batch_size, num_samples_per_hit, pixels);
  else
    volrender_fwd_kernel<RTXN_VR_COMPAT, true><<<grid, block, 0, rtxn::as_stream(stream)>>>(
        static_cast<const float4*>(radiance_half4), num_hits, indices, nullptr, batch_size, num_samples_per_hit, pixels);
  RTXN_LAUNCH_CHECK("volrender_fwd_kernel<compact>");
  return RTXN_OK;
}

extern "C" int rtxn_volrender_fwd_compact_nerf(const void* radiance_half4, const float* segment_step, const int* num_hits,
                                               const int* indices, int batch_size, int num_samples_per_hit, float* pixels,
                                               rtxn_stream_t stream) {
  RTXN_REQUIRE(batch_size >= 0, "rtxn_volrender_fwd_compact_nerf: batch_size = %d < 0", batch_size);
  RTXN_REQUIRE(num_samples_per_hit > 0, "rtxn_volrender_fwd_compact_nerf: num_samples_per_hit = %d", num_samples_per_hit);
  RTXN_DEVICE_OR_FAIL();
  if (batch_size == 0) return RTXN_OK;
  RTXN_REQUIRE(radiance_half4 && segment_step && num_hits && indices && pixels, "rtxn_volrender_fwd_compact_nerf: NULL buffer");
  RTXN_REQUIRE(((uintptr_t)radiance_half4 & 7) == 0, "rtxn_volrender_fwd_compact_nerf: radiance must be 8-byte aligned");
  dim3 grid((batch_size + 3) / 4), block(256);
  if (num_samples_per_hit % 2 == 0 && ((uintptr_t)radiance_half4 & 15) == 0)
    volrender_fwd_pair_kernel<RTXN_VR_NERF, true><<<dim3((batch_size + 4 * kRaysPerWave - 1) / (4 * kRaysPerWave)), block, 0, rtxn::as_stream(stream)>>>(
        static_cast<const float4*>(radiance_half4), num_hits, indices, segment_step, batch_size, num_samples_per_hit, pixels);
  else
    volrender_fwd_kernel<RTXN_VR_NERF, true><<<grid, block, 0, rtxn::as_stream(stream)>>>(
        static_cast<const float4*>(radiance_half4), num_hits, indices, segment_step, batch_size, num_samples_per_hit, pixels);
  RTXN_LAUNCH_CHECK("volrender_fwd_kernel<compact, nerf>");
  return RTXN_OK;
}

extern "C" int rtxn_volrender_bwd(const float* loss_values, const void* loss_gradients,
                                  const float* sampled_points_radiance, const float* t_hit, const int* num_hits,
                                  const int* indices, int batch_size, int num_samples_per_hit,
                                  void* radiance_gradients, int mode, rtxn_stream_t stream) {
  (void)loss_values;  // unused by the reference as well (vol_render.cu:75-143)
  RTXN_REQUIRE(batch_size >= 0, "rtxn_volrender_bwd: batch_size = %d < 0", batch_size);
  RTXN_REQUIRE(num_samples_per_hit > 0, "rtxn_volrender_bwd: num_samples_per_hit = %d", num_samples_per_hit);
  RTXN_REQUIRE(mode == RTXN_VR_COMPAT || mode == RTXN_VR_NERF, "rtxn_volrender_bwd: unknown mode %d", mode);
  RTXN_DEVICE_OR_FAIL();
  if (batch_size == 0) return RTXN_OK;
  RTXN_REQUIRE(loss_gradients && sampled_points_radiance && t_hit && num_hits && indices && radiance_gradients,
               "rtxn_volrender_bwd: NULL buffer");
  RTXN_REQUIRE(((uintptr_t)sampled_points_radiance & 15) == 0 && ((uintptr_t)radiance_gradients & 7) == 0,
               "rtxn_volrender_bwd: radiance must be 16-byte and gradients 8-byte aligned");
  hipStream_t s = rtxn::as_stream(stream);
  dim3 grid((batch_size + 3) / 4), block(256);
  const float4* rad = reinterpret_cast<const float4*>(sampled_points_radiance);
  const __half* lg = static_cast<const __half*>(loss_gradients);
  half4* out = static_cast<half4*>(radiance_gradients);
  if (mode == RTXN_VR_COMPAT)
    volrender_bwd_compat_kernel<<<grid, block, 0, s>>>(lg, rad, t_hit, num_hits, indices, batch_size,
                                                       num_samples_per_hit, out);
  else if (num_samples_per_hit % 2 == 0 && ((uintptr_t)t_hit & 7) == 0 && ((uintptr_t)radiance_gradients & 15) == 0)
    // the training compositor with the pixel gradients given instead of formed from a target: 512 samples per step, two per lane
    volrender_l2_fused_multi_kernel<4><<<grid, block, 0, s>>>(rad, t_hit, num_hits, indices, batch_size, num_samples_per_hit, nullptr, 0.0f,
                                                              nullptr, const_cast<__half*>(lg), nullptr, out);
  else
    volrender_bwd_nerf_kernel<<<grid, block, 0, s>>>(lg, rad, t_hit, num_hits, indices, batch_size,
                                                     num_samples_per_hit, out);
  RTXN_LAUNCH_CHECK("volrender_bwd_kernel");
  return RTXN_OK;
}

extern "C" int rtxn_volrender_l2_train(const float* network_outputs, const float* ray_hit, const int* num_hits, const int* indices,
                                       int batch_size, int num_samples_per_hit, const float* target, float loss_scale,
                                       float* pixels, void* loss_gradients_half, float* loss_sum, void* radiance_gradients,
                                       rtxn_stream_t stream) {
  RTXN_REQUIRE(batch_size >= 0, "rtxn_volrender_l2_train: batch_size = %d < 0", batch_size);
  RTXN_REQUIRE(num_samples_per_hit > 0, "rtxn_volrender_l2_train: num_samples_per_hit = %d", num_samples_per_hit);
  RTXN_DEVICE_OR_FAIL();
  hipStream_t s = rtxn::as_stream(stream);
  if (loss_sum) RTXN_HIP(rtxn::zero_words(loss_sum, 1, s));
  if (batch_size == 0) return RTXN_OK;
  RTXN_REQUIRE(network_outputs && ray_hit && num_hits && indices && target && pixels && radiance_gradients,
               "rtxn_volrender_l2_train: NULL buffer");
  RTXN_REQUIRE(((uintptr_t)network_outputs & 15) == 0 && ((uintptr_t)radiance_gradients & 7) == 0,
               "rtxn_volrender_l2_train: radiance must be 16-byte and gradients 8-byte aligned");
  const bool pairs = num_samples_per_hit % 2 == 0 && ((uintptr_t)ray_hit & 7) == 0 && ((uintptr_t)radiance_gradients & 15) == 0;
  if (pairs)
    volrender_l2_fused_multi_kernel<4><<<(batch_size + 3) / 4, 256, 0, s>>>(reinterpret_cast<const float4*>(network_outputs), ray_hit, num_hits,
                                                                        indices, batch_size, num_samples_per_hit, target, loss_scale, pixels,
                                                                        static_cast<__half*>(loss_gradients_half), loss_sum,
                                                                        static_cast<half4*>(radiance_gradients));
  else
    volrender_l2_fused_kernel<<<(batch_size + 3) / 4, 256, 0, s>>>(reinterpret_cast<const float4*>(network_outputs), ray_hit, num_hits,
                                                                   indices, batch_size, num_samples_per_hit, target, loss_scale, pixels,
                                                                   static_cast<__half*>(loss_gradients_half), loss_sum,
                                                                   static_cast<half4*>(radiance_gradients));
  RTXN_LAUNCH_CHECK("volrender_l2_fused_kernel");
  return RTXN_OK;
}
