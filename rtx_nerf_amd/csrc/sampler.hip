// Per-segment sampler.  Replaces generate_samples / launchSampler
// (reference sampler/sampler.cu:14-131).
//
// The reference runs one thread per ray and writes each 20-byte sample with
// five strided stores.  Here one 64-lane wavefront owns a ray and emits two
// segments (64 samples = 320 contiguous floats = 80 float4) per step: lane l
// writes float4 l of that 1280-byte run (lanes 0..15 a second one), so a store
// instruction covers 1 KiB of contiguous bytes.  Measured 3.4 TB/s on the bench
// frame (0.73 ms for 2.48 GB): bound by wave turnover, not HBM -- 70 % of the rays
// have no segment and the rest five on average.  Two balanced decompositions were
// tried and measured slower: 32 rays per wave over their contiguous segment range
// (2.5 ms: the heaviest wave sets the time) and fixed 64-segment chunks per wave
// with a 64-ary search for the owning ray (1.0 ms: the per-step owner bookkeeping
// costs more than the turnover it saves).  HBM-bound: 792 B per segment algorithmic
// (24 B read, 640 B samples + 128 B t_vals written).
//
// Arithmetic follows the restatement in oracle/rtxn_oracle.c bit for bit
// (explicit fmaf where nvcc's default -fmad contracts; file built with
// -ffp-contract=off).
#include "common.h"

namespace {

constexpr int K = RTXN_NUM_SAMPLES_PER_SEGMENT;

// thrust::minstd_rand: x <- 48271 x mod (2^31-1), default seed 1; the reference
// passes the engine by value (sampler.cu:25,117) so every ray replays the same
// stream, draw number (segment*32 + i + 1) belonging to sample (segment, i).
__device__ __forceinline__ uint32_t mulmod(uint32_t a, uint32_t b) {
  return (uint32_t)(((uint64_t)a * b) % 2147483647ull);
}
__device__ __forceinline__ uint32_t minstd_pow(uint32_t n) {  // 48271^n mod m
  uint32_t r = 1, b = 48271u;
  while (n) {
    if (n & 1u) r = mulmod(r, b);
    b = mulmod(b, b);
    n >>= 1;
  }
  return r;
}
__device__ __forceinline__ float thrust_uniform(uint32_t x, float a, float b) {
  float r = (float)(x - 1u);
  r /= (1.0f + (float)2147483645u);
  return fmaf(r, b - a, a);
}

template <int TYPE>
__global__ __launch_bounds__(256) void sample_kernel(const float* __restrict__ start_points,
                                                     const float* __restrict__ end_points,
                                                     const float* __restrict__ view_dirs,
                                                     float* __restrict__ t_vals, float* __restrict__ samples,
                                                     int batch_size, const int* __restrict__ num_hits,
                                                     const int* __restrict__ indices) {
  __shared__ __attribute__((aligned(16))) float strips[4][64 * 5];
  const int lane = threadIdx.x & 63;
  float* strip = strips[threadIdx.x >> 6];
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= batch_size) return;  // reference reads indices[x] before its guard (sampler.cu:32-37); we do not
  const int start_index = indices[ray];
  const int n_hits = num_hits[ray];
  const float theta = view_dirs[2 * ray], phi = view_dirs[2 * ray + 1];
  const float inc = 1.0f / K;
  const int i = lane & (K - 1);  // sample within segment
  static_assert(K == 32, "the random modes' draw numbering below assumes two 32-sample segments per 64-lane step");
  uint32_t lane_pow = 1, step_pow = 1, pow64 = 1;
  if (TYPE == RTXN_SAMPLING_STRATIFIED_JITTERING || TYPE == RTXN_SAMPLING_UNIFORM) {
    lane_pow = minstd_pow((uint32_t)lane + 1u);
    pow64 = minstd_pow(64u);
  }
  for (int j0 = 0; j0 < n_hits; j0 += 2) {
    const int nseg = min(2, n_hits - j0);
    const long seg0 = (long)start_index + j0;
    // this lane's own sample: (segment j0 + lane/32, i)
    float t, tv;
    if (TYPE == RTXN_SAMPLING_REGULAR) {
      t = (float)i * inc;  // i repeated additions of 2^-5 are exact
      tv = (float)(i + 1) * inc;
    } else if (TYPE == RTXN_SAMPLING_MIDPOINT_WORLD) {
      t = ((float)i + 0.5f) * inc;
      const long g = (seg0 + min(lane >> 5, nseg - 1)) * 3;
      const float dx = end_points[g] - start_points[g], dy = end_points[g + 1] - start_points[g + 1],
                  dz = end_points[g + 2] - start_points[g + 2];
      tv = sqrtf(fmaf(dz, dz, fmaf(dx, dx, dy * dy))) * inc;
    } else {
      // draw number (j0 + lane / 32) * K + i + 1 = 64 (j0 / 2) + lane + 1: 48271^draw = 48271^(lane + 1) (per lane, formed once)
      // times (48271^64)^(j0 / 2) (per wave, one modular multiply per step) -- the same residues as square-and-multiply per
      // sample, which took ~22 modular multiplies each
      const uint32_t x = mulmod(lane_pow, step_pow);  // seed 1
      step_pow = mulmod(step_pow, pow64);
      if (TYPE == RTXN_SAMPLING_UNIFORM) {
        t = thrust_uniform(x, 0.0f, 1.0f);
        tv = 0.0f;
      } else {
        float ti = (float)i * inc, tf = (float)(i + 1) * inc;
        t = thrust_uniform(x, ti, tf);
        tv = ti;
      }
    }
    if ((lane >> 5) < nseg) t_vals[(seg0 + (lane >> 5)) * K + i] = tv;
    // This lane's own sample (segment j0 + lane / 32, i): three multiply-adds from its segment's start and direction (two
    // distinct addresses per load instruction), written as 5 floats into the wave's LDS strip -- stride 5 words: conflict-free.
    // The strip is then read back linearly as float4 and stored: the 2-segment run is 320 contiguous floats = 80 float4, lane
    // l stores float4 l (1 KiB per store instruction), lanes 0..15 a second one.  (The first version had every lane compute
    // the four floats of ITS float4 -- a division by 5, selects between the two segments and a shuffle of t per value,
    // ~100 VALU instructions per step -- and ran at 0.42 of the HBM rate with the stores waiting on the arithmetic.)
    {
      const long g = (seg0 + min(lane >> 5, nseg - 1)) * 3;
      const float ox = start_points[g], oy = start_points[g + 1], oz = start_points[g + 2];
      float* my = strip + lane * 5;
      my[0] = fmaf(t, end_points[g] - ox, ox);
      my[1] = fmaf(t, end_points[g + 1] - oy, oy);
      my[2] = fmaf(t, end_points[g + 2] - oz, oz);
      my[3] = theta;
      my[4] = phi;
    }
    __builtin_amdgcn_wave_barrier();        // one wave, in-order LDS queue: the reads below see the writes above
    float4* out = reinterpret_cast<float4*>(samples + seg0 * (K * 5));
    const int nq = nseg * (K * 5 / 4);
    const float4* lin = reinterpret_cast<const float4*>(strip);
    if (lane < nq) out[lane] = lin[lane];
    if (lane + 64 < nq) out[lane + 64] = lin[lane + 64];
    __builtin_amdgcn_wave_barrier();        // the next step rewrites the strip
  }
}

}  // namespace

extern "C" int rtxn_sample(const float* start_points, const float* end_points, const float* view_dirs,
                           float* t_vals, float* sampled_points, int batch_size, int grid_res,
                           const int* num_hits, const int* indices, int sample_type, rtxn_stream_t stream) {
  (void)grid_res;
  RTXN_REQUIRE(batch_size >= 0, "rtxn_sample: batch_size = %d < 0", batch_size);
  RTXN_REQUIRE(sample_type >= 0 && sample_type <= 3, "rtxn_sample: unknown sample_type %d", sample_type);
  RTXN_DEVICE_OR_FAIL();
  if (batch_size == 0) return RTXN_OK;
  RTXN_REQUIRE(start_points && end_points && view_dirs && t_vals && sampled_points && num_hits && indices,
               "rtxn_sample: NULL buffer");
  hipStream_t s = rtxn::as_stream(stream);
  dim3 grid((batch_size + 3) / 4), block(256);
  switch (sample_type) {
    case RTXN_SAMPLING_REGULAR:
      sample_kernel<RTXN_SAMPLING_REGULAR><<<grid, block, 0, s>>>(start_points, end_points, view_dirs, t_vals,
                                                                   sampled_points, batch_size, num_hits, indices);
      break;
    case RTXN_SAMPLING_STRATIFIED_JITTERING:
      sample_kernel<RTXN_SAMPLING_STRATIFIED_JITTERING><<<grid, block, 0, s>>>(
          start_points, end_points, view_dirs, t_vals, sampled_points, batch_size, num_hits, indices);
      break;
    case RTXN_SAMPLING_UNIFORM:
      sample_kernel<RTXN_SAMPLING_UNIFORM><<<grid, block, 0, s>>>(start_points, end_points, view_dirs, t_vals,
                                                                   sampled_points, batch_size, num_hits, indices);
      break;
    default:
      sample_kernel<RTXN_SAMPLING_MIDPOINT_WORLD><<<grid, block, 0, s>>>(start_points, end_points, view_dirs, t_vals,
                                                                          sampled_points, batch_size, num_hits, indices);
      break;
  }
  RTXN_LAUNCH_CHECK("sample_kernel");
  return RTXN_OK;
}
