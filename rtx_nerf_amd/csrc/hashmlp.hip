// Hash-grid inference: launchSampler + HashGrid/Frequency encoding + network->forward + glue as ONE kernel over packed
// segments -- the hash-grid counterpart of mlp_fwd16_kernel<.., segments, half4>.  north_star names "the fully-fused MLP +
// hash-grid encoding"; the reference's own model is Composite-Frequency (main.cu:47-61), so the stage order is what is
// followed here (main.cu:703-737: sampler -> forward -> glue), the encoding is tiny-cuda-nn's published HashGrid (Mueller et
// al. 2022) as oracle/rtxn_oracle.c restates it (orc_encode_hg, orc_mlpe_forward).  PARITY UNPINNED.
//
// What bounds it.  32,768 FLOP per sample (4x64 model) against 16 levels x 8 corners = 128 four-byte table gathers per
// sample: on the bench frame 3.3 TFLOP (2 ms of MFMA) beside 12.8 G gathers.  The kernel is bound by the gather rate of the
// cache hierarchy (the 25-MB table lives in L2 / Infinity Cache, not HBM), so the design is about keeping gathers in flight:
//   * ONE SAMPLE PER LANE while gathering (lane = sample 0..63 of the wave's two segments, exactly the access pattern of
//     hashgrid_encode_f2_kernel): level constants are wave-uniform scalars, hashed/dense is a scalar branch, consecutive
//     lanes are consecutive samples of a segment, so on the coarse and middle levels they share cells and the texture
//     addresser coalesces them; a level's gathers (4 aligned 8-byte pairs + up to 4 singles) are all issued before the
//     first is used.
//   * NO BARRIER in the tile loop.  Every weight of the model (8 KiB per 64x64 layer, 34 KiB for 4x64) is staged into LDS
//     once per block and stays; a wave then runs gather -> transpose -> layers -> epilogue on its own 64 samples and takes the
//     next tile, so the 3 waves of a SIMD drift apart and one wave's MFMA phase covers another's gather latency.
//   * The MFMA pipeline wants the B operand as (16-sample column tile) x (8 features per lane group): the sample-per-lane
//     encoding goes through a 4.5-KiB per-wave LDS strip [level][sample] (row stride 288 B: conflict-free both ways) and comes
//     back in fragment order.  The 16 direction features are computed directly in fragment order (lane group g owns dimension
//     g >> 1, two octaves): no transposition.
//   * Layers: rtxn::pipe_layer16 (v_mfma_f32_16x16x32_f16, all-asm k-steps, activations in registers; mlp_internal.h), the
//     output layer as a plain 16-row tile: lane group 0 ends up with (r, g, b, sigma) of 4 x 16 samples and stores half4.
// Feature values are formed with the arithmetic of hashgrid_encode_f2_kernel / encode_freq_kernel (same fmaf chains, same
// roundings), so this kernel's layer-0 input is bit-identical to the staged encoders' encT.
#include "common.h"

#include <cstring>
#include <mutex>

#include "hashgrid_internal.h"
#include "mlp_internal.h"

namespace {

using rtxn::HgLevels;
using rtxn::hg_index_nodiv;
using rtxn::sin_turns;

constexpr int kWaves = 4, kThreads = 64 * kWaves;
constexpr int kW = 64, kRT = kW / 16, kKS = kW / 32, kCT = 4;
constexpr int kStripStride = 288;                 // bytes per level row of the transposition strip: 64 samples x 4 B + 32
constexpr int kStripBytes = 16 * kStripStride;    // per wave

struct HashMlpArgs {
  HgLevels lv;
  int n_dir_freqs;
  int E;                       // encoded width (multiple of 16, <= 64)
  const uint8_t* table;        // fp16 [n_params]
  const uint8_t* packed;       // A fragments: [layer 0: KS0 x 4][hidden: 2 x 4 each][output: 4 rotations x 2] KiB
  int n_hidden, out_act;
  const float* start;
  const float* end;
  const float* seg_view;
  const int* total_segments;
  long max_segments;
  int midpoint;                // RTXN_SAMPLING_MIDPOINT_WORLD
  float t_scale;
  _Float16* out_half4;         // [segments * 32][4]
  float* seg_step;             // [segments] or NULL
};

// class of one B-fragment dword (two consecutive features) of layer 0, per lane: see classify()
enum : int { kClsHash = 0, kClsDir = 1, kClsOne = 2, kClsZero = 3 };

// The layer stack of one wave tile (64 samples as four 16-column tiles) over weights resident in LDS at `smem`:
// [layer 0: KS0 x 4 KiB | hidden layers: 8 KiB each | output layer: 2 KiB].  bf holds layer 0's B fragments; the output layer's raw
// accumulators come back in acc2[0] (rows 4g .. 4g+3 of sample (ct, c) in lane (c, g)).  Nothing is staged and no barrier is
// taken: the caller's waves run independently.
template <int KS0, int NB>
__device__ __forceinline__ void run_layers(const uint8_t* smem, int n_layers, half8 (&bf)[NB][kCT], half8 (&bg)[NB][kCT],
                                           rtxn::floatx4 (&acc2)[2][kCT], int wave_u, int lane) {
  constexpr int L0_BYTES = KS0 * kRT * 1024, HID_BYTES = kKS * kRT * 1024;
  // pipe_layer16 counts its own LDS reads (s_waitcnt lgkmcnt(N)): every other LDS / scalar-memory access must have landed
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const rtxn::StageJob none{smem, const_cast<uint8_t*>(smem), 0};
  rtxn::pipe_layer16<kRT, KS0, NB, kCT, false>(smem, none, bf, bg, acc2, wave_u, lane);
  const uint8_t* w = smem + L0_BYTES;
  int l = 1;
  for (; l + 1 < n_layers - 1; l += 2) {
    rtxn::pipe_layer16<kRT, kKS, NB, kCT, true>(w, none, bg, bf, acc2, wave_u, lane);
    w += HID_BYTES;
    rtxn::pipe_layer16<kRT, kKS, NB, kCT, true>(w, none, bf, bg, acc2, wave_u, lane);
    w += HID_BYTES;
  }
  if (l < n_layers - 1) {
    rtxn::pipe_layer16<kRT, kKS, NB, kCT, true>(w, none, bg, bf, acc2, wave_u, lane);
    w += HID_BYTES;
    rtxn::pipe_layer16<0, kKS, NB, kCT, true>(w, none, bf, bg, acc2, wave_u, lane);
  } else {
    rtxn::pipe_layer16<0, kKS, NB, kCT, true>(w, none, bg, bf, acc2, wave_u, lane);
  }
}

template <int KS0>
__global__ __launch_bounds__(kThreads, 3) void hashmlp_fwd_kernel(HashMlpArgs a) {
  constexpr int NB = KS0 > kKS ? KS0 : kKS;
  constexpr int L0_BYTES = KS0 * kRT * 1024, HID_BYTES = kKS * kRT * 1024, OUT_BYTES = kKS * 1024;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // [weights | kWaves transposition strips]
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);

  long total_seg = *a.total_segments;
  if (total_seg > a.max_segments) total_seg = a.max_segments;
  const int n_tiles = __builtin_amdgcn_readfirstlane((int)((total_seg + 1) / 2));   // a wave tile = 2 segments = 64 samples
  if ((int)blockIdx.x * kWaves >= n_tiles) return;                                  // whole block idle (block-uniform)

  // ---- all weights -> LDS, once (the output layer: rotation 0 = the layer as it is) ----
  const int n_layers = a.n_hidden + 1;
  const int w_bytes = L0_BYTES + (a.n_hidden - 1) * HID_BYTES + OUT_BYTES;
  rtxn::stage_rt(a.packed, smem, w_bytes, tid);      // layer 0, hidden layers and output rotation 0 are contiguous in `packed`
  rtxn::staged_barrier();
  uint8_t* strip = smem + w_bytes + wave_u * kStripBytes;

  // ---- layer-0 dword classes of this lane (fragment order: k-step s, dword e holds features perm_feature16(s, g, 2e), +1) ----
  const int NH = a.lv.n_levels * 2, WIDTH = NH + 4 * a.n_dir_freqs;
  int cls[KS0][4], arg[KS0][4];                       // arg: hash level, or direction pair index q (dimension q / DF, octave q % DF)
#pragma unroll
  for (int s = 0; s < KS0; ++s)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int feat = rtxn::perm_feature16(s, g, 2 * e);
      if (feat < NH) { cls[s][e] = kClsHash; arg[s][e] = feat >> 1; }
      else if (feat < WIDTH) { cls[s][e] = kClsDir; arg[s][e] = (feat - NH) >> 1; }
      else { cls[s][e] = feat < a.E ? kClsOne : kClsZero; arg[s][e] = 0; }
    }

  typedef float f3v __attribute__((ext_vector_type(3)));
  typedef float f2v __attribute__((ext_vector_type(2)));
  // Segment records of the wave's two segments: this lane's own (sample-per-lane phase: segment lane >> 5) -- fetched one tile
  // ahead, formed at the top of their tile (mlp_fwd16_kernel explains why) -- and both segments' view angles for the fragment-order
  // direction features.
  f3v raw_s, raw_e;
  f2v raw_v[2];
  const int tile_step = (int)gridDim.x * kWaves;
  auto fetch = [&](int tile) {
    const long s0 = 2L * tile, mine = s0 + (lane >> 5);
    const long sg = mine < total_seg ? mine : 0;
    __builtin_memcpy(&raw_s, a.start + 3 * sg, 12);
    __builtin_memcpy(&raw_e, a.end + 3 * sg, 12);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const long q = s0 + k < total_seg ? s0 + k : 0;
      __builtin_memcpy(&raw_v[k], a.seg_view + 2 * q, 8);
    }
  };
  int tile = (int)blockIdx.x * kWaves + wave_u;
  if (tile < n_tiles) fetch(tile);

  for (; tile < n_tiles; tile += tile_step) {
    // ------------------------------------------------------------------ sample of this lane (sampler.cu:52-66 / MIDPOINT_WORLD)
    const long seg_mine = 2L * tile + (lane >> 5);
    const bool ok = seg_mine < total_seg;
    float x01[3];
    {
      const float t = ((float)(lane & 31) + (a.midpoint ? 0.5f : 0.0f)) * (1.0f / 32);
      float dd[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float og = raw_s[k];
        dd[k] = raw_e[k] - og;
        x01[k] = fmaf(fmaf(t, dd[k], og), 0.5f, 0.5f);
      }
      if (a.seg_step && a.midpoint && ok && (lane & 31) == 0) {          // rtxn_sample's MIDPOINT_WORLD t_vals, once per segment
        const float tv = sqrtf(fmaf(dd[2], dd[2], fmaf(dd[0], dd[0], dd[1] * dd[1]))) * (1.0f / 32);
        a.seg_step[seg_mine] = a.t_scale == 1.0f ? tv : tv * a.t_scale;
      }
    }
    const f2v view0 = raw_v[0], view1 = raw_v[1];
    const bool more = tile + tile_step < n_tiles;
    if (more) fetch(tile + tile_step);                 // raw records of the next tile: in flight under this tile's gathers

    // ------------------------------------------------------------------ hash levels, one sample per lane -> strip[level][lane]
    for (int l = 0; l < a.lv.n_levels; ++l) {
      const unsigned res = a.lv.res[l], size = a.lv.size[l];
      const float scale = a.lv.scale[l];
      const bool hashed = (unsigned long long)res * res * res > size;
      const uint8_t* base = a.table + (size_t)a.lv.offset[l] * 4;     // scalar
      float fr[3];
      unsigned gi[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float p = fmaf(x01[k], scale, 0.5f), fl = floorf(p);
        gi[k] = (unsigned)(int)fl;
        fr[k] = p - fl;
      }
      const bool shared = hashed && (gi[0] & 1u) == 0;
      unsigned e8[8], i_lo[4], i_hi[4];
      uint2 pr[4];
      rtxn::hg_corner_indices(gi, res, size, hashed, i_lo, i_hi);
#pragma unroll
      for (int yz = 0; yz < 4; ++yz) pr[yz] = *reinterpret_cast<const uint2*>(base + ((i_lo[yz] & ~1u) << 2));
      unsigned hi[4] = {0u, 0u, 0u, 0u};
      if (!shared) {
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) hi[yz] = *reinterpret_cast<const unsigned*>(base + (i_hi[yz] << 2));
      }
#pragma unroll
      for (int yz = 0; yz < 4; ++yz) {
        e8[2 * yz] = (i_lo[yz] & 1u) ? pr[yz].y : pr[yz].x;
        e8[2 * yz + 1] = shared ? ((i_lo[yz] & 1u) ? pr[yz].x : pr[yz].y) : hi[yz];
      }
      float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
      for (int corner = 0; corner < 8; ++corner) {      // weights ((x) y) z and corner order of hashgrid_encode_kernel
        float w = 1.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) w *= ((corner >> k) & 1) ? fr[k] : 1.0f - fr[k];
        const half2v v = __builtin_bit_cast(half2v, e8[corner]);
        acc0 = fmaf(w, (float)v[0], acc0);
        acc1 = fmaf(w, (float)v[1], acc1);
      }
      asm volatile("" : "+v"(acc0), "+v"(acc1));        // fp32 first, THEN fp16 (no v_fma_mixlo_f16): as the oracle and the staged encoder
      const half2v hv = {(_Float16)acc0, (_Float16)acc1};
      *reinterpret_cast<int*>(strip + l * kStripStride + lane * 4) = __builtin_bit_cast(int, hv);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ------------------------------------------------------------------ layer-0 B fragments (column tile ct = samples 16 ct + c)
    half8 bf[NB][kCT], bg[NB][kCT];
#pragma unroll
    for (int ct = 0; ct < kCT; ++ct) {
      const f2v vw = ct < 2 ? view0 : view1;            // column tiles 0, 1: segment 0; 2, 3: segment 1
#pragma unroll
      for (int s = 0; s < KS0; ++s) {
        rtxn::int4v t;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int d;
          const int k = cls[s][e];
          if (k == kClsHash) {
            d = *reinterpret_cast<const int*>(strip + arg[s][e] * kStripStride + (16 * ct + c) * 4);
          } else if (k == kClsDir) {
            const int q = arg[s][e], dim = q >= a.n_dir_freqs ? 1 : 0, f = q - dim * a.n_dir_freqs;
            const float x = dim ? vw[1] : vw[0];
            const half2v hv = {(_Float16)sin_turns(x, f, 0), (_Float16)sin_turns(x, f, 1)};
            d = __builtin_bit_cast(int, hv);
          } else {
            d = k == kClsOne ? 0x3c003c00 : 0;
          }
          t[e] = d;
        }
        bf[s][ct] = __builtin_bit_cast(half8, t);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    rtxn::floatx4 acc2[2][kCT];
    run_layers<KS0, NB>(smem, n_layers, bf, bg, acc2, wave_u, lane);
    // ------------------------------------------------------------------ epilogue: output rows 4g .. 4g+3 of sample (ct, c) are
    // this lane's accumulator registers; rows 0..3 = (r, g, b, sigma) sit in lane group 0
    if (g == 0) {
#pragma unroll
      for (int ct = 0; ct < kCT; ++ct) {
        const long seg = 2L * tile + (ct >> 1);
        if (seg < total_seg) {
          half4v o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float z = acc2[0][ct][e];
            o[e] = (_Float16)(a.out_act == RTXN_ACT_SIGMOID ? rtxn::sigmoidf_fast(z) : z);
          }
          *reinterpret_cast<half4v*>(a.out_half4 + (seg * 32 + 16 * (ct & 1) + c) * 4) = o;
        }
      }
    }
  }
}

// network->forward on PRE-ENCODED input (main.cu:721 for a model whose encoding ran as its own kernel): the same layer
// stack over encT[E][Sp] (feature-major fp16, as rtxn_hashgrid_encode_segments / rtxn_encode_frequency_segments write it),
// outputs only -- the forward half of the training step's recompute path and, fed by the staged hash encoder, the
// "staged" form the fused kernel above is held to bit for bit (same fragments, same k order, same asm k-steps).
struct EncFwdArgs {
  const uint8_t* packed;
  int n_hidden, out_act, E;
  long S, Sp;
  const int* total_segments;   // NULL: S is the batch's sample count; else the live count is read here (x32), clamped to `capacity`
  int capacity;
  const _Float16* encT;        // [E][Sp]
  _Float16* out_half;          // [S][16]
  float4* radiance;            // [S] or NULL
};

template <int KS0>
__global__ __launch_bounds__(kThreads, 3) void mlp_enc_fwd16_kernel(EncFwdArgs a) {
  constexpr int NB = KS0 > kKS ? KS0 : kKS;
  constexpr int L0_BYTES = KS0 * kRT * 1024, HID_BYTES = kKS * kRT * 1024, OUT_BYTES = kKS * 1024;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  long S = a.S;
  if (a.total_segments) {
    const int t = *a.total_segments;
    S = 32L * (t < a.capacity ? (t < 0 ? 0 : t) : a.capacity);
  }
  const int n_tiles = __builtin_amdgcn_readfirstlane((int)((S + 63) / 64));
  if ((int)blockIdx.x * kWaves >= n_tiles) return;
  const int n_layers = a.n_hidden + 1;
  const int w_bytes = L0_BYTES + (a.n_hidden - 1) * HID_BYTES + OUT_BYTES;
  rtxn::stage_rt(a.packed, smem, w_bytes, tid);
  rtxn::staged_barrier();
  const int tile_step = (int)gridDim.x * kWaves;
  for (int tile = (int)blockIdx.x * kWaves + wave_u; tile < n_tiles; tile += tile_step) {
    half8 bf[NB][kCT], bg[NB][kCT];
    // layer 0's B fragments straight from encT: lane (c, g) takes features perm_feature16(s, g, j) of sample 16 ct + c
    // (16 consecutive lanes = 32 contiguous bytes of a feature row); features >= E meet zero weights and are not read
#pragma unroll
    for (int ct = 0; ct < kCT; ++ct) {
      const long smp = 64L * tile + 16 * ct + c;
      const _Float16* col = a.encT + (smp < S ? smp : 0);
#pragma unroll
      for (int s = 0; s < KS0; ++s) {
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int feat = rtxn::perm_feature16(s, g, j);
          v[j] = feat < a.E ? col[(long)feat * a.Sp] : (_Float16)0.0f;
        }
        bf[s][ct] = v;
      }
    }
    rtxn::floatx4 acc2[2][kCT];
    run_layers<KS0, NB>(smem, n_layers, bf, bg, acc2, wave_u, lane);
    // output rows 4g .. 4g+3 of sample (ct, c) are this lane's four accumulator registers
#pragma unroll
    for (int ct = 0; ct < kCT; ++ct) {
      const long smp = 64L * tile + 16 * ct + c;
      if (smp < S) {
        half4v o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = acc2[0][ct][e];
          o[e] = (_Float16)(a.out_act == RTXN_ACT_SIGMOID ? rtxn::sigmoidf_fast(z) : z);
        }
        *reinterpret_cast<half4v*>(a.out_half + smp * 16 + 4 * g) = o;
        if (a.radiance && g == 0) a.radiance[smp] = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
      }
    }
  }
}

typedef void (*hashmlp_fn)(HashMlpArgs);
typedef void (*encfwd_fn)(EncFwdArgs);

// hipFuncSetAttribute once per (device, kernel, size): never inside a stream capture after the first, un-captured, call
hipError_t set_lds_once(const void* fn, size_t lds) {
  struct Seen { int dev; const void* fn; size_t lds; };
  static std::mutex mu;
  static Seen seen[128];
  static int n_seen = 0;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < n_seen; ++i)
    if (seen[i].dev == dev && seen[i].fn == fn && seen[i].lds >= lds) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess && n_seen < 128) seen[n_seen++] = Seen{dev, fn, lds};
  return e;
}

// persistent grid: as many blocks as stay resident (LDS- and register-bound: 3 per CU for the 4x64 model)
long persistent_grid(long n_wave_tiles, size_t lds, int reserved_cus) {
  int dev = 0, n_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
  const int by_lds = (int)((160 * 1024) / lds), per_cu = by_lds < 3 ? (by_lds > 0 ? by_lds : 1) : 3;
  long grid = (n_wave_tiles + kWaves - 1) / kWaves;
  const long cap = (long)(n_cu - reserved_cus > 1 ? n_cu - reserved_cus : 1) * per_cu;
  if (grid > cap) grid = cap;
  return grid < 1 ? 1 : grid;
}

}  // namespace

extern "C" int rtxn_hashmlp_supported(const rtxn_mlp* m, const rtxn_hashgrid* g, int n_dir_freqs) {
  if (!m || !g) return 0;
  return m->cfg.encoding == RTXN_ENC_EXTERNAL && m->cfg.n_neurons == 64 && m->cfg.n_hidden_layers >= 1 && m->cfg.n_hidden_layers <= 8 &&
         m->enc_padded <= 64 && g->cfg.n_features == 2 && g->cfg.n_levels % 2 == 0 && g->cfg.n_levels <= 16 && n_dir_freqs >= 0 &&
         rtxn_hashgrid_encoded_width(g, n_dir_freqs) == m->enc_padded;
}

extern "C" int rtxn_hashmlp_forward_segments(const rtxn_mlp* m, const rtxn_hashgrid* g, int n_dir_freqs, const void* table_fp16,
                                             const float* start_points, const float* end_points, const float* seg_view,
                                             const int* total_segments, long max_segments, int sample_type, float t_scale,
                                             void* radiance_half4, float* segment_step, rtxn_stream_t stream) {
  RTXN_REQUIRE(m && g, "rtxn_hashmlp_forward_segments: NULL model or grid");
  if (!rtxn_hashmlp_supported(m, g, n_dir_freqs)) {
    rtxn::set_error("rtxn_hashmlp_forward_segments: built for a pre-encoded 64-wide model with 1..8 hidden layers over a hash grid with 2 features "
                    "per level and an even number (<= 16) of levels whose encoded width (<= 64) matches the model's (this call: %d wide, %d layers, "
                    "model width %d; grid %d levels x %d features, %d direction octaves -> %d)", m->cfg.n_neurons, m->cfg.n_hidden_layers,
                    m->enc_padded, g->cfg.n_levels, g->cfg.n_features, n_dir_freqs, rtxn_hashgrid_encoded_width(g, n_dir_freqs));
    return RTXN_ERR_UNSUPPORTED;
  }
  RTXN_REQUIRE(sample_type == RTXN_SAMPLING_REGULAR || sample_type == RTXN_SAMPLING_MIDPOINT_WORLD,
               "rtxn_hashmlp_forward_segments: sample_type %d (the deterministic modes only: REGULAR, MIDPOINT_WORLD)", sample_type);
  RTXN_REQUIRE(max_segments >= 0 && max_segments <= (1L << 30), "rtxn_hashmlp_forward_segments: max_segments = %ld", max_segments);
  RTXN_REQUIRE(m->packed && m->inference_ready, "rtxn_hashmlp_forward_segments: rtxn_mlp_set_params has not been called");
  RTXN_DEVICE_OR_FAIL();
  if (max_segments == 0) return RTXN_OK;
  RTXN_REQUIRE(table_fp16 && start_points && end_points && seg_view && total_segments && radiance_half4,
               "rtxn_hashmlp_forward_segments: NULL buffer");
  RTXN_REQUIRE(((uintptr_t)table_fp16 & 7) == 0 && ((uintptr_t)radiance_half4 & 7) == 0,
               "rtxn_hashmlp_forward_segments: table and radiance must be 8-byte aligned");
  HashMlpArgs a;
  memset(&a, 0, sizeof(a));
  a.lv = rtxn::levels_of(g);
  a.n_dir_freqs = n_dir_freqs;
  a.E = m->enc_padded;
  a.table = static_cast<const uint8_t*>(table_fp16);
  a.packed = static_cast<const uint8_t*>(m->packed);
  a.n_hidden = m->cfg.n_hidden_layers;
  a.out_act = m->cfg.output_activation;
  a.start = start_points;
  a.end = end_points;
  a.seg_view = seg_view;
  a.total_segments = total_segments;
  a.max_segments = max_segments;
  a.midpoint = sample_type == RTXN_SAMPLING_MIDPOINT_WORLD;
  a.t_scale = t_scale;
  a.out_half4 = static_cast<_Float16*>(radiance_half4);
  a.seg_step = segment_step;
  const int KS0 = m->k0 / 32;                                   // 1 or 2
  const hashmlp_fn fn = KS0 == 1 ? hashmlp_fwd_kernel<1> : hashmlp_fwd_kernel<2>;
  const size_t lds = (size_t)(KS0 * kRT + (a.n_hidden - 1) * kKS * kRT + kKS) * 1024 + (size_t)kWaves * kStripBytes;
  RTXN_HIP(set_lds_once(reinterpret_cast<const void*>(fn), lds));
  const long grid = persistent_grid((max_segments + 1) / 2, lds, m->reserved_cus);   // each wave strides over the wave tiles
  hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kThreads), lds, rtxn::as_stream(stream), a);
  RTXN_LAUNCH_CHECK("hashmlp_fwd_kernel");
  return RTXN_OK;
}

// network->forward on pre-encoded input without saved activations (rtxn_mlp_train_forward_outputs, rtxn_train_gradients'
// recompute path): 64-wide models whose weights fit LDS whole.  Non-ABI: called from train.hip.
namespace rtxn {

bool enc_forward16_supported(const rtxn_mlp* m) {
  // RTXN_ENC_EXTERNAL only: a Composite-Frequency model's `packed` holds layer 0 in the fused frequency kernel's own input order
  return m->cfg.encoding == RTXN_ENC_EXTERNAL && m->cfg.n_neurons == 64 && m->cfg.n_hidden_layers >= 1 && m->cfg.n_hidden_layers <= 8 &&
         m->k0 >= 32 && m->k0 <= 64 && m->packed != nullptr;   // wider inputs spill at 3 waves per SIMD
}

int launch_enc_forward16(const rtxn_mlp* m, const void* encT, long n_samples, long Sp, const int* total_segments, int capacity,
                         void* output_half, float* radiance, hipStream_t stream) {
  EncFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.packed = static_cast<const uint8_t*>(m->packed);
  a.n_hidden = m->cfg.n_hidden_layers;
  a.out_act = m->cfg.output_activation;
  a.E = m->enc_padded;
  a.S = n_samples;
  a.Sp = Sp;
  a.total_segments = total_segments;
  a.capacity = capacity;
  a.encT = static_cast<const _Float16*>(encT);
  a.out_half = static_cast<_Float16*>(output_half);
  a.radiance = reinterpret_cast<float4*>(radiance);
  const int KS0 = m->k0 / 32;
  static const encfwd_fn table[2] = {mlp_enc_fwd16_kernel<1>, mlp_enc_fwd16_kernel<2>};
  const encfwd_fn fn = table[KS0 - 1];
  const size_t lds = (size_t)(KS0 * kRT + (a.n_hidden - 1) * kKS * kRT + kKS) * 1024;
  RTXN_HIP(set_lds_once(reinterpret_cast<const void*>(fn), lds));
  const long grid = persistent_grid((n_samples + 63) / 64, lds, 0);
  hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kThreads), lds, stream, a);
  RTXN_LAUNCH_CHECK("mlp_enc_fwd16_kernel");
  return RTXN_OK;
}

}  // namespace rtxn
