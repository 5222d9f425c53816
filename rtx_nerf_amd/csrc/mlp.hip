// Frequency encoding + fully-fused MLP forward on MFMA.
// Replaces the tiny-cuda-nn surface main.cu uses for inference: create_from_config
// (main.cu:35-69,325), n_params/set_params/initialize_params (:327-349),
// network->forward (:721) and the convertHalfToFloat glue (:203-208,723-728); the
// segment variant also folds launchSampler (REGULAR, sampler/sampler.cu:52-66) in.
// tiny-cuda-nn itself is an un-vendored, unpinned submodule: the numerics below are
// this build's restatement of its published algorithm (see oracle/rtxn_oracle.c).
//
// Design (gfx950, wave64, v_mfma_f32_32x32x16_f16)
//   * The network is evaluated TRANSPOSED: H_{l+1}^T [W x samples] = W_l [W x K] . H_l^T.
//     A 32x32 f32 accumulator tile has its column (= sample) on the lane and its rows
//     (= features) in the 16 registers, which is exactly the B-operand layout of the
//     next 32x32x16 MFMA once pairs of registers are packed to f16.  Activations
//     therefore never leave registers between layers: no LDS round trip, no barrier
//     on the activation path.  The k order this imposes (element j of lane-half h of
//     k-step s is feature 16s + 8(j>>2) + 4h + (j&3)) is baked into the weight packing.
//   * Weights are pre-packed (rtxn_mlp_set_params) into 1-KiB "A fragments": chunk
//     (layer, row-tile, k-step) holds lane l's 8 halves at byte l*16, so the LDS image
//     is lane-linear: staged with global_load_lds (16 B/lane, no VGPR round trip) and
//     read back with one conflict-free ds_read_b128 per fragment.
//   * One layer (<= 32 KiB) is resident per LDS buffer; layer l+1 streams into the other
//     buffer while layer l computes.  All blocks read the same 0.03-0.3 MB of packed
//     weights, so the stream is served by L2, not HBM.
//   * A wave owns 64 samples (two 32-column tiles sharing every A fragment); a 256-thread
//     block owns 256 samples; two blocks per CU (2 waves/SIMD from different blocks, so
//     one block's encode/convert VALU phase overlaps the other's MFMA phase).
//   * First layer: the encoding is computed straight into B fragments.  Lane-half h of
//     slot p = 8*kstep + j holds feature 2p+h, i.e. h selects sin/cos of one (dim, freq)
//     pair, so dim and frequency are compile-time per slot.  sin(2^f pi x + h pi/2) =
//     v_sin_f32(fract(x 2^(f-1)) + h/4): the argument reduction is exact.
//
// MFMA-bound: 2*(enc_padded*W + (L-1)*W^2 + 16*W) FLOP per sample (262,144 for the
// reference's 8x128 model), against 28 B/sample of HBM traffic in radiance mode.
#include "common.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include "mlp_internal.h"

namespace {

// Build-time knobs of the 64/128-wide inference kernel (defaults are the measured best; tools/ablate.sh builds variants):
//   RTXN_NW     waves per block, 8 (one block per CU) or 4 (two)            -- mlp_internal.h
//   RTXN_SKEW   1: the two wave groups of an 8-wave block run one stage apart (resident first/last layer, 3-slot ring)
//   RTXN_CT     column tiles per wave, 2 (4: one wave per SIMD with AGPRs, measured slower)
//   RTXN_PIPE   depth of the A-fragment register ring                        -- mlp_internal.h
//   RTXN_SHARE_DIR 1: segment input computes a column tile's direction features once per lane-half (DirShare)
//   RTXN_L0_PLAIN  encode all of layer 0's input up front instead of inside layer 0 (A/B timing)
//   RTXN_ILV16  1: a k-step of the 16x16x32 pipelines is one asm statement per MFMA pair, convert units inside (0: builtin
//               MFMAs + asm units, the form that came out wrong beside extra asm at the stage boundary)      -- mlp_internal.h
//   RTXN_STAMPS diagnostic build: per-stage s_memtime stamps of block 0 (tools/probe/stamps.py); never in the shipped library
// Every variant these knobs select computes the same values (tools/ablate.sh builds them side by side).  The round-1
// timing ablations that broke the results (no encoding / no barriers / no weight staging) are gone from this file; their
// measurements are recorded in DESIGN.md 3.4.
#ifndef RTXN_SKEW
#define RTXN_SKEW 1
#endif
#ifndef RTXN_SHARE_DIR
#define RTXN_SHARE_DIR 1
#endif
using rtxn::pipe_layer;
using rtxn::relu_pack;
using rtxn::stage;


struct FwdArgs {
  const uint8_t* packed;
  int n_hidden;     // hidden layers (>= 1); the first one consumes the encoding
  int out_act;
  // IN_MODE 0
  const float* input;
  long n;
  // IN_MODE 1
  const float* start;
  const float* end;
  const float* seg_view;
  const int* total_segments;
  long max_segments;
  // outputs
  _Float16* out_half;  // [n][16]
  float4* radiance;    // [n]
  float* t_vals;       // [n] or NULL (IN_MODE 1 only)
  // OUT_MODE 2 (IN_MODE 1 only): per-segment partial composite instead of per-sample radiance
  const uint8_t* seg_first;  // [segments] 1 = first segment of its ray (COMPAT only)
  float4* seg_out;           // [segments] (C_r, C_g, C_b, optical depth of the segment)
  int vr_mode;               // RTXN_VR_COMPAT / RTXN_VR_NERF
  float step_scale;          // NERF: world step multiplier (density scale)
};

// Per-segment partial composite of one 32-sample column tile (one segment), lanes col = 0..31 of a half-wave.
// The compositor factorises over segments: pixel = sum_seg exp(-T_before(seg)) * C_seg with
//   C_seg = sum_i w_i c_i,  w_i = exp(-T_loc_i) (1 - exp(-x_i)),  x_i = delta_i sigma_i,
// T_loc inclusive (COMPAT, vol_render.cu:60-63) or exclusive (NERF) WITHIN the segment, so only 16 bytes per
// segment (C_seg, sum x) leave the kernel instead of 20 bytes per sample.
__device__ __forceinline__ float4 seg_composite(float r, float g, float b, float sigma, int col, float d0, float dr,
                                                int vr_mode) {
  const float x = (col == 0 ? d0 : dr) * sigma;
  float incl = x;
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) {
    const float t = __shfl_up(incl, d, 32);
    if (col >= d) incl += t;
  }
  const float Tloc = vr_mode == RTXN_VR_COMPAT ? incl : incl - x;
  const float w = expf(-Tloc) * (1.0f - expf(-x));
  float cr = w * r, cg = w * g, cb = w * b;
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) {
    cr += __shfl_xor(cr, d, 32);
    cg += __shfl_xor(cg, d, 32);
    cb += __shfl_xor(cb, d, 32);
  }
  return make_float4(cr, cg, cb, __shfl(incl, 31, 32));
}

// ---------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------
// params (tcnn layout): layer 0 [W][E], hidden [W][W] x (L-1), out [16][W], row-major fp16 (E = enc_padded).
// A "fragment" is 1 KiB: 64 lanes x 8 halves, lane l = (r = l&31, h = l>>5) at byte 16*l.
//
// MODE 0 (inference): per layer chunks [rowtile][kstep];
//   layer 0  : element (r,h ; kk ; j) = W0[32rt + r][2*(8kk+j) + h]   (sin/cos pair slots; 0 beyond E), K = k0
//   others   : element = Wl[32rt + r][perm_feature(kk,h,j)]            (0 if row >= rows)
// MODE 1 (training forward): as MODE 0 but layer 0 uses perm_feature too, K = E.
// MODE 2 (training backward, TRANSPOSED layers, stored in backward order out, L-1, ..., 0):
//   layer l  : element = Wl[perm_feature(kk,h,j)][32rt + r], rows = in_width(l) padded to 32, K = out rows
//              (16 for the output layer: one k-step; W otherwise).
// MODE 3 (inference on v_mfma_f32_16x16x32_f16, mlp_fwd16_kernel): fragments are 16 rows x 32 k, lane l = (r = l&15,
//   g = l>>4) at byte 16*l, chunks [rowtile16][kstep32];
//   layer 0  : element (r,g ; kk ; j) = W0[16rt + r][enc16_feature(8kk + j, g)]  (0 where there is none), K = k0_16
//   others   : element = Wl[16rt + r][perm_feature16(kk,g,j)]                             (output layer: one 16-row tile)
struct Enc16Dims { int PD, PF, DD, DF, E, k0; };
// Layer-0 input order of the 16x16x32 kernel.  Lane group g (0..3) owns a BLOCK of consecutive frequencies of every
// dimension, FB = ceil(F / 4) of them starting at FB g.  A B-fragment dword is one (dimension, k) pair of the block:
// j-slot 2D holds sin, 2D+1 cos of 2^(FB g + k) pi x -- so a lane evaluates v_sin_f32 / v_cos_f32 once per dimension, at
// k = 0, and gets the block's other octaves by angle doubling (octave_unit).  Dword order: direction dims (DD x FB_D),
// position dims (PD x FB_P), padding dwords (features enc_width.. = 1.0), then nothing.  Frequencies beyond F (block 3 of a
// 10-frequency dimension holds f = 9 only) are computed and meet zero weights.  Returns the tcnn feature index (Composite:
// position block, then direction block, then padding) of j-slot u in lane group g, or -1.
__host__ __device__ inline int enc16_feature(const Enc16Dims& d, int u, int g) {
  const int FBD = (d.DF + 3) / 4, FBP = (d.PF + 3) / 4, ND = d.DD * FBD, NP = d.PD * FBP;
  const int width = 2 * (d.PD * d.PF + d.DD * d.DF);
  const int D = u >> 1, ph = u & 1;
  if (D < ND) { const int dd = D / FBD, f = FBD * g + D % FBD; return f < d.DF ? 2 * d.PD * d.PF + (dd * d.DF + f) * 2 + ph : -1; }
  if (D < ND + NP) { const int v = D - ND, dim = v / FBP, f = FBP * g + v % FBP; return f < d.PF ? (dim * d.PF + f) * 2 + ph : -1; }
  const int feat = width + 4 * (2 * (D - ND - NP) + ph) + g;
  return feat < d.E ? feat : -1;
}

//   output layer: FOUR variants of its single 16-row tile, [variant v][kstep]: variant 0 is the layer as it is (all 16
//   rows; what OUT_MODE 0 uses for every column tile); variant v > 0 holds output rows 0..3 at tile rows 4v..4v+3 and zeros
//   elsewhere, so that column tile v's MFMA leaves ITS (r, g, b, sigma) in lane group v -- the matrix core does the
//   transposition that lets all 64 lanes of a wave run the sigmoid epilogue on one sample each.
__global__ void pack16_kernel(const _Float16* __restrict__ params, _Float16* __restrict__ packed, int W, int n_hidden, Enc16Dims d) {
  const int RT = W / 16, KS = W / 32, KS0 = d.k0 / 32, E = d.E;
  const long l0 = (long)KS0 * RT * 512, hid = (long)KS * RT * 512, outl = 4L * KS * 512;
  const long total = l0 + (long)(n_hidden - 1) * hid + outl;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int layer;
    long q = e;
    if (q < l0) layer = 0;
    else {
      q -= l0;
      layer = 1 + (int)(q / hid);
      if (layer >= n_hidden) { layer = n_hidden; q -= (long)(n_hidden - 1) * hid; } else q -= (long)(layer - 1) * hid;
    }
    const int j = (int)(q & 7), lane = (int)((q >> 3) & 63);
    const long chunk = q >> 9;
    const int r = lane & 15, g = lane >> 4;
    const int ks_count = layer == 0 ? KS0 : KS;
    const int kk = (int)(chunk % ks_count), rt = (int)(chunk / ks_count);   // output layer: rt = rotation variant
    const int in_w = layer == 0 ? E : W, rows = layer == n_hidden ? 16 : W;
    int row = 16 * rt + r;
    if (layer == n_hidden) row = rt == 0 ? r : ((r >> 2) == rt ? (r & 3) : rows);   // variant v: rows 4v..4v+3 <- outputs 0..3
    const int feat = layer == 0 ? enc16_feature(d, 8 * kk + j, g) : rtxn::perm_feature16(kk, g, j);
    const long base = layer == 0 ? 0 : (long)W * E + (long)(layer - 1) * W * W;
    _Float16 v = (_Float16)0.0f;
    if (row < rows && feat >= 0 && feat < in_w) v = params[base + (long)row * in_w + feat];
    packed[e] = v;
  }
}

__global__ void pack_kernel(const _Float16* __restrict__ params, _Float16* __restrict__ packed, int W, int E, int k0,
                            int n_hidden, int mode) {
  const int RT = W / 32;
  const int l0_ks = mode == 0 ? k0 / 16 : E / 16;
  const int rt_e = (E + 31) / 32;
  auto layer_elems = [&](int l) -> long {  // l in forward numbering: 0..n_hidden (n_hidden = output layer)
    if (mode < 2) {
      if (l == 0) return (long)l0_ks * RT * 512;
      if (l < n_hidden) return (long)(W / 16) * RT * 512;
      return (long)(W / 16) * 512;
    }
    if (l == n_hidden) return (long)RT * 1 * 512;                 // rows W, K = 16
    if (l == 0) return (long)rt_e * (W / 16) * 512;               // rows E (padded to 32), K = W
    return (long)RT * (W / 16) * 512;
  };
  auto src_base = [&](int l) -> long {
    if (l == 0) return 0;
    return (long)W * E + (long)(l - 1) * W * W;
  };
  long total = 0;
  for (int l = 0; l <= n_hidden; ++l) total += layer_elems(l);
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    long q = e;
    int layer = -1;
    for (int i = 0; i <= n_hidden; ++i) {
      const int l = mode < 2 ? i : n_hidden - i;  // storage order
      const long n = layer_elems(l);
      if (q < n) { layer = l; break; }
      q -= n;
    }
    const int j = (int)(q & 7), lane = (int)((q >> 3) & 63);
    const long chunk = q >> 9;
    const int r = lane & 31, h = lane >> 5;
    const int in_w = layer == 0 ? E : W;
    const int rows = layer == n_hidden ? 16 : W;
    _Float16 v = (_Float16)0.0f;
    if (mode < 2) {
      const int ks_count = layer == 0 ? l0_ks : W / 16;
      const int kk = (int)(chunk % ks_count), rt = (int)(chunk / ks_count);
      const int row = 32 * rt + r;
      const int feat = (mode == 0 && layer == 0) ? 2 * (8 * kk + j) + h : rtxn::perm_feature(kk, h, j);
      if (row < rows && feat < in_w) v = params[src_base(layer) + (long)row * in_w + feat];
    } else {
      const int ks_count = layer == n_hidden ? 1 : W / 16;
      const int kk = (int)(chunk % ks_count), rt = (int)(chunk / ks_count);
      const int col = 32 * rt + r;                       // input feature of the layer = output row of W^T
      const int row = rtxn::perm_feature(kk, h, j);      // output feature of the layer = k of W^T
      if (row < rows && col < in_w) v = params[src_base(layer) + (long)row * in_w + col];
    }
    packed[e] = v;
  }
}

// ---------------------------------------------------------------------------
// forward kernel
// ---------------------------------------------------------------------------
// Encoding slot p (0..): pair (dim, freq) of Composite(Frequency(PD,PF), Frequency(DD,DF)).
template <int PD, int PF, int DD, int DF>
struct EncSpec {
  static constexpr int n_pairs = PD * PF + DD * DF;
  static constexpr int enc_width = 2 * n_pairs;
  static constexpr int enc_padded = (enc_width + 15) / 16 * 16;
  static constexpr int n_slots = enc_padded / 2;           // slots holding real or padding(=1) features
  static constexpr int k0 = (n_slots + 7) / 8 * 16;        // first-layer K as staged
};

template <class ES, int PD, int PF, int DD, int DF>
__device__ __forceinline__ _Float16 encode_slot(int p, const float (&x)[5], float phase) {
  // p is a compile-time constant after unrolling
  if (p < PD * PF) {
    const int dim = p / PF, f = p % PF;
    const float rev = __builtin_amdgcn_fractf(x[dim] * (0.5f * (float)(1u << f))) + phase;
    return (_Float16)__builtin_amdgcn_sinf(rev);
  } else if (p < ES::n_pairs) {
    const int q = p - PD * PF;
    const int dim = PD + q / DF, f = q % DF;
    const float rev = __builtin_amdgcn_fractf(x[dim] * (0.5f * (float)(1u << f))) + phase;
    return (_Float16)__builtin_amdgcn_sinf(rev);
  } else if (p < ES::n_slots) {
    return (_Float16)1.0f;
  }
  return (_Float16)0.0f;
}

// ---- layer 0 with the encoder inside it -------------------------------------------------------------------------
// Encoding all KS0 B fragments before layer 0 is 108 quarter-rate v_sin_f32 (+ range reduction) per sample pair during
// which the wave issues no MFMA.  Layer 0 therefore runs K-STEP-OUTER: the B fragments of k-step kk+1 are encoded in
// slices behind the MFMAs of k-step kk (all RT row tiles accumulate at once: RT x CT accumulator tiles, but only two
// k-steps of encoded input are ever live instead of KS0).  One encode unit = one dword of a B fragment = slots 2e, 2e+1
// of (kk, ct): v_mul, v_fract, v_add(phase), v_sin per slot -- the same four instructions encode_slot compiles to --
// then v_cvt_pk_f16_f32 (one wait state after the transcendental, which hipcc cannot insert inside asm).
template <int PD, int PF, int DD, int DF, int P>
struct SlotInfo {   // real slot P: input dimension and 2^(f-1)
  static constexpr int f = P < PD * PF ? P % PF : (P - PD * PF) % DF;
  static constexpr int dim = P < PD * PF ? P / PF : PD + (P - PD * PF) / DF;
};
// Segment input (IN_MODE 1): a 32-sample column tile is ONE segment, so its direction features (DD*DF of the n_pairs
// slots: 24 of 54 for the reference model) are the same for all 32 lanes of a lane-half.  Instead of every lane computing
// all of them, lane c of each half computes slot c alone (same four instructions, scale 2^(f-1) from v_ldexp), neighbouring
// lanes pack their two values with one DPP move + v_cvt_pk_f16_f32, and DD*DF/2 ds_bpermute broadcasts hand every lane the
// finished B-fragment dwords -- bit-identical values, 23 fewer v_sin_f32 per lane and column tile.
template <int PD, int PF, int DD, int DF>
struct DirShare {
  static constexpr bool possible = (PD * PF) % 2 == 0 && (DD * DF) % 2 == 0 && DD == 2 && DD * DF <= 32;
  static constexpr int n_dwords = possible ? DD * DF / 2 : 1;
};

template <int PD, int PF, int DD, int DF>
__device__ __forceinline__ void share_direction(float theta, float phi, float phase, int lane, int (&dirs)[DirShare<PD, PF, DD, DF>::n_dwords]) {
  const int c = lane & 31;
  const int q = c < DD * DF ? c : 0;                 // this lane's direction slot
  const float xs = q >= DF ? phi : theta;   // by value: a select on an array element made hipcc index the array in scratch
  const float rev = __builtin_amdgcn_fractf(xs * ldexpf(0.5f, q % DF)) + phase;
  const float v = __builtin_amdgcn_sinf(rev);
  // the neighbour's value (lanes 2j <-> 2j+1): quad_perm [1,0,3,2]
  const float w = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  float2v pr = {v, w};                                // correct order in the even lane of each pair
  const int packed = __builtin_bit_cast(int, __builtin_convertvector(pr, half2v));
#pragma unroll
  for (int j = 0; j < DirShare<PD, PF, DD, DF>::n_dwords; ++j)
    dirs[j] = __builtin_amdgcn_ds_bpermute(4 * ((lane & 32) + 2 * j), packed);
}

// dword E (0..3) of the B fragment of k-step KK for one column tile
template <class ES, int PD, int PF, int DD, int DF, bool SHARE, int KK, int E>
__device__ __forceinline__ int encode_unit(const float (&x)[5], float phase, const int (&dirs)[DirShare<PD, PF, DD, DF>::n_dwords]) {
  constexpr int p0 = 8 * KK + 2 * E, p1 = p0 + 1;
  if constexpr (SHARE && p0 >= PD * PF && p1 < ES::n_pairs) {
    return dirs[(p0 - PD * PF) / 2];
  } else if constexpr (p1 < ES::n_pairs) {
    int r;
    float t0, t1;
    using S0 = SlotInfo<PD, PF, DD, DF, p0>;
    using S1 = SlotInfo<PD, PF, DD, DF, p1>;
    const float c0 = 0.5f * (float)(1u << S0::f), c1 = 0.5f * (float)(1u << S1::f);
    asm volatile(
        "v_mul_f32 %1, %4, %3\n\t"
        "v_mul_f32 %2, %6, %5\n\t"
        "v_fract_f32 %1, %1\n\t"
        "v_fract_f32 %2, %2\n\t"
        "v_add_f32 %1, %1, %7\n\t"
        "v_add_f32 %2, %2, %7\n\t"
        "v_sin_f32 %1, %1\n\t"
        "v_sin_f32 %2, %2\n\t"
        "s_nop 0\n\t"
        "v_cvt_pk_f16_f32 %0, %1, %2"
        : "=v"(r), "=&v"(t0), "=&v"(t1)
        : "v"(x[S0::dim]), "s"(c0), "v"(x[S1::dim]), "s"(c1), "v"(phase));
    return r;
  } else {
    // padding slots (1.0 up to enc_padded, then 0) -- or a real slot next to a padding one, which the plain path handles
    half2v v;
    v[0] = encode_slot<ES, PD, PF, DD, DF>(p0, x, phase);
    v[1] = encode_slot<ES, PD, PF, DD, DF>(p1, x, phase);
    return __builtin_bit_cast(int, v);
  }
}
template <class ES, int PD, int PF, int DD, int DF, int CT, bool SHARE, int KK, int U0, int U1>
__device__ __forceinline__ void encode_units(const float (&xin)[CT][5], float phase, half8 (&b)[CT],
                                             const int (&dirs)[CT][DirShare<PD, PF, DD, DF>::n_dwords]) {
  if constexpr (U0 < U1) {   // unit U: column tile U / 4, dword U % 4
    constexpr int ct = U0 / 4, e = U0 % 4;
    rtxn::int4v t = __builtin_bit_cast(rtxn::int4v, b[ct]);
    t[e] = encode_unit<ES, PD, PF, DD, DF, SHARE, KK, e>(xin[ct], phase, dirs[ct]);
    b[ct] = __builtin_bit_cast(half8, t);
    encode_units<ES, PD, PF, DD, DF, CT, SHARE, KK, U0 + 1, U1>(xin, phase, b, dirs);
  }
}

template <class ES, int PD, int PF, int DD, int DF, int RT, int KS0, int NB, int CT, bool SHARE, int I>
struct Layer0Step {
  using DirTab = int[CT][DirShare<PD, PF, DD, DF>::n_dwords];
  static constexpr int D = RTXN_PIPE, N = RT * KS0, WAVES = RTXN_NW;
  static constexpr int CHUNKS = N < 32 / WAVES ? N : 32 / WAVES;
  static constexpr int UE = (4 * CT + RT - 1) / RT;   // encode units per (kk, rt) sub-step
  __device__ static __forceinline__ void run(unsigned addr, const float (&xin)[CT][5], float phase, half8 (&b)[2][CT],
                                             half8 (&out)[NB][CT], half8 (&ring)[D], floatx16 (&acc)[RT][CT],
                                             const rtxn::StageJob& sj, int wave_u, int lane, const DirTab& dirs) {
    constexpr int kk = I / RT, rt = I % RT, cur = kk & 1;
    constexpr int outstanding = (N - 1 - I) < (D - 1) ? (N - 1 - I) : (D - 1);
    rtxn::lds_wait<outstanding>();
    const half8 a = ring[I % D];
    if (kk == 0) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[rt][ct][e] = 0.0f;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[cur][ct], acc[rt][ct], 0, 0, 0);
    if constexpr (kk + 1 < KS0) {
      constexpr int u0 = rt * UE < 4 * CT ? rt * UE : 4 * CT, u1 = (rt + 1) * UE < 4 * CT ? (rt + 1) * UE : 4 * CT;
      encode_units<ES, PD, PF, DD, DF, CT, SHARE, kk + 1, u0, u1>(xin, phase, b[cur ^ 1], dirs);
    } else if constexpr (rt > 0) {
      rtxn::convert_units<NB, CT, 0, 8 * CT>(acc[rt - 1], out, 2 * (rt - 1));   // last k-step: row tile rt-1 is complete
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I + D < N) rtxn::lds_read_frag<(((I + D) % RT) * KS0 + (I + D) / RT) * 1024>(ring[I % D], addr);
    if constexpr (I < CHUNKS) {
      rtxn::stage_chunk<I, WAVES>(sj, wave_u, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (I + 1 < N)
      Layer0Step<ES, PD, PF, DD, DF, RT, KS0, NB, CT, SHARE, I + 1>::run(addr, xin, phase, b, out, ring, acc, sj, wave_u, lane, dirs);
  }
};

// Leaves row tiles 0..RT-2 converted in out[] and the last one pending in pend[] (= acc2[1] of the kernel).
template <class ES, int PD, int PF, int DD, int DF, int RT, int KS0, int NB, int CT, bool SHARE>
__device__ __forceinline__ void pipe_layer0(const uint8_t* lds_buf, const rtxn::StageJob& sj, const float (&xin)[CT][5], float phase,
                                            const int (&dirs)[CT][DirShare<PD, PF, DD, DF>::n_dwords],
                                            half8 (&out)[NB][CT], floatx16 (&pend)[CT], int wave_u, int lane) {
  constexpr int D = RTXN_PIPE, N = RT * KS0;
  static_assert(N * 1024 <= 65535 + 1024, "fragment offsets must fit the 16-bit ds offset");
  half8 ring[D];
  const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)lds_buf + lane * 16;
  // fragment of sub-step I = (kk, rt) sits at ((rt * KS0 + kk) * 64 + lane) * 16
  rtxn::lds_read_frag<0>(ring[0], addr);
  if constexpr (D > 1) rtxn::lds_read_frag<((1 % RT) * KS0 + 1 / RT) * 1024>(ring[1 % D], addr);
  if constexpr (D > 2) rtxn::lds_read_frag<((2 % RT) * KS0 + 2 / RT) * 1024>(ring[2 % D], addr);
  if constexpr (D > 3) rtxn::lds_read_frag<((3 % RT) * KS0 + 3 / RT) * 1024>(ring[3 % D], addr);
  half8 b[2][CT];
  encode_units<ES, PD, PF, DD, DF, CT, SHARE, 0, 0, 4 * CT>(xin, phase, b[0], dirs);   // k-step 0: nothing to hide behind yet
  floatx16 acc[RT][CT];
  Layer0Step<ES, PD, PF, DD, DF, RT, KS0, NB, CT, SHARE, 0>::run(addr, xin, phase, b, out, ring, acc, sj, wave_u, lane, dirs);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) pend[ct] = acc[RT - 1][ct];
}

// CT = 32-sample column tiles per wave.  CT = 2: 4 waves x 64 samples, 2 blocks/CU (2 waves/SIMD, ~244 VGPRs), every
// A fragment feeds two MFMAs.  CT = 1: 8 waves x 32 samples, 2 blocks/CU (4 waves/SIMD, <= 128 VGPRs): more waves to
// cover each other's encode/convert phases, one LDS read per MFMA.
template <int W, int PD, int PF, int DD, int DF, int IN_MODE, int OUT_MODE, int CT>
__global__ __launch_bounds__(64 * RTXN_NW, CT == 2 ? 2 : 1) void mlp_fwd_kernel(FwdArgs a) {
  constexpr int THREADS = 64 * RTXN_NW;
  constexpr int TILE = 32 * RTXN_NW * CT, TILE_SEGS = RTXN_NW * CT;   // samples / segments per block per iteration
  using ES = EncSpec<PD, PF, DD, DF>;
  constexpr int RT = W / 32, KS = W / 16, KS0 = ES::k0 / 16;
  constexpr int NB = KS0 > KS ? KS0 : KS;
  constexpr int L0_BYTES = KS0 * RT * 1024, HID_BYTES = KS * RT * 1024, OUT_BYTES = KS * 1024;
  constexpr int BUF = L0_BYTES > HID_BYTES ? L0_BYTES : HID_BYTES;
  // SKEW (8-wave blocks): waves 0-3 (group A) and 4-7 (group B, the other wave of each SIMD) run the same program one
  // stage apart, so that one group's VALU-bound layer 0 (the encoder) always overlaps the other group's MFMA-bound
  // stages instead of both hitting it together.  The offset costs nothing to arrange: B executes one extra barrier
  // before its first stage and A one after its last -- s_barrier only counts arrivals.  What it needs is LDS: layer 0
  // and the output layer stay resident (fetched once per launch), hidden layers stream through a ring of THREE slots
  // (A's stage, B's stage, the one being fetched); every wave fetches its share of the stage group A needs NEXT.
  constexpr bool SKEW = RTXN_SKEW && RTXN_NW == 8;
  constexpr int RES_BYTES = L0_BYTES + OUT_BYTES;   // SKEW: [layer 0 | output layer | 3 x HID_BYTES]
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // SKEW ? RES_BYTES + 3 * HID_BYTES : 2 * BUF

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // the same number, known to hipcc as wave-uniform
  static_assert(CT == 2 || CT == 4, "pipe_layer: 4 waves x CT column tiles");
  long n_tiles;
  long total_seg = 0;
  if (IN_MODE == 1) {
    total_seg = *a.total_segments;
    if (total_seg > a.max_segments) total_seg = a.max_segments;
    n_tiles = (total_seg + TILE_SEGS - 1) / TILE_SEGS;
  } else {
    n_tiles = (a.n + TILE - 1) / TILE;
  }
  if ((long)blockIdx.x >= n_tiles) return;

  const int n_layers = a.n_hidden + 1;  // hidden layers + output layer
  // byte offset of layer l in the packed buffer
  auto layer_off = [&](int l) -> long {
    if (l == 0) return 0;
    return (long)L0_BYTES + (long)(l - 1) * HID_BYTES;
  };

  const int grp = SKEW ? wave_u >> 2 : 0;   // 0: leading group, 1: one stage behind
  const int n_hid = n_layers - 2;           // streamed (hidden) stages per tile
  // prologue: layer 0 of the first tile (SKEW: the two resident layers)
  stage<L0_BYTES, THREADS>(a.packed, smem, tid);
  if (SKEW) stage<OUT_BYTES, THREADS>(a.packed + layer_off(n_layers - 1), smem + L0_BYTES, tid);
  int q = 0;  // global stage counter: buffer = q & 1 (SKEW: hidden-stage counter of this wave, slot = q % 3)

  // Inputs are fetched ONE TILE AHEAD: the loads for tile t+1 are issued right after tile t's
  // encoding and have the whole layer stack of tile t to land (the first barrier drains them).
  float xin[CT][5];
  float d0_n[CT], dr_n[CT];  // OUT_MODE 2: step of sample 0 / of the other samples of the segment
  // sample index / validity of this lane's column in column tile ct of a tile: recomputed where needed (the epilogue)
  // rather than carried through the layer stack in registers
  auto sample_of = [&](long tile, int ct, bool& valid) -> long {
    if (IN_MODE == 1) {
      const long seg = tile * TILE_SEGS + wave_u * CT + ct;
      valid = seg < total_seg;
      return seg * 32 + col;
    }
    const long sidx = tile * TILE + wave_u * (32 * CT) + ct * 32 + col;
    valid = sidx < a.n;
    return sidx;
  };
  // fetched raw one tile ahead, formed into samples at the top of their own tile (mlp_fwd16_kernel explains why: forming them
  // where the loads are issued makes the wave wait for HBM there, once per tile)
  typedef float f3v __attribute__((ext_vector_type(3)));
  typedef float f2v __attribute__((ext_vector_type(2)));
  f3v raw_s[CT], raw_e[CT];                 // segment input: start, end, view (and first-of-ray flag) of the wave's CT segments
  f2v raw_v[CT];
  unsigned char raw_first[CT];
  float raw_x[IN_MODE == 1 ? 1 : CT][5];    // sample input
  auto fetch_inputs = [&](long tile) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      bool valid_in;
      const long samp_in = sample_of(tile, ct, valid_in);
      if (IN_MODE == 1) {
        const long sg = valid_in ? (samp_in >> 5) : 0;
        __builtin_memcpy(&raw_s[ct], a.start + 3 * sg, 12);
        __builtin_memcpy(&raw_e[ct], a.end + 3 * sg, 12);
        __builtin_memcpy(&raw_v[ct], a.seg_view + 2 * sg, 8);
        raw_first[ct] = 0;
        if (OUT_MODE == 2 && a.vr_mode == RTXN_VR_COMPAT) raw_first[ct] = a.seg_first[sg];
      } else {
        const long sidx = valid_in ? samp_in : 0;
#pragma unroll
        for (int c = 0; c < 5; ++c) raw_x[ct][c] = a.input[5 * sidx + c];
      }
    }
  };
  auto form_inputs = [&]() {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      if (IN_MODE == 1) {
        const bool mid = OUT_MODE == 2 && a.vr_mode == RTXN_VR_NERF;   // NERF composite samples sub-interval midpoints
        const float t = ((float)col + (mid ? 0.5f : 0.0f)) * (1.0f / 32);
        float dd[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float og = raw_s[ct][c];
          dd[c] = raw_e[ct][c] - og;
          xin[ct][c] = fmaf(t, dd[c], og);   // REGULAR sample, sampler.cu:52-66
        }
        const float len2 = fmaf(dd[2], dd[2], fmaf(dd[0], dd[0], dd[1] * dd[1]));   // as the MIDPOINT_WORLD sampler
        xin[ct][3] = raw_v[ct][0];
        xin[ct][4] = raw_v[ct][1];
        if (OUT_MODE == 2) {
          if (a.vr_mode == RTXN_VR_COMPAT) {
            dr_n[ct] = 1.0f / 32;
            d0_n[ct] = raw_first[ct] ? 1.0f / 32 : 31.0f / 32;   // t_prev is not reset per segment (vol_render.cu:56)
          } else {
            d0_n[ct] = dr_n[ct] = sqrtf(len2) * (1.0f / 32) * a.step_scale;
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < 5; ++c) xin[ct][c] = raw_x[ct][c];
      }
    }
  };
  fetch_inputs(blockIdx.x);

  if (SKEW && grp == 1) {
    // group B's bubble: its share of the first streamed stage, then the barrier that puts it one stage behind
    if (n_hid > 0) {
      rtxn::StageJob sj0{a.packed + layer_off(1), smem + RES_BYTES, HID_BYTES / 1024};
      rtxn::stage_chunk<0, 8>(sj0, wave_u, lane);
      rtxn::stage_chunk<1, 8>(sj0, wave_u, lane);
      rtxn::stage_chunk<2, 8>(sj0, wave_u, lane);
      rtxn::stage_chunk<3, 8>(sj0, wave_u, lane);
    }
    rtxn::staged_barrier();
  }
  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    // ---- per-tile state; the encoding itself happens inside layer 0 (pipe_layer0) ----
    half8 bf[NB][CT];
    float d0[CT], dr[CT];
    const float phase = 0.25f * (float)h;
    form_inputs();
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      d0[ct] = d0_n[ct];
      dr[ct] = dr_n[ct];
      if (IN_MODE == 1 && OUT_MODE != 2 && a.t_vals && h == 0) {
        bool valid;
        const long samp = sample_of(tile, ct, valid);
        if (valid) a.t_vals[samp] = (float)(col + 1) * (1.0f / 32);
      }
    }

#ifdef RTXN_L0_PLAIN
    // all of layer 0's B fragments encoded up front (the variant pipe_layer0 replaces; kept for A/B timing)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int kk = 0; kk < KS0; ++kk) {
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = encode_slot<ES, PD, PF, DD, DF>(8 * kk + j, xin[ct], phase);
        bf[kk][ct] = v;
      }
#endif

    // segment input: the column tile's direction features, computed once per lane-half and broadcast (DirShare)
    constexpr bool SHARE = RTXN_SHARE_DIR && IN_MODE == 1 && DirShare<PD, PF, DD, DF>::possible;
    int dirs[CT][DirShare<PD, PF, DD, DF>::n_dwords];
    if constexpr (SHARE) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) share_direction<PD, PF, DD, DF>(xin[ct][PD], xin[ct][PD + 1], phase, lane, dirs[ct]);
    }

    // ---- layers ----  (two fragment sets used ping-pong: no register copies between layers)
    half8 bg[NB][CT];
    floatx16 acc2[2][CT];
    // barrier, then: the LDS buffer holding layer l, and the job (this wave's share) that fetches a following stage
    rtxn::StageJob sj;
    auto begin_stage = [&](int l) -> const uint8_t* {
      const uint8_t* cur;
      rtxn::staged_barrier();  // every wave's share of this stage has landed; the slot fetched into next is free
      if (SKEW) {
        // q = hidden stages this wave has begun = instance number of its next hidden stage; instance i lives in ring
        // slot i % 3.  This wave runs layer l; the stage group A needs next is layer l + 1 + grp (wrapping into the next
        // tile), and this wave fetches its share of it if it is a streamed one.
        const bool hidden = l > 0 && l < n_layers - 1;
        cur = l == 0 ? smem : (hidden ? smem + RES_BYTES + (q % 3) * HID_BYTES : smem + L0_BYTES);
        const int h0 = l == 0 ? 0 : (hidden ? l - 1 : n_hid);      // hidden stages of this tile begun before this stage
        int lk = l + 1 + grp, inst = q - h0;                        // inst: instance number of this tile's first hidden stage
        bool exists = true;
        if (lk >= n_layers) { lk -= n_layers; inst += n_hid; exists = tile + gridDim.x < n_tiles; }
        const bool fetch = exists && lk > 0 && lk < n_layers - 1;
        sj.g = a.packed + layer_off(fetch ? lk : 0);
        sj.lds = smem + RES_BYTES + ((inst + lk - 1) % 3) * HID_BYTES;
        sj.nfrags = fetch ? HID_BYTES / 1024 : 0;
        if (hidden) ++q;
      } else {
        cur = smem + (q & 1) * BUF;
        sj.lds = smem + ((q + 1) & 1) * BUF;
        if (l + 1 < n_layers) {
          sj.g = a.packed + layer_off(l + 1), sj.nfrags = (l + 1 == n_layers - 1 ? OUT_BYTES : HID_BYTES) / 1024;
        } else {
          sj.g = a.packed, sj.nfrags = tile + gridDim.x < n_tiles ? L0_BYTES / 1024 : 0;
        }
        ++q;
      }
      // next tile's inputs, one tile ahead: issued behind the first barrier after layer 0 (which still reads this tile's),
      // so that no staged_barrier ever waits on them before they have had a whole layer to land
      if (l == 1 && tile + gridDim.x < n_tiles) fetch_inputs(tile + gridDim.x);
      return cur;
    };
    auto finish = [&](half8 (&in)[NB][CT], half8 (&other)[NB][CT]) {
      const uint8_t* w = begin_stage(n_layers - 1);
      pipe_layer<0, KS, NB, CT, true>(w, sj, in, other, acc2, wave_u, lane);
      floatx16 (&acc)[CT] = acc2[0];
      // rows 4h..4h+3 are regs 0..3, rows 8+4h..8+4h+3 are regs 4..7 of this lane
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float z = acc[ct][e];
          y[e] = a.out_act == RTXN_ACT_SIGMOID ? rtxn::sigmoidf_fast(z) : z;
        }
        bool valid;
        const long samp = sample_of(tile, ct, valid);
        if (OUT_MODE == 2) {
          const float4 c = seg_composite((float)(_Float16)y[0], (float)(_Float16)y[1], (float)(_Float16)y[2],
                                         (float)(_Float16)y[3], col, d0[ct], dr[ct], a.vr_mode);
          if (valid && lane == 0) a.seg_out[samp >> 5] = c;
        } else if (valid) {
          if (OUT_MODE == 0) {
            half4v lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = (_Float16)y[e]; hi[e] = (_Float16)y[4 + e]; }
            _Float16* o = a.out_half + samp * 16;
            *reinterpret_cast<half4v*>(o + 4 * h) = lo;
            *reinterpret_cast<half4v*>(o + 8 + 4 * h) = hi;
          } else if (OUT_MODE == 3) {
            // compact: the four half outputs themselves (8 B/sample); the consumer widens them (rtxn_volrender_fwd_compact)
            if (h == 0) {
              half4v o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (_Float16)y[e];
              *reinterpret_cast<half4v*>(a.out_half + samp * 4) = o;
            }
          } else if (h == 0) {
            // radiance = fp32(fp16(y)): the half output of network->forward, then convertHalfToFloat
            a.radiance[samp] = make_float4((float)(_Float16)y[0], (float)(_Float16)y[1],
                                               (float)(_Float16)y[2], (float)(_Float16)y[3]);
          }
        }
      }
    };
    // every layer leaves its last row tile pending in acc2[1]; the next one converts it under its first MFMAs
    {
      const uint8_t* w = begin_stage(0);
#if defined(RTXN_L0_PLAIN)
      pipe_layer<RT, KS0, NB, CT, false>(w, sj, bf, bg, acc2, wave_u, lane);
#else
      pipe_layer0<ES, PD, PF, DD, DF, RT, KS0, NB, CT, SHARE>(w, sj, xin, phase, dirs, bg, acc2[1], wave_u, lane);
#endif
    }
    int l = 1;
    for (; l + 1 < n_layers - 1; l += 2) {  // activations in bg at the top
      const uint8_t* w = begin_stage(l);
      pipe_layer<RT, KS, NB, CT, true>(w, sj, bg, bf, acc2, wave_u, lane);
      w = begin_stage(l + 1);
      pipe_layer<RT, KS, NB, CT, true>(w, sj, bf, bg, acc2, wave_u, lane);
    }
    if (l < n_layers - 1) {
      const uint8_t* w = begin_stage(l);
      pipe_layer<RT, KS, NB, CT, true>(w, sj, bg, bf, acc2, wave_u, lane);
      finish(bf, bg);
    } else {
      finish(bg, bf);
    }
  }
  if (SKEW && grp == 0) rtxn::staged_barrier();   // pairs with group B's last stage barrier (B began one barrier late)
}

// ---------------------------------------------------------------------------
// The 64/128-wide kernel on v_mfma_f32_16x16x32_f16 (mlp_internal.h, pipe_layer16)
// ---------------------------------------------------------------------------
// Same block geometry, LDS plan, staging protocol and wave-group skew as mlp_fwd_kernel; what changes is the fragment
// shape and with it who holds what: lane (c = l & 15, g = l >> 4) owns sample 16 ct + c of the wave's four 16-column tiles
// and, of every 32 features, the eight perm_feature16 gives its lane group.
// Layer 0 / the encoder.  Lane group g evaluates a block of FB = ceil(F/4) consecutive frequencies of every input dimension
// (enc16_feature): the inputs are pre-scaled once per tile by 2^(FB g) (exact), the block's lowest octave comes from ONE
// v_sin_f32 + ONE v_cos_f32 (exact range reduction as before) and its other octaves from the double-angle identities
// s' = 2 s c, c' = 1 - 2 s^2 -- three 4-cycle instructions per octave instead of two 8-cycle transcendentals with their
// range reductions: 64 issue cycles per (dimension, column tile) for six features, where the direct form took 120.  The error
// of an octave doubles the previous one's: two doublings on a ~1e-6 seed stay under 1e-5, a fortieth of an fp16 ulp of
// these values (tests/test_gpu_parity.py reports the measured maximum per octave).
template <int PD, int PF, int DD, int DF>
struct EncSpec16 {
  static constexpr int FBP = (PF + 3) / 4, FBD = (DF + 3) / 4;     // octaves per lane group and dimension
  static_assert(FBP >= 1 && FBP <= 3 && FBD >= 1 && FBD <= 3, "octave_unit is written for 1-3 octaves per block");
  static constexpr int enc_width = 2 * (PD * PF + DD * DF);
  static constexpr int enc_padded = (enc_width + 15) / 16 * 16;
  static constexpr int ND = DD * FBD, NP = PD * FBP;               // dwords: direction, position
  static constexpr int NPADW = (enc_padded - enc_width + 7) / 8;   // padding dwords (8 features each over the four groups)
  static constexpr int n_dwords = ND + NP + NPADW;
  static constexpr int k0 = (n_dwords + 3) / 4 * 32;               // first-layer K as staged (4 dwords = 32 k per k-step)
  static_assert(ND <= 8, "direction dwords are placed before the pipeline starts: k-steps 0 and 1 only");
};

// One dimension of one sample: NK dwords {sin, cos} of octaves 0..NK-1 of the lane group's block.  xg = x 2^(FB g).
template <int NK>
__device__ __forceinline__ void octave_unit(float xg, int (&d)[3]) {
  float t, u, sn, cs;
  if constexpr (NK == 3) {
    asm volatile(
        "v_mul_f32 %3, 0.5, %7\n\t"
        "v_fract_f32 %3, %3\n\t"
        "v_sin_f32 %5, %3\n\t"
        "v_cos_f32 %6, %3\n\t"
        "s_nop 0\n\t"
        "v_cvt_pk_f16_f32 %0, %5, %6\n\t"
        "v_add_f32 %3, %5, %5\n\t"          // 2 s0
        "v_mul_f32 %4, %3, %6\n\t"          // s1 = 2 s0 c0
        "v_fma_f32 %6, -%3, %5, 1.0\n\t"    // c1 = 1 - 2 s0^2
        "v_cvt_pk_f16_f32 %1, %4, %6\n\t"
        "v_add_f32 %3, %4, %4\n\t"          // 2 s1
        "v_mul_f32 %5, %3, %6\n\t"          // s2 = 2 s1 c1
        "v_fma_f32 %6, -%3, %4, 1.0\n\t"    // c2 = 1 - 2 s1^2
        "v_cvt_pk_f16_f32 %2, %5, %6"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(t), "=&v"(u), "=&v"(sn), "=&v"(cs)
        : "v"(xg));
  } else if constexpr (NK == 2) {
    asm volatile(
        "v_mul_f32 %2, 0.5, %6\n\t"
        "v_fract_f32 %2, %2\n\t"
        "v_sin_f32 %4, %2\n\t"
        "v_cos_f32 %5, %2\n\t"
        "s_nop 0\n\t"
        "v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_add_f32 %2, %4, %4\n\t"
        "v_mul_f32 %3, %2, %5\n\t"
        "v_fma_f32 %5, -%2, %4, 1.0\n\t"
        "v_cvt_pk_f16_f32 %1, %3, %5"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(t), "=&v"(u), "=&v"(sn), "=&v"(cs)
        : "v"(xg));
    d[2] = 0;
  } else {
    asm volatile(
        "v_mul_f32 %1, 0.5, %4\n\t"
        "v_fract_f32 %1, %1\n\t"
        "v_sin_f32 %2, %1\n\t"
        "v_cos_f32 %3, %1\n\t"
        "s_nop 0\n\t"
        "v_cvt_pk_f16_f32 %0, %2, %3"
        : "=&v"(d[0]), "=&v"(t), "=&v"(sn), "=&v"(cs)
        : "v"(xg));
    d[1] = d[2] = 0;
  }
}

template <int PD, int PF, int DD, int DF>
struct DirShare16 {
  static constexpr bool possible = DD <= 16;
  static constexpr int n_dwords = DD * EncSpec16<PD, PF, DD, DF>::FBD;
};

// Segment input: a segment is TWO adjacent 16-column tiles and its direction features are the same for all 32 samples.
// Lane (c, g), c < DD, runs the octave unit of direction dimension c for its own lane group; DD * FB_D ds_bpermute broadcasts
// hand every lane of the group the finished B-fragment dwords (dword dd * FB_D + k from lane (dd, g)).  The inputs come in
// already scaled by 2^(FB_D g).
template <int PD, int PF, int DD, int DF>
__device__ __forceinline__ void share_direction16(const float (&dir_g)[DD], int lane, int (&dirs)[DirShare16<PD, PF, DD, DF>::n_dwords]) {
  constexpr int FBD = EncSpec16<PD, PF, DD, DF>::FBD;
  const int c = lane & 15;
  float xs = dir_g[0];
#pragma unroll
  for (int dd = 1; dd < DD; ++dd) xs = c == dd ? dir_g[dd] : xs;     // by value (a select on an array element once went through scratch)
  int d[3];
  octave_unit<FBD>(xs, d);
#pragma unroll
  for (int dd = 0; dd < DD; ++dd)
#pragma unroll
    for (int k = 0; k < FBD; ++k) dirs[dd * FBD + k] = __builtin_amdgcn_ds_bpermute(4 * ((lane & 48) + dd), d[k]);
}

// The whole encoded input of a column tile as layer 0's B fragments: global dword D sits in k-step D / 4, dword D % 4.
// Layer 0 then runs on the same row-tile-outer pipeline as the hidden layers, with the encoding done BEFORE it rather than in
// slices behind its MFMAs.  Measured against the k-step-outer form that hid the encoder inside layer 0 (round 1's design,
// carried over to this kernel first): the same kernel time within noise -- the other wave group keeps the matrix core busy
// during the encode, and the chip is issue- and power-bound, not latency-bound -- at 189 instead of 231-251 VGPRs (32
// accumulator registers live in layer 0 instead of 128).  That is what lets the traversal and compositor kernels of the
// neighbouring frames co-reside with two of these waves per SIMD (render.py, render_async): MLP-to-MLP gaps 12-20 us.
template <class ES, int PD, int PF, int DD, int DF, int KS0, int NB, int CT, bool SHARE>
__device__ __forceinline__ void encode_layer0_input(const float (&xq)[CT][5], const int (&dirs)[CT / 2][DirShare16<PD, PF, DD, DF>::n_dwords],
                                                    half8 (&bf)[NB][CT]) {
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    rtxn::int4v t[KS0];
#pragma unroll
    for (int kk = 0; kk < KS0; ++kk)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int D = 4 * kk + e;   // padding dwords {1.0, 1.0} (features beyond enc_padded meet zero weights); beyond n_dwords: 0
        t[kk][e] = (D >= ES::ND + ES::NP && D < ES::n_dwords) ? 0x3c003c00 : 0;
      }
    if constexpr (SHARE) {
#pragma unroll
      for (int D = 0; D < ES::ND; ++D) t[D / 4][D % 4] = dirs[ct / 2][D];
    } else {
#pragma unroll
      for (int dd = 0; dd < DD; ++dd) {
        int d[3];
        octave_unit<ES::FBD>(xq[ct][PD + dd], d);
#pragma unroll
        for (int k = 0; k < ES::FBD; ++k) t[(dd * ES::FBD + k) / 4][(dd * ES::FBD + k) % 4] = d[k];
      }
    }
#pragma unroll
    for (int dim = 0; dim < PD; ++dim) {
      int d[3];
      octave_unit<ES::FBP>(xq[ct][dim], d);
#pragma unroll
      for (int k = 0; k < ES::FBP; ++k) t[(ES::ND + ES::FBP * dim + k) / 4][(ES::ND + ES::FBP * dim + k) % 4] = d[k];
    }
#pragma unroll
    for (int kk = 0; kk < KS0; ++kk) bf[kk][ct] = __builtin_bit_cast(half8, t[kk]);
    // Segment input: the two column tiles of a segment get IDENTICAL k-step-0 fragments when that k-step holds direction
    // dwords only, and hipcc then merges their MFMAs (one result feeding both accumulator chains: an out-of-place MFMA for one
    // tile, in-place for the other).  Legal for the compiler -- but that kernel came out wrong on the hardware, and not
    // reproducibly so (first column tile of a wave's first segment), while every build that keeps the four accumulator
    // chains separate is exact and bit-deterministic.  The hand-placed asm slices of the pipeline rely on the chains being
    // what the source says, so the fragment is made opaque to value numbering.
    if constexpr (SHARE) asm volatile("" : "+v"(bf[0][ct]));
  }
}

// Diagnostic build only (-DRTXN_STAMPS, tools/probe/stamps.py; never in the shipped library): block 0 records s_memtime at
// the stage boundaries of its tiles 2..5, every wave its own, into LDS and copies them out when it is done.  The values go
// to a buffer nothing else reads.  Stamp k of a tile: 0 top, 1 encoded, 2 + 2l past the barrier of stage l, 3 + 2l stage l done.
#ifdef RTXN_STAMPS
constexpr int kStampTiles = 4, kStampSlots = 24;
__device__ unsigned g_stamps[8 * kStampTiles * kStampSlots];
#define RTXN_STAMP(k)                                                                                     \
  do {                                                                                                    \
    if (blockIdx.x == 0 && tile_it >= 2 && tile_it < 2 + kStampTiles) {                                   \
      const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime();                                         \
      if (lane == 0) stamp_lds[(wave_u * kStampTiles + (tile_it - 2)) * kStampSlots + (k)] = t_;          \
    }                                                                                                     \
  } while (0)
#else
#define RTXN_STAMP(k)
#endif

// OUT_MODE 0: half[n][16]; 1: float4 radiance (+ t_vals); 3: compact half4.  (The per-segment compositor epilogue,
// OUT_MODE 2, exists only in the 32x32 kernel.)
template <int W, int PD, int PF, int DD, int DF, int IN_MODE, int OUT_MODE>
__global__ __launch_bounds__(512, 2) void mlp_fwd16_kernel(FwdArgs a) {
  static_assert(RTXN_NW == 8, "8-wave blocks");
  static_assert(OUT_MODE != 2, "segment-composite epilogue: 32x32 kernel only");
  constexpr int CT = 4, THREADS = 512;
  constexpr int TILE = 512, TILE_SEGS = 16;             // samples / segments per block per iteration
  using ES = EncSpec16<PD, PF, DD, DF>;
  constexpr int RT = W / 16, KS = W / 32, KS0 = ES::k0 / 32;
  constexpr int NB = KS0 > KS ? KS0 : KS;
  constexpr int L0_BYTES = KS0 * RT * 1024, HID_BYTES = KS * RT * 1024, OUT_BYTES = 4 * KS * 1024;
  constexpr int RES_BYTES = L0_BYTES + OUT_BYTES;       // [layer 0 | output layer x 4 rotations | 3 x HID_BYTES]
  constexpr bool ROT = OUT_MODE != 0;                   // 4-output epilogue: column tile v's outputs land in lane group v
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
#ifdef RTXN_STAMPS
  __shared__ unsigned stamp_lds[8 * kStampTiles * kStampSlots];
  int tile_it = 0;
#endif

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // Tile bookkeeping in 32-bit scalars: gfx950 has no 64-bit scalar compare, so `long` counters put every uniform loop
  // and fetch decision through VCC and the staging jobs behind vector branches (measured: +35 % VALU instructions).
  long total_seg = 0;
  int n_tiles;
  if (IN_MODE == 1) {
    total_seg = *a.total_segments;
    if (total_seg > a.max_segments) total_seg = a.max_segments;
    n_tiles = (int)((total_seg + TILE_SEGS - 1) / TILE_SEGS);
  } else {
    n_tiles = (int)((a.n + TILE - 1) / TILE);
  }
  n_tiles = __builtin_amdgcn_readfirstlane(n_tiles);
  const int tile_step = (int)gridDim.x;
  if ((int)blockIdx.x >= n_tiles) return;

  const int n_layers = a.n_hidden + 1;
  auto layer_off = [&](int l) -> unsigned { return l == 0 ? 0u : (unsigned)L0_BYTES + (unsigned)(l - 1) * HID_BYTES; };
  const int grp = wave_u >> 2;              // 0: leading wave group, 1: one stage behind (see mlp_fwd_kernel, SKEW)
  const int n_hid = n_layers - 2;
  stage<L0_BYTES, THREADS>(a.packed, smem, tid);
  stage<OUT_BYTES, THREADS>(a.packed + layer_off(n_layers - 1), smem + L0_BYTES, tid);
  int qs = 0;                               // hidden stages this wave has begun (ring slot = qs % 3)

  const float pos_scale = (float)(1u << (ES::FBP * g)), dir_scale = (float)(1u << (ES::FBD * g));   // 2^(FB g): see EncSpec16
  float xq[CT][5];                          // inputs of the lane's four samples, already scaled for its lane group
  auto sample_of = [&](int tile, int ct, bool& valid) -> long {
    if (IN_MODE == 1) {
      const long seg = (long)tile * TILE_SEGS + wave_u * 2 + (ct >> 1);
      valid = seg < total_seg;
      return seg * 32 + 16 * (ct & 1) + c;
    }
    const long sidx = (long)tile * TILE + wave_u * 64 + ct * 16 + c;
    valid = sidx < a.n;
    return sidx;
  };
  // The next tile's inputs are FETCHED one tile ahead (behind the barrier of stage 1) and only turned into samples at the top
  // of their own tile.  Forming the samples where the loads are issued -- as this kernel did until the in-kernel stamps
  // (tools/probe/stamps.py) showed a 5,000-cycle "barrier" at stage 1 -- makes hipcc wait for the loads right there: both
  // wave groups stood still for an HBM round trip once per tile, 10 % of the tile's 46,600 cycles.
  // (The carried values are kept in the vector types the loads produce: as a float array they were copied element by
  // element right behind the loads -- register tuples versus loop phis -- and the copies waited for the data all the same.)
  typedef float f3v __attribute__((ext_vector_type(3)));
  typedef float f2v __attribute__((ext_vector_type(2)));
  f3v raw_s[CT / 2], raw_e[CT / 2];         // segments: start, end, view of the wave's two
  f2v raw_v[CT / 2];
  float raw_x[IN_MODE == 1 ? 1 : CT][5];    // samples: x[5] of the lane's four
  auto fetch_inputs = [&](int tile) {
    if (IN_MODE == 1) {
#pragma unroll
      for (int sgi = 0; sgi < CT / 2; ++sgi) {
        bool valid_in;
        const long samp_in = sample_of(tile, 2 * sgi, valid_in);
        const long sg = valid_in ? (samp_in >> 5) : 0;
        __builtin_memcpy(&raw_s[sgi], a.start + 3 * sg, 12);
        __builtin_memcpy(&raw_e[sgi], a.end + 3 * sg, 12);
        __builtin_memcpy(&raw_v[sgi], a.seg_view + 2 * sg, 8);
      }
    } else {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        bool valid_in;
        const long samp_in = sample_of(tile, ct, valid_in);
        const long sidx = valid_in ? samp_in : 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) raw_x[ct][k] = a.input[5 * sidx + k];
      }
    }
  };
  auto form_inputs = [&]() {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      if (IN_MODE == 1) {
        const int sgi = ct >> 1;
        const float t = (float)(16 * (ct & 1) + c) * (1.0f / 32);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float og = raw_s[sgi][k];
          xq[ct][k] = fmaf(t, raw_e[sgi][k] - og, og) * pos_scale;   // REGULAR sample, sampler.cu:52-66; exact scaling
        }
        xq[ct][3] = raw_v[sgi][0] * dir_scale;
        xq[ct][4] = raw_v[sgi][1] * dir_scale;
      } else {
#pragma unroll
        for (int k = 0; k < 5; ++k) xq[ct][k] = raw_x[ct][k] * (k < PD ? pos_scale : dir_scale);
      }
    }
  };
  fetch_inputs((int)blockIdx.x);

  if (grp == 1) {
    if (n_hid > 0) {
      rtxn::StageJob sj0{a.packed + layer_off(1), smem + RES_BYTES, HID_BYTES / 1024};
      rtxn::stage_chunk<0, 8>(sj0, wave_u, lane);
      rtxn::stage_chunk<1, 8>(sj0, wave_u, lane);
      rtxn::stage_chunk<2, 8>(sj0, wave_u, lane);
      rtxn::stage_chunk<3, 8>(sj0, wave_u, lane);
    }
    rtxn::staged_barrier();
  }
  for (int tile = (int)blockIdx.x; tile < n_tiles; tile += tile_step) {
    const bool more = tile + tile_step < n_tiles;       // this block has another tile after this one
    RTXN_STAMP(0);
    form_inputs();
    if (IN_MODE == 1 && OUT_MODE == 1 && a.t_vals) {
      // REGULAR t_vals (sampler.cu:65: post-increment) of the wave's 64 samples, one per lane: lane l is sample l of the two
      // segments.  The lane index goes through an empty asm so that the per-lane address is formed here, not hoisted out of
      // the tile loop and carried in VGPRs through layer 0's register peak (it was, and got spilled).
      int lane_t = lane;
      asm volatile("" : "+v"(lane_t));
      const long seg = (long)tile * TILE_SEGS + wave_u * 2 + (lane_t >> 5);
      if (seg < total_seg) a.t_vals[seg * 32 + (lane_t & 31)] = (float)((lane_t & 31) + 1) * (1.0f / 32);
    }
    constexpr bool SHARE = RTXN_SHARE_DIR && IN_MODE == 1 && DirShare16<PD, PF, DD, DF>::possible;
    int dirs[CT / 2][DirShare16<PD, PF, DD, DF>::n_dwords];
    if constexpr (SHARE) {
#pragma unroll
      for (int sg = 0; sg < CT / 2; ++sg) {
        float dg[DD];
#pragma unroll
        for (int dd = 0; dd < DD; ++dd) dg[dd] = xq[2 * sg][PD + dd];
        share_direction16<PD, PF, DD, DF>(dg, lane, dirs[sg]);
      }
    }
    half8 bf[NB][CT], bg[NB][CT];
    rtxn::floatx4 acc2[2][CT];
    rtxn::StageJob sj;
    auto begin_stage = [&](int l) -> const uint8_t* {
      rtxn::staged_barrier();
      const bool hidden = l > 0 && l < n_layers - 1;
      const uint8_t* cur = l == 0 ? smem : (hidden ? smem + RES_BYTES + (qs % 3) * HID_BYTES : smem + L0_BYTES);
      const int h0 = l == 0 ? 0 : (hidden ? l - 1 : n_hid);
      int lk = l + 1 + grp, inst = qs - h0;
      bool exists = true;
      if (lk >= n_layers) { lk -= n_layers; inst += n_hid; exists = more; }
      const bool fetch = exists && lk > 0 && lk < n_layers - 1;
      // everything in the job is wave-uniform, and says so: scalar registers, scalar branches around the fetches
      sj.g = a.packed + (unsigned)__builtin_amdgcn_readfirstlane((int)layer_off(fetch ? lk : 0));
      sj.lds = smem + RES_BYTES + (unsigned)__builtin_amdgcn_readfirstlane(((inst + lk - 1) % 3) * HID_BYTES);
      sj.nfrags = __builtin_amdgcn_readfirstlane(fetch ? HID_BYTES / 1024 : 0);
      if (hidden) ++qs;
      if (l == 1 && more) fetch_inputs(tile + tile_step);
      return cur;
    };
    auto finish = [&](half8 (&in)[NB][CT], half8 (&other)[NB][CT]) {
      const uint8_t* w = begin_stage(n_layers - 1);
      RTXN_STAMP(2 + 2 * (n_layers - 1));
      if constexpr (!ROT) {
        rtxn::pipe_layer16<0, KS, NB, CT, true>(w, sj, in, other, acc2, wave_u, lane);
        // output rows 4g .. 4g+3 of sample (ct, c) are this lane's four accumulator registers
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          bool valid;
          const long samp = sample_of(tile, ct, valid);
          if (valid) {
            half4v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float z = acc2[0][ct][e];
              o[e] = (_Float16)(a.out_act == RTXN_ACT_SIGMOID ? rtxn::sigmoidf_fast(z) : z);
            }
            *reinterpret_cast<half4v*>(a.out_half + samp * 16 + 4 * g) = o;
          }
        }
      } else {
        // Rotated output layer: column tile ct multiplies by variant ct of the layer (pack16_kernel), so its four outputs
        // arrive in lane group ct.  16 of the layer stack's 1040 MFMAs: compiler-scheduled, the pending tile converted up front.
        rtxn::stage_chunk<0, 8>(sj, wave_u, lane);
        rtxn::stage_chunk<1, 8>(sj, wave_u, lane);
        rtxn::stage_chunk<2, 8>(sj, wave_u, lane);
        rtxn::stage_chunk<3, 8>(sj, wave_u, lane);
        rtxn::convert_units16<NB, CT, 2 * KS - 1, 0, 2 * CT>(acc2[1], in);
        rtxn::floatx4 z4[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int e = 0; e < 4; ++e) z4[ct][e] = 0.0f;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const half8 af = *reinterpret_cast<const half8*>(w + ((ct * KS + kk) * 64 + lane) * 16);
            z4[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, in[kk][ct], z4[ct], 0, 0, 0);
          }
          // two k-steps' fragments in flight at a time (one: four exposed LDS latencies in an epilogue that is on the block's
          // critical path; all 16 hoisted = 64 VGPRs on top of the layer's peak)
          if (kk & 1) __builtin_amdgcn_sched_barrier(0);
        }
        // lane (c, g): sample 16 g + c of the wave's 64 -- consecutive lanes, consecutive samples
        rtxn::floatx4 z = g == 0 ? z4[0] : (g == 1 ? z4[1] : (g == 2 ? z4[2] : z4[3]));
        bool valid;
        long samp;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));   // form the store address HERE (see t_vals above): not a loop invariant to carry around
        if (IN_MODE == 1) {
          const long seg = (long)tile * TILE_SEGS + wave_u * 2 + (lane_e >> 5);
          valid = seg < total_seg;
          samp = seg * 32 + (lane_e & 31);
        } else {
          samp = (long)tile * TILE + wave_u * 64 + lane_e;
          valid = samp < a.n;
        }
        if (valid) {
          half4v o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (_Float16)(a.out_act == RTXN_ACT_SIGMOID ? rtxn::sigmoidf_fast(z[e]) : z[e]);
          if (OUT_MODE == 3) *reinterpret_cast<half4v*>(a.out_half + samp * 4) = o;
          else a.radiance[samp] = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
        }
      }
    };
    {
      encode_layer0_input<ES, PD, PF, DD, DF, KS0, NB, CT, SHARE>(xq, dirs, bf);   // VALU only: before the stage barrier, not behind it
      RTXN_STAMP(1);
      const uint8_t* w = begin_stage(0);
      RTXN_STAMP(2);
      rtxn::pipe_layer16<RT, KS0, NB, CT, false>(w, sj, bf, bg, acc2, wave_u, lane);
      RTXN_STAMP(3);
    }
    int l = 1;
    for (; l + 1 < n_layers - 1; l += 2) {
      const uint8_t* w = begin_stage(l);
      RTXN_STAMP(2 + 2 * l);
      rtxn::pipe_layer16<RT, KS, NB, CT, true>(w, sj, bg, bf, acc2, wave_u, lane);
      RTXN_STAMP(3 + 2 * l);
      w = begin_stage(l + 1);
      RTXN_STAMP(4 + 2 * l);
      rtxn::pipe_layer16<RT, KS, NB, CT, true>(w, sj, bf, bg, acc2, wave_u, lane);
      RTXN_STAMP(5 + 2 * l);
    }
    if (l < n_layers - 1) {
      const uint8_t* w = begin_stage(l);
      RTXN_STAMP(2 + 2 * l);
      rtxn::pipe_layer16<RT, KS, NB, CT, true>(w, sj, bg, bf, acc2, wave_u, lane);
      RTXN_STAMP(3 + 2 * l);
      finish(bf, bg);
    } else {
      finish(bg, bf);
    }
    RTXN_STAMP(3 + 2 * (n_layers - 1));
#ifdef RTXN_STAMPS
    ++tile_it;
#endif
  }
  if (grp == 0) rtxn::staged_barrier();
#ifdef RTXN_STAMPS
  if (blockIdx.x == 0)
    for (int i = lane; i < kStampTiles * kStampSlots; i += 64)
      g_stamps[wave_u * kStampTiles * kStampSlots + i] = stamp_lds[wave_u * kStampTiles * kStampSlots + i];
#endif
}

// ---------------------------------------------------------------------------
// 256-wide variant (BASELINE config 5: 8x256)
// ---------------------------------------------------------------------------
// One 256x256 layer is 128 KiB of A fragments -- it cannot be double-buffered in 160 KiB of LDS.
// The row-tile-outer loop only ever needs ONE row tile's fragments at a time, so the weights
// stream through a ring of three 32-KiB slots in chunks of two row tiles (2 x 16 k-steps x 1 KiB),
// two chunks ahead of the MFMAs; one barrier per chunk.  A wave owns one 32-sample column tile
// (bf + nbf = 128 VGPRs at K = 256), a 512-thread block owns 256 samples; 8 waves share every
// staged chunk.
constexpr int kThreads256 = 512;
constexpr int kSlot256 = 32 * 1024;

__device__ __forceinline__ void stage512(const uint8_t* __restrict__ g, uint8_t* lds_buf, int bytes, int tid) {
  for (int off = (tid >> 6) * 1024; off < bytes; off += 8 * 1024) {
    const uint8_t* src = g + off + (tid & 63) * 16;
    uint8_t* dst = lds_buf + off;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
}

template <int PD, int PF, int DD, int DF, int IN_MODE, int OUT_MODE>
__global__ __launch_bounds__(kThreads256, 2) void mlp_fwd256_kernel(FwdArgs a) {
  using ES = EncSpec<PD, PF, DD, DF>;
  constexpr int KS = 16, KS0 = ES::k0 / 16;
  static_assert(KS0 <= KS, "first-layer K must not exceed the width");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 3 slots
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
  long n_tiles, total_seg = 0;
  if (IN_MODE == 1) {
    total_seg = *a.total_segments;
    if (total_seg > a.max_segments) total_seg = a.max_segments;
    n_tiles = (total_seg + 7) / 8;
  } else {
    n_tiles = (a.n + 255) / 256;
  }
  if ((long)blockIdx.x >= n_tiles) return;
  const long my_tiles = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
  const int n_chunks = 4 + 4 * (a.n_hidden - 1) + 1;
  const long g_end = my_tiles * n_chunks;
  long g = 0;  // chunks consumed so far by this block
  float xin[5];
  bool valid_n;
  long samp_n;
  float d0_n = 0.0f, dr_n = 0.0f;
  auto load_inputs = [&](long tile) {
    if (IN_MODE == 1) {
      const long seg = tile * 8 + wave;
      valid_n = seg < total_seg;
      samp_n = seg * 32 + col;
      const long sg = valid_n ? seg : 0;
      const bool mid = OUT_MODE == 2 && a.vr_mode == RTXN_VR_NERF;
      const float t = ((float)col + (mid ? 0.5f : 0.0f)) * (1.0f / 32);
      float dd[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float og = a.start[3 * sg + c];
        dd[c] = a.end[3 * sg + c] - og;
        xin[c] = fmaf(t, dd[c], og);
      }
      xin[3] = a.seg_view[2 * sg];
      xin[4] = a.seg_view[2 * sg + 1];
      if (OUT_MODE == 2) {
        if (a.vr_mode == RTXN_VR_COMPAT) {
          dr_n = 1.0f / 32;
          d0_n = a.seg_first[sg] ? 1.0f / 32 : 31.0f / 32;
        } else {
          d0_n = dr_n = sqrtf(fmaf(dd[2], dd[2], fmaf(dd[0], dd[0], dd[1] * dd[1]))) * (1.0f / 32) * a.step_scale;
        }
      }
    } else {
      samp_n = tile * 256 + wave * 32 + col;
      valid_n = samp_n < a.n;
      const long sidx = valid_n ? samp_n : 0;
#pragma unroll
      for (int c = 0; c < 5; ++c) xin[c] = a.input[5 * sidx + c];
    }
  };
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // chunk gi of this block's stream: offset and size in the packed buffer
  auto chunk_src = [&](long gi, int& size) -> long {
    const int i = (int)(gi % n_chunks);
    if (i < 4) { size = 2 * KS0 * 1024; return (long)i * size; }
    if (i < n_chunks - 1) { size = 2 * KS * 1024; return 4L * 2 * KS0 * 1024 + (long)(i - 4) * size; }
    size = KS * 1024;
    return 4L * 2 * KS0 * 1024 + (long)(n_chunks - 5) * 2 * KS * 1024;
  };
  auto issue = [&](long gi) {
    int size;
    const long off = chunk_src(gi, size);
    stage512(a.packed + off, smem + (gi % 3) * kSlot256, size, tid);
  };
  issue(0);
  if (g_end > 1) issue(1);
  // barrier, then: the slot holding chunk g, and the job that fetches chunk g+2 into the slot chunk g-1 just left
  rtxn::StageJob sj;
  int chunk_in_tile = 0;
  long prefetch_tile = -1;   // tile whose inputs are fetched one tile ahead
  auto next_chunk = [&]() -> const uint8_t* {
    rtxn::staged_barrier();  // chunk g landed; everyone is done with chunk g-1
    if (chunk_in_tile++ == 1 && prefetch_tile >= 0) load_inputs(prefetch_tile);   // behind a barrier: never waited on early
    int size = 0;
    const long off = g + 2 < g_end ? chunk_src(g + 2, size) : 0;
    sj.g = a.packed + off;
    sj.lds = smem + ((g + 2) % 3) * kSlot256;
    sj.nfrags = size / 1024;
    const uint8_t* p = smem + (g % 3) * kSlot256;
    ++g;
    return p;
  };

  load_inputs(blockIdx.x);

  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    half8 bf[KS][1], bg[KS][1];
    const bool valid = valid_n;
    const long samp = samp_n;
    const float d0 = d0_n, dr = dr_n;
    const float phase = 0.25f * (float)h;
    constexpr bool SHARE = RTXN_SHARE_DIR && IN_MODE == 1 && DirShare<PD, PF, DD, DF>::possible;
    int dirs[DirShare<PD, PF, DD, DF>::n_dwords];
    if constexpr (SHARE) share_direction<PD, PF, DD, DF>(xin[PD], xin[PD + 1], phase, lane, dirs);   // one segment per column tile
#pragma unroll
    for (int kk = 0; kk < KS0; ++kk) {
      rtxn::int4v v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p0 = 8 * kk + 2 * e;
        if (SHARE && p0 >= PD * PF && p0 + 1 < ES::n_pairs) {
          v[e] = dirs[(p0 - PD * PF) / 2];
        } else {
          half2v h2;
          h2[0] = encode_slot<ES, PD, PF, DD, DF>(p0, xin, phase);
          h2[1] = encode_slot<ES, PD, PF, DD, DF>(p0 + 1, xin, phase);
          v[e] = __builtin_bit_cast(int, h2);
        }
      }
      bf[kk][0] = __builtin_bit_cast(half8, v);
    }
    if (IN_MODE == 1 && OUT_MODE != 2 && a.t_vals && valid && h == 0) a.t_vals[samp] = (float)(col + 1) * (1.0f / 32);
    chunk_in_tile = 0;
    prefetch_tile = tile + gridDim.x < n_tiles ? tile + gridDim.x : -1;

    // A layer = 4 chunks of two row tiles; the second row tile of every chunk stays pending in acc2[1] and is converted
    // under the next chunk's MFMAs (mlp_internal.h, PipeStep256).
    floatx16 acc2[2][1];
    auto layer = [&](auto ks_tag, auto pend_tag, half8 (&in)[KS][1], half8 (&out)[KS][1]) {
      constexpr int KSL = decltype(ks_tag)::value;
      constexpr bool PEND0 = decltype(pend_tag)::value;
      const uint8_t* w = next_chunk();
      rtxn::pipe_chunk256<KSL, KS, 2, 0, PEND0>(w, sj, in, out, acc2, wave_u, lane);
      w = next_chunk();
      rtxn::pipe_chunk256<KSL, KS, 2, 2, true>(w, sj, in, out, acc2, wave_u, lane);
      w = next_chunk();
      rtxn::pipe_chunk256<KSL, KS, 2, 4, true>(w, sj, in, out, acc2, wave_u, lane);
      w = next_chunk();
      rtxn::pipe_chunk256<KSL, KS, 2, 6, true>(w, sj, in, out, acc2, wave_u, lane);
    };
    using std::integral_constant;
    layer(integral_constant<int, KS0>{}, integral_constant<bool, false>{}, bf, bg);   // layer 0: K = 16*KS0
    // hidden layers 1..n_hidden-1, activations ping-pong bg -> bf -> bg
    int l = 1;
    for (; l + 1 < a.n_hidden; l += 2) {
      layer(integral_constant<int, KS>{}, integral_constant<bool, true>{}, bg, bf);
      layer(integral_constant<int, KS>{}, integral_constant<bool, true>{}, bf, bg);
    }
    const bool odd = l < a.n_hidden;
    if (odd) layer(integral_constant<int, KS>{}, integral_constant<bool, true>{}, bg, bf);
    // output layer: one row tile, raw accumulators in acc2[0]
    {
      const uint8_t* w = next_chunk();
      if (odd) rtxn::pipe_chunk256<KS, KS, 1, 0, true>(w, sj, bf, bg, acc2, wave_u, lane);
      else rtxn::pipe_chunk256<KS, KS, 1, 0, true>(w, sj, bg, bf, acc2, wave_u, lane);
    }
    const floatx16& acc = acc2[0][0];
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = a.out_act == RTXN_ACT_SIGMOID ? rtxn::sigmoidf_fast(acc[e]) : acc[e];
    if (OUT_MODE == 2) {
      const float4 c = seg_composite((float)(_Float16)y[0], (float)(_Float16)y[1], (float)(_Float16)y[2],
                                     (float)(_Float16)y[3], col, d0, dr, a.vr_mode);
      if (valid && lane == 0) a.seg_out[samp >> 5] = c;
    } else if (valid) {
      if (OUT_MODE == 0) {
        half4v lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[e] = (_Float16)y[e]; hi[e] = (_Float16)y[4 + e]; }
        _Float16* o = a.out_half + samp * 16;
        *reinterpret_cast<half4v*>(o + 4 * h) = lo;
        *reinterpret_cast<half4v*>(o + 8 + 4 * h) = hi;
      } else if (OUT_MODE == 3) {
        if (h == 0) {
          half4v o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (_Float16)y[e];
          *reinterpret_cast<half4v*>(a.out_half + samp * 4) = o;
        }
      } else if (h == 0) {
        a.radiance[samp] = make_float4((float)(_Float16)y[0], (float)(_Float16)y[1], (float)(_Float16)y[2],
                                       (float)(_Float16)y[3]);
      }
    }
  }
}

// The 256-wide kernel on v_mfma_f32_16x16x32_f16 (the default; RTXN_MFMA_SHAPE=32 selects mlp_fwd256_kernel).  Same block
// (8 waves = 8 segments = 256 samples), same three-slot 32-KiB weight ring and one barrier per chunk; what changes is the
// fragment shape: a wave's 32 samples are two 16-column tiles, lane (c, g) owns samples c and 16 + c and of every 32 features
// the eight perm_feature16 gives its lane group; a chunk is four 16-row tiles x eight 32-wide k-steps (pipe_chunk16), the
// encoder is mlp_fwd16_kernel's (lane-group frequency blocks, angle doubling, direction shared across the segment), and the
// output layer is multiplied in two row-rotated variants so that column tile v's (r, g, b, sigma) land in lane group v.
// LDS reads per FLOP are those of the 32x32x16 kernel (one 1-KiB fragment per 2 x 8 MFMA passes); the gain is the shape's.
template <int PD, int PF, int DD, int DF, int IN_MODE, int OUT_MODE>
__global__ __launch_bounds__(kThreads256, 2) void mlp_fwd256x16_kernel(FwdArgs a) {
  static_assert(OUT_MODE != 2, "segment-composite epilogue: 32x32 kernel only");
  using ES = EncSpec16<PD, PF, DD, DF>;
  constexpr int CT = 2, KS = 8, NB = 8, KS0 = ES::k0 / 32;
  constexpr bool ROT = OUT_MODE != 0;
  static_assert(KS0 <= KS, "first-layer K must not exceed the width");
  constexpr int L0_CHUNK = 4 * KS0 * 1024, HID_CHUNK = 4 * KS * 1024, OUT_CHUNK = 2 * KS * 1024;   // output: rotations 0 and 1
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 3 slots
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  long total_seg = 0;
  int n_tiles;
  if (IN_MODE == 1) {
    total_seg = *a.total_segments;
    if (total_seg > a.max_segments) total_seg = a.max_segments;
    n_tiles = (int)((total_seg + 7) / 8);
  } else {
    n_tiles = (int)((a.n + 255) / 256);
  }
  n_tiles = __builtin_amdgcn_readfirstlane(n_tiles);
  if ((int)blockIdx.x >= n_tiles) return;
  const int tile_step = (int)gridDim.x;
  const int my_tiles = (n_tiles - (int)blockIdx.x + tile_step - 1) / tile_step;
  const int n_chunks = 4 + 4 * (a.n_hidden - 1) + 1;
  const int g_end = my_tiles * n_chunks;
  int gq = 0;  // chunks consumed so far by this block

  const float pos_scale = (float)(1u << (ES::FBP * g)), dir_scale = (float)(1u << (ES::FBD * g));   // 2^(FB g): see EncSpec16
  float xq[CT][5];
  auto sample_of = [&](int tile, int ct, bool& valid) -> long {
    if (IN_MODE == 1) {
      const long seg = (long)tile * 8 + wave_u;
      valid = seg < total_seg;
      return seg * 32 + 16 * ct + c;
    }
    const long sidx = (long)tile * 256 + wave_u * 32 + 16 * ct + c;
    valid = sidx < a.n;
    return sidx;
  };
  // fetched one tile ahead as the loads deliver them, formed into samples at the top of their own tile (mlp_fwd16_kernel
  // explains why: forming them where the loads are issued makes the wave wait for HBM there)
  typedef float f3v __attribute__((ext_vector_type(3)));
  typedef float f2v __attribute__((ext_vector_type(2)));
  f3v raw_s, raw_e;                         // segment input: the wave's one segment
  f2v raw_v;
  float raw_x[IN_MODE == 1 ? 1 : CT][5];    // sample input: x[5] of the lane's two
  auto fetch_inputs = [&](int tile) {
    if (IN_MODE == 1) {
      bool valid_in;
      const long samp_in = sample_of(tile, 0, valid_in);
      const long sg = valid_in ? (samp_in >> 5) : 0;
      __builtin_memcpy(&raw_s, a.start + 3 * sg, 12);
      __builtin_memcpy(&raw_e, a.end + 3 * sg, 12);
      __builtin_memcpy(&raw_v, a.seg_view + 2 * sg, 8);
    } else {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        bool valid_in;
        const long samp_in = sample_of(tile, ct, valid_in);
        const long sidx = valid_in ? samp_in : 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) raw_x[ct][k] = a.input[5 * sidx + k];
      }
    }
  };
  auto form_inputs = [&]() {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      if (IN_MODE == 1) {
        const float t = (float)(16 * ct + c) * (1.0f / 32);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float og = raw_s[k];
          xq[ct][k] = fmaf(t, raw_e[k] - og, og) * pos_scale;   // REGULAR sample, sampler.cu:52-66; exact scaling
        }
        xq[ct][3] = raw_v[0] * dir_scale;
        xq[ct][4] = raw_v[1] * dir_scale;
      } else {
#pragma unroll
        for (int k = 0; k < 5; ++k) xq[ct][k] = raw_x[ct][k] * (k < PD ? pos_scale : dir_scale);
      }
    }
  };
  // chunk gi of this block's stream: offset and size in the packed buffer ([layer 0: 4 chunks][hidden: 4 each][output])
  auto chunk_src = [&](int gi, int& size) -> unsigned {
    const int i = gi % n_chunks;
    if (i < 4) { size = L0_CHUNK; return (unsigned)i * L0_CHUNK; }
    if (i < n_chunks - 1) { size = HID_CHUNK; return 4u * L0_CHUNK + (unsigned)(i - 4) * HID_CHUNK; }
    size = OUT_CHUNK;
    return 4u * L0_CHUNK + (unsigned)(n_chunks - 5) * HID_CHUNK;
  };
  auto issue = [&](int gi) {
    int size;
    const unsigned off = chunk_src(gi, size);
    stage512(a.packed + off, smem + (gi % 3) * kSlot256, size, tid);
  };
  issue(0);
  if (g_end > 1) issue(1);
  rtxn::StageJob sj;
  int chunk_in_tile = 0, prefetch_tile = -1;
  auto next_chunk = [&]() -> const uint8_t* {
    rtxn::staged_barrier();  // chunk gq landed; everyone is done with chunk gq-1
    if (chunk_in_tile++ == 1 && prefetch_tile >= 0) fetch_inputs(prefetch_tile);   // behind a barrier; consumed at the next tile's top
    int size = 0;
    const unsigned off = gq + 2 < g_end ? chunk_src(gq + 2, size) : 0u;
    sj.g = a.packed + (unsigned)__builtin_amdgcn_readfirstlane((int)off);
    sj.lds = smem + (unsigned)__builtin_amdgcn_readfirstlane(((gq + 2) % 3) * kSlot256);
    sj.nfrags = __builtin_amdgcn_readfirstlane(size / 1024);
    const uint8_t* p = smem + (gq % 3) * kSlot256;
    ++gq;
    return p;
  };

  fetch_inputs((int)blockIdx.x);

  for (int tile = (int)blockIdx.x; tile < n_tiles; tile += tile_step) {
    form_inputs();
    if (IN_MODE == 1 && OUT_MODE == 1 && a.t_vals) {
      int lane_t = lane;
      asm volatile("" : "+v"(lane_t));
      const long seg = (long)tile * 8 + wave_u;
      if (seg < total_seg && lane_t < 32) a.t_vals[seg * 32 + lane_t] = (float)(lane_t + 1) * (1.0f / 32);
    }
    constexpr bool SHARE = RTXN_SHARE_DIR && IN_MODE == 1 && DirShare16<PD, PF, DD, DF>::possible;
    int dirs[1][DirShare16<PD, PF, DD, DF>::n_dwords];
    if constexpr (SHARE) {
      float dg[DD];
#pragma unroll
      for (int dd = 0; dd < DD; ++dd) dg[dd] = xq[0][PD + dd];
      share_direction16<PD, PF, DD, DF>(dg, lane, dirs[0]);
    }
    half8 bf[NB][CT], bg[NB][CT];
    rtxn::floatx4 acc2[2][CT];
    encode_layer0_input<ES, PD, PF, DD, DF, KS0, NB, CT, SHARE>(xq, dirs, bf);
    chunk_in_tile = 0;
    prefetch_tile = tile + tile_step < n_tiles ? tile + tile_step : -1;

    auto layer = [&](auto ks_tag, auto pend_tag, half8 (&in)[NB][CT], half8 (&out)[NB][CT]) {
      constexpr int KSL = decltype(ks_tag)::value;
      constexpr bool PEND0 = decltype(pend_tag)::value;
      const uint8_t* w = next_chunk();
      rtxn::pipe_chunk16<KSL, NB, CT, 4, 0, PEND0>(w, sj, in, out, acc2, wave_u, lane);
      w = next_chunk();
      rtxn::pipe_chunk16<KSL, NB, CT, 4, 4, true>(w, sj, in, out, acc2, wave_u, lane);
      w = next_chunk();
      rtxn::pipe_chunk16<KSL, NB, CT, 4, 8, true>(w, sj, in, out, acc2, wave_u, lane);
      w = next_chunk();
      rtxn::pipe_chunk16<KSL, NB, CT, 4, 12, true>(w, sj, in, out, acc2, wave_u, lane);
    };
    auto finish = [&](half8 (&in)[NB][CT]) {
      const uint8_t* w = next_chunk();
      rtxn::stage_chunk<0, 8>(sj, wave_u, lane);
      rtxn::stage_chunk<1, 8>(sj, wave_u, lane);
      rtxn::stage_chunk<2, 8>(sj, wave_u, lane);
      rtxn::stage_chunk<3, 8>(sj, wave_u, lane);
      rtxn::convert_units16<NB, CT, 2 * NB - 1, 0, 2 * CT>(acc2[1], in);     // the last hidden layer's pending row tile
      rtxn::floatx4 z4[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) z4[ct][e] = 0.0f;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          // ROT: column tile ct multiplies by rotation ct of the output layer; otherwise both by the layer as it is
          const half8 af = *reinterpret_cast<const half8*>(w + (((ROT ? ct : 0) * KS + kk) * 64 + lane) * 16);
          z4[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, in[kk][ct], z4[ct], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (!ROT) {
        // output rows 4g .. 4g+3 of sample (ct, c) are this lane's four accumulator registers
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          bool valid;
          const long samp = sample_of(tile, ct, valid);
          if (valid) {
            half4v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float z = z4[ct][e];
              o[e] = (_Float16)(a.out_act == RTXN_ACT_SIGMOID ? rtxn::sigmoidf_fast(z) : z);
            }
            *reinterpret_cast<half4v*>(a.out_half + samp * 16 + 4 * g) = o;
          }
        }
      } else {
        // lane (c, g < 2): sample 16 g + c of the wave's 32 -- lanes 0..31 in sample order
        const rtxn::floatx4 z = g == 0 ? z4[0] : z4[1];
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        bool valid;
        long samp;
        if (IN_MODE == 1) {
          const long seg = (long)tile * 8 + wave_u;
          valid = seg < total_seg && lane_e < 32;
          samp = seg * 32 + lane_e;
        } else {
          samp = (long)tile * 256 + wave_u * 32 + lane_e;
          valid = samp < a.n && lane_e < 32;
        }
        if (valid) {
          half4v o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (_Float16)(a.out_act == RTXN_ACT_SIGMOID ? rtxn::sigmoidf_fast(z[e]) : z[e]);
          if (OUT_MODE == 3) *reinterpret_cast<half4v*>(a.out_half + samp * 4) = o;
          else a.radiance[samp] = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
        }
      }
    };
    using std::integral_constant;
    layer(integral_constant<int, KS0>{}, integral_constant<bool, false>{}, bf, bg);   // layer 0: K = 32*KS0
    int l = 1;
    for (; l + 1 < a.n_hidden; l += 2) {
      layer(integral_constant<int, KS>{}, integral_constant<bool, true>{}, bg, bf);
      layer(integral_constant<int, KS>{}, integral_constant<bool, true>{}, bf, bg);
    }
    if (l < a.n_hidden) {
      layer(integral_constant<int, KS>{}, integral_constant<bool, true>{}, bg, bf);
      finish(bf);
    } else {
      finish(bg);
    }
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
typedef void (*fwd_fn)(FwdArgs);

struct Variant {
  int W, PD, PF, DD, DF;
  fwd_fn fn[2][4];  // [IN_MODE][OUT_MODE]; OUT_MODE 2 (segment composite) and 3 (compact half4) exist for IN_MODE 1 only
  int k0;
  size_t lds;
  int threads;        // block size
  int blocks_per_cu;  // persistent grid = CUs x this
  int tile;           // samples per block per iteration (segments: tile / 32)
  // the same model on v_mfma_f32_16x16x32_f16 (mlp_fwd16_kernel / mlp_fwd256x16_kernel): [IN_MODE][OUT_MODE], no OUT_MODE 2
  fwd_fn fn16[2][4];
  int k0_16;          // first-layer K as staged for that kernel
  size_t lds16;
};

#ifndef RTXN_CT
#define RTXN_CT 2
#endif
template <int W, int PD, int PF, int DD, int DF, int CT = RTXN_CT>
Variant make_variant() {
  using ES = EncSpec<PD, PF, DD, DF>;
  constexpr int RT = W / 32, KS = W / 16, KS0 = ES::k0 / 16;
  constexpr int L0 = KS0 * RT * 1024, HID = KS * RT * 1024;
  Variant v;
  v.W = W; v.PD = PD; v.PF = PF; v.DD = DD; v.DF = DF;
  v.fn[0][0] = mlp_fwd_kernel<W, PD, PF, DD, DF, 0, 0, CT>;
  v.fn[0][1] = mlp_fwd_kernel<W, PD, PF, DD, DF, 0, 1, CT>;
  v.fn[1][0] = mlp_fwd_kernel<W, PD, PF, DD, DF, 1, 0, CT>;
  v.fn[1][1] = mlp_fwd_kernel<W, PD, PF, DD, DF, 1, 1, CT>;
  v.fn[0][2] = nullptr;
  v.fn[1][2] = mlp_fwd_kernel<W, PD, PF, DD, DF, 1, 2, CT>;
  v.fn[0][3] = nullptr;
  v.fn[1][3] = mlp_fwd_kernel<W, PD, PF, DD, DF, 1, 3, CT>;
  v.k0 = ES::k0;
  v.lds = (RTXN_SKEW && RTXN_NW == 8) ? (size_t)L0 + KS * 1024 + 3 * (size_t)HID : 2 * (size_t)(L0 > HID ? L0 : HID);
  v.threads = 64 * RTXN_NW;
  v.blocks_per_cu = (CT == 2 ? 2 : 1) * 4 / RTXN_NW;
  v.tile = 32 * RTXN_NW * CT;
  memset(v.fn16, 0, sizeof(v.fn16));
  v.k0_16 = 0;
  v.lds16 = 0;
  // same 512-sample tiles per block as the 32x32 kernel
  if constexpr (RTXN_NW == 8 && CT == 2 && PF % 2 == 0 && DF % 2 == 0) {
    using ES16 = EncSpec16<PD, PF, DD, DF>;
    v.fn16[0][0] = mlp_fwd16_kernel<W, PD, PF, DD, DF, 0, 0>;
    v.fn16[0][1] = mlp_fwd16_kernel<W, PD, PF, DD, DF, 0, 1>;
    v.fn16[1][0] = mlp_fwd16_kernel<W, PD, PF, DD, DF, 1, 0>;
    v.fn16[1][1] = mlp_fwd16_kernel<W, PD, PF, DD, DF, 1, 1>;
    v.fn16[1][3] = mlp_fwd16_kernel<W, PD, PF, DD, DF, 1, 3>;
    v.k0_16 = ES16::k0;
    v.lds16 = (size_t)(ES16::k0 / 32) * (W / 16) * 1024 + 4 * (size_t)(W / 32) * 1024 + 3 * (size_t)(W / 32) * (W / 16) * 1024;
  }
  return v;
}

template <int PD, int PF, int DD, int DF>
Variant make_variant256() {
  using ES = EncSpec<PD, PF, DD, DF>;
  Variant v;
  v.W = 256; v.PD = PD; v.PF = PF; v.DD = DD; v.DF = DF;
  v.fn[0][0] = mlp_fwd256_kernel<PD, PF, DD, DF, 0, 0>;
  v.fn[0][1] = mlp_fwd256_kernel<PD, PF, DD, DF, 0, 1>;
  v.fn[1][0] = mlp_fwd256_kernel<PD, PF, DD, DF, 1, 0>;
  v.fn[1][1] = mlp_fwd256_kernel<PD, PF, DD, DF, 1, 1>;
  v.fn[0][2] = nullptr;
  v.fn[1][2] = mlp_fwd256_kernel<PD, PF, DD, DF, 1, 2>;
  v.fn[0][3] = nullptr;
  v.fn[1][3] = mlp_fwd256_kernel<PD, PF, DD, DF, 1, 3>;
  v.k0 = ES::k0;
  v.lds = 3 * (size_t)kSlot256;
  v.threads = kThreads256;
  v.blocks_per_cu = 1;
  v.tile = 256;
  memset(v.fn16, 0, sizeof(v.fn16));
  v.fn16[0][0] = mlp_fwd256x16_kernel<PD, PF, DD, DF, 0, 0>;
  v.fn16[0][1] = mlp_fwd256x16_kernel<PD, PF, DD, DF, 0, 1>;
  v.fn16[1][0] = mlp_fwd256x16_kernel<PD, PF, DD, DF, 1, 0>;
  v.fn16[1][1] = mlp_fwd256x16_kernel<PD, PF, DD, DF, 1, 1>;
  v.fn16[1][3] = mlp_fwd256x16_kernel<PD, PF, DD, DF, 1, 3>;
  v.k0_16 = EncSpec16<PD, PF, DD, DF>::k0;
  v.lds16 = 3 * (size_t)kSlot256;
  return v;
}

const std::vector<Variant>& variants() {
  static const std::vector<Variant> v = {
      make_variant<128, 3, 10, 2, 12>(),  // the reference model (main.cu:47-68)
      make_variant<64, 3, 10, 2, 12>(),   // BASELINE config 1 (2x64)
      make_variant<128, 3, 10, 2, 4>(),
      make_variant<64, 3, 10, 2, 4>(),
      make_variant256<3, 10, 2, 12>(),    // BASELINE config 5 (8x256)
  };
  return v;
}

// PCG32 (O'Neill), the generator tiny-cuda-nn seeds its initialisers with.
struct Pcg32 {
  uint64_t state, inc;
  explicit Pcg32(uint64_t seed, uint64_t seq = 1) {
    state = 0u;
    inc = (seq << 1u) | 1u;
    next_uint();
    state += seed;
    next_uint();
  }
  uint32_t next_uint() {
    uint64_t old = state;
    state = old * 6364136223846793005ull + inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
  }
  float next_float() {
    union { uint32_t u; float f; } x;
    x.u = (next_uint() >> 9) | 0x3f800000u;
    return x.f - 1.0f;
  }
};

// n_units: samples (in_mode 0) or segments (in_mode 1) the launch may have to cover
int launch_fwd(const rtxn_mlp* m, FwdArgs& a, int in_mode, int out_mode, long n_units, hipStream_t s) {
  const Variant& v = variants()[m->variant];
  const long per_tile = in_mode == 1 ? v.tile / 32 : v.tile;
  const long n_tiles = (n_units + per_tile - 1) / per_tile;
  const bool use16 = m->mfma16 && v.fn16[in_mode][out_mode] != nullptr;   // same tile size, grid and block shape either way
  a.packed = static_cast<const uint8_t*>(use16 ? m->packed16 : m->packed);
  a.n_hidden = m->cfg.n_hidden_layers;
  a.out_act = m->cfg.output_activation;
  // CU count and the dynamic-LDS attribute are per DEVICE: a process may drive several GPUs (and they need not be alike)
  int dev = 0, n_cu = 0;
  RTXN_HIP(hipGetDevice(&dev));
  RTXN_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  if (n_cu <= 0) n_cu = 256;
  const int cus = n_cu - m->reserved_cus > 1 ? n_cu - m->reserved_cus : 1;
  long grid = n_tiles < (long)cus * v.blocks_per_cu ? n_tiles : (long)cus * v.blocks_per_cu;  // persistent grid
  if (grid < 1) grid = 1;
  fwd_fn fn = use16 ? v.fn16[in_mode][out_mode] : v.fn[in_mode][out_mode];
  const size_t lds = use16 ? v.lds16 : v.lds;
  {
    constexpr int kMaxDev = 64;
    static std::mutex mu;
    static bool attr_set[kMaxDev][16][2][2][4] = {};
    std::lock_guard<std::mutex> lock(mu);
    const bool known = dev >= 0 && dev < kMaxDev && attr_set[dev][m->variant][use16][in_mode][out_mode];
    if (!known) {
      RTXN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      if (dev >= 0 && dev < kMaxDev) attr_set[dev][m->variant][use16][in_mode][out_mode] = true;
    }
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3((unsigned)v.threads), lds, s, a);
  RTXN_LAUNCH_CHECK("mlp_fwd_kernel");
  return RTXN_OK;
}

}  // namespace

extern "C" int rtxn_mlp_create(const rtxn_mlp_config* cfg, rtxn_mlp** out) {
  RTXN_REQUIRE(cfg && out, "rtxn_mlp_create: NULL argument");
  RTXN_REQUIRE(cfg->n_hidden_layers >= 1 && cfg->n_hidden_layers <= 64, "rtxn_mlp_create: n_hidden_layers = %d",
               cfg->n_hidden_layers);
  RTXN_REQUIRE(cfg->n_output_dims >= 1 && cfg->n_output_dims <= 16, "rtxn_mlp_create: n_output_dims = %d",
               cfg->n_output_dims);
  RTXN_REQUIRE(cfg->output_activation == RTXN_ACT_NONE || cfg->output_activation == RTXN_ACT_SIGMOID,
               "rtxn_mlp_create: unknown output_activation %d", cfg->output_activation);
  RTXN_REQUIRE(cfg->encoding == RTXN_ENC_FREQUENCY || cfg->encoding == RTXN_ENC_EXTERNAL,
               "rtxn_mlp_create: unknown encoding %d", cfg->encoding);
  int variant = -1;
  const auto& vs = variants();
  if (cfg->encoding == RTXN_ENC_FREQUENCY) {
    for (size_t i = 0; i < vs.size(); ++i)
      if (vs[i].W == cfg->n_neurons && vs[i].PD == cfg->n_pos_dims && vs[i].PF == cfg->n_pos_freqs &&
          vs[i].DD == cfg->n_dir_dims && vs[i].DF == cfg->n_dir_freqs)
        variant = (int)i;
    if (variant < 0) {
      rtxn::set_error("rtxn_mlp_create: no kernel for n_neurons=%d enc=(%d x %d, %d x %d); built: 64/128 wide (3x10, 2x12|2x4), 256 wide (3x10, 2x12)",
                      cfg->n_neurons, cfg->n_pos_dims, cfg->n_pos_freqs, cfg->n_dir_dims, cfg->n_dir_freqs);
      return RTXN_ERR_UNSUPPORTED;
    }
  } else {
    if (cfg->n_neurons != 64 && cfg->n_neurons != 128) {
      rtxn::set_error("rtxn_mlp_create: n_neurons = %d; built: 64, 128", cfg->n_neurons);
      return RTXN_ERR_UNSUPPORTED;
    }
    RTXN_REQUIRE(cfg->n_encoded_features >= 16 && cfg->n_encoded_features <= 256 && cfg->n_encoded_features % 16 == 0,
                 "rtxn_mlp_create: n_encoded_features = %d must be a multiple of 16 in [16,256]", cfg->n_encoded_features);
  }
  rtxn_mlp* m = new rtxn_mlp();
  m->cfg = *cfg;
  m->variant = variant;
  if (cfg->encoding == RTXN_ENC_FREQUENCY) {
    m->enc_width = 2 * (cfg->n_pos_dims * cfg->n_pos_freqs + cfg->n_dir_dims * cfg->n_dir_freqs);
    m->enc_padded = (m->enc_width + 15) / 16 * 16;
    m->k0 = vs[variant].k0;
  } else {
    m->enc_width = m->enc_padded = cfg->n_encoded_features;
    m->k0 = m->enc_padded;
  }
  const long W = cfg->n_neurons, E = m->enc_padded, L = cfg->n_hidden_layers;
  m->n_params = W * E + (L - 1) * W * W + 16 * W;
  const long RT = W / 32, KS = W / 16;
  m->packed_bytes = variant >= 0 ? (size_t)((m->k0 / 16) * RT + (L - 1) * KS * RT + KS) * 1024 : 0;
  // MFMA shape of the fused inference kernel: 16x16x32 where the variant has it, unless RTXN_MFMA_SHAPE=32 asks for the
  // 32x32x16 kernel (kept for A/B runs and for the per-segment compositor epilogue); read when the model is created
  m->mfma16 = 0;
  m->packed16 = nullptr;
  m->packed16_bytes = 0;
  if (variant >= 0 && vs[variant].k0_16 > 0) {
    const char* shape = getenv("RTXN_MFMA_SHAPE");
    m->mfma16 = !(shape && atoi(shape) == 32);
    m->packed16_bytes = (size_t)((vs[variant].k0_16 / 32) * (W / 16) + (L - 1) * (W / 32) * (W / 16) + 4 * (W / 32)) * 1024;
  }
  m->packed_train_bytes = (size_t)((E / 16) * RT + (L - 1) * KS * RT + KS) * 1024;
  m->packed_t_bytes = (size_t)(RT + (L - 1) * RT * KS + ((E + 31) / 32) * KS) * 1024;
  m->packed = m->packed_train = m->packed_t = nullptr;
  m->inference_ready = 0;
  *out = m;
  return RTXN_OK;
}

extern "C" int rtxn_mlp_destroy(rtxn_mlp* m) {
  if (!m) return RTXN_OK;
  if (m->packed) (void)hipFree(m->packed);
  if (m->packed_train) (void)hipFree(m->packed_train);
  if (m->packed_t) (void)hipFree(m->packed_t);
  if (m->packed16) (void)hipFree(m->packed16);
  delete m;
  return RTXN_OK;
}

extern "C" int rtxn_mlp_set_reserved_cus(rtxn_mlp* m, int n_cus) {
  RTXN_REQUIRE(m != nullptr, "rtxn_mlp_set_reserved_cus: NULL model");
  RTXN_REQUIRE(n_cus >= 0 && n_cus <= 64, "rtxn_mlp_set_reserved_cus: n_cus = %d out of [0,64]", n_cus);
  m->reserved_cus = n_cus;
  return RTXN_OK;
}

extern "C" int rtxn_mlp_mfma_shape(const rtxn_mlp* m) {
  if (!m || m->variant < 0) return 0;
  return m->mfma16 ? 16 : 32;
}

extern "C" long rtxn_mlp_n_params(const rtxn_mlp* m) { return m ? m->n_params : -1; }
extern "C" int rtxn_mlp_padded_output_width(const rtxn_mlp* m) { return m ? 16 : -1; }
extern "C" int rtxn_mlp_encoded_width(const rtxn_mlp* m) { return m ? m->enc_padded : -1; }

extern "C" int rtxn_mlp_initialize_params(const rtxn_mlp* m, uint64_t seed, float* host_params_fp32) {
  RTXN_REQUIRE(m && host_params_fp32, "rtxn_mlp_initialize_params: NULL argument");
  Pcg32 rng(seed);
  const long W = m->cfg.n_neurons;
  float* p = host_params_fp32;
  auto fill = [&](long rows, long cols) {
    const float scale = std::sqrt(6.0f / (float)(rows + cols));  // Xavier uniform
    for (long i = 0; i < rows * cols; ++i) *p++ = (rng.next_float() * 2.0f - 1.0f) * scale;
  };
  fill(W, m->enc_padded);
  for (int l = 1; l < m->cfg.n_hidden_layers; ++l) fill(W, W);
  fill(16, W);
  return RTXN_OK;
}

static int set_params_impl(rtxn_mlp* m, const void* params_fp16, rtxn_stream_t stream, bool inference);

extern "C" int rtxn_mlp_set_params(rtxn_mlp* m, const void* params_fp16, rtxn_stream_t stream) {
  RTXN_REQUIRE(m && params_fp16, "rtxn_mlp_set_params: NULL argument");
  return set_params_impl(m, params_fp16, stream, true);
}

extern "C" int rtxn_mlp_set_params_training(rtxn_mlp* m, const void* params_fp16, rtxn_stream_t stream) {
  RTXN_REQUIRE(m && params_fp16, "rtxn_mlp_set_params_training: NULL argument");
  return set_params_impl(m, params_fp16, stream, false);
}

static int set_params_impl(rtxn_mlp* m, const void* params_fp16, rtxn_stream_t stream, bool inference) {
  RTXN_DEVICE_OR_FAIL();
  // training-only update: the inference packings are neither allocated nor refreshed, and say so (check_ready)
  m->inference_ready = 0;
  if (inference && m->packed_bytes && !m->packed) RTXN_HIP(hipMalloc(&m->packed, m->packed_bytes));
  if (!m->packed_train) RTXN_HIP(hipMalloc(&m->packed_train, m->packed_train_bytes));
  if (!m->packed_t) RTXN_HIP(hipMalloc(&m->packed_t, m->packed_t_bytes));
  void* dst[3] = {m->packed, m->packed_train, m->packed_t};
  const size_t bytes[3] = {m->packed_bytes, m->packed_train_bytes, m->packed_t_bytes};
  for (int mode = inference ? 0 : 1; mode < 3; ++mode) {
    if (!bytes[mode]) continue;
    const long total = (long)(bytes[mode] / 2);
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    pack_kernel<<<blocks, 256, 0, rtxn::as_stream(stream)>>>(static_cast<const _Float16*>(params_fp16),
                                                             static_cast<_Float16*>(dst[mode]), m->cfg.n_neurons,
                                                             m->enc_padded, m->k0, m->cfg.n_hidden_layers, mode);
    RTXN_LAUNCH_CHECK("pack_kernel");
  }
  if (m->packed16_bytes && inference) {
    if (!m->packed16) RTXN_HIP(hipMalloc(&m->packed16, m->packed16_bytes));
    const long total = (long)(m->packed16_bytes / 2);
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    const Enc16Dims d{m->cfg.n_pos_dims, m->cfg.n_pos_freqs, m->cfg.n_dir_dims, m->cfg.n_dir_freqs, m->enc_padded,
                      variants()[m->variant].k0_16};
    pack16_kernel<<<blocks, 256, 0, rtxn::as_stream(stream)>>>(static_cast<const _Float16*>(params_fp16),
                                                               static_cast<_Float16*>(m->packed16), m->cfg.n_neurons,
                                                               m->cfg.n_hidden_layers, d);
    RTXN_LAUNCH_CHECK("pack16_kernel");
  }
  if (inference) m->inference_ready = 1;
  return RTXN_OK;
}

static int check_ready(const rtxn_mlp* m, const char* who) {
  if (!m) { rtxn::set_error("%s: NULL model", who); return RTXN_ERR_INVALID; }
  if (m->variant < 0) {
    rtxn::set_error("%s: this model takes pre-encoded input (RTXN_ENC_EXTERNAL); use rtxn_mlp_train_forward", who);
    return RTXN_ERR_UNSUPPORTED;
  }
  if (!m->packed || (m->mfma16 && !m->packed16)) { rtxn::set_error("%s: rtxn_mlp_set_params has not been called", who); return RTXN_ERR_INVALID; }
  if (!m->inference_ready) {
    rtxn::set_error("%s: the parameters were last set with rtxn_mlp_set_params_training, which leaves the fused inference kernels' "
                    "weights stale; call rtxn_mlp_set_params before rendering", who);
    return RTXN_ERR_INVALID;
  }
  return RTXN_OK;
}

extern "C" int rtxn_mlp_forward(const rtxn_mlp* m, const float* input, void* output_half, long n,
                                rtxn_stream_t stream) {
  int rc = check_ready(m, "rtxn_mlp_forward");
  if (rc != RTXN_OK) return rc;
  RTXN_REQUIRE(n >= 0, "rtxn_mlp_forward: n = %ld < 0", n);
  RTXN_DEVICE_OR_FAIL();
  if (n == 0) return RTXN_OK;
  RTXN_REQUIRE(input && output_half, "rtxn_mlp_forward: NULL buffer");
  RTXN_REQUIRE(((uintptr_t)output_half & 7) == 0, "rtxn_mlp_forward: output must be 8-byte aligned");
  FwdArgs a;
  memset(&a, 0, sizeof(a));
  a.input = input;
  a.n = n;
  a.out_half = static_cast<_Float16*>(output_half);
  return launch_fwd(m, a, 0, 0, n, rtxn::as_stream(stream));
}

extern "C" int rtxn_mlp_forward_radiance(const rtxn_mlp* m, const float* input, float* radiance, long n,
                                         rtxn_stream_t stream) {
  int rc = check_ready(m, "rtxn_mlp_forward_radiance");
  if (rc != RTXN_OK) return rc;
  RTXN_REQUIRE(n >= 0, "rtxn_mlp_forward_radiance: n = %ld < 0", n);
  RTXN_DEVICE_OR_FAIL();
  if (n == 0) return RTXN_OK;
  RTXN_REQUIRE(input && radiance, "rtxn_mlp_forward_radiance: NULL buffer");
  RTXN_REQUIRE(((uintptr_t)radiance & 15) == 0, "rtxn_mlp_forward_radiance: radiance must be 16-byte aligned");
  FwdArgs a;
  memset(&a, 0, sizeof(a));
  a.input = input;
  a.n = n;
  a.radiance = reinterpret_cast<float4*>(radiance);
  return launch_fwd(m, a, 0, 1, n, rtxn::as_stream(stream));
}

extern "C" int rtxn_mlp_forward_segments(const rtxn_mlp* m, const float* start_points, const float* end_points,
                                         const float* seg_view, const int* total_segments, long max_segments,
                                         float* radiance, float* t_vals, rtxn_stream_t stream) {
  int rc = check_ready(m, "rtxn_mlp_forward_segments");
  if (rc != RTXN_OK) return rc;
  RTXN_REQUIRE(max_segments >= 0, "rtxn_mlp_forward_segments: max_segments = %ld < 0", max_segments);
  RTXN_DEVICE_OR_FAIL();
  if (max_segments == 0) return RTXN_OK;
  RTXN_REQUIRE(start_points && end_points && seg_view && total_segments && radiance,
               "rtxn_mlp_forward_segments: NULL buffer");
  RTXN_REQUIRE(((uintptr_t)radiance & 15) == 0, "rtxn_mlp_forward_segments: radiance must be 16-byte aligned");
  FwdArgs a;
  memset(&a, 0, sizeof(a));
  a.start = start_points;
  a.end = end_points;
  a.seg_view = seg_view;
  a.total_segments = total_segments;
  a.max_segments = max_segments;
  a.radiance = reinterpret_cast<float4*>(radiance);
  a.t_vals = t_vals;
  return launch_fwd(m, a, 1, 1, max_segments, rtxn::as_stream(stream));
}

extern "C" int rtxn_mlp_forward_segments_compact(const rtxn_mlp* m, const float* start_points, const float* end_points,
                                                 const float* seg_view, const int* total_segments, long max_segments,
                                                 void* radiance_half4, rtxn_stream_t stream) {
  int rc = check_ready(m, "rtxn_mlp_forward_segments_compact");
  if (rc != RTXN_OK) return rc;
  RTXN_REQUIRE(max_segments >= 0, "rtxn_mlp_forward_segments_compact: max_segments = %ld < 0", max_segments);
  RTXN_DEVICE_OR_FAIL();
  if (max_segments == 0) return RTXN_OK;
  RTXN_REQUIRE(start_points && end_points && seg_view && total_segments && radiance_half4,
               "rtxn_mlp_forward_segments_compact: NULL buffer");
  RTXN_REQUIRE(((uintptr_t)radiance_half4 & 7) == 0, "rtxn_mlp_forward_segments_compact: radiance must be 8-byte aligned");
  FwdArgs a;
  memset(&a, 0, sizeof(a));
  a.start = start_points;
  a.end = end_points;
  a.seg_view = seg_view;
  a.total_segments = total_segments;
  a.max_segments = max_segments;
  a.out_half = static_cast<_Float16*>(radiance_half4);
  return launch_fwd(m, a, 1, 3, max_segments, rtxn::as_stream(stream));
}

extern "C" int rtxn_mlp_forward_segments_composite(const rtxn_mlp* m, const float* start_points, const float* end_points,
                                                   const float* seg_view, const uint8_t* seg_first,
                                                   const int* total_segments, long max_segments, float* seg_out, int mode,
                                                   float step_scale, rtxn_stream_t stream) {
  int rc = check_ready(m, "rtxn_mlp_forward_segments_composite");
  if (rc != RTXN_OK) return rc;
  RTXN_REQUIRE(max_segments >= 0, "rtxn_mlp_forward_segments_composite: max_segments = %ld < 0", max_segments);
  RTXN_REQUIRE(mode == RTXN_VR_COMPAT || mode == RTXN_VR_NERF, "rtxn_mlp_forward_segments_composite: unknown mode %d", mode);
  RTXN_DEVICE_OR_FAIL();
  if (max_segments == 0) return RTXN_OK;
  RTXN_REQUIRE(start_points && end_points && seg_view && total_segments && seg_out,
               "rtxn_mlp_forward_segments_composite: NULL buffer");
  RTXN_REQUIRE(mode != RTXN_VR_COMPAT || seg_first, "rtxn_mlp_forward_segments_composite: COMPAT mode needs seg_first");
  RTXN_REQUIRE(((uintptr_t)seg_out & 15) == 0, "rtxn_mlp_forward_segments_composite: seg_out must be 16-byte aligned");
  FwdArgs a;
  memset(&a, 0, sizeof(a));
  a.start = start_points;
  a.end = end_points;
  a.seg_view = seg_view;
  a.seg_first = seg_first;
  a.total_segments = total_segments;
  a.max_segments = max_segments;
  a.seg_out = reinterpret_cast<float4*>(seg_out);
  a.vr_mode = mode;
  a.step_scale = step_scale;
  return launch_fwd(m, a, 1, 2, max_segments, rtxn::as_stream(stream));
}

#ifdef RTXN_STAMPS
// diagnostic builds only: the stamps of the last launch of mlp_fwd16_kernel (8 waves x 4 tiles x 24 slots)
extern "C" int rtxn_debug_read_stamps(unsigned* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), sizeof(unsigned) * 8 * kStampTiles * kStampSlots) == hipSuccess ? 0 : 1;
}
#endif
