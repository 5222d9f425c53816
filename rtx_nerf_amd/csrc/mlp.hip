// Frequency encoding + fully-fused MLP forward on MFMA.
// Replaces the tiny-cuda-nn surface main.cu uses for inference: create_from_config
// (main.cu:35-69,325), n_params/set_params/initialize_params (:327-349),
// network->forward (:721) and the convertHalfToFloat glue (:203-208,723-728); the
// segment variant also folds launchSampler (REGULAR, sampler/sampler.cu:52-66) in.
// tiny-cuda-nn itself is an un-vendored, unpinned submodule: the numerics below are
// this build's restatement of its published algorithm (see oracle/rtxn_oracle.c).
//
// Design (gfx950, wave64, v_mfma_f32_16x16x32_f16)
//   * The network is evaluated TRANSPOSED: H_{l+1}^T [W x samples] = W_l [W x K] . H_l^T.
//     A 16x16 f32 accumulator tile has its column (= sample) on the lane and four rows
//     (= features) in its registers; two consecutive row tiles are exactly the B operand of
//     one k-step of the next MFMA once pairs of registers are packed to f16 (mlp_internal.h,
//     perm_feature16).  Activations therefore never leave registers between layers: no LDS
//     round trip, no barrier on the activation path; the k order this imposes is baked into
//     the weight packing.
//   * Weights are pre-packed (rtxn_mlp_set_params) into 1-KiB "A fragments": chunk
//     (layer, row-tile, k-step) holds lane l's 8 halves at byte l*16, so the LDS image
//     is lane-linear: staged with global_load_lds (16 B/lane, no VGPR round trip) and
//     read back with one conflict-free ds_read_b128 per fragment.  All blocks read the same
//     0.03-1 MB of packed weights, so the stream is served by L2, not HBM.
//   * 64/128 wide: one 512-thread block per CU, a wave owns 64 samples as four 16-column tiles.
//     Waves 0-3 (group A) and 4-7 (group B, the other wave of each SIMD) run the same program
//     ONE STAGE APART, so that one group's VALU-bound encode / epilogue always meets the other
//     group's MFMA-bound layers.  The offset costs nothing to arrange: B executes one extra
//     barrier before its first stage and A one after its last (s_barrier only counts arrivals).
//     What it needs is LDS: layer 0 and the output layer stay resident (fetched once per launch),
//     hidden layers stream through a ring of THREE slots (A's stage, B's stage, the one being
//     fetched); every wave fetches its share of the stage group A needs next.
//   * 256 wide: a layer is 128 KiB of fragments; it streams through three 32-KiB slots in chunks
//     of four row tiles, one barrier per chunk (mlp_fwd256x16_kernel).
//   * First layer: the encoding is computed straight into B fragments (EncSpec16, octave_unit).
//
// MFMA-bound: 2*(enc_padded*W + (L-1)*W^2 + 16*W) FLOP per sample (262,144 for the
// reference's 8x128 model), against 8-28 B/sample of HBM traffic.
//
// (Rounds 1-2 also carried the same kernels on v_mfma_f32_32x32x16_f16 -- builtin MFMAs beside asm convert units --
// and a per-segment compositor epilogue on them; both are gone: the 16x16x32 shape measured 3-5 % faster on this
// power-limited chip, the epilogue 6.5 % slower than the separate compositor, and the builtin + asm construct is the
// one DESIGN.md 3.4 records as having produced timing-dependent wrong values.)
#include "common.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include "mlp_internal.h"

namespace {

// Build-time knobs of the inference kernels (defaults are the measured best):
//   RTXN_SHARE_DIR 1: segment input computes a segment's direction features once per lane group (DirShare16)
//   RTXN_PIPE16 depth of the A-fragment register ring                        -- mlp_internal.h
//   RTXN_STAMPS diagnostic build: per-stage s_memtime stamps of block 0 (tools/probe/stamps.py); never in the shipped library
#ifndef RTXN_SHARE_DIR
#define RTXN_SHARE_DIR 1
#endif
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
  _Float16* out_half;  // [n][16] (OUT_MODE 0) or [n][4] (OUT_MODE 3)
  float4* radiance;    // [n]
  float* t_vals;       // [n] or NULL (IN_MODE 1 only)
};

// ---------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------
// params (tcnn layout): layer 0 [W][E], hidden [W][W] x (L-1), out [16][W], row-major fp16 (E = enc_padded).
// A "fragment" is 1 KiB: 64 lanes x 8 halves at byte 16*lane.
//
// pack16_kernel (inference, v_mfma_f32_16x16x32_f16): fragments are 16 rows x 32 k, lane l = (r = l&15, g = l>>4),
//   chunks [rowtile16][kstep32];
//   layer 0  : element (r,g ; kk ; j) = W0[16rt + r][enc16_feature(8kk + j, g)]  (0 where there is none), K = k0_16
//   others   : element = Wl[16rt + r][perm_feature16(kk,g,j)]                             (output layer: one 16-row tile)
// pack_kernel (training, v_mfma_f32_32x32x16_f16): fragments are 32 rows x 16 k, lane l = (r = l&31, h = l>>5);
//   MODE 1 (forward): per layer chunks [rowtile][kstep], element = Wl[32rt + r][perm_feature(kk,h,j)], K = E / W
//   MODE 2 (backward, TRANSPOSED layers, stored in backward order out, L-1, ..., 0):
//     layer l: element = Wl[perm_feature(kk,h,j)][32rt + r], rows = in_width(l) padded to 32, K = out rows
//              (16 for the output layer: one k-step; W otherwise).
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
//   d.PD < 0: a model on PRE-ENCODED input (RTXN_ENC_EXTERNAL): layer 0 takes its E features in the accumulator-permuted k
//   order perm_feature16 like every other layer (hashmlp.hip builds its B fragments in that order), K = d.k0 = E rounded up to 32.
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
    const int feat = (layer == 0 && d.PD >= 0) ? enc16_feature(d, 8 * kk + j, g) : rtxn::perm_feature16(kk, g, j);
    const long base = layer == 0 ? 0 : (long)W * E + (long)(layer - 1) * W * W;
    _Float16 v = (_Float16)0.0f;
    if (row < rows && feat >= 0 && feat < in_w) v = params[base + (long)row * in_w + feat];
    packed[e] = v;
  }
}

__global__ void pack_kernel(const _Float16* __restrict__ params, _Float16* __restrict__ packed, int W, int E,
                            int n_hidden, int mode) {
  const int RT = W / 32;
  const int l0_ks = E / 16;
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
      const int feat = rtxn::perm_feature(kk, h, j);
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
// The 64/128-wide kernel on v_mfma_f32_16x16x32_f16 (mlp_internal.h, pipe_layer16)
// ---------------------------------------------------------------------------
// Block geometry, LDS plan, staging protocol and wave-group skew: see the file header.  Who holds what: lane (c = l & 15, g = l >> 4) owns sample 16 ct + c of the wave's four 16-column tiles
// and, of every 32 features, the eight perm_feature16 gives its lane group.
// Layer 0 / the encoder.  Lane group g evaluates a block of FB = ceil(F/4) consecutive frequencies of every input dimension
// (enc16_feature): the inputs are pre-scaled once per tile by 2^(FB g) / 2 (exact), the block's lowest octave comes from ONE
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

// One dimension of one sample: NK dwords {sin, cos} of octaves 0..NK-1 of the lane group's block.  xg = x 2^(FB g) / 2: the
// argument of the block's lowest octave in TURNS (sin(2^k pi x) = sin(2 pi (2^k x / 2))), the halving folded into the tile's one
// exact pre-scaling (round 3; v_fract stays: v_sin_f32 returns 0 outside +-256 turns, and the C ABI takes any position).
template <int NK>
__device__ __forceinline__ void octave_unit(float xg, int (&d)[3]) {
  float t, u, sn, cs;
  if constexpr (NK == 3) {
    asm volatile(
        "v_fract_f32 %3, %7\n\t"
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
        "v_fract_f32 %2, %6\n\t"
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
        "v_fract_f32 %1, %4\n\t"
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

// OUT_MODE 0: half[n][16]; 1: float4 radiance (+ t_vals); 3: compact half4.
template <int W, int PD, int PF, int DD, int DF, int IN_MODE, int OUT_MODE>
__global__ __launch_bounds__(512, 2) void mlp_fwd16_kernel(FwdArgs a) {
  static_assert(RTXN_NW == 8, "8-wave blocks");
  static_assert(OUT_MODE == 0 || OUT_MODE == 1 || OUT_MODE == 3, "output modes: half16, radiance, compact half4");
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
  const int grp = wave_u >> 2;              // 0: leading wave group, 1: one stage behind (file header)
  const int n_hid = n_layers - 2;
  stage<L0_BYTES, THREADS>(a.packed, smem, tid);
  stage<OUT_BYTES, THREADS>(a.packed + layer_off(n_layers - 1), smem + L0_BYTES, tid);
  int qs = 0;                               // hidden stages this wave has begun (ring slot = qs % 3)

  const float pos_scale = 0.5f * (float)(1u << (ES::FBP * g)), dir_scale = 0.5f * (float)(1u << (ES::FBD * g));   // 2^(FB g) / 2 turns per unit: see EncSpec16
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
// stream through a ring of three 32-KiB slots in chunks of four 16-row tiles (4 x 8 k-steps x 1 KiB),
// two chunks ahead of the MFMAs; one barrier per chunk.  A wave owns one 32-sample segment
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

// The 256-wide kernel.  A block is 8 waves = 8 segments = 256 samples; a wave's 32 samples are two 16-column tiles, lane (c, g)
// owns samples c and 16 + c and of every 32 features the eight perm_feature16 gives its lane group; a chunk is four 16-row
// tiles x eight 32-wide k-steps (pipe_chunk16), the encoder is mlp_fwd16_kernel's (lane-group frequency blocks, angle doubling,
// direction shared across the segment), and the output layer is multiplied in two row-rotated variants so that column tile
// v's (r, g, b, sigma) land in lane group v.  One 1-KiB A fragment feeds 2 x 8 MFMA passes.
template <int PD, int PF, int DD, int DF, int IN_MODE, int OUT_MODE>
__global__ __launch_bounds__(kThreads256, 2) void mlp_fwd256x16_kernel(FwdArgs a) {
  static_assert(OUT_MODE == 0 || OUT_MODE == 1 || OUT_MODE == 3, "output modes: half16, radiance, compact half4");
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

  const float pos_scale = 0.5f * (float)(1u << (ES::FBP * g)), dir_scale = 0.5f * (float)(1u << (ES::FBD * g));   // 2^(FB g) / 2 turns per unit: see EncSpec16
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
  fwd_fn fn[2][4];    // [IN_MODE][OUT_MODE]: OUT_MODE 0 half[n][16], 1 float4 radiance (+ t_vals), 3 compact half4 (IN_MODE 1 only)
  int k0;             // first-layer K as staged
  size_t lds;
  int threads;        // block size
  int blocks_per_cu;  // persistent grid = CUs x this
  int tile;           // samples per block per iteration (segments: tile / 32)
};

template <int W, int PD, int PF, int DD, int DF>
Variant make_variant() {
  static_assert(RTXN_NW == 8, "8-wave blocks");
  using ES = EncSpec16<PD, PF, DD, DF>;
  Variant v;
  memset(&v, 0, sizeof(v));
  v.W = W; v.PD = PD; v.PF = PF; v.DD = DD; v.DF = DF;
  v.fn[0][0] = mlp_fwd16_kernel<W, PD, PF, DD, DF, 0, 0>;
  v.fn[0][1] = mlp_fwd16_kernel<W, PD, PF, DD, DF, 0, 1>;
  v.fn[1][0] = mlp_fwd16_kernel<W, PD, PF, DD, DF, 1, 0>;
  v.fn[1][1] = mlp_fwd16_kernel<W, PD, PF, DD, DF, 1, 1>;
  v.fn[1][3] = mlp_fwd16_kernel<W, PD, PF, DD, DF, 1, 3>;
  v.k0 = ES::k0;
  // [layer 0 | output layer x 4 rotations | 3 ring slots of one hidden layer]
  v.lds = (size_t)(ES::k0 / 32) * (W / 16) * 1024 + 4 * (size_t)(W / 32) * 1024 + 3 * (size_t)(W / 32) * (W / 16) * 1024;
  v.threads = 512;
  v.blocks_per_cu = 1;
  v.tile = 512;
  return v;
}

template <int PD, int PF, int DD, int DF>
Variant make_variant256() {
  Variant v;
  memset(&v, 0, sizeof(v));
  v.W = 256; v.PD = PD; v.PF = PF; v.DD = DD; v.DF = DF;
  v.fn[0][0] = mlp_fwd256x16_kernel<PD, PF, DD, DF, 0, 0>;
  v.fn[0][1] = mlp_fwd256x16_kernel<PD, PF, DD, DF, 0, 1>;
  v.fn[1][0] = mlp_fwd256x16_kernel<PD, PF, DD, DF, 1, 0>;
  v.fn[1][1] = mlp_fwd256x16_kernel<PD, PF, DD, DF, 1, 1>;
  v.fn[1][3] = mlp_fwd256x16_kernel<PD, PF, DD, DF, 1, 3>;
  v.k0 = EncSpec16<PD, PF, DD, DF>::k0;
  v.lds = 3 * (size_t)kSlot256;
  v.threads = kThreads256;
  v.blocks_per_cu = 1;
  v.tile = 256;
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
  a.packed = static_cast<const uint8_t*>(m->packed);
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
  fwd_fn fn = v.fn[in_mode][out_mode];
  if (!fn) { rtxn::set_error("mlp forward: no kernel for input mode %d / output mode %d", in_mode, out_mode); return RTXN_ERR_UNSUPPORTED; }
  {
    constexpr int kMaxDev = 64;
    static std::mutex mu;
    static bool attr_set[kMaxDev][16][2][4] = {};
    std::lock_guard<std::mutex> lock(mu);
    const bool known = dev >= 0 && dev < kMaxDev && attr_set[dev][m->variant][in_mode][out_mode];
    if (!known) {
      RTXN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds));
      if (dev >= 0 && dev < kMaxDev) attr_set[dev][m->variant][in_mode][out_mode] = true;
    }
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3((unsigned)v.threads), v.lds, s, a);
  RTXN_LAUNCH_CHECK("mlp_fwd16_kernel");
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
    m->k0 = (m->enc_padded + 31) / 32 * 32;
  }
  const long W = cfg->n_neurons, E = m->enc_padded, L = cfg->n_hidden_layers;
  m->n_params = W * E + (L - 1) * W * W + 16 * W;
  const long RT = W / 32, KS = W / 16;
  // fused inference kernels: [layer 0: k0/32 k-steps x W/16 row tiles | hidden layers | output layer x 4 rotations] KiB
  // (RTXN_ENC_EXTERNAL: the same layout with layer 0 in perm_feature16 order, read by hashmlp.hip)
  m->packed_bytes = (size_t)((m->k0 / 32) * (W / 16) + (L - 1) * (W / 32) * (W / 16) + 4 * (W / 32)) * 1024;
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
  delete m;
  return RTXN_OK;
}

extern "C" int rtxn_mlp_set_reserved_cus(rtxn_mlp* m, int n_cus) {
  RTXN_REQUIRE(m != nullptr, "rtxn_mlp_set_reserved_cus: NULL model");
  RTXN_REQUIRE(n_cus >= 0 && n_cus <= 64, "rtxn_mlp_set_reserved_cus: n_cus = %d out of [0,64]", n_cus);
  m->reserved_cus = n_cus;
  return RTXN_OK;
}

extern "C" int rtxn_mlp_mfma_shape(const rtxn_mlp* m) { return (!m || m->variant < 0) ? 0 : 16; }

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
  // training-only update: the inference packing is neither allocated nor refreshed, and says so (check_ready)
  m->inference_ready = 0;
  if (!m->packed_train) RTXN_HIP(hipMalloc(&m->packed_train, m->packed_train_bytes));
  if (!m->packed_t) RTXN_HIP(hipMalloc(&m->packed_t, m->packed_t_bytes));
  void* dst[3] = {nullptr, m->packed_train, m->packed_t};
  const size_t bytes[3] = {0, m->packed_train_bytes, m->packed_t_bytes};
  for (int mode = 1; mode < 3; ++mode) {
    const long total = (long)(bytes[mode] / 2);
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    pack_kernel<<<blocks, 256, 0, rtxn::as_stream(stream)>>>(static_cast<const _Float16*>(params_fp16),
                                                             static_cast<_Float16*>(dst[mode]), m->cfg.n_neurons,
                                                             m->enc_padded, m->cfg.n_hidden_layers, mode);
    RTXN_LAUNCH_CHECK("pack_kernel");
  }
  // a pre-encoded model's 16x16x32 packing is what BOTH its inference kernel (hashmlp.hip) and a training loop's renders read:
  // refreshed by either entry point (40 KiB for the 4x64 model)
  const bool external = m->cfg.encoding == RTXN_ENC_EXTERNAL;
  if (m->packed_bytes && (inference || external)) {
    if (!m->packed) RTXN_HIP(hipMalloc(&m->packed, m->packed_bytes));
    const long total = (long)(m->packed_bytes / 2);
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    const Enc16Dims d{external ? -1 : m->cfg.n_pos_dims, m->cfg.n_pos_freqs, m->cfg.n_dir_dims, m->cfg.n_dir_freqs, m->enc_padded, m->k0};
    pack16_kernel<<<blocks, 256, 0, rtxn::as_stream(stream)>>>(static_cast<const _Float16*>(params_fp16),
                                                               static_cast<_Float16*>(m->packed), m->cfg.n_neurons,
                                                               m->cfg.n_hidden_layers, d);
    RTXN_LAUNCH_CHECK("pack16_kernel");
  }
  if (inference || external) m->inference_ready = 1;
  return RTXN_OK;
}

static int check_ready(const rtxn_mlp* m, const char* who) {
  if (!m) { rtxn::set_error("%s: NULL model", who); return RTXN_ERR_INVALID; }
  if (m->variant < 0) {
    rtxn::set_error("%s: this model takes pre-encoded input (RTXN_ENC_EXTERNAL); use rtxn_mlp_train_forward", who);
    return RTXN_ERR_UNSUPPORTED;
  }
  if (!m->packed) { rtxn::set_error("%s: rtxn_mlp_set_params has not been called", who); return RTXN_ERR_INVALID; }
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

#ifdef RTXN_STAMPS
// diagnostic builds only: the stamps of the last launch of mlp_fwd16_kernel (8 waves x 4 tiles x 24 slots)
extern "C" int rtxn_debug_read_stamps(unsigned* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), sizeof(unsigned) * 8 * kStampTiles * kStampSlots) == hipSuccess ? 0 : 1;
}
#endif
