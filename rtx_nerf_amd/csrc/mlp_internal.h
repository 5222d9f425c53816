// Internal (non-ABI) declarations shared by mlp.hip (inference) and train.hip (training).
#pragma once
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef short short2v __attribute__((ext_vector_type(2)));

struct rtxn_mlp {
  rtxn_mlp_config cfg;
  int enc_width;    // encoded features before padding
  int enc_padded;   // multiple of 16, padding features are 1.0 (= first-layer in_width of the params)
  int k0;           // inference kernel: first-layer K as staged (multiple of 32 covering all encode dwords)
  long n_params;
  int variant;      // index into the inference kernel table, -1 if this model has no fused inference kernel
  int reserved_cus; // CUs the persistent inference grid leaves free (rtxn_mlp_set_reserved_cus)
  // device buffers owned by the model, (re)built by rtxn_mlp_set_params
  void* packed;       // inference (v_mfma_f32_16x16x32_f16): A fragments, layer 0 in the lane-group frequency-block order (pack16_kernel)
  size_t packed_bytes;
  void* packed_train; // training forward: A fragments, every layer in the accumulator-permuted k order
  size_t packed_train_bytes;
  void* packed_t;     // training backward: A fragments of the TRANSPOSED layers (dA = W^T dZ)
  size_t packed_t_bytes;
  int inference_ready; // the fused inference kernels' packings (packed / packed16) hold the CURRENT parameters: set by
                       // rtxn_mlp_set_params, cleared by rtxn_mlp_set_params_training (which re-packs the training layouts only)
};

namespace rtxn {

// hashmlp.hip: network->forward on pre-encoded input, outputs only, on the 16x16x32 all-asm pipeline with every weight resident
// in LDS (64-wide models).  total_segments != NULL: the live sample count is read on the device (32 x *total_segments,
// clamped to `capacity` segments) and n_samples is the capacity the grid is sized for.
bool enc_forward16_supported(const rtxn_mlp* m);
int launch_enc_forward16(const rtxn_mlp* m, const void* encT, long n_samples, long Sp, const int* total_segments, int capacity,
                         void* output_half, float* radiance, hipStream_t stream);

// The output activation of every MLP kernel: 1 / (1 + exp(-z)) with the hardware exponential and reciprocal (v_exp_f32,
// v_rcp_f32: 1 ulp each).  A plain `1.0f / x` is an IEEE division, ten instructions per value in an epilogue that sits on the
// block's critical path; the result is rounded to fp16 right behind it.
__device__ __forceinline__ float sigmoidf_fast(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }

// Stage BYTES of lane-linear A fragments from global memory into LDS with LDS-DMA
// (global_load_lds, 16 B per lane).  All 256 threads of the block call it.
template <int BYTES, int THREADS = 256>
__device__ __forceinline__ void stage(const uint8_t* __restrict__ g, uint8_t* lds_buf, int tid) {
  static_assert(BYTES % 1024 == 0, "layer bytes must be whole 1-KiB fragments");
  constexpr int ROUND = THREADS * 16;  // bytes per round: one 1-KiB fragment per wave
#pragma unroll
  for (int i = 0; i < BYTES / ROUND; ++i) {
    const uint8_t* src = g + i * ROUND + tid * 16;
    uint8_t* dst = lds_buf + i * ROUND + (tid & ~63) * 16;  // wave-uniform base; HW adds lane*16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
  constexpr int TAIL = (BYTES % ROUND) / 1024;  // whole fragments left: one wave each
  if (TAIL > 0 && (tid >> 6) < TAIL) {
    const int off = (BYTES / ROUND) * ROUND;
    const uint8_t* src = g + off + tid * 16;
    uint8_t* dst = lds_buf + off + (tid & ~63) * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
}

// runtime-sized variant (bytes % 1024 == 0)
__device__ __forceinline__ void stage_rt(const uint8_t* __restrict__ g, uint8_t* lds_buf, int bytes, int tid) {
  for (int off = (tid >> 6) * 1024; off < bytes; off += 4096) {
    const uint8_t* src = g + off + (tid & 63) * 16;
    uint8_t* dst = lds_buf + off;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
}

// Barrier that publishes LDS-DMA-staged fragments to the workgroup.  LDS-DMA completion is tracked by vmcnt only, and
// hipcc does NOT reliably drain vmcnt in front of s_barrier (gfx950 has back-off barriers; the wait appeared only where
// an unrelated dependency forced one, and a re-ordered kernel lost it: rare whole-tile errors, found by the re-render
// determinism test).  Every wave must have its own DMA pieces landed BEFORE it arrives, so the wait is explicit.
__device__ __forceinline__ void staged_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// f32 pair -> packed f16 (v_cvt_pk_f16_f32), optional ReLU as a signed-integer max on the packed
// halves (v_pk_max_i16): a negative half is a negative int16 and rounding is monotone, so this is
// fp16(max(x,0)) at one VALU op per two values and without the canonicalising v_max hipcc puts in
// front of fmaxf on MFMA results.
template <bool RELU>
__device__ __forceinline__ half2v pack2(float a, float b) {
  float2v f = {a, b};
  half2v hv = __builtin_convertvector(f, half2v);
  if (RELU) {
    short2v sv = __builtin_bit_cast(short2v, hv);
    sv = __builtin_elementwise_max(sv, (short2v)0);
    hv = __builtin_bit_cast(half2v, sv);
  }
  return hv;
}

// registers 8s..8s+7 of a 32x32 accumulator tile -> the B fragment of k-step s of the next MFMA
template <bool RELU>
__device__ __forceinline__ half8 pack8(const floatx16& c, int s) {
  const half2v p0 = pack2<RELU>(c[8 * s + 0], c[8 * s + 1]);
  const half2v p1 = pack2<RELU>(c[8 * s + 2], c[8 * s + 3]);
  const half2v p2 = pack2<RELU>(c[8 * s + 4], c[8 * s + 5]);
  const half2v p3 = pack2<RELU>(c[8 * s + 6], c[8 * s + 7]);
  const half4v q0 = __builtin_shufflevector(p0, p1, 0, 1, 2, 3);
  const half4v q1 = __builtin_shufflevector(p2, p3, 0, 1, 2, 3);
  return __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ half8 relu_pack(const floatx16& c, int s) { return pack8<true>(c, s); }

// feature (row) index held in element j of lane-half h of k-step kk: the order in which a
// 32x32 accumulator tile hands its rows to the next MFMA as a B operand
__host__ __device__ __forceinline__ int perm_feature(int kk, int h, int j) { return 16 * kk + 8 * (j >> 2) + 4 * h + (j & 3); }

// One layer: out rows [32*rt, 32*rt+32) for rt < RT, K = 16*KS, for the wave's two column
// tiles.  Row-tile-outer: an accumulator is live for one row tile only, and its ReLU/convert
// (VALU) overlaps the next row tile's MFMAs.  A fragments: chunk (rt, kk) at ((rt*KS+kk)*64+lane)*16.
// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined layers for the inference kernel (cdna_hip_programming.md 5.7 form (iii)).
// Left to itself hipcc (a) sinks every A-fragment LDS load down to its consumer (ds_read; s_waitcnt lgkmcnt(0); mfma)
// and (b) keeps all RT accumulator tiles live and converts them after the layer's last MFMA, so inside one wave
// neither the LDS latency nor the ReLU/convert VALU work runs in the shadow of an MFMA.  Here the order is fixed by
// hand:
//   * A fragments come through a register ring of RTXN_PIPE slots filled with inline-asm ds_read_b128 that many
//     k-steps ahead, with a counted s_waitcnt lgkmcnt(N) in front of each consumer (nothing else in these functions
//     may touch LDS or SMEM);
//   * accumulators are double-buffered by row-tile parity: acc[rt & 1] collects row tile rt while the finished tile
//     rt-1 in acc[~rt & 1] is converted in slices of a few pack2 units issued right behind each k-step's MFMAs;
//   * the LAST row tile of a layer stays pending in acc[1] and is converted during the first k-steps of the NEXT
//     layer's (or the output layer's) row tile 0, which needs those two B fragments only at k-steps KS-2, KS-1;
//   * sched_barrier(0) between k-steps keeps hipcc from regrouping them.
// The converts are asm volatile: as plain code they are pure value computations and instruction selection places them
// after the layer's last MFMA wherever they are written.  hipcc's hazard recogniser cannot see into asm, so the
// MFMA-result -> VALU-read wait states are guaranteed by construction instead: a unit only ever reads an accumulator
// whose last MFMA is followed by at least one full k-step of MFMAs on the same (in-order, one matrix core) SIMD.
#ifndef RTXN_PIPE
#define RTXN_PIPE 3
#endif
#ifndef RTXN_NW
#define RTXN_NW 8   // waves per block of the 64/128-wide inference kernel (8: one 8-wave block per CU shares each staged layer)
#endif
typedef int int4v __attribute__((ext_vector_type(4)));

template <int OFF>
__device__ __forceinline__ void lds_read_frag(half8& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N));
  __builtin_amdgcn_sched_barrier(0);
}
// The same wait as a real instruction (gfx9 encoding: vmcnt and expcnt fields all ones).  Between two asm statements that
// share a register (ds_read -> the MFMA statement) hipcc's hazard recogniser assumes a forwarding hazard and pads with s_nop
// unless an instruction it can see sits between them; an asm s_waitcnt does not count, this one does.
template <int N>
__device__ __forceinline__ void lds_wait_insn() {
  __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8));
  __builtin_amdgcn_sched_barrier(0);
}

// The weights of the FOLLOWING stage are fetched while this one computes: one 1-KiB fragment per wave per k-step
// (LDS-DMA), issued behind that k-step's MFMAs so that neither the address arithmetic nor the M0 set-up costs a slot in
// which the matrix core idles.  Everything in a StageJob is wave-uniform.
struct StageJob {
  const uint8_t* g;   // fragments of the next stage in the packed buffer
  uint8_t* lds;       // the LDS buffer that is free during this stage
  int nfrags;         // 0: nothing to fetch
};
template <int C, int WAVES>
__device__ __forceinline__ void stage_chunk(const StageJob& sj, int wave_u, int lane) {
  const int frag = C * WAVES + wave_u;
  if (frag < sj.nfrags) {
    const uint8_t* base = sj.g + (size_t)frag * 1024;   // SGPR pair; the lane part stays a 32-bit VGPR offset (saddr form)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (unsigned)(lane << 4)),
                                     (__attribute__((address_space(3))) void*)(sj.lds + frag * 1024), 16, 0, 0);
  }
}

// Accumulators written by asm MFMAs and read next by ORDINARY code (a caller's conversion or epilogue): hipcc cannot see the
// MFMAs, so the MFMA-result -> VALU-read wait states (16 passes + margin) are spent by hand -- and the accumulators must go
// THROUGH the statement that spends them ("+v").  A bare `asm volatile("s_nop ..." ::: "memory")` orders nothing that lives in
// registers: in the outputs-only 128-wide training forward hipcc hoisted the caller's first v_cvt_pk_f16_f32 above it, and
// that one conversion read elements 0, 1 of the last column tile's pending row tile before the last k-step had landed
// (features 96, 97, 100, 101 of column tile 1 one k-step short, every layer; found with tools/probe/fwd_hidden_dump.py,
// profiles/r03/asm_tail_hazard.txt).  As operands of the statement no read of them can be scheduled above it.
template <typename Acc, int CT>
__device__ __forceinline__ void mfma_results_settle(Acc (&acc)[CT]) {
  static_assert(CT == 2 || CT == 4, "operand lists written for 2 or 4 column tiles");
  if constexpr (CT == 2) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
  else asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
}

// One k-step of the 32x32x16 pipeline (training forward, CT = 2) with NOTHING left to the compiler: the two MFMAs, accumulating
// in place, and the NUP ReLU/convert units of the finished row tile as ONE asm statement, units interleaved with the MFMAs
// (M0 u.. M1 u..).  Builtin MFMAs that hipcc may move, merge or re-allocate around asm units it cannot see into produced wrong,
// timing-dependent values three times in this kernel family (history of 3.4); as one statement per k-step the step
// is what the source says.  ZERO: first k-step of a row tile, C = 0.  Hazards, by construction: a unit reads a tile whose last
// MFMA is at least two 16-pass MFMAs behind it; nothing here reads `dst`.
#define RTXN_M32(ACC, B, C) "v_mfma_f32_32x32x16_f16 %[" ACC "], %[a], %[" B "], " C "\n\t"
// operands exchanged: the product comes out TRANSPOSED (lane = the weight tile's row, registers = samples) -- same k order, same values
#define RTXN_M32T(ACC, B, C) "v_mfma_f32_32x32x16_f16 %[" ACC "], %[" B "], %[a], " C "\n\t"
#define RTXN_PAIR32_CASES(M0, M1, ACC0, ACC1)                                                                                          \
  if constexpr (NUP == 0) asm volatile(M0 M1 : ACC0, ACC1 : RTXN_IN32);                                                                  \
  else if constexpr (NUP == 2)                                                                                                    \
    asm volatile(M0 RTXN_UNIT32("0") M1 RTXN_UNIT32("1") : ACC0, ACC1, RTXN_UOUT(0, 0), RTXN_UOUT(1, 1) : RTXN_IN32, RTXN_UIN(0, 0), RTXN_UIN(1, 1)); \
  else                                                                                                                            \
    asm volatile(M0 RTXN_UNIT32("0") RTXN_UNIT32("1") M1 RTXN_UNIT32("2") RTXN_UNIT32("3")                                         \
                 : ACC0, ACC1, RTXN_UOUT(0, 0), RTXN_UOUT(1, 1), RTXN_UOUT(2, 2), RTXN_UOUT(3, 3)                                         \
                 : RTXN_IN32, RTXN_UIN(0, 0), RTXN_UIN(1, 1), RTXN_UIN(2, 2), RTXN_UIN(3, 3));
#define RTXN_UNIT32(U) "v_cvt_pk_f16_f32 %[r" U "], %[x" U "], %[y" U "]\n\tv_pk_max_i16 %[r" U "], %[r" U "], 0\n\t"
#define RTXN_IN32 [a] "v"(a), [b0] "v"(b0), [b1] "v"(b1)
#define RTXN_UOUT(U, I) [r##U] "=&v"(r[I])
#define RTXN_UIN(U, I) [x##U] "v"(x[I]), [y##U] "v"(y[I])
template <bool ZERO, int NUP, bool SWAP = false>
__device__ __forceinline__ void pair32(floatx16& acc0, floatx16& acc1, const half8& a, const half8& b0, const half8& b1,
                                       int (&r)[NUP > 0 ? NUP : 1], const float (&x)[NUP > 0 ? NUP : 1], const float (&y)[NUP > 0 ? NUP : 1]) {
  static_assert(NUP == 0 || NUP == 2 || NUP == 4, "unit pattern not written");
  if constexpr (SWAP) {
    if constexpr (ZERO) {
      RTXN_PAIR32_CASES(RTXN_M32T("c0", "b0", "0"), RTXN_M32T("c1", "b1", "0"), [c0] "=&v"(acc0), [c1] "=&v"(acc1))
    } else {
      RTXN_PAIR32_CASES(RTXN_M32T("c0", "b0", "%[c0]"), RTXN_M32T("c1", "b1", "%[c1]"), [c0] "+v"(acc0), [c1] "+v"(acc1))
    }
  } else if constexpr (ZERO) {
    RTXN_PAIR32_CASES(RTXN_M32("c0", "b0", "0"), RTXN_M32("c1", "b1", "0"), [c0] "=&v"(acc0), [c1] "=&v"(acc1))
  } else {
    RTXN_PAIR32_CASES(RTXN_M32("c0", "b0", "%[c0]"), RTXN_M32("c1", "b1", "%[c1]"), [c0] "+v"(acc0), [c1] "+v"(acc1))
  }
}

// RT row tiles of 32 rows (RT even: the LAST one is left unconverted in acc[1] for the caller), KS k-steps of 16, two
// 32-sample column tiles.  A fragments come through a register ring of RTXN_PIPE slots filled with inline-asm ds_read_b128
// that many k-steps ahead, with a counted s_waitcnt lgkmcnt(N) in front of each consumer (nothing else in this function may
// touch LDS or SMEM); accumulators are double-buffered by row-tile parity: acc[rt & 1] collects row tile rt while the
// finished tile rt-1 in acc[~rt & 1] is converted, U pack units per k-step, inside that k-step's asm statement.
template <int RT, int KS, int NB, int I, bool SWAP = false>
struct PipeStep {
  static constexpr int CT = 2, D = RTXN_PIPE, N = RT * KS;
  static constexpr int U = (8 * CT + KS - 1) / KS;                 // units per k-step, row tiles 1..: 2 (KS = 8) or 4 (KS = 4)
  __device__ static __forceinline__ void run(unsigned addr, half8 (&bf)[NB][CT], half8 (&nbf)[NB][CT], half8 (&ring)[D],
                                             floatx16 (&acc)[2][CT]) {
    constexpr int rt = I / KS, kk = I % KS, cur = rt & 1;
    constexpr int outstanding = (N - 1 - I) < (D - 1) ? (N - 1 - I) : (D - 1);   // reads issued after fragment I
    lds_wait_insn<outstanding>();
    constexpr int p0 = rt > 0 ? (kk * U < 8 * CT ? kk * U : 8 * CT) : 0, p1 = rt > 0 ? ((kk + 1) * U < 8 * CT ? (kk + 1) * U : 8 * CT) : 0;
    constexpr int NUP = p1 - p0;
    int r[NUP > 0 ? NUP : 1];
    float x[NUP > 0 ? NUP : 1], y[NUP > 0 ? NUP : 1];
#pragma unroll
    for (int i = 0; i < NUP; ++i) {      // pack unit P: registers 8s + 2e, 8s + 2e + 1 of column tile P / 8 (s = (P % 8) / 4, e = P % 4)
      const int P = p0 + i, ct = P / 8, q = P % 8;
      x[i] = acc[cur ^ 1][ct][8 * (q / 4) + 2 * (q % 4)];
      y[i] = acc[cur ^ 1][ct][8 * (q / 4) + 2 * (q % 4) + 1];
    }
    pair32<kk == 0, NUP, SWAP>(acc[cur][0], acc[cur][1], ring[I % D], bf[kk][0], bf[kk][1], r, x, y);
#pragma unroll
    for (int i = 0; i < NUP; ++i) {      // -> dword e of B fragment nbf[2 (rt - 1) + s][ct]
      const int P = p0 + i, ct = P / 8, q = P % 8;
      int4v t = __builtin_bit_cast(int4v, nbf[2 * (rt - 1) + q / 4][ct]);
      t[q % 4] = r[i];
      nbf[2 * (rt - 1) + q / 4][ct] = __builtin_bit_cast(half8, t);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I + D < N) lds_read_frag<(I + D) * 1024>(ring[I % D], addr);
    if constexpr (I + 1 < N) PipeStep<RT, KS, NB, I + 1, SWAP>::run(addr, bf, nbf, ring, acc);
  }
};

// SWAP: every MFMA with its operands exchanged -- tile (rt, ct) comes out transposed: lane (col, h) holds FEATURE 32 rt + col of
// the 16 samples 32 ct + (e & 3) + 8 (e >> 2) + 4 h, and the unchanged pack units leave in nbf[2 rt + s][ct] that feature's
// values (ReLU, fp16) at the eight samples 32 ct + 16 s + {4 h .. 4 h + 3, 8 + 4 h .. 8 + 4 h + 3}: an A operand whose k runs
// over SAMPLES (the lean weight-gradient kernel's output layer).
template <int RT, int KS, int NB, bool SWAP = false>
__device__ __forceinline__ void pipe_layer(const uint8_t* lds_buf, half8 (&bf)[NB][2], half8 (&nbf)[NB][2], floatx16 (&acc)[2][2], int lane) {
  constexpr int D = RTXN_PIPE, N = RT * KS;
  static_assert(D >= 1 && D <= 4, "ring depth");
  static_assert(RT % 2 == 0 && RT >= 2, "the pending row tile must land in acc[1]");
  static_assert(KS == 4 || KS == 8, "unit patterns written for 2 or 4 units per k-step");
  static_assert(N * 1024 <= 65535 + 1024, "fragment offsets must fit the 16-bit ds offset");
  half8 ring[D];
  const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)lds_buf + lane * 16;
  lds_read_frag<0>(ring[0], addr);
  if constexpr (D > 1 && N > 1) lds_read_frag<1024>(ring[1 % D], addr);
  if constexpr (D > 2 && N > 2) lds_read_frag<2048>(ring[2 % D], addr);
  if constexpr (D > 3 && N > 3) lds_read_frag<3072>(ring[3 % D], addr);
  PipeStep<RT, KS, NB, 0, SWAP>::run(addr, bf, nbf, ring, acc);
  // the caller converts the pending tile in acc[1] with ordinary code: see mfma_results_settle
  mfma_results_settle(acc[1]);
}

// ---------------------------------------------------------------------------------------------------------------
// The same pipeline on v_mfma_f32_16x16x32_f16 (MI355X_MICROARCH.md, DVFS give-back item 7: in power-limited loops the
// chip holds a higher clock on this shape at equal cycles per FLOP).
//   A (weights):  lane (r = l & 15, g = l >> 4) holds W[16 rt + r][k = 8 g + j], j = 0..7 of a 32-wide k-step;
//   B (samples):  lane (c = l & 15, g)          holds act[k = 8 g + j][sample c];
//   D:            lane (c, g) holds rows 4 g + r (r = 0..3) of column c.
// Two consecutive 16-row tiles 2s, 2s+1 therefore leave in lane (c, g) the features 32 s + 4 g + r and 32 s + 16 + 4 g + r:
// eight values of ONE sample = the B operand of k-step s of the next layer under the k order perm_feature16, which is
// baked into the weight packing (pack_kernel mode 3).  Activations stay in registers exactly as in the 32x32x16 chain.
// A wave owns 64 samples as FOUR 16-column tiles (CT = 4), so a 1-KiB A fragment still feeds 64 kFLOP (4 MFMAs x 16
// cycles = the 2 x 32 cycles of the other shape) and the LDS traffic per FLOP is unchanged.
typedef float floatx4 __attribute__((ext_vector_type(4)));
#ifndef RTXN_PRIO_BANDS
#define RTXN_PRIO_BANDS 1   // 0: no s_setprio; 1: 3, 2, 1, 0 over a layer's quarters (default); 2: 1, 0 over its halves; 3: rising (the control)
#endif
#ifndef RTXN_PRIO_BANDS256
#define RTXN_PRIO_BANDS256 1   // the same over the four row tiles of a chunk of the 256-wide kernel (config5 13.69 -> 13.45 ms); the training
#endif                         // forward's 32x32x16 pipeline measured no change with it and carries none
#ifndef RTXN_PIPE16
#define RTXN_PIPE16 2   // A-fragment ring depth of the 16x16x32 pipeline: a step is 4 MFMAs = 64 cycles, so two steps ahead covers
#endif                  // the LDS latency, and the third slot's 4 VGPRs are what keeps the 128-wide segment variants from spilling

__host__ __device__ __forceinline__ int perm_feature16(int s, int g, int j) { return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3); }

// unit P (0 .. 2*CT-1) of finished 16-row tile RTI: accumulator registers 2e, 2e+1 of column tile P/2 -> dword
// 2*(RTI & 1) + e of the B fragment dst[RTI >> 1][P/2]
template <int NB, int CT, int RTI, int P>
__device__ __forceinline__ void convert_unit16(const floatx4 (&acc)[CT], half8 (&dst)[NB][CT]) {
  constexpr int ct = P / 2, e = P % 2;
  int4v t = __builtin_bit_cast(int4v, dst[RTI >> 1][ct]);
  int r;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0" : "=v"(r) : "v"(acc[ct][2 * e]), "v"(acc[ct][2 * e + 1]));
  t[2 * (RTI & 1) + e] = r;
  dst[RTI >> 1][ct] = __builtin_bit_cast(half8, t);
}
template <int NB, int CT, int RTI, int P0, int P1>
__device__ __forceinline__ void convert_units16(const floatx4 (&acc)[CT], half8 (&dst)[NB][CT]) {
  if constexpr (P0 < P1) {
    convert_unit16<NB, CT, RTI, P0>(acc, dst);
    convert_units16<NB, CT, RTI, P0 + 1, P1>(acc, dst);
  }
}
template <int NB, int CT, int RTI, int U, int KK>
__device__ __forceinline__ void convert_slice16(const floatx4 (&acc)[CT], half8 (&dst)[NB][CT]) {
  constexpr int p0 = KK * U < 2 * CT ? KK * U : 2 * CT, p1 = (KK + 1) * U < 2 * CT ? (KK + 1) * U : 2 * CT;
  convert_units16<NB, CT, RTI, p0, p1>(acc, dst);
}

// One k-step of the 16x16x32 pipeline with NOTHING left to the compiler: the CT MFMAs, accumulating
// in place, and the ReLU/convert units [P0, P1) of the finished tile RTI (accumulators `fin`) as asm, the units interleaved
// with the MFMAs (M cvt M max ...).  Why: builtin MFMAs that hipcc may move, merge or re-allocate around asm units it cannot
// see into produced wrong, timing-dependent values three times in this kernel family; as one asm statement per
// MFMA pair the step is what the source says.  Speed: the same as the builtin form -- tools/probe/mfma_cadence.hip: a VALU
// instruction next to the MFMAs of its own wave costs ~4 cycles wherever it stands (units behind the first MFMA 81 cycles per
// step, interleaved 80, behind the last 98); it is the partner wave of the SIMD that hides it.
// ZERO: first k-step of a row tile, C = 0.  Hazards, by construction as before: a unit reads a tile whose last MFMA is at
// least CT MFMAs behind it; a chain accumulator meets its next MFMA CT - 1 MFMAs later; nothing here reads `dst`.
// Two MFMAs and the NUP units behind them as ONE asm statement: between separate statements hipcc's hazard recogniser pads
// a cvt -> max pair it cannot see through with s_nop (4 issue cycles).  NUP = 1: M0 cvt M1 max; 2: M0 cvt max M1 cvt max.
#define RTXN_M16(ACC, B, C) "v_mfma_f32_16x16x32_f16 %[" ACC "], %[a], %[" B "], " C "\n\t"
#define RTXN_CVT16(U) "v_cvt_pk_f16_f32 %[r" U "], %[x" U "], %[y" U "]\n\t"
#define RTXN_MAX16(U) "v_pk_max_i16 %[r" U "], %[r" U "], 0\n\t"
#define RTXN_UNIT16(U) RTXN_CVT16(U) RTXN_MAX16(U)
#define RTXN_IN16 [a] "v"(a), [b0] "v"(b0), [b1] "v"(b1)
#define RTXN_PAIR16_CASES(M0, M1, ACC0, ACC1)                                                                                          \
  if constexpr (NUP == 0) asm volatile(M0 M1 : ACC0, ACC1 : RTXN_IN16);                                                                  \
  else if constexpr (NUP == 1) asm volatile(M0 RTXN_CVT16("0") M1 RTXN_MAX16("0") : ACC0, ACC1, RTXN_UOUT(0, 0) : RTXN_IN16, RTXN_UIN(0, 0)); \
  else if constexpr (NUP == 2)                                                                                                    \
    asm volatile(M0 RTXN_UNIT16("0") M1 RTXN_UNIT16("1") : ACC0, ACC1, RTXN_UOUT(0, 0), RTXN_UOUT(1, 1) : RTXN_IN16, RTXN_UIN(0, 0), RTXN_UIN(1, 1)); \
  else                                                                                                                            \
    asm volatile(M0 RTXN_UNIT16("0") RTXN_UNIT16("1") M1 RTXN_UNIT16("2") RTXN_UNIT16("3")                                         \
                 : ACC0, ACC1, RTXN_UOUT(0, 0), RTXN_UOUT(1, 1), RTXN_UOUT(2, 2), RTXN_UOUT(3, 3)                                         \
                 : RTXN_IN16, RTXN_UIN(0, 0), RTXN_UIN(1, 1), RTXN_UIN(2, 2), RTXN_UIN(3, 3));
template <bool ZERO, int NUP>
__device__ __forceinline__ void pair16(floatx4& acc0, floatx4& acc1, const half8& a, const half8& b0, const half8& b1,
                                       int (&r)[NUP > 0 ? NUP : 1], const float (&x)[NUP > 0 ? NUP : 1], const float (&y)[NUP > 0 ? NUP : 1]) {
  static_assert(NUP == 0 || NUP == 1 || NUP == 2 || NUP == 4, "unit pattern not written");
  if constexpr (ZERO) {
    RTXN_PAIR16_CASES(RTXN_M16("c0", "b0", "0"), RTXN_M16("c1", "b1", "0"), [c0] "=&v"(acc0), [c1] "=&v"(acc1))
  } else {
    RTXN_PAIR16_CASES(RTXN_M16("c0", "b0", "%[c0]"), RTXN_M16("c1", "b1", "%[c1]"), [c0] "+v"(acc0), [c1] "+v"(acc1))
  }
}
// units [P0, P0 + NU) of the finished tile RTI go behind the step's CT MFMAs, pair by pair
template <int NB, int CT, int RTI, int P0, int NU, bool ZERO, int PAIR>
struct PairRun16 {
  static constexpr int NPAIR = CT / 2;
  static constexpr int u0 = (NU * PAIR + NPAIR - 1) / NPAIR, u1 = (NU * (PAIR + 1) + NPAIR - 1) / NPAIR, NUP = u1 - u0;   // earlier pairs take the odd one
  __device__ static __forceinline__ void run(const half8& a, const half8 (&b)[CT], floatx4 (&acc)[CT], const floatx4 (&fin)[CT],
                                             half8 (&dst)[NB][CT]) {
    int r[NUP > 0 ? NUP : 1];
    float x[NUP > 0 ? NUP : 1], y[NUP > 0 ? NUP : 1];
#pragma unroll
    for (int i = 0; i < NUP; ++i) {
      const int P = P0 + u0 + i;
      x[i] = fin[P / 2][2 * (P % 2)];
      y[i] = fin[P / 2][2 * (P % 2) + 1];
    }
    pair16<ZERO, NUP>(acc[2 * PAIR], acc[2 * PAIR + 1], a, b[2 * PAIR], b[2 * PAIR + 1], r, x, y);
#pragma unroll
    for (int i = 0; i < NUP; ++i) {
      const int P = P0 + u0 + i, ct = P / 2, e = P % 2;
      int4v t = __builtin_bit_cast(int4v, dst[RTI >> 1][ct]);
      t[2 * (RTI & 1) + e] = r[i];
      dst[RTI >> 1][ct] = __builtin_bit_cast(half8, t);
    }
    if constexpr (PAIR + 1 < NPAIR) PairRun16<NB, CT, RTI, P0, NU, ZERO, PAIR + 1>::run(a, b, acc, fin, dst);
  }
};
template <int NB, int CT, int RTI, int P0, int P1, bool ZERO>
__device__ __forceinline__ void mfma_convert_step16(const half8& a, const half8 (&b)[CT], floatx4 (&acc)[CT], const floatx4 (&fin)[CT],
                                                    half8 (&dst)[NB][CT]) {
  static_assert(CT % 2 == 0, "MFMAs go in pairs");
  PairRun16<NB, CT, RTI, P0, (P1 > P0 ? P1 - P0 : 0), ZERO, 0>::run(a, b, acc, fin, dst);
}
template <int CT, int U, int KK>
struct UnitRange16 {
  static constexpr int p0 = KK * U < 2 * CT ? KK * U : 2 * CT, p1 = (KK + 1) * U < 2 * CT ? (KK + 1) * U : 2 * CT;
};

// RT 16-row tiles (RT == 0: the output layer's single tile, left raw in acc[0]); KS 32-wide k-steps.
// PEND: acc[1] holds the previous layer's last row tile (an odd tile: dwords 2, 3 of bf[KS-1]).
template <int RT, int KS, int NB, int CT, bool PEND, int I>
struct PipeStep16 {
  static constexpr int D = RTXN_PIPE16, N = (RT ? RT : 1) * KS;
  static constexpr int U = (2 * CT + KS - 1) / KS;                 // units per k-step, row tiles 1..
  static constexpr int WIN = KS - 1 > 1 ? KS - 1 : 1;              // k-steps of row tile 0 the pending tile is spread over
  static constexpr int UP = (2 * CT + WIN - 1) / WIN;
  static constexpr int WAVES = RTXN_NW;
  static constexpr int CHUNKS = N < 32 / WAVES ? N : 32 / WAVES;
  __device__ static __forceinline__ void run(unsigned addr, half8 (&bf)[NB][CT], half8 (&nbf)[NB][CT], half8 (&ring)[D],
                                             floatx4 (&acc)[2][CT], const StageJob& sj, int wave_u, int lane) {
    constexpr int rt = I / KS, kk = I % KS, cur = rt & 1;
    constexpr int outstanding = (N - 1 - I) < (D - 1) ? (N - 1 - I) : (D - 1);
#if RTXN_PRIO_BANDS
    // Issue priority falls with progress through the layer (3, 2, 1, 0 over its four quarters): of the two waves that share a
    // SIMD the one that is behind wins the arbitration, so neither runs the end of the stage alone (a lone wave issues its
    // MFMAs at 75 % of the rate two interleaved waves reach -- the older wave group used to finish a stage at 3,050 cycles and
    // leave the other until 4,700; with the bands 4,000 and 4,500, tile 44,400 -> 42,800 cycles.  Giving the trailing group one level more at
    // equal progress, or eighths against quarters, measured slightly worse than the symmetric form).
#if RTXN_PRIO_BANDS == 1
    if constexpr (kk == 0 && RT >= 4 && rt % (RT / 4) == 0) asm volatile("s_setprio %0" ::"n"(3 - rt / (RT / 4)));
#elif RTXN_PRIO_BANDS == 2
    if constexpr (kk == 0 && RT >= 2 && rt % (RT / 2) == 0) asm volatile("s_setprio %0" ::"n"(1 - rt / (RT / 2)));
#elif RTXN_PRIO_BANDS == 3
    if constexpr (kk == 0 && RT >= 4 && rt % (RT / 4) == 0) asm volatile("s_setprio %0" ::"n"(rt / (RT / 4)));
#endif
#endif
    lds_wait_insn<outstanding>();
    if constexpr (rt > 0) {
      using R = UnitRange16<CT, U, kk>;
      mfma_convert_step16<NB, CT, rt - 1, R::p0, R::p1, kk == 0>(ring[I % D], bf[kk], acc[cur], acc[cur ^ 1], nbf);
    } else if constexpr (PEND && kk < WIN) {
      using R = UnitRange16<CT, UP, kk>;
      mfma_convert_step16<NB, CT, 2 * KS - 1, R::p0, R::p1, kk == 0>(ring[I % D], bf[kk], acc[cur], acc[1], bf);
    } else {
      mfma_convert_step16<NB, CT, 0, 0, 0, kk == 0>(ring[I % D], bf[kk], acc[cur], acc[cur], nbf);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I + D < N) lds_read_frag<(I + D) * 1024>(ring[I % D], addr);
    if constexpr (I < CHUNKS) {
      stage_chunk<I, WAVES>(sj, wave_u, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (I + 1 < N) PipeStep16<RT, KS, NB, CT, PEND, I + 1>::run(addr, bf, nbf, ring, acc, sj, wave_u, lane);
  }
};

template <int RT, int KS, int NB, int CT, bool PEND>
__device__ __forceinline__ void pipe_layer16(const uint8_t* lds_buf, const StageJob& sj, half8 (&bf)[NB][CT], half8 (&nbf)[NB][CT],
                                             floatx4 (&acc)[2][CT], int wave_u, int lane) {
  constexpr int D = RTXN_PIPE16, N = (RT ? RT : 1) * KS;
  static_assert(D >= 1 && D <= 4, "ring depth");
  static_assert(RT % 2 == 0, "the pending row tile must land in acc[1] and be the odd tile of its pair");
  static_assert(CT >= 2, "a unit must never read the accumulator of the MFMA issued just before it");
  static_assert(!PEND || KS >= 2, "the pending tile's fragment is first read at k-step KS-1; it is written during k-steps < KS-1");
  static_assert(NB >= KS, "fragment set too small");
  static_assert(N * 1024 <= 65535 + 1024, "fragment offsets must fit the 16-bit ds offset");
  half8 ring[D];
  const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)lds_buf + lane * 16;
  lds_read_frag<0>(ring[0], addr);
  if constexpr (D > 1 && N > 1) lds_read_frag<1024>(ring[1 % D], addr);
  if constexpr (D > 2 && N > 2) lds_read_frag<2048>(ring[2 % D], addr);
  if constexpr (D > 3 && N > 3) lds_read_frag<3072>(ring[3 % D], addr);
  PipeStep16<RT, KS, NB, CT, PEND, 0>::run(addr, bf, nbf, ring, acc, sj, wave_u, lane);
  // RT == 0: the caller reads acc[0] with ordinary code right behind this: see mfma_results_settle
  if constexpr (RT == 0) mfma_results_settle(acc[0]);
}

// The same 16x16x32 pipeline over one CHUNK of a streamed layer (the 256-wide kernel: a layer is 128 KiB of fragments and goes
// through LDS in four 32-KiB chunks of four 16-row tiles): NR row tiles starting at the layer's row tile RT0 (even), KS
// 32-wide k-steps, 8 waves.  acc[1] arrives holding the row tile before RT0 -- this layer's tile RT0-1 (-> out) or, for a
// layer's first chunk, the previous layer's last tile 2*NB-1 (-> in[NB-1], dwords 2 and 3, first read at k-step NB-1) -- if
// PEND; the chunk's own last tile (odd) is left pending in acc[1] in turn.
template <int KS, int NB, int CT, int NR, int RT0, bool PEND, int I>
struct PipeStep16c {
  static constexpr int D = RTXN_PIPE16, N = NR * KS, WAVES = 8;
  // A finished tile's units run in k-steps 1 .. KS-1 of the NEXT tile, never in its k-step 0: with two column tiles a k-step
  // is only two 8-pass MFMAs, and a unit hoisted above them by the compiler (asm volatile orders it against other asm only)
  // would read an accumulator one MFMA after its last write -- partial sums, timing-dependent.  One k-step later the
  // writer is at least three MFMAs behind whatever the scheduler does inside the step.
  static constexpr int SPAN = KS - 1;
  static constexpr int U = (2 * CT + SPAN - 1) / SPAN;
  static constexpr int WIN = KS - 1 > 1 ? KS - 1 : 1;
  static constexpr int UP = (2 * CT + WIN - 1) / WIN;
  static constexpr int CHUNKS = 4;                                  // 32 KiB / (8 waves x 1 KiB)
  __device__ static __forceinline__ void run(unsigned addr, half8 (&in)[NB][CT], half8 (&out)[NB][CT], half8 (&ring)[D],
                                             floatx4 (&acc)[2][CT], const StageJob& sj, int wave_u, int lane) {
    constexpr int r = I / KS, kk = I % KS, cur = r & 1;
    constexpr int outstanding = (N - 1 - I) < (D - 1) ? (N - 1 - I) : (D - 1);
#if RTXN_PRIO_BANDS256
    if constexpr (kk == 0 && NR >= 4 && r % (NR / 4) == 0) asm volatile("s_setprio %0" ::"n"(3 - r / (NR / 4)));     // see PipeStep16
#endif
    lds_wait_insn<outstanding>();
    if constexpr (r > 0 && kk >= 1) {
      using R = UnitRange16<CT, U, kk - 1>;
      mfma_convert_step16<NB, CT, RT0 + r - 1, R::p0, R::p1, false>(ring[I % D], in[kk], acc[cur], acc[cur ^ 1], out);
    } else if constexpr (r == 0 && PEND && kk < WIN) {
      using R = UnitRange16<CT, UP, kk>;
      if constexpr (RT0 > 0) mfma_convert_step16<NB, CT, RT0 - 1, R::p0, R::p1, kk == 0>(ring[I % D], in[kk], acc[cur], acc[1], out);
      else mfma_convert_step16<NB, CT, 2 * NB - 1, R::p0, R::p1, kk == 0>(ring[I % D], in[kk], acc[cur], acc[1], in);
    } else {
      mfma_convert_step16<NB, CT, 0, 0, 0, kk == 0>(ring[I % D], in[kk], acc[cur], acc[cur], out);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I + D < N) lds_read_frag<(I + D) * 1024>(ring[I % D], addr);
    if constexpr (I < CHUNKS) {
      stage_chunk<I, WAVES>(sj, wave_u, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (I + 1 < N) PipeStep16c<KS, NB, CT, NR, RT0, PEND, I + 1>::run(addr, in, out, ring, acc, sj, wave_u, lane);
  }
};

template <int KS, int NB, int CT, int NR, int RT0, bool PEND>
__device__ __forceinline__ void pipe_chunk16(const uint8_t* lds_buf, const StageJob& sj, half8 (&in)[NB][CT], half8 (&out)[NB][CT],
                                             floatx4 (&acc)[2][CT], int wave_u, int lane) {
  constexpr int D = RTXN_PIPE16, N = NR * KS;
  static_assert(RT0 % 2 == 0 && NR % 2 == 0, "chunks start on an even row tile and leave an odd one pending");
  static_assert(CT >= 2, "a unit must never read the accumulator of the MFMA issued just before it");
  static_assert(!PEND || RT0 > 0 || KS == NB, "a tile pending across a layer boundary lands in in[NB-1]");
  static_assert(!PEND || KS >= 2, "the pending tile is converted during k-steps < KS-1");
  static_assert(N * 1024 <= 65535 + 1024, "fragment offsets must fit the 16-bit ds offset");
  half8 ring[D];
  const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)lds_buf + lane * 16;
  lds_read_frag<0>(ring[0], addr);
  if constexpr (D > 1 && N > 1) lds_read_frag<1024>(ring[1 % D], addr);
  if constexpr (D > 2 && N > 2) lds_read_frag<2048>(ring[2 % D], addr);
  if constexpr (D > 3 && N > 3) lds_read_frag<3072>(ring[3 % D], addr);
  PipeStep16c<KS, NB, CT, NR, RT0, PEND, 0>::run(addr, in, out, ring, acc, sj, wave_u, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// Compiler-scheduled layers (training kernels)
// One layer: out rows [32*rt, 32*rt+32) for rt < RT, K = 16*KS, for the wave's two column
// tiles.  Row-tile-outer: an accumulator is live for one row tile only, and its ReLU/convert
// (VALU) overlaps the next row tile's MFMAs.  A fragments: chunk (rt, kk) at ((rt*KS+kk)*64+lane)*16.
template <int RT, int KS, int NB, int CT>
__device__ __forceinline__ void layer_mma(const uint8_t* lds_buf, const half8 (&bf)[NB][CT], half8 (&nbf)[NB][CT],
                                          int lane) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    floatx16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ct][e] = 0.0f;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const half8 a = *reinterpret_cast<const half8*>(lds_buf + ((rt * KS + kk) * 64 + lane) * 16);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bf[kk][ct], acc[ct], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) nbf[2 * rt + s][ct] = relu_pack(acc[ct], s);
  }
}
// Output layer: 32 rows (16 real), raw accumulators returned.
template <int KS, int NB, int CT>
__device__ __forceinline__ void out_mma(const uint8_t* lds_buf, const half8 (&bf)[NB][CT], floatx16 (&acc)[CT], int lane) {
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[ct][e] = 0.0f;
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const half8 a = *reinterpret_cast<const half8*>(lds_buf + (kk * 64 + lane) * 16);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bf[kk][ct], acc[ct], 0, 0, 0);
  }
}


}  // namespace rtxn
