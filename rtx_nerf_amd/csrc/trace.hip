// Ray generation + ray/grid traversal.  Replaces the OptiX pipeline of the
// reference: __raygen__/__intersection__/__closesthit__/__miss__ray_march
// (rtx/src/optixPrograms.cu:43-248), the AABB grid + GAS build
// (main.cu:154-174, rtx/src/rtxFunctions.cpp:293-351) and optixLaunch
// (main.cu:506-508).
//
// OptiX finds "the closest AABB" with a full BVH traversal per crossed cell.
// The primitives are the cells of a uniform grid, so the same ordered list of
// (entry, exit) points falls out of an analytic walk, one thread per ray:
//   RTXN_TRACE_COMPAT  the reference's arithmetic: slab test and exit planes
//                      from the re-launched origin, next origin = exit point;
//   RTXN_TRACE_DDA     global-t 3D-DDA whose plane crossings are pure
//                      functions of the integer cell index, so a two-level
//                      walk over a coarse occupancy mip (staged in LDS) skips
//                      empty 4^3 blocks and still reproduces the flat walk
//                      bit for bit.
// Both follow oracle/rtxn_oracle.c operation for operation (explicit fmaf,
// -ffp-contract=off, IEEE division) so segment end points match bit-exactly;
// only atan2f (theta, phi) differs between libm and the device library.
//
// HBM-bound on its outputs: 24 B/ray (origin 12 + direction 8 + count 4) and
// 32 B/segment in packed mode (start 12 + end 12 + seg_view 8; t_start/t_end,
// which nothing downstream reads, optional +8).  The occupancy bitfield
// (R^3/8 bytes: 256 KiB at 128^3, 2 MiB at 256^3) is L2-resident.
#include "common.h"

namespace {

struct Sink {
  float* start;
  float* end;
  float* t0;
  float* t1;
  int* seg_ray;
  float* seg_view;
  uint8_t* seg_first;
  float view[2];
  long base;
  long limit;  // first slot this ray may not write
  int ray;
  int n;
  int prior;   // segments of this ray emitted by earlier sub-ray lanes (seg_first)
  __device__ __forceinline__ void emit(const float (&p0)[3], const float (&p1)[3], float a, float b) {
    const long k = base + n;
    if (k < limit) {
      if (start) { start[3 * k] = p0[0]; start[3 * k + 1] = p0[1]; start[3 * k + 2] = p0[2]; }
      if (end) { end[3 * k] = p1[0]; end[3 * k + 1] = p1[1]; end[3 * k + 2] = p1[2]; }
      if (t0) t0[k] = a;
      if (t1) t1[k] = b;
      if (seg_ray) seg_ray[k] = ray;
      if (seg_view) { seg_view[2 * k] = view[0]; seg_view[2 * k + 1] = view[1]; }
      if (seg_first) seg_first[k] = prior + n == 0;
    }
    ++n;
  }
};

__device__ __forceinline__ float cell_lo(int i, float L) { return -1.0f + (float)i * L; }
__device__ __forceinline__ float cell_hi(int i, float L) { return -1.0f + (float)i * L + L; }

__device__ __forceinline__ bool occ_test(const uint32_t* __restrict__ occ, int R, int x, int y, int z) {
  if (!occ) return true;
  const uint32_t idx = ((uint32_t)x * (uint32_t)R + (uint32_t)y) * (uint32_t)R + (uint32_t)z;
  return (occ[idx >> 5] >> (idx & 31)) & 1u;
}

// a2: optixPrograms.cu:43-82
__device__ __forceinline__ void make_ray(const float* __restrict__ la, float focal_length, float aspect_ratio,
                                         unsigned width, unsigned height, unsigned px, unsigned py, float (&o)[3],
                                         float (&d)[3], float (&v)[2]) {
  const float u = (float)((2 * (px + 0.5) / width - 1) * aspect_ratio);
  const float vv = (float)(2 * (py + 0.5) / height - 1);
  const float nf0 = la[2] * -1.0f, nf1 = la[6] * -1.0f, nf2 = la[10] * -1.0f;
  float xd = fmaf(nf0, focal_length, fmaf(la[0], u, la[1] * vv));
  float yd = fmaf(nf1, focal_length, fmaf(la[4], u, la[5] * vv));
  float zd = fmaf(nf2, focal_length, fmaf(la[8], u, la[9] * vv));
  const float norm = sqrtf(fmaf(zd, zd, fmaf(xd, xd, yd * yd)));
  xd /= norm;
  yd /= norm;
  zd /= norm;
  v[0] = atan2f(sqrtf(fmaf(xd, xd, yd * yd)), zd);
  v[1] = atan2f(yd, xd);
  d[0] = xd; d[1] = yd; d[2] = zd;
  o[0] = la[3] / 10;
  o[1] = la[7] / 10;
  o[2] = la[11] / 10;
}

__device__ __forceinline__ bool grid_entry(const float (&o)[3], const float (&d)[3], int R, float L, int (&cell)[3],
                                           float& t_enter) {
  bool inside = true;
#pragma unroll
  for (int a = 0; a < 3; ++a)
    if (!(o[a] >= -1.0f && o[a] <= 1.0f)) inside = false;
  float t0 = 0.0f;
  int enter_axis = -1;
  if (!inside) {
    float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (d[a] == 0.0f) {  // parallel to this slab of the grid: inside it or a miss
        if (!(o[a] >= -1.0f && o[a] <= 1.0f)) return false;
        continue;
      }
      const float t1 = (-1.0f - o[a]) / d[a];
      const float t2 = (1.0f - o[a]) / d[a];
      const float tn = fminf(t1, t2), tf = fmaxf(t1, t2);
      if (tn > tmin) { tmin = tn; enter_axis = a; }
      tmax = fminf(tmax, tf);
    }
    if (!(tmax > tmin) || tmin < 0.0f) return false;
    t0 = tmin;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float p = fmaf(t0, d[a], o[a]);
    const int c = (int)floorf((p + 1.0f) / L);
    cell[a] = min(max(c, 0), R - 1);
    if (a == enter_axis) cell[a] = d[a] > 0 ? 0 : R - 1;
  }
  t_enter = t0;
  return true;
}

// a3-a5 restated as a cell walk (see oracle march_compat for the derivation).
__device__ void march_compat(const float (&o0)[3], const float (&d)[3], int R, const uint32_t* __restrict__ occ,
                             Sink& s) {
  const float L = 2.0f / (float)R;
  int c[3];
  float tE;
  if (!grid_entry(o0, d, R, L, c, tE)) return;
  float o[3] = {o0[0], o0[1], o0[2]};
  for (int guard = 0; guard < 3 * R + 8; ++guard) {
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = cell_lo(c[a], L); hi[a] = cell_hi(c[a], L); }
    // __intersection__ray_march: optixPrograms.cu:132-169
    float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (d[a] == 0.0f) {
        // parallel to this slab: the reference divides by zero (undefined result); defined here as
        // "no constraint if the origin is inside the slab, no hit otherwise"
        if (!(o[a] >= lo[a] && o[a] <= hi[a])) tmin = INFINITY;
        continue;
      }
      const float t1 = (lo[a] - o[a]) / d[a];
      const float t2 = (hi[a] - o[a]) / d[a];
      tmin = fmaxf(tmin, fminf(t1, t2));
      tmax = fminf(tmax, fmaxf(t1, t2));
    }
    const bool rep = tmax > tmin;
    if (rep && tmin < 0 && (double)tmax > 1e-6) tmin = 0;
    const float t_hit = tmin;
    // __closesthit__ray_march: optixPrograms.cu:180-248
    float te[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float plane = d[a] < 0 ? lo[a] : hi[a];
      te[a] = d[a] == 0.0f ? INFINITY : (plane - o[a]) / d[a];  // a parallel axis is never the exit axis
    }
    const float t_e = fminf(fminf(te[0], te[1]), te[2]);
    float p0[3], p1[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { p0[a] = fmaf(t_hit, d[a], o[a]); p1[a] = fmaf(t_e, d[a], o[a]); }
    if (rep && t_hit >= 0.0f && occ_test(occ, R, c[0], c[1], c[2])) s.emit(p0, p1, t_hit, t_e);
    bool out = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (te[a] == t_e) {
        c[a] += d[a] < 0 ? -1 : 1;
        if (c[a] < 0 || c[a] >= R) out = true;
      }
    }
    if (out) break;
    o[0] = p1[0]; o[1] = p1[1]; o[2] = p1[2];
  }
}

__device__ __forceinline__ float plane_t(int i, float L, float o, float inv) { return (cell_lo(i, L) - o) * inv; }

// Global-t DDA.  `coarse` (LDS or global, may be NULL): bit ((X*Rc+Y)*Rc+Z) of the
// (R/4)^3 mip.  An empty coarse block is crossed in one step: every axis advances
// while its exit plane's t is <= the block's exit t, which is exactly the state
// the flat walk reaches (plane_t depends only on the cell index).
// The state the flat walk is in once every plane with plane_t <= tc has been crossed, starting from cell c: each axis
// is estimated from the position and fixed up with the walk's own exact comparisons (exit plane of cell i is plane
// i + up; crossed iff <= tc).  Returns false if the walk has left the grid by then.
__device__ __forceinline__ bool land_at(int (&c)[3], float tc, const float (&o)[3], const float (&d)[3], const float (&inv)[3],
                                        const int (&step)[3], const int (&up)[3], int R, float L) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (d[a] != 0.0f) {
      const int c0 = c[a];
      int est = (int)floorf((fmaf(tc, d[a], o[a]) + 1.0f) / L);
      est = step[a] > 0 ? max(est, c0) : min(est, c0);
      est = min(max(est, 0), R - 1);
      while (est != c0 && plane_t(est - step[a] + up[a], L, o[a], inv[a]) > tc) est -= step[a];
      while (plane_t(est + up[a], L, o[a], inv[a]) <= tc) {
        est += step[a];
        if (est < 0 || est >= R) return false;
      }
      c[a] = est;
    }
  }
  return true;
}

// sub > 1: this lane is piece `piece` of `sub` of the ray.  The ray's parameter range inside the grid is cut at
// T_k = t_enter + (t_exit - t_enter) * k / sub; a cell belongs to the piece with T_piece < t_out <= T_piece+1 (first and
// last piece unbounded), so every cell is emitted by exactly one lane, in ray order across the lanes.  A piece starts in
// the cell the flat walk is in at T_piece (land_at) with the entry time the flat walk has there: t_in is always the t of
// the last plane crossed, which is an entry plane of the current cell, and both are pure functions of integer cell
// indices -- so the pieces reproduce the one-thread walk bit for bit.
__device__ void march_dda(const float (&o)[3], const float (&d)[3], int R, const uint32_t* __restrict__ occ,
                          const uint32_t* coarse, const uint32_t* super, const unsigned long long* __restrict__ bricks,
                          int piece, int sub, Sink& s) {
  const float L = 2.0f / (float)R;
  int c[3];
  float t_in;
  if (!grid_entry(o, d, R, L, c, t_in)) return;
  float inv[3];
  int step[3], up[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    inv[a] = 1.0f / d[a];
    step[a] = d[a] < 0 ? -1 : 1;
    up[a] = d[a] < 0 ? 0 : 1;
  }
  float t_stop = INFINITY;   // cells leaving after t_stop belong to the next piece
  if (sub > 1) {
    float t_exit = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a)
      if (d[a] != 0.0f) t_exit = fminf(t_exit, plane_t(up[a] ? R : 0, L, o[a], inv[a]));
    const float span = t_exit - t_in, t_enter = t_in;
    if (piece + 1 < sub) t_stop = fmaf(span, (float)(piece + 1) / (float)sub, t_enter);
    if (piece > 0) {
      const float t_from = fmaf(span, (float)piece / (float)sub, t_enter);   // == the previous piece's t_stop
      const int c_in[3] = {c[0], c[1], c[2]};
      if (!land_at(c, t_from, o, d, inv, step, up, R, L)) return;
#pragma unroll
      for (int a = 0; a < 3; ++a)
        if (d[a] != 0.0f && c[a] != c_in[a]) t_in = fmaxf(t_in, plane_t(c[a] + 1 - up[a], L, o[a], inv[a]));
    }
  }
  const int Rc = R >> 2;
  // brick cache: the 64 fine bits of the occupied 4^3 block the walk is currently in (one 8-byte load per block
  // instead of one dependent 4-byte load per fine cell)
  uint32_t brick_id = 0xffffffffu;
  unsigned long long brick = 0;
  for (int guard = 0; guard < 3 * R + 8; ++guard) {
    bool out = false;
    if (coarse) {
      // empty-space skipping, coarsest level first: a 16^3 (super) or 4^3 (coarse) block without occupied cells is
      // crossed in one step -- every axis advances while its exit plane's t <= the block's exit t, which is exactly
      // the state the flat walk reaches (plane_t depends only on the integer cell index)
      int shift = 0;
      if (super) {
        const int Rs = R >> 4;
        const uint32_t si = ((uint32_t)(c[0] >> 4) * Rs + (uint32_t)(c[1] >> 4)) * Rs + (uint32_t)(c[2] >> 4);
        if (!((super[si >> 5] >> (si & 31)) & 1u)) shift = 4;
      }
      if (!shift) {
        const uint32_t ci = ((uint32_t)(c[0] >> 2) * Rc + (uint32_t)(c[1] >> 2)) * Rc + (uint32_t)(c[2] >> 2);
        const bool block_on = (coarse[ci >> 5] >> (ci & 31)) & 1u;
        if (block_on && bricks && ci != brick_id) {
          brick_id = ci;
          brick = bricks[ci];
        }
        if (!block_on) shift = 2;
      }
      if (shift) {
        float tc = INFINITY;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          if (d[a] != 0.0f) tc = fminf(tc, plane_t(((c[a] >> shift) + up[a]) << shift, L, o[a], inv[a]));
        }
        // land directly in the cell the flat walk would be in at tc
        if (!land_at(c, tc, o, d, inv, step, up, R, L)) break;
        if (tc > t_in) t_in = tc;
        if (t_in >= t_stop) break;   // everything from here on leaves after t_stop
        continue;
      }
    }
    float te[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) te[a] = d[a] == 0.0f ? INFINITY : plane_t(c[a] + up[a], L, o[a], inv[a]);
    const float t_out = fminf(fminf(te[0], te[1]), te[2]);
    if (t_out > t_stop) break;   // this cell and all later ones belong to the following pieces
    bool on;
    if (coarse && bricks) on = (brick >> (((c[0] & 3) << 4) | ((c[1] & 3) << 2) | (c[2] & 3))) & 1ull;
    else on = occ_test(occ, R, c[0], c[1], c[2]);
    if (t_out > t_in && on) {
      float p0[3], p1[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) { p0[a] = fmaf(t_in, d[a], o[a]); p1[a] = fmaf(t_out, d[a], o[a]); }
      s.emit(p0, p1, t_in, t_out);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (te[a] == t_out) {
        c[a] += step[a];
        if (c[a] < 0 || c[a] >= R) out = true;
      }
    }
    if (out) break;
    if (t_out > t_in) t_in = t_out;
  }
}

constexpr int kCoarseLdsWords = 8192;  // 32 KiB: coarse mip of up to 256^3 (64^3 bits)
constexpr int kSuperLdsWords = 512;    // 2 KiB: super mip (16^3-cell blocks) of up to 400^3

// LDS copy of the two coarse levels, sized by the launch to the grid at hand (128^3: 4 KiB + 64 B) rather than to the
// largest grid supported: a traversal block then fits on a CU beside a resident block of the MLP kernel (132 KiB of the
// 160), which is what lets the next frame's traversal run underneath this frame's MLP (render.py, render_async).
struct TraceLds { int coarse_words, super_words; };   // words staged in LDS (0: read that level from global memory)
__host__ __device__ inline TraceLds trace_lds(const rtxn_trace_params& p) {
  TraceLds t{0, 0};
  if (p.mode != RTXN_TRACE_DDA || !p.occupancy_coarse) return t;
  const int Rc = p.grid_res >> 2, Rs = p.grid_res >> 4;
  const int words = (Rc * Rc * Rc + 31) >> 5, swords = p.occupancy_super ? (Rs * Rs * Rs + 31) >> 5 : 0;
  if (words <= kCoarseLdsWords) t.coarse_words = words;
  if (swords > 0 && swords <= kSuperLdsWords) t.super_words = swords;
  return t;
}

template <int MODE>
__global__ __launch_bounds__(256) void trace_kernel(rtxn_trace_params p) {
  extern __shared__ uint32_t trace_smem[];   // [coarse_words | super_words]
  const uint32_t* coarse = nullptr;
  const uint32_t* super = nullptr;
  if (MODE == RTXN_TRACE_DDA && p.occupancy_coarse) {
    const TraceLds tl = trace_lds(p);
    const int Rs = p.grid_res >> 4;
    const int words = tl.coarse_words;
    const int swords = p.occupancy_super ? (Rs * Rs * Rs + 31) >> 5 : 0;
    const bool c_lds = tl.coarse_words > 0, s_lds = tl.super_words > 0;
    uint32_t* coarse_lds = trace_smem;
    uint32_t* super_lds = trace_smem + tl.coarse_words;
    if (c_lds)
      for (int i = threadIdx.x; i < words; i += blockDim.x) coarse_lds[i] = p.occupancy_coarse[i];
    if (s_lds)
      for (int i = threadIdx.x; i < swords; i += blockDim.x) super_lds[i] = p.occupancy_super[i];
    if (c_lds || s_lds) __syncthreads();
    coarse = c_lds ? coarse_lds : p.occupancy_coarse;
    if (swords) super = s_lds ? super_lds : p.occupancy_super;
  }
  // sub_rays = Q > 1: Q adjacent lanes share a ray (thread = ray * Q + piece); Q divides 64, so a ray never straddles waves
  const int Q = MODE == RTXN_TRACE_DDA && p.sub_rays > 1 ? p.sub_rays : 1;
  const unsigned tid = blockIdx.x * 256u + threadIdx.x;
  const unsigned r = tid / (unsigned)Q;
  const int piece = (int)(tid % (unsigned)Q);
  const bool live = r < p.ray_count;   // dead lanes stay for the wave-level sums below
  const unsigned rr = live ? r : 0;
  const unsigned gid = p.window_chunk ? p.ray_begin + (rr / p.window_chunk) * p.window_stride + (rr % p.window_chunk)
                                      : p.ray_begin + rr;
  float o[3], d[3], v[2];
  if (p.look_at) {
    make_ray(p.look_at, p.focal_length, p.aspect_ratio, p.width, p.height, gid % p.width, gid / p.width, o, d, v);
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a) { o[a] = p.rays_o[3 * (size_t)gid + a]; d[a] = p.rays_d[3 * (size_t)gid + a]; }
    v[0] = atan2f(sqrtf(fmaf(d[0], d[0], d[1] * d[1])), d[2]);
    v[1] = atan2f(d[1], d[0]);
  }
  if (live && piece == 0) {
    if (p.ray_origins) { p.ray_origins[3 * (size_t)r] = o[0]; p.ray_origins[3 * (size_t)r + 1] = o[1]; p.ray_origins[3 * (size_t)r + 2] = o[2]; }
    if (p.viewing_direction) { p.viewing_direction[2 * (size_t)r] = v[0]; p.viewing_direction[2 * (size_t)r + 1] = v[1]; }
  }
  const bool writing = p.start_points || p.end_points || p.t_start || p.t_end || p.seg_ray || p.seg_view || p.seg_first;
  // sub-ray write pass: this piece's segments follow those of the earlier pieces (their counts come from the counting pass)
  int prior = 0;
  if (Q > 1 && writing) {
    const int mine = live ? p.sub_hits[(size_t)r * Q + piece] : 0;
    int incl = mine;
    for (int dlt = 1; dlt < Q; dlt <<= 1) {
      const int t = __shfl_up(incl, dlt, 64);
      if (piece >= dlt) incl += t;
    }
    prior = incl - mine;
  }
  Sink s;
  s.start = p.start_points; s.end = p.end_points; s.t0 = p.t_start; s.t1 = p.t_end; s.seg_ray = p.seg_ray; s.seg_view = p.seg_view; s.seg_first = p.seg_first; s.view[0] = v[0]; s.view[1] = v[1];
  s.ray = (int)r; s.n = 0; s.prior = prior;
  if (p.indices) { s.base = (long)p.indices[rr] + prior; s.limit = p.segment_capacity > 0 ? p.segment_capacity : 0x7fffffffffffffffL; }
  else { s.base = (long)r * p.intersection_arr_size + prior; s.limit = ((long)r + 1) * p.intersection_arr_size; }
  if (!writing || !live) s.limit = 0;
  if (live) {
    if (MODE == RTXN_TRACE_COMPAT) march_compat(o, d, p.grid_res, p.occupancy, s);
    else march_dda(o, d, p.grid_res, p.occupancy, coarse, super, reinterpret_cast<const unsigned long long*>(p.occupancy_bricks), piece, Q, s);
  }
  const long room = s.limit - s.base;
  int hits = s.n, stored = (int)(room <= 0 ? 0 : (room < s.n ? room : s.n));
  if (Q > 1) {
    if (live && !writing) p.sub_hits[(size_t)r * Q + piece] = s.n;
    for (int dlt = 1; dlt < Q; dlt <<= 1) {   // sum over the ray's Q lanes (aligned group)
      hits += __shfl_xor(hits, dlt, 64);
      stored += __shfl_xor(stored, dlt, 64);
    }
  }
  if (live && piece == 0) {
    p.num_hits[r] = hits;
    if (p.num_stored) p.num_stored[r] = stored;
  }
}

// one thread per coarse WORD (32 coarse cells), deterministic, no pre-zeroing
__global__ void mip_kernel(const uint32_t* __restrict__ occ, int R, uint32_t* __restrict__ coarse) {
  const int Rc = R >> 2;
  const int ncell = Rc * Rc * Rc;
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w * 32 >= ncell) return;
  uint32_t word = 0;
  for (int b = 0; b < 32; ++b) {
    const int ci = w * 32 + b;
    if (ci >= ncell) break;
    const int Z = ci % Rc, Y = (ci / Rc) % Rc, X = ci / (Rc * Rc);
    uint32_t any = 0;
    for (int dx = 0; dx < 4; ++dx)
      for (int dy = 0; dy < 4; ++dy) {
        // the 4 z-cells of this row are 4 consecutive bits inside one word (R % 4 == 0)
        const uint32_t idx = ((uint32_t)(4 * X + dx) * R + (uint32_t)(4 * Y + dy)) * R + (uint32_t)(4 * Z);
        any |= (occ[idx >> 5] >> (idx & 31)) & 0xfu;
      }
    if (any) word |= 1u << b;
  }
  coarse[w] = word;
}

// one thread per 4^3 block: its 64 fine bits, bit ((x&3)<<4 | (y&3)<<2 | (z&3)), gathered from 16 four-bit z-runs
__global__ void brick_kernel(const uint32_t* __restrict__ occ, int R, unsigned long long* __restrict__ bricks) {
  const int Rc = R >> 2;
  const int ci = blockIdx.x * blockDim.x + threadIdx.x;
  if (ci >= Rc * Rc * Rc) return;
  const int Z = ci % Rc, Y = (ci / Rc) % Rc, X = ci / (Rc * Rc);
  unsigned long long m = 0;
  for (int dx = 0; dx < 4; ++dx)
    for (int dy = 0; dy < 4; ++dy) {
      const uint32_t idx = ((uint32_t)(4 * X + dx) * R + (uint32_t)(4 * Y + dy)) * R + (uint32_t)(4 * Z);
      const unsigned long long run = (occ[idx >> 5] >> (idx & 31)) & 0xfu;   // z = 4Z..4Z+3, bit z&3
      // run bit k is cell z = 4Z+k -> brick bit (dx<<4 | dy<<2 | k)
      m |= run << ((dx << 4) | (dy << 2));
    }
  bricks[ci] = m;
}

// density[R^3] (index (x*R+y)*R+z) -> occupancy bits: one wave ballot = two 32-bit words
__global__ __launch_bounds__(256) void occupancy_kernel(const float* __restrict__ density, float threshold, long n,
                                                        uint32_t* __restrict__ bits) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const bool on = i < n && density[i] > threshold;
  const unsigned long long m = __ballot(on);
  const int lane = threadIdx.x & 63;
  const long word = i >> 5;
  if ((lane & 31) == 0 && (i < n || (i & ~31L) < n)) bits[word] = (uint32_t)(m >> (lane & 32));
}

}  // namespace

extern "C" int rtxn_build_occupancy_bricks(const uint32_t* occupancy, int grid_res, uint64_t* bricks, rtxn_stream_t stream) {
  RTXN_REQUIRE(occupancy && bricks, "rtxn_build_occupancy_bricks: NULL buffer");
  RTXN_REQUIRE(grid_res >= 4 && grid_res % 4 == 0 && grid_res <= 1024,
               "rtxn_build_occupancy_bricks: grid_res = %d must be a multiple of 4 in [4,1024]", grid_res);
  RTXN_DEVICE_OR_FAIL();
  const int Rc = grid_res / 4;
  const int n = Rc * Rc * Rc;
  brick_kernel<<<(n + 255) / 256, 256, 0, rtxn::as_stream(stream)>>>(occupancy, grid_res,
                                                                      reinterpret_cast<unsigned long long*>(bricks));
  RTXN_LAUNCH_CHECK("brick_kernel");
  return RTXN_OK;
}

extern "C" int rtxn_occupancy_from_density(const float* density, float threshold, int grid_res, uint32_t* occupancy,
                                           rtxn_stream_t stream) {
  RTXN_REQUIRE(density && occupancy, "rtxn_occupancy_from_density: NULL buffer");
  RTXN_REQUIRE(grid_res >= 1 && grid_res <= 1024, "rtxn_occupancy_from_density: grid_res = %d", grid_res);
  RTXN_DEVICE_OR_FAIL();
  const long n = (long)grid_res * grid_res * grid_res;
  const long padded_n = (n + 31) / 32 * 32;
  occupancy_kernel<<<(unsigned)((padded_n + 255) / 256), 256, 0, rtxn::as_stream(stream)>>>(density, threshold, n, occupancy);
  RTXN_LAUNCH_CHECK("occupancy_kernel");
  return RTXN_OK;
}

extern "C" int rtxn_trace_grid(const rtxn_trace_params* p, rtxn_stream_t stream) {
  RTXN_REQUIRE(p != nullptr, "rtxn_trace_grid: params is NULL");
  RTXN_REQUIRE(p->grid_res >= 1 && p->grid_res <= 1024, "rtxn_trace_grid: grid_res = %d out of [1,1024]", p->grid_res);
  RTXN_REQUIRE(p->mode == RTXN_TRACE_COMPAT || p->mode == RTXN_TRACE_DDA, "rtxn_trace_grid: unknown mode %d", p->mode);
  RTXN_REQUIRE(p->look_at || (p->rays_o && p->rays_d), "rtxn_trace_grid: neither look_at nor rays_o/rays_d given");
  {
    uint64_t last = p->ray_count ? (uint64_t)p->ray_count - 1 : 0;
    if (p->window_chunk) last = (last / p->window_chunk) * (uint64_t)p->window_stride + last % p->window_chunk;
    RTXN_REQUIRE(p->ray_count == 0 || (uint64_t)p->ray_begin + last < (uint64_t)p->width * p->height,
                 "rtxn_trace_grid: ray window [%u,+%u) (chunk %u stride %u) exceeds the %ux%u launch", p->ray_begin,
                 p->ray_count, p->window_chunk, p->window_stride, p->width, p->height);
    RTXN_REQUIRE(!p->window_chunk || p->window_stride >= p->window_chunk,
                 "rtxn_trace_grid: window_stride %u < window_chunk %u", p->window_stride, p->window_chunk);
  }
  RTXN_REQUIRE(!p->occupancy_coarse || (p->occupancy && p->grid_res % 4 == 0),
               "rtxn_trace_grid: occupancy_coarse needs occupancy and grid_res %% 4 == 0");
  RTXN_REQUIRE(!p->occupancy_bricks || p->occupancy_coarse, "rtxn_trace_grid: occupancy_bricks needs occupancy_coarse");
  RTXN_REQUIRE(!p->occupancy_super || (p->occupancy_coarse && p->grid_res % 16 == 0),
               "rtxn_trace_grid: occupancy_super needs occupancy_coarse and grid_res %% 16 == 0");
  const bool wants_segments = p->start_points || p->end_points || p->t_start || p->t_end || p->seg_ray || p->seg_view || p->seg_first;
  RTXN_REQUIRE(!wants_segments || p->indices || p->intersection_arr_size > 0,
               "rtxn_trace_grid: segment outputs need indices (packed) or intersection_arr_size > 0 (strided)");
  RTXN_DEVICE_OR_FAIL();
  if (p->ray_count == 0) return RTXN_OK;
  RTXN_REQUIRE(p->num_hits != nullptr, "rtxn_trace_grid: num_hits is NULL");
  hipStream_t s = rtxn::as_stream(stream);
  const int Q = p->mode == RTXN_TRACE_DDA && p->sub_rays > 1 ? p->sub_rays : 1;
  RTXN_REQUIRE(p->sub_rays >= 0 && Q >= 1 && Q <= 64 && (Q & (Q - 1)) == 0 && (p->sub_rays <= 1 || p->mode == RTXN_TRACE_DDA),
               "rtxn_trace_grid: sub_rays = %d must be 0 or a power of two up to 64 (RTXN_TRACE_DDA only)", p->sub_rays);
  RTXN_REQUIRE(Q == 1 || p->sub_hits != nullptr, "rtxn_trace_grid: sub_rays = %d needs the sub_hits scratch", p->sub_rays);
  dim3 grid((unsigned)(((size_t)p->ray_count * Q + 255) / 256)), block(256);
  const TraceLds tl = trace_lds(*p);
  const size_t lds = (size_t)(tl.coarse_words + tl.super_words) * sizeof(uint32_t);
  if (p->mode == RTXN_TRACE_COMPAT) trace_kernel<RTXN_TRACE_COMPAT><<<grid, block, 0, s>>>(*p);
  else trace_kernel<RTXN_TRACE_DDA><<<grid, block, lds, s>>>(*p);
  RTXN_LAUNCH_CHECK("trace_kernel");
  return RTXN_OK;
}

extern "C" int rtxn_build_occupancy_mip(const uint32_t* occupancy, int grid_res, uint32_t* coarse,
                                        rtxn_stream_t stream) {
  RTXN_REQUIRE(occupancy && coarse, "rtxn_build_occupancy_mip: NULL buffer");
  RTXN_REQUIRE(grid_res >= 4 && grid_res % 4 == 0 && grid_res <= 1024,
               "rtxn_build_occupancy_mip: grid_res = %d must be a multiple of 4 in [4,1024]", grid_res);
  RTXN_DEVICE_OR_FAIL();
  const int Rc = grid_res / 4;
  const int words = (Rc * Rc * Rc + 31) / 32;
  mip_kernel<<<(words + 255) / 256, 256, 0, rtxn::as_stream(stream)>>>(occupancy, grid_res, coarse);
  RTXN_LAUNCH_CHECK("mip_kernel");
  return RTXN_OK;
}
