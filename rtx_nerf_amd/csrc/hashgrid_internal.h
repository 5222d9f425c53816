// Internal (non-ABI) declarations of the multiresolution hash grid shared by train.hip (encode / scatter kernels) and
// hashmlp.hip (the fused hash-encode + MLP inference kernel).
#pragma once
#include "common.h"

#include <cstring>

struct rtxn_hashgrid {
  rtxn_hashgrid_config cfg;
  long n_params;
  float scale[32];
  unsigned res[32], size[32], offset[32];
};

namespace rtxn {

struct HgLevels {
  float scale[16];
  unsigned res[16], size[16], offset[16];
  int n_levels, n_features;
};

__device__ __forceinline__ unsigned hg_index(unsigned x, unsigned y, unsigned z, unsigned res, unsigned size) {
  const unsigned long long dense = (unsigned long long)res * res * res;
  // the +1 corner of a boundary cell indexes one past the level's extent; like tcnn's grid_index the
  // result is reduced modulo the level size, so it wraps instead of leaving the level
  if (dense <= size) return (x + y * res + z * res * res) % size;
  return ((x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u)) % size;
}

// hg_index without the integer division, for cell coordinates INSIDE the level (x, y, z <= res: a position in [-1, 1]^3 and
// its +1 corners): a hashed level's size is the table cap, a power of two (mask); a densely stored level's index is then
// below twice its size, so one conditional subtract is the modulo.  `hashed` is uniform per launch row / loop iteration (one
// level), so the choice is a scalar branch and the rest is branch-free.
__device__ __forceinline__ unsigned hg_index_nodiv(unsigned x, unsigned y, unsigned z, unsigned res, unsigned size, bool hashed) {
  if (hashed) return ((x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u)) & (size - 1u);
  const unsigned i = x + y * res + z * res * res;
  return i >= size ? i - size : i;
}
// A position OUTSIDE the domain (or an Inf / NaN) reaching the public encode / backward entry points gives a base cell
// coordinate beyond res - 1 (negative coordinates convert to huge unsigned values): the single subtract above would then leave
// the level.  The kernels test the base cell ONCE per sample and level -- three compares, wave-reduced with a ballot so that
// the decision is a scalar branch -- and send such a wave through hg_index's real `% size`, the wrap tcnn's grid_index and the
// oracle perform; in-domain waves never take it.  (Hashed levels are masked: always in range.)
__device__ __forceinline__ bool hg_wave_out_of_domain(const unsigned (&g)[3], unsigned res) {
  return __ballot((g[0] >= res) | (g[1] >= res) | (g[2] >= res)) != 0ull;
}
// The eight corner indices of base cell g, chosen by ONE scalar branch per level: lo[yz] = index of (g0, g1 + (yz & 1),
// g2 + (yz >> 1)), hi[yz] the same with g0 + 1 (corner c of the trilinear loops = (c & 1 ? hi : lo)[c >> 1]).  Same values as
// hg_index for every input.
__device__ __forceinline__ void hg_corner_indices(const unsigned (&g)[3], unsigned res, unsigned size, bool hashed, unsigned (&lo)[4],
                                                  unsigned (&hi)[4]) {
  if (hashed) {
    const unsigned m = size - 1u;
#pragma unroll
    for (int yz = 0; yz < 4; ++yz) {
      const unsigned h = ((g[1] + (unsigned)(yz & 1)) * 2654435761u) ^ ((g[2] + (unsigned)(yz >> 1)) * 805459861u);
      lo[yz] = (g[0] ^ h) & m;
      hi[yz] = ((g[0] + 1u) ^ h) & m;
    }
  } else if (!hg_wave_out_of_domain(g, res)) {
#pragma unroll
    for (int yz = 0; yz < 4; ++yz) {
      const unsigned i = g[0] + (g[1] + (unsigned)(yz & 1)) * res + (g[2] + (unsigned)(yz >> 1)) * res * res, j = i + 1u;
      lo[yz] = i >= size ? i - size : i;
      hi[yz] = j >= size ? j - size : j;
    }
  } else {
#pragma unroll
    for (int yz = 0; yz < 4; ++yz) {
      lo[yz] = hg_index(g[0], g[1] + (unsigned)(yz & 1), g[2] + (unsigned)(yz >> 1), res, size);
      hi[yz] = hg_index(g[0] + 1u, g[1] + (unsigned)(yz & 1), g[2] + (unsigned)(yz >> 1), res, size);
    }
  }
}

__device__ __forceinline__ float sin_turns(float x, int f, int ph) {
  // sin(pi * 2^f * x + ph*pi/2) with an exact argument reduction
  return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(ldexpf(x, f - 1)) + 0.25f * (float)ph);
}

inline HgLevels levels_of(const rtxn_hashgrid* g) {
  HgLevels lv;
  memset(&lv, 0, sizeof(lv));
  lv.n_levels = g->cfg.n_levels;
  lv.n_features = g->cfg.n_features;
  for (int l = 0; l < g->cfg.n_levels; ++l) { lv.scale[l] = g->scale[l]; lv.res[l] = g->res[l]; lv.size[l] = g->size[l]; lv.offset[l] = g->offset[l]; }
  return lv;
}

}  // namespace rtxn
