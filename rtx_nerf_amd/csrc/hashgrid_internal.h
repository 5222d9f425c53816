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

// hg_index without the integer division: a hashed level's size is the table cap, a power of two (mask); a densely stored
// level's index is below twice its size for a position inside the domain (x, y, z <= res), so one conditional subtract is the
// modulo.  A position OUTSIDE [-1, 1]^3 (or an Inf / NaN) reaching the public encode / backward entry points gives cell
// coordinates beyond res: the second compare then falls back to the real `% size`, so the index stays inside the level
// whatever the input is -- the same wrap hg_index (and tcnn's grid_index) performs; never taken for in-domain samples.
// `hashed` is uniform per launch row (one level), so the choice is a scalar branch.
__device__ __forceinline__ unsigned hg_index_nodiv(unsigned x, unsigned y, unsigned z, unsigned res, unsigned size, bool hashed) {
  if (hashed) return ((x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u)) & (size - 1u);
  unsigned i = x + y * res + z * res * res;
  if (i >= size) {
    i -= size;
    if (__builtin_expect(i >= size, 0)) i %= size;
  }
  return i;
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
