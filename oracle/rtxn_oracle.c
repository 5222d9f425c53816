/*
 * rtxn_oracle.c -- CPU restatement of the owensgroup/rtx_nerf hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke()
 * check of __graft_entry__.py and the cpu_baseline leg of bench.py may load
 * it.  The shipped path (rtx_nerf_amd/librtxn.so, HIP) never links, loads or
 * calls anything in this directory and has no CPU fallback.
 *
 * PARITY UNPINNED.  The reference (read-only at /root/reference) has no tests,
 * no golden vectors and no fixtures, cannot be built in
 * this image (needs nvcc + OptiX 7.7 + tiny-cuda-nn + jsoncpp; main.cu:778 does
 * not compile) and the task rules forbid building it behind stand-in headers.
 * Every function below therefore restates the reference by reading its source,
 * citing file:line, and is pinned only by hand-derived known-answer tests in
 * tests/test_oracle_kat.py.  Floating-point contraction: nvcc contracts a*b+c
 * by default (-fmad=true); this file is compiled with -ffp-contract=off and
 * spells the contractions it assumes as explicit fmaf(), so the HIP kernels can
 * match it bit for bit by spelling the same ones.
 *
 * tiny-cuda-nn (un-vendored submodule, .gitmodules:5-6, version unpinned) is
 * restated from its published algorithm (Frequency encoding; FullyFusedMLP:
 * fp16 weights/activations, fp32 accumulate, no biases) -- see orc_mlp_*.
 *
 * Build: oracle/Makefile  (gcc -O2 -fopenmp -ffp-contract=off -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_K 32 /* NUM_SAMPLES_PER_SEGMENT, sampler/sampler.h:4 */

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* fp16 helpers (gcc 11 has no _Float16 on x86): round-to-nearest-even        */
/* ------------------------------------------------------------------------- */
uint16_t orc_f32_to_f16_bits(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) { /* inf / nan */
    return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0));
  }
  if (ax >= 0x477ff000u) { /* >= 65520 -> inf */
    return (uint16_t)(sign | 0x7c00u);
  }
  if (ax < 0x38800000u) { /* subnormal half or zero */
    if (ax < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 */
    int e = (int)(ax >> 23);                     /* biased exp */
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    int shift = 126 - e; /* shift so that result is in units of 2^-24 */
    /* value = m * 2^(e-150); half subnormal unit 2^-24 => q = m * 2^(e-126) */
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    return (uint16_t)(sign | q);
  }
  /* normal */
  uint32_t e = (ax >> 23) - 112u;
  uint32_t m = ax & 0x7fffffu;
  uint32_t h = (e << 10) | (m >> 13);
  uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
  return (uint16_t)(sign | h);
}

float orc_f16_bits_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu;
  uint32_t m = h & 0x3ffu;
  uint32_t x;
  if (e == 0) {
    if (m == 0) {
      x = sign;
    } else { /* subnormal */
      int s = 0;
      while (!(m & 0x400u)) { m <<= 1; s++; }
      m &= 0x3ffu;
      x = sign | ((uint32_t)(113 - s) << 23) | (m << 13);
    }
  } else if (e == 31) {
    x = sign | 0x7f800000u | (m << 13);
  } else {
    x = sign | ((e + 112u) << 23) | (m << 13);
  }
  float f;
  memcpy(&f, &x, 4);
  return f;
}

static inline float rh(float f) { return orc_f16_bits_to_f32(orc_f32_to_f16_bits(f)); }

/* ------------------------------------------------------------------------- */
/* a2  ray generation -- rtx/src/optixPrograms.cu:43-82                       */
/* ------------------------------------------------------------------------- */
/* look_at: 16 floats row-major 4x4, translation at [3],[7],[11] (:75).
 * Outputs: origin[3] (= translation/10, :76-78), dir[3] (normalised, :62-69),
 * view[2] = (theta, phi) (:71-73).  u,v are formed in double as the reference's
 * mixed int/double/float expression does (:56-57); no image-y flip (quirk Q3). */
void orc_make_ray(const float* la, float focal_length, float aspect_ratio,
                  unsigned width, unsigned height, unsigned px, unsigned py,
                  float* origin, float* dir, float* view) {
  float u = (float)((2 * (px + 0.5) / width - 1) * aspect_ratio);
  float v = (float)(2 * (py + 0.5) / height - 1);
  float nf0 = la[2] * -1.0f, nf1 = la[6] * -1.0f, nf2 = la[10] * -1.0f;
  float xd = fmaf(nf0, focal_length, fmaf(la[0], u, la[1] * v));
  float yd = fmaf(nf1, focal_length, fmaf(la[4], u, la[5] * v));
  float zd = fmaf(nf2, focal_length, fmaf(la[8], u, la[9] * v));
  float norm = sqrtf(fmaf(zd, zd, fmaf(xd, xd, yd * yd)));
  xd /= norm;
  yd /= norm;
  zd /= norm;
  view[0] = atan2f(sqrtf(fmaf(xd, xd, yd * yd)), zd);
  view[1] = atan2f(yd, xd);
  dir[0] = xd;
  dir[1] = yd;
  dir[2] = zd;
  origin[0] = la[3] / 10;
  origin[1] = la[7] / 10;
  origin[2] = la[11] / 10;
}

/* ------------------------------------------------------------------------- */
/* a6  grid -- main.cu:154-174 (make_grid)                                    */
/* ------------------------------------------------------------------------- */
/* Cell i along one axis spans [-1 + i*L, -1 + i*L + L], L = 2/R, evaluated in
 * fp32 in exactly that order (main.cu:161-166).  Primitive index of cell
 * (x,y,z) is (x*R + y)*R + z (loop order :158-160); the occupancy bitfield of
 * this build uses the same index, bit (idx & 31) of word idx >> 5. */
static inline float cell_lo(int i, float L) { return -1.0f + (float)i * L; }
static inline float cell_hi(int i, float L) { return -1.0f + (float)i * L + L; }

static inline int occ_test(const uint32_t* occ, int R, int x, int y, int z) {
  if (!occ) return 1;
  uint32_t idx = ((uint32_t)x * (uint32_t)R + (uint32_t)y) * (uint32_t)R + (uint32_t)z;
  return (occ[idx >> 5] >> (idx & 31)) & 1u;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* slab test of one box, __intersection__ray_march, optixPrograms.cu:132-169.
 * Returns 1 when a hit would be reported (tmax > tmin) and stores the reported
 * t (with the origin-inside clamp, :164-165). */
static inline int slab_box(const float* o, const float* d, const float* lo, const float* hi,
                           float* t_rep, float* t_far) {
  float tmin = -INFINITY, tmax = INFINITY;
  for (int a = 0; a < 3; ++a) {
    if (d[a] == 0.0f) {
      /* ray parallel to this slab.  The reference divides by zero here (+-inf, or NaN when the origin lies on
       * the plane, and -inf exit times for d = -0.0): its result is undefined.  This build defines it: the slab
       * constrains nothing if the origin is inside it and rejects the box otherwise. */
      if (!(o[a] >= lo[a] && o[a] <= hi[a])) { tmin = INFINITY; }
      continue;
    }
    float t1 = (lo[a] - o[a]) / d[a];
    float t2 = (hi[a] - o[a]) / d[a];
    tmin = fmaxf(tmin, fminf(t1, t2));
    tmax = fminf(tmax, fmaxf(t1, t2));
  }
  *t_far = tmax;
  if (tmax > tmin) {
    if (tmin < 0 && (double)tmax > 1e-6) tmin = 0;
    *t_rep = tmin;
    return 1;
  }
  *t_rep = tmin;
  return 0;
}

/* Segment sink shared by both traversal modes. */
typedef struct {
  float* start; /* float3 per slot or NULL */
  float* end;
  float* t0;
  float* t1;
  int* seg_ray;
  int base;     /* first slot of this ray */
  int cap;      /* slots available to this ray */
  int ray;
  int n;
} orc_sink;

static inline void sink_emit(orc_sink* s, const float* p0, const float* p1, float t0, float t1) {
  if (s->n < s->cap) {
    int k = s->base + s->n;
    if (s->start) { s->start[3 * k] = p0[0]; s->start[3 * k + 1] = p0[1]; s->start[3 * k + 2] = p0[2]; }
    if (s->end) { s->end[3 * k] = p1[0]; s->end[3 * k + 1] = p1[1]; s->end[3 * k + 2] = p1[2]; }
    if (s->t0) s->t0[k] = t0;
    if (s->t1) s->t1[k] = t1;
    if (s->seg_ray) s->seg_ray[k] = s->ray;
  }
  s->n++;
}

/* Entry of the ray into the whole grid [-1,1]^3.  Returns 0 on a miss.  On a
 * hit stores the first cell.  Shared by both modes (the reference gets there
 * by closest-hit over all boxes, optixPrograms.cu:99-115). */
static int grid_entry(const float* o, const float* d, int R, float L, int* cell, float* t_enter) {
  int inside = 1;
  for (int a = 0; a < 3; ++a)
    if (!(o[a] >= -1.0f && o[a] <= 1.0f)) inside = 0;
  float t0 = 0.0f;
  int enter_axis = -1;
  if (!inside) {
    float tmin = -INFINITY, tmax = INFINITY;
    for (int a = 0; a < 3; ++a) {
      if (d[a] == 0.0f) { /* parallel to this slab of the grid: inside it or a miss */
        if (!(o[a] >= -1.0f && o[a] <= 1.0f)) return 0;
        continue;
      }
      float t1 = (-1.0f - o[a]) / d[a];
      float t2 = (1.0f - o[a]) / d[a];
      float tn = fminf(t1, t2), tf = fmaxf(t1, t2);
      if (tn > tmin) { tmin = tn; enter_axis = a; }
      tmax = fminf(tmax, tf);
    }
    if (!(tmax > tmin) || tmin < 0.0f) return 0;
    t0 = tmin;
  }
  for (int a = 0; a < 3; ++a) {
    float p = fmaf(t0, d[a], o[a]);
    int c = (int)floorf((p + 1.0f) / L);
    cell[a] = clampi(c, 0, R - 1);
  }
  if (enter_axis >= 0) cell[enter_axis] = d[enter_axis] > 0 ? 0 : R - 1;
  *t_enter = t0;
  return 1;
}

/* a2-a5 COMPAT march -- the reference's "closest hit, then re-launch from the
 * exit point" chain (optixPrograms.cu:99-115,180-248) restated as a cell walk:
 * per crossed cell, t_hit is the slab test from the CURRENT (re-launched)
 * origin (:132-169), start = o + t_hit*d (:194-196), the exit plane is picked
 * by the sign of d (:201-203), t_e = min(t_x,t_y,t_z) (:204-207),
 * end = o + t_e*d (:209-211), and the next origin is `end` (:245-247).  Hence
 * t_start ~ 0 and t_end = segment length after the first segment.
 * Where OptiX's order among equal-t candidates is unspecified (rays through a
 * shared edge/corner) this walk steps every tied axis at once and emits no
 * sliver segment.  occ == NULL is the reference's dense grid; with occ only
 * occupied cells are emitted, the chain itself is unchanged. */
static void march_compat(const float* o0, const float* d, int R, const uint32_t* occ, orc_sink* s) {
  float L = 2.0f / (float)R;
  int c[3];
  float tE;
  if (!grid_entry(o0, d, R, L, c, &tE)) return;
  float o[3] = {o0[0], o0[1], o0[2]};
  for (int guard = 0; guard < 3 * R + 8; ++guard) {
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = cell_lo(c[a], L); hi[a] = cell_hi(c[a], L); }
    float t_hit, t_far;
    int rep = slab_box(o, d, lo, hi, &t_hit, &t_far);
    float te[3];
    for (int a = 0; a < 3; ++a) {
      float plane = d[a] < 0 ? lo[a] : hi[a];
      te[a] = d[a] == 0.0f ? INFINITY : (plane - o[a]) / d[a];   /* a parallel axis is never the exit axis */
    }
    float t_e = fminf(fminf(te[0], te[1]), te[2]);
    float p0[3], p1[3];
    for (int a = 0; a < 3; ++a) { p0[a] = fmaf(t_hit, d[a], o[a]); p1[a] = fmaf(t_e, d[a], o[a]); }
    if (rep && t_hit >= 0.0f && occ_test(occ, R, c[0], c[1], c[2])) sink_emit(s, p0, p1, t_hit, t_e);
    int out = 0;
    for (int a = 0; a < 3; ++a) {
      if (te[a] == t_e) {
        c[a] += d[a] < 0 ? -1 : 1;
        if (c[a] < 0 || c[a] >= R) out = 1;
      }
    }
    if (out) break;
    o[0] = p1[0]; o[1] = p1[1]; o[2] = p1[2];
  }
}

/* DDA march (this build's fast mode, not in the reference): same cells, but t
 * is a GLOBAL ray parameter from the original origin and every plane crossing
 * is a pure function of the integer cell index, t_a(i) = (plane_a(i) - o_a) *
 * (1/d_a), so a hierarchical walk that skips empty space reproduces a flat one
 * bit for bit.  start = o + t_in*d, end = o + t_out*d. */
static inline float plane_t(int i, float L, float o, float inv) { return (cell_lo(i, L) - o) * inv; }

static void march_dda(const float* o, const float* d, int R, const uint32_t* occ, orc_sink* s) {
  float L = 2.0f / (float)R;
  int c[3];
  float t_in;
  if (!grid_entry(o, d, R, L, c, &t_in)) return;
  float inv[3];
  int step[3];
  for (int a = 0; a < 3; ++a) { inv[a] = 1.0f / d[a]; step[a] = d[a] < 0 ? -1 : 1; }
  for (int guard = 0; guard < 3 * R + 8; ++guard) {
    float te[3];
    for (int a = 0; a < 3; ++a) {
      if (d[a] == 0.0f) te[a] = INFINITY;
      else te[a] = plane_t(c[a] + (step[a] > 0 ? 1 : 0), L, o[a], inv[a]);
    }
    float t_out = fminf(fminf(te[0], te[1]), te[2]);
    if (t_out > t_in && occ_test(occ, R, c[0], c[1], c[2])) {
      float p0[3], p1[3];
      for (int a = 0; a < 3; ++a) { p0[a] = fmaf(t_in, d[a], o[a]); p1[a] = fmaf(t_out, d[a], o[a]); }
      sink_emit(s, p0, p1, t_in, t_out);
    }
    int out = 0;
    for (int a = 0; a < 3; ++a) {
      if (te[a] == t_out) {
        c[a] += step[a];
        if (c[a] < 0 || c[a] >= R) out = 1;
      }
    }
    if (out) break;
    if (t_out > t_in) t_in = t_out;
  }
}

/* Trace a window of the W x H launch.  mode 0 = COMPAT, 1 = DDA.
 * look_at != NULL: pinhole rays as a2; look_at == NULL: rays_o/rays_d[gid].
 * Layout A (reference, strided): indices == NULL, slot = local_ray*S + k with
 * S = intersection_arr_size (main.cu:486, optixPrograms.cu:184).
 * Layout B (packed CSR): indices != NULL, slot = indices[local_ray] + k.
 * num_hits is always written (:241); any of the segment outputs may be NULL
 * (count-only pass). */
void orc_trace(const float* look_at, const float* rays_o, const float* rays_d, float focal_length,
               float aspect_ratio, unsigned width, unsigned height, int R, const uint32_t* occ, int mode,
               unsigned ray_begin, unsigned ray_count, unsigned window_chunk, unsigned window_stride, int S,
               const int* indices, float* ray_origins,
               float* view_dirs, int* num_hits, float* start_points, float* end_points, float* t_start,
               float* t_end, int* seg_ray) {
#pragma omp parallel for schedule(dynamic, 64)
  for (long r = 0; r < (long)ray_count; ++r) {
    unsigned gid = window_chunk ? ray_begin + ((unsigned)r / window_chunk) * window_stride + (unsigned)r % window_chunk
                                : ray_begin + (unsigned)r;
    float o[3], d[3], v[2];
    if (look_at) {
      unsigned px = gid % width, py = gid / width;
      orc_make_ray(look_at, focal_length, aspect_ratio, width, height, px, py, o, d, v);
    } else { /* explicit rays (this build's extension; theta/phi as :71-73) */
      for (int a = 0; a < 3; ++a) { o[a] = rays_o[3 * (size_t)gid + a]; d[a] = rays_d[3 * (size_t)gid + a]; }
      v[0] = atan2f(sqrtf(fmaf(d[0], d[0], d[1] * d[1])), d[2]);
      v[1] = atan2f(d[1], d[0]);
    }
    if (ray_origins) { ray_origins[3 * r] = o[0]; ray_origins[3 * r + 1] = o[1]; ray_origins[3 * r + 2] = o[2]; }
    if (view_dirs) { view_dirs[2 * r] = v[0]; view_dirs[2 * r + 1] = v[1]; }
    orc_sink s;
    s.start = start_points; s.end = end_points; s.t0 = t_start; s.t1 = t_end; s.seg_ray = seg_ray;
    s.ray = (int)r; s.n = 0;
    if (indices) { s.base = indices[r]; s.cap = 0x7fffffff; }
    else { s.base = (int)r * S; s.cap = S; }
    if (!start_points && !end_points && !t_start && !t_end && !seg_ray) s.cap = 0;
    if (mode == 0) march_compat(o, d, R, occ, &s);
    else march_dda(o, d, R, occ, &s);
    num_hits[r] = s.n;
  }
}

/* ------------------------------------------------------------------------- */
/* a7  CSR compaction -- main.cu:631-637 (thrust::reduce + exclusive_scan)    */
/* ------------------------------------------------------------------------- */
int orc_scan_hits(const int* num_hits, int* indices, int n) {
  int acc = 0;
  for (int i = 0; i < n; ++i) { indices[i] = acc; acc += num_hits[i]; }
  return acc;
}

/* ------------------------------------------------------------------------- */
/* a8  sampler -- sampler/sampler.cu:14-103                                   */
/* ------------------------------------------------------------------------- */
/* thrust::minstd_rand (default seed 1): x <- 48271*x mod (2^31-1), passed BY
 * VALUE into the kernel (sampler.cu:25,117) so every ray draws the identical
 * sequence, one draw per sample in (segment, i) order.
 * thrust::uniform_real_distribution<float>(a,b):
 *   r = float(x - 1) / (1.0f + float(2147483645)) ; r*(b-a)+a
 * (rocThrust/thrust uniform_real_distribution.inl). */
static inline uint32_t minstd_next(uint32_t x) { return (uint32_t)(((uint64_t)x * 48271u) % 2147483647u); }
static inline float thrust_uniform(uint32_t x, float a, float b) {
  float r = (float)(x - 1u);
  r /= (1.0f + (float)2147483645u);
  return fmaf(r, b - a, a);
}

void orc_sample(const float* start_points, const float* end_points, const float* view_dirs,
                float* t_vals, float* samples, int batch_size, int grid_res, const int* num_hits,
                const int* indices, int sample_type) {
  (void)grid_res; /* unused by the reference too */
  const float inc = 1.0f / ORC_K;
#pragma omp parallel for schedule(dynamic, 64)
  for (int x = 0; x < batch_size; ++x) {
    int start_index = indices[x];
    int n_hits = num_hits[x];
    float theta = view_dirs[2 * x], phi = view_dirs[2 * x + 1];
    uint32_t rng = 1u;
    for (int j = 0; j < n_hits; ++j) {
      const float* og = start_points + 3 * (size_t)(start_index + j);
      const float* fn = end_points + 3 * (size_t)(start_index + j);
      float dir[3] = {fn[0] - og[0], fn[1] - og[1], fn[2] - og[2]};
      float t_initial = 0.0f, t_final = inc;
      for (int i = 0; i < ORC_K; ++i) {
        size_t n = (size_t)(start_index + j) * ORC_K + i;
        float t, tv;
        if (sample_type == 0) { /* REGULAR :52-66 */
          t = t_initial;
          t_initial += inc;
          tv = t_initial;
        } else if (sample_type == 3) { /* MIDPOINT_WORLD (this build, for RTXN_VR_NERF) */
          t = ((float)i + 0.5f) * inc;
          tv = sqrtf(fmaf(dir[2], dir[2], fmaf(dir[0], dir[0], dir[1] * dir[1]))) * inc;
        } else if (sample_type == 2) { /* UNIFORM :67-80 */
          rng = minstd_next(rng);
          t = thrust_uniform(rng, 0.0f, 1.0f);
          tv = t_initial; /* never advanced: always 0 (:71) */
        } else { /* STRATIFIED_JITTERING :81-98 */
          rng = minstd_next(rng);
          t = thrust_uniform(rng, t_initial, t_final);
          tv = t_initial;
          t_initial = t_final;
          t_final += inc;
        }
        samples[n * 5 + 0] = fmaf(t, dir[0], og[0]);
        samples[n * 5 + 1] = fmaf(t, dir[1], og[1]);
        samples[n * 5 + 2] = fmaf(t, dir[2], og[2]);
        samples[n * 5 + 3] = theta;
        samples[n * 5 + 4] = phi;
        t_vals[n] = tv;
      }
    }
  }
}

/* ------------------------------------------------------------------------- */
/* a9  volume render forward -- vol_render/vol_render.cu:19-73                */
/* ------------------------------------------------------------------------- */
/* mode 0 (COMPAT) is the reference: delta = |t - t_prev| with t_prev starting
 * at 0 and NOT reset per segment (:56-57, the FIXME), transmittance += delta*
 * sigma BEFORE the weight (inclusive, :60), w = exp(-T)*(1-exp(-delta*sigma))
 * (:61-63).  The serial fp32 accumulation order is the reference's; the HIP
 * kernel scans in a different order, hence the 1e-5 abs tolerance. */
void orc_volrender_fwd(const float* network_outputs, const int* num_hits, const int* indices,
                       const float* ray_hit, int batch_size, int K, float* pixels) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int x = 0; x < batch_size; ++x) {
    int start_index = indices[x];
    int n = num_hits[x];
    float T = 0.0f, t_prev = 0.0f;
    float acc[3] = {0, 0, 0};
    for (int j = 0; j < n; ++j) {
      for (int i = 0; i < K; ++i) {
        size_t s = (size_t)(start_index + j) * K + i;
        const float* c = network_outputs + 4 * s;
        float sigma = c[3];
        float t = ray_hit[s];
        float delta = fabsf(t - t_prev);
        t_prev = t;
        T = fmaf(delta, sigma, T);
        float w = expf(-T) * (1 - expf(-delta * sigma));
        acc[0] += w * c[0];
        acc[1] += w * c[1];
        acc[2] += w * c[2];
      }
    }
    pixels[3 * x] = acc[0];
    pixels[3 * x + 1] = acc[1];
    pixels[3 * x + 2] = acc[2];
  }
}

/* ------------------------------------------------------------------------- */
/* a10 volume render backward -- vol_render/vol_render.cu:75-143              */
/* ------------------------------------------------------------------------- */
/* Reference semantics, including that `transmittance` is ASSIGNED delta*sigma
 * (:118) and that the result is not the analytic gradient of a9.
 * loss_gradients: half[B*3] as raw bits; radiance_gradients: half[N*4] bits. */
void orc_volrender_bwd(const uint16_t* loss_gradients, const float* radiance, const float* t_hit,
                       const int* num_hits, const int* indices, int batch_size, int K,
                       uint16_t* radiance_gradients) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int x = 0; x < batch_size; ++x) {
    int start_index = indices[x];
    int n = num_hits[x];
    float t_prev = 0.0f;
    float g[3] = {orc_f16_bits_to_f32(loss_gradients[3 * x]), orc_f16_bits_to_f32(loss_gradients[3 * x + 1]),
                  orc_f16_bits_to_f32(loss_gradients[3 * x + 2])};
    for (int j = 0; j < n; ++j) {
      for (int i = 0; i < K; ++i) {
        size_t s = (size_t)(start_index + j) * K + i;
        const float* c = radiance + 4 * s;
        float sigma = c[3];
        float t = t_hit[s];
        float delta = fabsf(t - t_prev);
        t_prev = t;
        float tr = delta * sigma;
        float e = expf(-sigma * delta);
        float dg = 0.0f;
        dg += g[0] * tr * c[0] * delta * e;
        dg += g[1] * tr * c[1] * delta * e;
        dg += g[2] * tr * c[2] * delta * e;
        float om = 1 - expf(-delta * sigma);
        radiance_gradients[4 * s + 0] = orc_f32_to_f16_bits(g[0] * tr * om);
        radiance_gradients[4 * s + 1] = orc_f32_to_f16_bits(g[1] * tr * om);
        radiance_gradients[4 * s + 2] = orc_f32_to_f16_bits(g[2] * tr * om);
        radiance_gradients[4 * s + 3] = orc_f32_to_f16_bits(dg);
      }
    }
  }
}

/* ------------------------------------------------------------------------- */
/* a11 tiny-cuda-nn model, restated from the published algorithm              */
/* ------------------------------------------------------------------------- */
/* Composite(Frequency(n_pos_dims, n_pos_freqs), Frequency(n_dir_dims,
 * n_dir_freqs)) -> FullyFusedMLP(n_hidden_layers x n_neurons, ReLU) -> out
 * (padded to 16) -> Sigmoid (main.cu:35-69).  Frequency feature j of an
 * encoding with F frequencies: dim = j/(2F), f = (j/2)%F,
 * value = sin(2^f * pi * x_dim + (j&1)*pi/2), rounded to fp16.  The encoded
 * width is padded to a multiple of 16 with ones.  Weights: fp16, no biases,
 * layer l is a row-major [out][in] matrix, layers concatenated in order
 * (first: n_neurons x enc_padded; hidden: n_neurons x n_neurons; last:
 * 16 x n_neurons).  Hidden activations are rounded to fp16 after ReLU; dot
 * products accumulate in fp32 in k order.  Output c of sample n is written to
 * out[n*16 + c] as fp16 bits (column-major 16 x N, as tcnn's GPUMatrix). */
typedef struct {
  int n_pos_dims, n_pos_freqs, n_dir_dims, n_dir_freqs;
  int n_neurons, n_hidden_layers, n_output_dims;
  int output_activation; /* 0 none, 1 sigmoid */
} orc_mlp_cfg;

int orc_mlp_enc_width(const orc_mlp_cfg* c) { return 2 * (c->n_pos_dims * c->n_pos_freqs + c->n_dir_dims * c->n_dir_freqs); }
int orc_mlp_enc_padded(const orc_mlp_cfg* c) { return (orc_mlp_enc_width(c) + 15) / 16 * 16; }
long orc_mlp_n_params(const orc_mlp_cfg* c) {
  long W = c->n_neurons;
  return W * orc_mlp_enc_padded(c) + (long)(c->n_hidden_layers - 1) * W * W + 16 * W;
}

void orc_freq_encode(const orc_mlp_cfg* c, const float* in5, float* enc /* enc_padded, fp16-rounded */) {
  int j = 0;
  for (int part = 0; part < 2; ++part) {
    int nd = part == 0 ? c->n_pos_dims : c->n_dir_dims;
    int F = part == 0 ? c->n_pos_freqs : c->n_dir_freqs;
    int off = part == 0 ? 0 : c->n_pos_dims;
    for (int k = 0; k < 2 * nd * F; ++k, ++j) {
      int dim = k / (2 * F), f = (k / 2) % F;
      /* the published formula evaluated without argument-rounding noise: the
       * product x*2^f is exact, pi and the sine are taken in double */
      double arg = M_PI * ldexp((double)in5[off + dim], f) + (double)(k & 1) * (M_PI / 2);
      enc[j] = rh((float)sin(arg));
    }
  }
  int P = orc_mlp_enc_padded(c);
  for (; j < P; ++j) enc[j] = 1.0f;
}

/* Weights widened to fp32 and TRANSPOSED to [in][out] per layer so the inner
 * loop runs over output rows (vectorisable) while every row still accumulates
 * its products sequentially in k order with one fmaf each -- the result is
 * bit-identical to the textbook row-by-row loop. */
static float* net_prepare(const orc_mlp_cfg* c, const uint16_t* params) {
  int W = c->n_neurons, P = orc_mlp_enc_padded(c), nh = c->n_hidden_layers;
  long np_ = orc_mlp_n_params(c);
  float* wT = (float*)malloc(sizeof(float) * np_);
  long off = 0;
  int in_w = P;
  for (int l = 0; l <= nh; ++l) {
    int rows = l == nh ? 16 : W;
    for (int r = 0; r < rows; ++r)
      for (int k = 0; k < in_w; ++k) wT[off + (long)k * rows + r] = orc_f16_bits_to_f32(params[off + (long)r * in_w + k]);
    off += (long)rows * in_w;
    in_w = W;
  }
  return wT;
}

/* a: encoded input (fp16-rounded floats), overwritten; b: scratch; y: 16 activated outputs (fp32, unrounded) */
static void net_eval(const orc_mlp_cfg* c, const float* wT, float* a, float* b, float* y) {
  int W = c->n_neurons, P = orc_mlp_enc_padded(c), nh = c->n_hidden_layers;
  const float* w = wT;
  int in_w = P;
  float acc[256];
  for (int l = 0; l < nh; ++l) {
    for (int r = 0; r < W; ++r) acc[r] = 0.0f;
    for (int k = 0; k < in_w; ++k) {
      const float ak = a[k];
      const float* wk = w + (long)k * W;
      for (int r = 0; r < W; ++r) acc[r] = fmaf(wk[r], ak, acc[r]);
    }
    for (int r = 0; r < W; ++r) b[r] = rh(acc[r] > 0.0f ? acc[r] : 0.0f);
    w += (long)W * in_w;
    in_w = W;
    float* t = a; a = b; b = t;
  }
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  for (int k = 0; k < W; ++k) {
    const float ak = a[k];
    const float* wk = w + (long)k * 16;
    for (int r = 0; r < 16; ++r) acc[r] = fmaf(wk[r], ak, acc[r]);
  }
  for (int r = 0; r < 16; ++r) y[r] = c->output_activation == 1 ? 1.0f / (1.0f + expf(-acc[r])) : acc[r];
}

void orc_mlp_forward(const orc_mlp_cfg* c, const uint16_t* params, const float* inputs /*[N][5]*/,
                     uint16_t* out /*[N][16]*/, long N) {
  int W = c->n_neurons, P = orc_mlp_enc_padded(c);
  float* wT = net_prepare(c, params);
  int nin = c->n_pos_dims + c->n_dir_dims;
#pragma omp parallel
  {
    float* a = (float*)malloc(sizeof(float) * (P > W ? P : W));
    float* b = (float*)malloc(sizeof(float) * (P > W ? P : W));
#pragma omp for schedule(static)
    for (long n = 0; n < N; ++n) {
      float y[16];
      orc_freq_encode(c, inputs + n * nin, a);
      net_eval(c, wT, a, b, y);
      for (int r = 0; r < 16; ++r) out[n * 16 + r] = orc_f32_to_f16_bits(y[r]);
    }
    free(a);
    free(b);
  }
  free(wT);
}

/* a13 glue: rows 0..3 of the half output -> float radiance AoS (the evident
 * intent of convertHalfToFloat, main.cu:203-208,723-728). */
void orc_radiance_from_half(const uint16_t* out16, float* radiance, long N) {
  for (long n = 0; n < N; ++n)
    for (int c = 0; c < 4; ++c) radiance[4 * n + c] = orc_f16_bits_to_f32(out16[16 * n + c]);
}

/* ------------------------------------------------------------------------- */
/* End-to-end host ray-march (config 1 and the cpu_baseline of bench.py):     */
/* trace -> scan -> sample(REGULAR) -> MLP -> composite, per ray, no big      */
/* intermediate buffers.  Same arithmetic as the staged functions above.      */
/* ------------------------------------------------------------------------- */
void orc_render(const float* look_at, float focal_length, float aspect_ratio, unsigned width,
                unsigned height, int R, const uint32_t* occ, int trace_mode, const orc_mlp_cfg* cfg,
                const uint16_t* params, const unsigned* ray_ids, long n_rays, float* pixels,
                long* total_samples) {
  int W = cfg->n_neurons, P = orc_mlp_enc_padded(cfg);
  float* wT = net_prepare(cfg, params);
  long tot = 0;
  const int S = 3 * R + 8;
#pragma omp parallel reduction(+ : tot)
  {
    float* sp = (float*)malloc(sizeof(float) * 3 * S);
    float* ep = (float*)malloc(sizeof(float) * 3 * S);
    float* a = (float*)malloc(sizeof(float) * (P > W ? P : W));
    float* b = (float*)malloc(sizeof(float) * (P > W ? P : W));
#pragma omp for schedule(dynamic, 4)
    for (long r = 0; r < n_rays; ++r) {
      unsigned gid = ray_ids[r];
      float o[3], d[3], v[2];
      orc_make_ray(look_at, focal_length, aspect_ratio, width, height, gid % width, gid / width, o, d, v);
      orc_sink s;
      s.start = sp; s.end = ep; s.t0 = 0; s.t1 = 0; s.seg_ray = 0; s.base = 0; s.cap = S; s.ray = 0; s.n = 0;
      if (trace_mode == 0) march_compat(o, d, R, occ, &s);
      else march_dda(o, d, R, occ, &s);
      int nseg = s.n < S ? s.n : S;
      float T = 0.0f, t_prev = 0.0f, acc3[3] = {0, 0, 0};
      const float inc = 1.0f / ORC_K;
      for (int j = 0; j < nseg; ++j) {
        float dir[3] = {ep[3 * j] - sp[3 * j], ep[3 * j + 1] - sp[3 * j + 1], ep[3 * j + 2] - sp[3 * j + 2]};
        float t_initial = 0.0f;
        for (int i = 0; i < ORC_K; ++i) {
          float t = t_initial;
          t_initial += inc;
          float in5[5] = {fmaf(t, dir[0], sp[3 * j]), fmaf(t, dir[1], sp[3 * j + 1]), fmaf(t, dir[2], sp[3 * j + 2]), v[0], v[1]};
          float y[16], rad[4];
          orc_freq_encode(cfg, in5, a);
          net_eval(cfg, wT, a, b, y);
          for (int q = 0; q < 4; ++q) rad[q] = rh(y[q]);
          float tv = t_initial;
          float delta = fabsf(tv - t_prev);
          t_prev = tv;
          T = fmaf(delta, rad[3], T);
          float wgt = expf(-T) * (1 - expf(-delta * rad[3]));
          acc3[0] += wgt * rad[0];
          acc3[1] += wgt * rad[1];
          acc3[2] += wgt * rad[2];
        }
      }
      tot += (long)nseg * ORC_K;
      pixels[3 * r] = acc3[0];
      pixels[3 * r + 1] = acc3[1];
      pixels[3 * r + 2] = acc3[2];
    }
    free(sp); free(ep); free(a); free(b);
  }
  free(wT);
  if (total_samples) *total_samples = tot;
}

/* ------------------------------------------------------------------------- */
/* cpu_baseline leg of bench.py: the same host ray-march with the network     */
/* evaluated for 64 samples at a time (two segments) as small GEMMs --        */
/* AVX2 + FMA micro-kernel of 4 samples x 16 outputs, weights reused across   */
/* the tile, activations rounded to fp16 by the F16C conversion instructions, */
/* the encoding's sines in single precision on an exactly reduced argument.   */
/* Same algorithm and roundings as orc_render (fp16 weights, fp32 accumulate, */
/* fp16 activations); the summation ORDER over k is the same too, so the only */
/* differences are the float sine (<= 1 fp16 ulp of a feature, now and then)  */
/* -- validated against orc_render in tests/test_oracle_kat.py.  It exists so */
/* that the CPU figure beside the GPU one is a tuned one*/
/* ------------------------------------------------------------------------- */
#include <immintrin.h>

static inline float rh_hw(float f) { return _cvtsh_ss(_cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT)); }

/* out[s][r] = act( sum_k wT[k][r] * in[s][k] ), s < 64 (padded tile), r < rows (multiple of 16); relu_round: ReLU + fp16 */
static void tiled_layer(const float* wT, int in_w, int rows, const float* in, int in_stride, float* out, int out_stride, int relu_round) {
  for (int s0 = 0; s0 < 64; s0 += 4) {
    const float* a0 = in + (long)s0 * in_stride;
    for (int r0 = 0; r0 < rows; r0 += 16) {
      __m256 c00 = _mm256_setzero_ps(), c01 = c00, c10 = c00, c11 = c00, c20 = c00, c21 = c00, c30 = c00, c31 = c00;
      const float* w = wT + r0;
      for (int k = 0; k < in_w; ++k, w += rows) {
        const __m256 w0 = _mm256_loadu_ps(w), w1 = _mm256_loadu_ps(w + 8);
        const __m256 x0 = _mm256_broadcast_ss(a0 + k), x1 = _mm256_broadcast_ss(a0 + in_stride + k);
        const __m256 x2 = _mm256_broadcast_ss(a0 + 2 * in_stride + k), x3 = _mm256_broadcast_ss(a0 + 3 * in_stride + k);
        c00 = _mm256_fmadd_ps(w0, x0, c00); c01 = _mm256_fmadd_ps(w1, x0, c01);
        c10 = _mm256_fmadd_ps(w0, x1, c10); c11 = _mm256_fmadd_ps(w1, x1, c11);
        c20 = _mm256_fmadd_ps(w0, x2, c20); c21 = _mm256_fmadd_ps(w1, x2, c21);
        c30 = _mm256_fmadd_ps(w0, x3, c30); c31 = _mm256_fmadd_ps(w1, x3, c31);
      }
      __m256 c[4][2] = {{c00, c01}, {c10, c11}, {c20, c21}, {c30, c31}};
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j) {
          __m256 v = c[i][j];
          if (relu_round) {
            v = _mm256_max_ps(v, _mm256_setzero_ps());
            v = _mm256_cvtph_ps(_mm256_cvtps_ph(v, _MM_FROUND_TO_NEAREST_INT));
          }
          _mm256_storeu_ps(out + (long)(s0 + i) * out_stride + r0 + 8 * j, v);
        }
    }
  }
}

void orc_render_tiled(const float* look_at, float focal_length, float aspect_ratio, unsigned width, unsigned height, int R,
                      const uint32_t* occ, int trace_mode, const orc_mlp_cfg* cfg, const uint16_t* params, const unsigned* ray_ids,
                      long n_rays, float* pixels, long* total_samples) {
  const int W = cfg->n_neurons, P = orc_mlp_enc_padded(cfg), nh = cfg->n_hidden_layers;
  const int ST = P > W ? P : W;                      /* row stride of the activation tiles */
  float* wT = net_prepare(cfg, params);
  long tot = 0;
  const int S = 3 * R + 8;
#pragma omp parallel reduction(+ : tot)
  {
    float* sp = (float*)malloc(sizeof(float) * 3 * S);
    float* ep = (float*)malloc(sizeof(float) * 3 * S);
    float* ta = (float*)aligned_alloc(64, sizeof(float) * 64 * ST);
    float* tb = (float*)aligned_alloc(64, sizeof(float) * 64 * ST);
    float* rad = (float*)malloc(sizeof(float) * 4 * ORC_K * S);
#pragma omp for schedule(dynamic, 4)
    for (long r = 0; r < n_rays; ++r) {
      unsigned gid = ray_ids[r];
      float o[3], d[3], v[2];
      orc_make_ray(look_at, focal_length, aspect_ratio, width, height, gid % width, gid / width, o, d, v);
      orc_sink s;
      s.start = sp; s.end = ep; s.t0 = 0; s.t1 = 0; s.seg_ray = 0; s.base = 0; s.cap = S; s.ray = 0; s.n = 0;
      if (trace_mode == 0) march_compat(o, d, R, occ, &s);
      else march_dda(o, d, R, occ, &s);
      const int nseg = s.n < S ? s.n : S;
      const float inc = 1.0f / ORC_K;
      for (int j0 = 0; j0 < nseg; j0 += 2) {          /* a tile: two segments = 64 samples (the second may be padding) */
        for (int q = 0; q < 64; ++q) {
          const int j = j0 + q / ORC_K < nseg ? j0 + q / ORC_K : j0, i = q % ORC_K;
          float t = 0.0f;
          for (int u = 0; u < i; ++u) t += inc;         /* the reference accumulates t by repeated += 1/32 (sampler.cu:65) */
          float in5[5];
          for (int a = 0; a < 3; ++a) in5[a] = fmaf(t, ep[3 * j + a] - sp[3 * j + a], sp[3 * j + a]);
          in5[3] = v[0]; in5[4] = v[1];
          float* e = ta + (long)q * ST;
          int f_idx = 0;
          for (int part = 0; part < 2; ++part) {
            const int nd = part == 0 ? cfg->n_pos_dims : cfg->n_dir_dims, F = part == 0 ? cfg->n_pos_freqs : cfg->n_dir_freqs;
            const int off = part == 0 ? 0 : cfg->n_pos_dims;
            for (int dim = 0; dim < nd; ++dim)
              for (int f = 0; f < F; ++f) {
                float tt = ldexpf(in5[off + dim], f);             /* exact */
                tt -= 2.0f * floorf(tt * 0.5f);                   /* exact: sin(pi t) has period 2 */
                e[f_idx++] = rh_hw(sinf((float)M_PI * tt));
                e[f_idx++] = rh_hw(sinf((float)M_PI * tt + (float)(M_PI / 2)));
              }
          }
          for (; f_idx < P; ++f_idx) e[f_idx] = 1.0f;
        }
        const float* w = wT;
        int in_w = P;
        float *a = ta, *b = tb;
        for (int l = 0; l < nh; ++l) {
          tiled_layer(w, in_w, W, a, ST, b, ST, 1);
          w += (long)W * in_w;
          in_w = W;
          float* t2 = a; a = b; b = t2;
        }
        tiled_layer(w, W, 16, a, ST, b, ST, 0);
        const int live = (nseg - j0 >= 2 ? 2 : 1) * ORC_K;
        for (int q = 0; q < live; ++q)
          for (int c = 0; c < 4; ++c) {
            const float z = b[(long)q * ST + c];
            rad[4 * ((long)j0 * ORC_K + q) + c] = rh_hw(cfg->output_activation == 1 ? 1.0f / (1.0f + expf(-z)) : z);
          }
      }
      float T = 0.0f, t_prev = 0.0f, acc3[3] = {0, 0, 0};
      for (int j = 0; j < nseg; ++j) {
        float t_initial = 0.0f;
        for (int i = 0; i < ORC_K; ++i) {
          t_initial += inc;
          const float* rd = rad + 4 * ((long)j * ORC_K + i);
          const float delta = fabsf(t_initial - t_prev);
          t_prev = t_initial;
          T = fmaf(delta, rd[3], T);
          const float wgt = expf(-T) * (1 - expf(-delta * rd[3]));
          acc3[0] += wgt * rd[0];
          acc3[1] += wgt * rd[1];
          acc3[2] += wgt * rd[2];
        }
      }
      tot += (long)nseg * ORC_K;
      pixels[3 * r] = acc3[0];
      pixels[3 * r + 1] = acc3[1];
      pixels[3 * r + 2] = acc3[2];
    }
    free(sp); free(ep); free(ta); free(tb); free(rad);
  }
  free(wT);
  if (total_samples) *total_samples = tot;
}

/* ------------------------------------------------------------------------- */
/* RTXN_VR_NERF (this build's corrected mode, not in the reference):          */
/* canonical quadrature, step = world-space length of each sample's interval, */
/*   C = sum_i T_i (1-exp(-x_i)) c_i, x_i = step_i sigma_i,                   */
/*   T_i = exp(-sum_{k<i} x_k); evaluated in double.                          */
/* ------------------------------------------------------------------------- */
void orc_volrender_fwd_nerf(const float* radiance, const int* num_hits, const int* indices,
                            const float* step, int batch_size, int K, float* pixels) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int x = 0; x < batch_size; ++x) {
    size_t base = (size_t)indices[x] * K, n = (size_t)num_hits[x] * K;
    double T = 0.0, acc[3] = {0, 0, 0};
    for (size_t s = 0; s < n; ++s) {
      const float* c = radiance + 4 * (base + s);
      double xs = (double)step[base + s] * c[3];
      double w = exp(-T) * (1.0 - exp(-xs));
      acc[0] += w * c[0]; acc[1] += w * c[1]; acc[2] += w * c[2];
      T += xs;
    }
    pixels[3 * x] = (float)acc[0]; pixels[3 * x + 1] = (float)acc[1]; pixels[3 * x + 2] = (float)acc[2];
  }
}

/* exact gradient of the above w.r.t. (r,g,b,sigma) per sample, contracted with
 * dL/dpixel g (fp16 in, fp32 out so the test can state the fp16 rounding). */
void orc_volrender_bwd_nerf(const uint16_t* loss_gradients, const float* radiance, const float* step,
                            const int* num_hits, const int* indices, int batch_size, int K,
                            float* grads /* [N][4] fp32 */) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int x = 0; x < batch_size; ++x) {
    size_t base = (size_t)indices[x] * K, n = (size_t)num_hits[x] * K;
    double g[3] = {orc_f16_bits_to_f32(loss_gradients[3 * x]), orc_f16_bits_to_f32(loss_gradients[3 * x + 1]),
                   orc_f16_bits_to_f32(loss_gradients[3 * x + 2])};
    double S = 0.0, T = 0.0;
    for (size_t s = 0; s < n; ++s) {
      const float* c = radiance + 4 * (base + s);
      double xs = (double)step[base + s] * c[3];
      S += exp(-T) * (1.0 - exp(-xs)) * (g[0] * c[0] + g[1] * c[1] + g[2] * c[2]);
      T += xs;
    }
    double P = 0.0;
    T = 0.0;
    for (size_t s = 0; s < n; ++s) {
      const float* c = radiance + 4 * (base + s);
      double d = step[base + s], xs = d * c[3];
      double Ti = exp(-T), ex = exp(-xs), a = 1.0 - ex;
      double gc = g[0] * c[0] + g[1] * c[1] + g[2] * c[2];
      P += Ti * a * gc;
      float* o = grads + 4 * (base + s);
      o[0] = (float)(g[0] * Ti * a); o[1] = (float)(g[1] * Ti * a); o[2] = (float)(g[2] * Ti * a);
      o[3] = (float)(d * (Ti * ex * gc - (S - P)));
      T += xs;
    }
  }
}

/* ========================================================================= */
/* TRAINING PATH                                                              */
/* Reference call sites: loss->evaluate (main.cu:759), network->backward       */
/* (:781), optimizer->step (:787), all inside tiny-cuda-nn (un-vendored,       */
/* unpinned): restated from the published algorithms.  PARITY UNPINNED.        */
/* ========================================================================= */

/* ---- multiresolution hash grid (Mueller et al. 2022; tcnn "HashGrid") ---- */
typedef struct {
  int n_levels, n_features, log2_hashmap_size, base_resolution;
  float per_level_scale;
} orc_hg_cfg;

static void hg_level(const orc_hg_cfg* c, int l, float* scale, unsigned* res, unsigned* size, unsigned* offset) {
  /* scale_l = base * per_level_scale^l - 1 ; res_l = ceil(scale_l) + 1 ;
   * params_l = min(res_l^3 rounded up to 8, 2^log2_hashmap_size) */
  unsigned off = 0;
  for (int i = 0; i <= l; ++i) {
    float s = exp2f((float)i * log2f(c->per_level_scale)) * (float)c->base_resolution - 1.0f;
    unsigned r = (unsigned)ceilf(s) + 1u;
    unsigned long long dense = (unsigned long long)r * r * r;
    dense = (dense + 7ull) / 8ull * 8ull;
    unsigned long long cap = 1ull << c->log2_hashmap_size;
    unsigned sz = (unsigned)(dense < cap ? dense : cap);
    if (i == l) { *scale = s; *res = r; *size = sz; *offset = off; }
    off += sz;
  }
}

long orc_hg_n_params(const orc_hg_cfg* c) {
  float s = 0; unsigned r = 0, sz = 0, off = 0;
  hg_level(c, c->n_levels - 1, &s, &r, &sz, &off);
  return ((long)off + sz) * c->n_features;
}

static inline unsigned hg_index(unsigned x, unsigned y, unsigned z, unsigned res, unsigned size) {
  unsigned long long dense = (unsigned long long)res * res * res;
  if (dense <= size) return (x + y * res + z * res * res) % size; /* dense level; the +1 corner of a boundary cell wraps (tcnn: index % hashmap_size) */
  return ((x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u)) % size; /* coherent prime hash */
}

/* xyz01: position in [0,1]^3.  table: fp16 bits [n_params].  out: n_levels*n_features floats (fp16-rounded). */
void orc_hg_encode_one(const orc_hg_cfg* c, const uint16_t* table, const float* xyz01, float* out) {
  for (int l = 0; l < c->n_levels; ++l) {
    float scale = 0; unsigned res = 0, size = 0, off = 0;
    hg_level(c, l, &scale, &res, &size, &off);
    float pos[3], fr[3]; unsigned g[3];
    for (int a = 0; a < 3; ++a) {
      pos[a] = fmaf(xyz01[a], scale, 0.5f);
      float fl = floorf(pos[a]);
      g[a] = (unsigned)(int)fl;
      fr[a] = pos[a] - fl;
    }
    for (int f = 0; f < c->n_features; ++f) {
      float acc = 0.0f;
      for (int corner = 0; corner < 8; ++corner) {
        float w = 1.0f; unsigned p[3];
        for (int a = 0; a < 3; ++a) {
          int hi = (corner >> a) & 1;
          w *= hi ? fr[a] : 1.0f - fr[a];
          p[a] = g[a] + (unsigned)hi;
        }
        unsigned idx = hg_index(p[0], p[1], p[2], res, size);
        acc = fmaf(w, orc_f16_bits_to_f32(table[((size_t)off + idx) * c->n_features + f]), acc);
      }
      out[l * c->n_features + f] = rh(acc);
    }
  }
}

/* Composite training encoding: HashGrid(xyz mapped from [-1,1] to [0,1]) (+) Frequency(dirs, n_dir_freqs),
 * padded with ones to a multiple of 16.  in: [S][5]; out: fp16 bits [S][E]. */
int orc_enc_hg_width(const orc_hg_cfg* c, int n_dir_freqs) {
  int w = c->n_levels * c->n_features + 2 * 2 * n_dir_freqs;
  return (w + 15) / 16 * 16;
}

void orc_encode_hg(const orc_hg_cfg* c, int n_dir_freqs, const uint16_t* table, const float* in5, long S, uint16_t* out) {
  int E = orc_enc_hg_width(c, n_dir_freqs), nh = c->n_levels * c->n_features;
#pragma omp parallel for schedule(static)
  for (long s = 0; s < S; ++s) {
    float tmp[64];
    float x01[3] = {fmaf(in5[5 * s], 0.5f, 0.5f), fmaf(in5[5 * s + 1], 0.5f, 0.5f), fmaf(in5[5 * s + 2], 0.5f, 0.5f)};
    orc_hg_encode_one(c, table, x01, tmp);
    uint16_t* o = out + s * E;
    int j = 0;
    for (; j < nh; ++j) o[j] = orc_f32_to_f16_bits(tmp[j]);
    for (int d = 0; d < 2; ++d)
      for (int f = 0; f < n_dir_freqs; ++f)
        for (int ph = 0; ph < 2; ++ph) {
          double arg = M_PI * ldexp((double)in5[5 * s + 3 + d], f) + ph * (M_PI / 2);
          o[j++] = orc_f32_to_f16_bits((float)sin(arg));
        }
    for (; j < E; ++j) o[j] = orc_f32_to_f16_bits(1.0f);
  }
}

/* hash-grid backward: dtable[idx][f] += w * denc[s][l*F+f]; double accumulation -> float out */
void orc_hg_backward(const orc_hg_cfg* c, const float* in5, long S, const uint16_t* denc, int E, float* dtable) {
  long np_ = orc_hg_n_params(c);
  double* acc = (double*)calloc((size_t)np_, sizeof(double));
  for (long s = 0; s < S; ++s) {
    float x01[3] = {fmaf(in5[5 * s], 0.5f, 0.5f), fmaf(in5[5 * s + 1], 0.5f, 0.5f), fmaf(in5[5 * s + 2], 0.5f, 0.5f)};
    for (int l = 0; l < c->n_levels; ++l) {
      float scale = 0; unsigned res = 0, size = 0, off = 0;
      hg_level(c, l, &scale, &res, &size, &off);
      float fr[3]; unsigned g[3];
      for (int a = 0; a < 3; ++a) {
        float p = fmaf(x01[a], scale, 0.5f), fl = floorf(p);
        g[a] = (unsigned)(int)fl; fr[a] = p - fl;
      }
      for (int corner = 0; corner < 8; ++corner) {
        float w = 1.0f; unsigned p[3];
        for (int a = 0; a < 3; ++a) { int hi = (corner >> a) & 1; w *= hi ? fr[a] : 1.0f - fr[a]; p[a] = g[a] + (unsigned)hi; }
        unsigned idx = hg_index(p[0], p[1], p[2], res, size);
        for (int f = 0; f < c->n_features; ++f)
          acc[((size_t)off + idx) * c->n_features + f] += (double)w * orc_f16_bits_to_f32(denc[s * E + l * c->n_features + f]);
      }
    }
  }
  for (long i = 0; i < np_; ++i) dtable[i] = (float)acc[i];
  free(acc);
}

/* Frequency composite as a standalone encoder: [S][5] -> fp16 bits [S][enc_padded] */
void orc_encode_freq(const orc_mlp_cfg* c, const float* in5, long S, uint16_t* out) {
  int P = orc_mlp_enc_padded(c);
#pragma omp parallel for schedule(static)
  for (long s = 0; s < S; ++s) {
    float tmp[256];
    orc_freq_encode(c, in5 + 5 * s, tmp);
    for (int j = 0; j < P; ++j) out[s * P + j] = orc_f32_to_f16_bits(tmp[j]);
  }
}

/* ---- MLP on pre-encoded input, with saved activations ---- */
/* enc: fp16 bits [S][E]; params: tcnn layout with first layer [W][E]; acts: fp16 bits [L][S][W]
 * (post-ReLU, sample-major here; the HIP side keeps its own layout); out: fp16 bits [S][16]. */
long orc_mlpe_n_params(int W, int L, int E) { return (long)W * E + (long)(L - 1) * W * W + 16L * W; }

void orc_mlpe_forward(int W, int L, int E, int out_act, const uint16_t* params, const uint16_t* enc, long S,
                      uint16_t* acts, uint16_t* out) {
  long np_ = orc_mlpe_n_params(W, L, E);
  float* wf = (float*)malloc(sizeof(float) * np_);
  for (long i = 0; i < np_; ++i) wf[i] = orc_f16_bits_to_f32(params[i]);
#pragma omp parallel for schedule(static)
  for (long s = 0; s < S; ++s) {
    float a[256], b[256];
    for (int k = 0; k < E; ++k) a[k] = orc_f16_bits_to_f32(enc[s * E + k]);
    const float* w = wf;
    int in_w = E;
    float *pa = a, *pb = b;
    for (int l = 0; l < L; ++l) {
      for (int r = 0; r < W; ++r) {
        float acc = 0.0f;
        for (int k = 0; k < in_w; ++k) acc = fmaf(w[(long)r * in_w + k], pa[k], acc);
        pb[r] = rh(acc > 0.0f ? acc : 0.0f);
        if (acts) acts[((long)l * S + s) * W + r] = orc_f32_to_f16_bits(pb[r]);
      }
      w += (long)W * in_w; in_w = W;
      float* t = pa; pa = pb; pb = t;
    }
    for (int r = 0; r < 16; ++r) {
      float acc = 0.0f;
      for (int k = 0; k < W; ++k) acc = fmaf(w[(long)r * W + k], pa[k], acc);
      float y = out_act == 1 ? 1.0f / (1.0f + expf(-acc)) : acc;
      out[s * 16 + r] = orc_f32_to_f16_bits(y);
    }
  }
  free(wf);
}

/* Backward.  dout: fp16 bits [S][4] (rows 4..15 of the output carry no gradient), the layout
 * launch_volrender_backward_cuda writes (vol_render.cu:136-139).  dparams: fp32 [n_params] (double
 * accumulation); denc: fp32 [S][E] or NULL.  Intermediate gradients are rounded to fp16 between layers
 * as the fp16 backward pass does (dZ_l stored in half). */
void orc_mlpe_backward(int W, int L, int E, int out_act, const uint16_t* params, const uint16_t* enc,
                       const uint16_t* acts, const uint16_t* out, const uint16_t* dout, long S, float* dparams,
                       float* denc) {
  long np_ = orc_mlpe_n_params(W, L, E);
  float* wf = (float*)malloc(sizeof(float) * np_);
  for (long i = 0; i < np_; ++i) wf[i] = orc_f16_bits_to_f32(params[i]);
  double* dp = (double*)calloc((size_t)np_, sizeof(double));
  long* loff = (long*)malloc(sizeof(long) * (L + 1));
  loff[0] = 0;
  for (int l = 1; l <= L; ++l) loff[l] = loff[l - 1] + (long)W * (l == 1 ? E : W);
  for (long s = 0; s < S; ++s) {
    float dz[256], da[256];
    /* output layer */
    float dzo[16];
    for (int r = 0; r < 16; ++r) {
      float g = r < 4 ? orc_f16_bits_to_f32(dout[s * 4 + r]) : 0.0f;
      if (out_act == 1) { float y = orc_f16_bits_to_f32(out[s * 16 + r]); g = g * y * (1.0f - y); }
      dzo[r] = rh(g);
    }
    const uint16_t* aprev = acts + ((long)(L - 1) * S + s) * W;
    for (int r = 0; r < 16; ++r)
      for (int k = 0; k < W; ++k) dp[loff[L] + (long)r * W + k] += (double)dzo[r] * orc_f16_bits_to_f32(aprev[k]);
    for (int k = 0; k < W; ++k) {
      float acc = 0.0f;
      for (int r = 0; r < 16; ++r) acc = fmaf(wf[loff[L] + (long)r * W + k], dzo[r], acc);
      da[k] = acc;
    }
    for (int l = L - 1; l >= 0; --l) {
      const uint16_t* al = acts + ((long)l * S + s) * W;
      for (int r = 0; r < W; ++r) dz[r] = rh(orc_f16_bits_to_f32(al[r]) > 0.0f ? da[r] : 0.0f);
      int in_w = l == 0 ? E : W;
      for (int r = 0; r < W; ++r) {
        if (dz[r] == 0.0f) continue;
        for (int k = 0; k < in_w; ++k) {
          float x = l == 0 ? orc_f16_bits_to_f32(enc[s * E + k]) : orc_f16_bits_to_f32(acts[((long)(l - 1) * S + s) * W + k]);
          dp[loff[l] + (long)r * in_w + k] += (double)dz[r] * x;
        }
      }
      if (l > 0 || denc) {
        for (int k = 0; k < in_w; ++k) {
          float acc = 0.0f;
          for (int r = 0; r < W; ++r) acc = fmaf(wf[loff[l] + (long)r * in_w + k], dz[r], acc);
          if (l > 0) da[k] = acc; else denc[s * E + k] = acc;
        }
      }
    }
  }
  for (long i = 0; i < np_; ++i) dparams[i] = (float)dp[i];
  free(dp); free(wf); free(loff);
}

/* ---- L2 loss (tcnn "L2"; main.cu:36-38,759): values = d^2/n, grads = scale*2d/n, n = B*3 ---- */
double orc_l2_loss(const float* pred, const float* target, long n, float scale, float* values, uint16_t* grads_f16,
                   float* grads_f32) {
  double sum = 0.0;
  for (long i = 0; i < n; ++i) {
    float d = pred[i] - target[i];
    float v = d * d / (float)n;
    float g = scale * 2.0f * d / (float)n;
    if (values) values[i] = v;
    if (grads_f16) grads_f16[i] = orc_f32_to_f16_bits(g);
    if (grads_f32) grads_f32[i] = g;
    sum += v;
  }
  return sum;
}

/* ---- Adam (tcnn "Adam"; main.cu:40-46,787) ---- */
void orc_adam_step(long n, float* master, uint16_t* params_f16, const float* grads, float* m, float* v, int step,
                   float lr, float beta1, float beta2, float eps, float loss_scale) {
  float lr_eff = lr * sqrtf(1.0f - powf(beta2, (float)step)) / (1.0f - powf(beta1, (float)step));
  for (long i = 0; i < n; ++i) {
    float g = grads[i] / loss_scale;
    m[i] = beta1 * m[i] + (1.0f - beta1) * g;
    v[i] = beta2 * v[i] + (1.0f - beta2) * g * g;
    master[i] -= lr_eff * m[i] / (sqrtf(v[i]) + eps);
    params_f16[i] = orc_f32_to_f16_bits(master[i]);
  }
}

/* tiny-cuda-nn's adam_step for its NON-MATRIX parameters (the hash table; optimizers/adam.h [upstream], the optimizer the
 * reference configures at main.cu:36-46 and steps at :787): `if (gradient == 0) return;`, and the bias correction uses the
 * parameter's own update count, `current_step = ++param_steps[i]`. */
void orc_adam_step_sparse(long n, float* master, uint16_t* params_f16, const float* grads, float* m, float* v, uint32_t* steps,
                          float lr, float beta1, float beta2, float eps, float loss_scale) {
  for (long i = 0; i < n; ++i) {
    float g = grads[i] / loss_scale;
    if (g == 0.0f) continue;
    m[i] = beta1 * m[i] + (1.0f - beta1) * g;
    v[i] = beta2 * v[i] + (1.0f - beta2) * g * g;
    float t = (float)(++steps[i]);
    float lr_eff = lr * sqrtf(1.0f - powf(beta2, t)) / (1.0f - powf(beta1, t));
    master[i] -= lr_eff * m[i] / (sqrtf(v[i]) + eps);
    params_f16[i] = orc_f32_to_f16_bits(master[i]);
  }
}
