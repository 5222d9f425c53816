/*
 * rtxn.h -- C ABI of librtxn.so, the MI355X (gfx950) implementation of the
 * owensgroup/rtx_nerf hot path: ray generation + ray/grid traversal, CSR
 * compaction, per-segment sampling, frequency-encoded fully-fused MLP and
 * alpha-compositing volume rendering (forward and backward).
 *
 * The reference has no FFI layer; its "operator API" is a handful of C++ free
 * functions and PODs called from main.cu.  Each entry point below names the
 * reference interface it replaces (paths relative to the reference root).  The
 * C++ drop-in headers in include/rtxn_dropin/ (sampler/sampler.h, vol_render/vol_render.h,
 * rtx/include/params.h, loader/data_loader.h) keep the reference's own names and argument order and forward
 * to these symbols.  INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - the caller owns every buffer; the library allocates nothing per call
 *     (rtxn_mlp_create owns its packed-weight buffer until rtxn_mlp_destroy);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all
 *     work is enqueued asynchronously on it, nothing synchronises the host, so
 *     every call is hipGraph-capturable;
 *   - return value: RTXN_OK or an error code; rtxn_last_error() (host,
 *     thread-local) describes the last failure.  The reference's functions
 *     return void and print-and-continue (common/common.h:38-50).
 *   - there is no CPU fallback: without a HIP device every compute entry
 *     point fails with RTXN_ERR_HIP.
 */
#ifndef RTXN_H
#define RTXN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTXN_VERSION 100

typedef void* rtxn_stream_t;

enum rtxn_status {
  RTXN_OK = 0,
  RTXN_ERR_INVALID = 1,     /* bad argument (null pointer, negative size, unsupported width) */
  RTXN_ERR_HIP = 2,         /* a HIP runtime call or kernel launch failed */
  RTXN_ERR_UNSUPPORTED = 3, /* valid request this build does not implement */
  RTXN_ERR_IO = 4           /* loader: missing/invalid file */
};

int rtxn_version(void);
const char* rtxn_last_error(void);

/* ---- sampler/sampler.h:4-9 ------------------------------------------------ */
#define RTXN_NUM_SAMPLES_PER_SEGMENT 32
enum rtxn_sampling_type {
  RTXN_SAMPLING_REGULAR = 0,
  RTXN_SAMPLING_STRATIFIED_JITTERING = 1,
  RTXN_SAMPLING_UNIFORM = 2,
  /* not in the reference: sample i at the MIDPOINT t = (i+0.5)/32 of its sub-interval, and t_vals = the
   * sub-interval's world-space length |end-start|/32 -- the inputs RTXN_VR_NERF expects */
  RTXN_SAMPLING_MIDPOINT_WORLD = 3
};

/* ---- traversal ------------------------------------------------------------ */
/* Replaces struct Params (rtx/include/params.h:14-42), the RTXDataHolder
 * lifecycle (rtx/include/rtxFunctions.h:56-100: initContext .. buildSBT,
 * initAccelerationStructure) and optixLaunch(pipeline_ray_march, ..., W, H, 1)
 * (main.cu:506-508).  There is no acceleration structure to build: the grid of
 * make_grid (main.cu:154-174), R^3 cells over [-1,1]^3, is walked analytically;
 * `handle`, `aabb` and `num_primitives` have no counterpart. */
enum rtxn_trace_mode {
  RTXN_TRACE_COMPAT = 0, /* reference arithmetic: re-launch from each exit point (optixPrograms.cu:99-115,180-248) */
  RTXN_TRACE_DDA = 1     /* global-t 3D-DDA with hierarchical empty-space skipping; same cells, points within 1e-5 */
};

typedef struct rtxn_trace_params {
  /* rays: either a pinhole camera (look_at != NULL; params.h:17,33-34,41) ... */
  const float* look_at;      /* 16 floats, row-major 4x4, translation at [3],[7],[11] */
  float focal_length;        /* params.h:33 */
  float aspect_ratio;        /* params.h:34 */
  uint32_t width, height;    /* params.h:41 */
  /* ... or explicit rays (look_at == NULL): float3[width*height] each, d normalised */
  const float* rays_o;
  const float* rays_d;
  /* window of the launch handled by this call: rays [ray_begin, ray_begin+ray_count)
   * of the width*height launch (row-major, ray = x + y*width, optixPrograms.cu:45).
   * Outputs are indexed by the LOCAL ray number (ray - ray_begin).  This is how
   * a launch is sharded across GPUs. */
  uint32_t ray_begin, ray_count;
  /* optional interleave: local ray i maps to launch ray
   * ray_begin + (i / window_chunk) * window_stride + (i % window_chunk); 0,0 = contiguous.
   * (chunk = width, stride = n_gpus*width gives GPU g every n_gpus-th image row.) */
  uint32_t window_chunk, window_stride;
  /* grid */
  int grid_res;              /* R; cell = 2/R (main.cu:156,482) */
  const uint32_t* occupancy; /* R^3 bits, bit ((x*R+y)*R+z); NULL = dense (reference) */
  const uint32_t* occupancy_coarse; /* (R/4)^3 bits, OR of 4^3 blocks; NULL = none (from rtxn_build_occupancy_mip) */
  int mode;                  /* enum rtxn_trace_mode */
  /* outputs */
  float* ray_origins;        /* float3[ray_count]  (params.h:25), may be NULL */
  float* viewing_direction;  /* float2[ray_count]  (params.h:32), may be NULL */
  int* num_hits;             /* int[ray_count]     (params.h:31) */
  /* segment outputs; any may be NULL.  Layout: strided (reference) when
   * indices == NULL: slot = local_ray*intersection_arr_size + k
   * (optixPrograms.cu:184, main.cu:486); packed CSR when indices != NULL:
   * slot = indices[local_ray] + k (the layout main.cu:646-673 builds on the
   * host for the sampler). */
  int intersection_arr_size; /* S (params.h:19); segments beyond S are counted, not stored */
  const int* indices;
  float* start_points;       /* float3 per slot (params.h:23) */
  float* end_points;         /* float3 per slot (params.h:24) */
  float* t_start;            /* float per slot  (params.h:28) */
  float* t_end;              /* float per slot  (params.h:29) */
  int* seg_ray;              /* int per slot: local ray of the segment (packed layout; new) */
  float* seg_view;           /* float2 per slot: (theta, phi) of the segment's ray (packed layout; new) */
  long segment_capacity;     /* packed layout: slots >= capacity are not written (0 = unbounded) */
  uint8_t* seg_first;        /* byte per slot: 1 if the segment is the first of its ray (packed layout; new) */
  const uint64_t* occupancy_bricks; /* (R/4)^3 words from rtxn_build_occupancy_bricks, or NULL: the 64 fine bits of each 4^3 block
                                     * (RTXN_TRACE_DDA + occupancy_coarse only; same segments, one load per block instead of per cell) */
  const uint32_t* occupancy_super;  /* (R/16)^3 bits = rtxn_build_occupancy_mip(occupancy_coarse, R/4, .), or NULL: third level of the
                                     * empty-space hierarchy (needs occupancy_coarse, R % 16 == 0) */
  int* num_stored;           /* int[ray_count] or NULL: segments actually WRITTEN for the ray (< num_hits when
                              * intersection_arr_size / segment_capacity cut it off): the count downstream stages may read */
  int sub_rays;              /* RTXN_TRACE_DDA only; 0 or 1: one thread walks a ray.  2, 4, ... 64 (a power of two): that many adjacent lanes walk
                              * consecutive pieces of the ray's parameter range (same segments, same order, bit for bit) -- for
                              * small batches, where the launch takes as long as the longest ray's walk */
  int* sub_hits;             /* int[ray_count * sub_rays] scratch, required when sub_rays > 1: written by the counting pass,
                              * read by the write pass of the same launch geometry */
} rtxn_trace_params;

/* One launch: ray generation + grid march.  With every segment pointer NULL it
 * is the counting pass of the two-pass packed pipeline (count -> rtxn_scan_hits
 * -> write). */
int rtxn_trace_grid(const rtxn_trace_params* p, rtxn_stream_t stream);

/* Coarse occupancy mip for RTXN_TRACE_DDA: bit of coarse cell (X,Y,Z) of a
 * (R/4)^3 grid = OR over its 4^3 fine cells.  R must be a multiple of 4.
 * coarse: (R/4)^3 bits rounded up to whole uint32 words. */
int rtxn_build_occupancy_mip(const uint32_t* occupancy, int grid_res, uint32_t* coarse, rtxn_stream_t stream);

/* Brick copy of the occupancy for RTXN_TRACE_DDA: bricks[(X*Rc+Y)*Rc+Z] (Rc = R/4) holds the 64 fine bits of block
 * (X,Y,Z), bit ((x&3)<<4 | (y&3)<<2 | (z&3)). */
int rtxn_build_occupancy_bricks(const uint32_t* occupancy, int grid_res, uint64_t* bricks, rtxn_stream_t stream);

/* Occupancy maintenance (the reference only builds the dense grid once,
 * main.cu:393-399): occupancy bit of cell i = density[i] > threshold, i = (x*R+y)*R+z.
 * occupancy: ceil(R^3/32) words. */
int rtxn_occupancy_from_density(const float* density, float threshold, int grid_res, uint32_t* occupancy,
                                rtxn_stream_t stream);

/* ---- CSR compaction -------------------------------------------------------- */
/* Replaces thrust::reduce + thrust::exclusive_scan over num_hits
 * (main.cu:631-637).  indices[i] = sum_{j<i} num_hits[j]; *total = sum of all.
 * `total` is a device int: nothing is copied to the host.  workspace: at least
 * rtxn_scan_workspace_bytes(n) bytes of device scratch. */
size_t rtxn_scan_workspace_bytes(int n);
int rtxn_scan_hits(const int* num_hits, int* indices, int* total, int n, void* workspace,
                   size_t workspace_bytes, rtxn_stream_t stream);

/* ---- sampler ---------------------------------------------------------------- */
/* Replaces launchSampler (sampler/sampler.h:19-30, sampler/sampler.cu:105-131).
 * start/end: float3[P] packed; view_dirs: float2[B]; t_vals: float[P*32];
 * sampled_points: float[P*32*5] AoS (x,y,z,theta,phi); num_hits, indices:
 * int[B].  grid_res is accepted and ignored, as in the reference. */
int rtxn_sample(const float* start_points, const float* end_points, const float* view_dirs,
                float* t_vals, float* sampled_points, int batch_size, int grid_res,
                const int* num_hits, const int* indices, int sample_type, rtxn_stream_t stream);

/* ---- volume rendering --------------------------------------------------------- */
enum rtxn_volrender_mode {
  RTXN_VR_COMPAT = 0,   /* reference arithmetic, vol_render.cu:19-143, quirks included */
  RTXN_VR_NERF = 1      /* canonical NeRF quadrature: exclusive transmittance, exact analytic backward */
};

/* Replaces launch_volrender_cuda (vol_render/vol_render.h:5-13,
 * vol_render.cu:144-164).  network_inputs is accepted and ignored (the
 * reference never reads it).  network_outputs: float[P*K*4] AoS (r,g,b,sigma);
 * ray_hit: float[P*K] (the sampler's t_vals); pixels: float[B*3]. */
int rtxn_volrender_fwd(const float* network_inputs, const float* network_outputs, const int* num_hits,
                       const int* indices, const float* ray_hit, int batch_size,
                       int num_samples_per_hit, float* pixels, int mode, rtxn_stream_t stream);

/* Replaces launch_volrender_backward_cuda (vol_render/vol_render.h:15-25,
 * vol_render.cu:166-190).  loss_values is accepted and ignored.  loss_gradients:
 * half[B*3]; radiance_gradients: half[P*K*4] (stride 4, as the reference
 * writes them, vol_render.cu:108,136-139). */
int rtxn_volrender_bwd(const float* loss_values, const void* loss_gradients,
                       const float* sampled_points_radiance, const float* t_hit, const int* num_hits,
                       const int* indices, int batch_size, int num_samples_per_hit,
                       void* radiance_gradients, int mode, rtxn_stream_t stream);

/* launch_volrender_cuda + loss->evaluate (L2) + launch_volrender_backward_cuda of a training batch (main.cu:737-767) in one
 * launch, RTXN_VR_NERF arithmetic: pixels float[B*3], loss_gradients_half half[B*3] (may be NULL), *loss_sum (device float,
 * may be NULL) = sum of (pixel - target)^2 / (3B), radiance_gradients half[P*K*4].  Same values as the three entry points
 * called one after the other up to fp32 rounding of one dot product. */
int rtxn_volrender_l2_train(const float* network_outputs, const float* ray_hit, const int* num_hits, const int* indices,
                            int batch_size, int num_samples_per_hit, const float* target, float loss_scale, float* pixels,
                            void* loss_gradients_half, float* loss_sum, void* radiance_gradients, rtxn_stream_t stream);

/* ---- MLP (tiny-cuda-nn surface used by main.cu) ------------------------------- */
/* Replaces tcnn::create_from_config(5, 4, config) (main.cu:35-69,325),
 * network->n_params / set_params / initialize_params (main.cu:327-349),
 * network->forward (main.cu:721) and the convertHalfToFloat glue
 * (main.cu:203-208,723-728).  Model: Composite(Frequency(n_pos_dims,
 * n_pos_freqs), Frequency(n_dir_dims, n_dir_freqs)) -> n_hidden_layers x
 * n_neurons ReLU -> 16 (n_output_dims used), fp16 weights, no biases. */
enum rtxn_activation { RTXN_ACT_NONE = 0, RTXN_ACT_SIGMOID = 1 };
enum rtxn_encoding {
  RTXN_ENC_FREQUENCY = 0, /* Composite(Frequency, Frequency) computed inside the kernels (main.cu:47-61) */
  RTXN_ENC_EXTERNAL = 1   /* the caller encodes (e.g. rtxn_hashgrid_encode); only the rtxn_mlp_train_* entry points apply */
};

typedef struct rtxn_mlp_config {
  int n_pos_dims, n_pos_freqs;   /* 3, 10 (main.cu:52-54) */
  int n_dir_dims, n_dir_freqs;   /* 2, 12 (main.cu:57-59: "n_bins" is not a Frequency key -> default 12) */
  int n_neurons;                 /* 64 or 128 (main.cu:66) */
  int n_hidden_layers;           /* >= 1 (main.cu:67) */
  int n_output_dims;             /* <= 16 (main.cu:323) */
  int output_activation;         /* enum rtxn_activation (main.cu:65) */
  int encoding;                  /* enum rtxn_encoding; 0 = the reference's Composite-Frequency */
  int n_encoded_features;        /* RTXN_ENC_EXTERNAL only: width of the pre-encoded input, multiple of 16 */
} rtxn_mlp_config;

typedef struct rtxn_mlp rtxn_mlp;
typedef struct rtxn_hashgrid rtxn_hashgrid;   /* multiresolution hash grid: rtxn_hashgrid_create, below */

int rtxn_mlp_create(const rtxn_mlp_config* cfg, rtxn_mlp** out);
int rtxn_mlp_destroy(rtxn_mlp* m);
/* number of fp16 parameters, tcnn layout: per layer a row-major [out][in]
 * matrix (first: n_neurons x enc_padded; hidden: n_neurons^2; last: 16 x
 * n_neurons), layers concatenated. */
/* The fused inference kernels run a persistent grid that fills every CU.  Work launched beside them on other streams
 * co-resides only if it fits the CU's left-over registers/LDS (this library's traversal, scan and compositor kernels do);
 * a collective library's kernels may not.  n_cus > 0 keeps that many CUs free of MLP blocks (default 0). */
int rtxn_mlp_set_reserved_cus(rtxn_mlp* m, int n_cus);
/* MFMA shape of the model's fused inference kernel: 16 = v_mfma_f32_16x16x32_f16 (every Composite-Frequency model);
 * 0 = no fused inference kernel (RTXN_ENC_EXTERNAL: use rtxn_hashmlp_forward_segments or the rtxn_mlp_train_* entry points). */
int rtxn_mlp_mfma_shape(const rtxn_mlp* m);
long rtxn_mlp_n_params(const rtxn_mlp* m);
int rtxn_mlp_padded_output_width(const rtxn_mlp* m); /* 16 (main.cu:715) */
int rtxn_mlp_encoded_width(const rtxn_mlp* m);       /* padded to a multiple of 16 */
/* Xavier-uniform init from a PCG32 stream into HOST fp32 (main.cu:344-349). */
int rtxn_mlp_initialize_params(const rtxn_mlp* m, uint64_t seed, float* host_params_fp32);
/* Point the model at device fp16 params (tcnn layout); re-packs them into the
 * MFMA fragment order the kernels read.  Call again after every update. */
int rtxn_mlp_set_params(rtxn_mlp* m, const void* params_fp16, rtxn_stream_t stream);
/* The same for a training loop's per-step update: re-packs what the rtxn_mlp_train_* entry points read and leaves the fused
 * inference kernels' copy as it was -- call rtxn_mlp_set_params before rendering with the rtxn_mlp_forward* family again. */
int rtxn_mlp_set_params_training(rtxn_mlp* m, const void* params_fp16, rtxn_stream_t stream);
/* network->forward: input float[N*5] (column-major 5xN = the sampler's AoS),
 * output half[N*16] (column-major 16xN). */
int rtxn_mlp_forward(const rtxn_mlp* m, const float* input, void* output_half, long n, rtxn_stream_t stream);
/* forward fused with the fp16->fp32 radiance glue: radiance float[N*4]. */
int rtxn_mlp_forward_radiance(const rtxn_mlp* m, const float* input, float* radiance, long n,
                              rtxn_stream_t stream);
/* Sampler (REGULAR) + encoding + MLP + glue fused: reads packed segments,
 * never materialises the 5-float samples.  seg_view: float2 per segment, the
 * (theta, phi) of its ray as written by rtxn_trace_grid (packed layout), so a
 * segment's inputs are three independent 12/12/8-byte records.
 * *total_segments is the device int written by rtxn_scan_hits; the launch is
 * sized by max_segments (capacity of the caller's buffers) and exits early
 * beyond *total_segments.
 * radiance: float[max_segments*32*4]; t_vals: float[max_segments*32] or NULL. */
int rtxn_mlp_forward_segments(const rtxn_mlp* m, const float* start_points, const float* end_points,
                              const float* seg_view, const int* total_segments, long max_segments,
                              float* radiance, float* t_vals, rtxn_stream_t stream);

/* Compact form of the two calls above for the render pipeline: the kernel stores the network's own four half outputs per
 * sample (half[N][4], 8 B instead of the 16-B float4 of convertHalfToFloat, main.cu:203-208) and no t_vals -- REGULAR
 * sampling makes them (i + 1)/32 of the sample index -- and rtxn_volrender_fwd_compact consumes exactly that.  Pixels are
 * bit-identical to rtxn_mlp_forward_segments + rtxn_volrender_fwd(RTXN_VR_COMPAT) at 40 % of the HBM traffic. */
int rtxn_mlp_forward_segments_compact(const rtxn_mlp* m, const float* start_points, const float* end_points,
                                      const float* seg_view, const int* total_segments, long max_segments,
                                      void* radiance_half4, rtxn_stream_t stream);
int rtxn_volrender_fwd_compact(const void* radiance_half4, const int* num_hits, const int* indices, int batch_size,
                               int num_samples_per_hit, float* pixels, rtxn_stream_t stream);
/* The same hand-over for RTXN_VR_NERF (midpoint samples, exclusive transmittance): every sample of a segment has the same
 * world-space step |end - start| / K (x density scale), so the compositor takes ONE float per segment (segment_step[P], written
 * by rtxn_hashmlp_forward_segments) instead of launch_volrender_cuda's ray_hit[P*K].  Pixels bit-identical to
 * rtxn_volrender_fwd(RTXN_VR_NERF) on the widened radiance and the expanded steps. */
int rtxn_volrender_fwd_compact_nerf(const void* radiance_half4, const float* segment_step, const int* num_hits,
                                    const int* indices, int batch_size, int num_samples_per_hit, float* pixels,
                                    rtxn_stream_t stream);

/* ---- hash-grid inference -------------------------------------------------------------------------------------------
 * launchSampler + HashGrid(position) (+) Frequency(direction) encoding + network->forward + the half outputs' glue
 * (main.cu:703-728 with the hash-grid model north_star names) as ONE kernel over packed segments: the hash-grid counterpart of
 * rtxn_mlp_forward_segments_compact.  `m` is the pre-encoded (RTXN_ENC_EXTERNAL) model a training loop trains with
 * rtxn_hashgrid_encode_segments + rtxn_mlp_train_forward_outputs; this entry point computes the SAME values as that staged
 * pair in one pass, without the half[E][S] encoding in memory.  sample_type RTXN_SAMPLING_REGULAR | RTXN_SAMPLING_MIDPOINT_WORLD.
 * radiance_half4: half[max_segments*32][4]; segment_step (may be NULL): float[max_segments], for MIDPOINT_WORLD the
 * world-space step |end - start| / 32 x t_scale of each segment -- what rtxn_volrender_fwd_compact_nerf consumes.
 * *total_segments: device int (rtxn_scan_hits' total), clamped to max_segments.
 * rtxn_hashmlp_supported: 1 if the fused kernel is built for this model / grid pair (64 wide, 1..8 hidden layers, 2 features
 * per level, an even number of levels, encoded width <= 64), else 0 (the entry point then returns RTXN_ERR_UNSUPPORTED). */
int rtxn_hashmlp_supported(const rtxn_mlp* m, const rtxn_hashgrid* g, int n_dir_freqs);
int rtxn_hashmlp_forward_segments(const rtxn_mlp* m, const rtxn_hashgrid* g, int n_dir_freqs, const void* table_fp16,
                                  const float* start_points, const float* end_points, const float* seg_view,
                                  const int* total_segments, long max_segments, int sample_type, float t_scale,
                                  void* radiance_half4, float* segment_step, rtxn_stream_t stream);

/* ---- one frame behind one call ------------------------------------------------------------------------------------
 * The per-image host sequence of the reference -- fill Params and optixLaunch (main.cu:473-508), copy every traversal
 * buffer to the host and re-pack it (:510-543, :646-673), thrust compaction (:631-637), launchSampler (:704),
 * network->forward (:721), convertHalfToFloat (:723-728), launch_volrender_cuda (:737) -- as ONE entry point that
 * enqueues   trace(count) -> scan -> trace(write packed CSR) -> sampler+encode+MLP (one kernel) -> composite
 * for a window of the width x height launch, with the segment count, the capacity clamp and the overflow flag all in
 * device memory: nothing synchronises the host, so rtxn_render_frame is hipGraph-capturable.
 *
 * rtxn_render owns no device memory of its own: the caller hands rtxn_render_create one workspace of
 * rtxn_render_workspace_bytes(cfg) bytes (256-byte aligned) in which the per-frame buffers of `n_slots` frames in flight
 * and the occupancy hierarchy (4^3 mip, bricks, 16^3 mip: rtxn_build_occupancy_*) are laid out.  Streams and events of the
 * pipelined form are created by rtxn_render_create and released by rtxn_render_destroy.
 *
 * Radiance model: `mlp` alone = a Composite-Frequency model (the fused inference kernels; rtxn_mlp_set_params must have run);
 * `mlp` + `grid` + `table_fp16` = multiresolution hash grid + Frequency(n_dir_freqs) directions feeding a pre-encoded
 * (RTXN_ENC_EXTERNAL) 64-wide model through rtxn_hashmlp_forward_segments (rtxn_mlp_set_params[_training] must have run;
 * the table is read at frame time, so a training loop may keep updating it in place). */
enum rtxn_render_flags {
  RTXN_RENDER_FLOAT4 = 1,   /* hand the compositor the reference's float4 radiance + float t_vals (convertHalfToFloat layout,
                             * 20 B/sample) instead of the network's own half4 outputs (8 B/sample); same pixels bit for bit */
  RTXN_RENDER_STABLE_INPUTS = 2   /* rtxn_render_frame_async only: the caller promises that a frame's device inputs (the 16
                             * look_at floats, the occupancy bits) are complete BEFORE the call and stay untouched until the
                             * frame's traversal has run (pre-uploaded poses, one buffer per frame in flight).  The traversal
                             * then does not wait for the caller's stream and overlaps the previous frame's MLP kernel. */
};
typedef struct rtxn_render_config {
  const rtxn_mlp* mlp;
  const rtxn_hashgrid* grid;      /* NULL: frequency model */
  const void* table_fp16;         /* hash grid only (device) */
  int n_dir_freqs;                /* hash grid only */
  uint32_t width, height;         /* the launch (params.h:41) */
  float focal_length, aspect_ratio;
  uint32_t max_rays;              /* largest ray window a frame call will ask for (0 = width*height) */
  uint32_t window_chunk, window_stride;  /* ray interleave of this shard, as rtxn_trace_params */
  int grid_res;
  const uint32_t* occupancy;      /* R^3 bits (device) or NULL = dense; read at rtxn_render_create / rtxn_render_set_occupancy
                                   * (hierarchy build) and by every frame */
  int trace_mode;                 /* enum rtxn_trace_mode */
  int sub_rays;                   /* RTXN_TRACE_DDA: lanes per ray (rtxn_trace_params.sub_rays); 0 = 1 */
  int vr_mode;                    /* enum rtxn_volrender_mode */
  int sample_type;                /* RTXN_SAMPLING_REGULAR (frequency model: the only one) | RTXN_SAMPLING_MIDPOINT_WORLD */
  float step_scale;               /* RTXN_VR_NERF: density scale on the world-space step */
  long max_segments;              /* capacity of the packed segment buffers; a frame that needs more is truncated on the
                                   * device (never out of bounds) and reported by rtxn_render_status */
  int n_slots;                    /* frames in flight for rtxn_render_frame_async: 1..4 */
  int flags;                      /* enum rtxn_render_flags */
} rtxn_render_config;
typedef struct rtxn_render rtxn_render;

typedef struct rtxn_render_stats {
  long frames;              /* frames enqueued so far */
  long frames_checked;      /* of those, frames whose segment count has reached the host */
  long overflow_frames;     /* checked frames that needed more than max_segments (delivered truncated) */
  long max_segments_needed; /* largest segment count seen */
  long last_segments;       /* segment count of the most recently checked frame */
  long max_segments;        /* the capacity */
} rtxn_render_stats;

size_t rtxn_render_workspace_bytes(const rtxn_render_config* cfg);
int rtxn_render_create(const rtxn_render_config* cfg, void* workspace, size_t workspace_bytes, rtxn_render** out);
int rtxn_render_destroy(rtxn_render* r);
/* Rebuild the occupancy hierarchy after the bits behind cfg->occupancy changed (or point the renderer at another bitfield
 * of the same resolution).  The rebuild runs on `stream` behind whatever the caller enqueued there (the copy of the new
 * bits); `stream` first waits for pipelined traversals still in flight, and the next rtxn_render_frame_async traversal waits
 * for the rebuild.  A serial rtxn_render_frame on a DIFFERENT stream is the caller's to order. */
int rtxn_render_set_occupancy(rtxn_render* r, const uint32_t* occupancy, rtxn_stream_t stream);
/* Counting pass + scan for one pose: *segments (HOST) = packed segments the window needs.  Synchronises `stream`; for sizing
 * max_segments outside any timed region (the reference sizes by 3R slots per ray, main.cu:486). */
int rtxn_render_count_segments(rtxn_render* r, const float* look_at, uint32_t ray_begin, uint32_t ray_count, long* segments,
                               rtxn_stream_t stream);
/* One frame (or one shard of it: rays [ray_begin, ray_begin + ray_count) of the launch under the configured interleave) on
 * ONE stream, using buffer slot `slot`.  look_at: 16 floats on the DEVICE, copied into the slot first, on `stream` (so
 * ordered behind the caller's writes there).  pixels: float[ray_count][3].  Nothing synchronises; capturable.  Every slot
 * has its own buffers incl. the scan workspace: two calls on two streams with two DIFFERENT slots may run concurrently (the
 * host calls themselves one at a time). */
int rtxn_render_frame(rtxn_render* r, int slot, const float* look_at, uint32_t ray_begin, uint32_t ray_count, float* pixels,
                      rtxn_stream_t stream);
/* The same frame software-pipelined against its neighbours: traversal on an internal stream, the MLP kernel on `stream`,
 * the compositor on a second internal stream, rotating through the n_slots buffer slots, so that under the MLP kernel of
 * frame i the chip also traverses frame i+1 and composites frame i-1.  *composite_stream (may be NULL) receives the stream
 * the pixels are complete on: enqueue follow-up work on the pixels there (a gather, a copy), or call rtxn_render_drain.
 * ORDERING CONTRACT.  look_at (device) is read on the internal traversal stream.  By default that stream first waits for
 * everything the caller has enqueued on `stream` up to this call, so a pose written on `stream` (one buffer rewritten per
 * frame), new occupancy bits copied there, or a table update are all seen -- and the traversal of frame i+1 therefore starts
 * only when the MLP kernel of frame i has finished (about 3 % of an 800x800 frame).  Callers whose inputs are stable declare
 * it with RTXN_RENDER_STABLE_INPUTS (see there) and get the full overlap; then the traversal waits for `stream` only on a
 * slot's first use and after rtxn_render_set_occupancy.  Work on OTHER streams is the caller's to order.  One thread at a
 * time per renderer. */
int rtxn_render_frame_async(rtxn_render* r, const float* look_at, uint32_t ray_begin, uint32_t ray_count, float* pixels,
                            rtxn_stream_t stream, rtxn_stream_t* composite_stream);
/* As rtxn_render_frame_async with the pose in HOST memory, where the reference keeps it (main.cu:481-501 fills Params from a
 * host look_at): the 16 floats are copied into pinned staging owned by the slot before the call returns (the caller may
 * reuse its array at once) and uploaded on the traversal stream, so the pipeline overlaps fully without any promise. */
int rtxn_render_frame_async_host(rtxn_render* r, const float* look_at_host, uint32_t ray_begin, uint32_t ray_count,
                                 float* pixels, rtxn_stream_t stream, rtxn_stream_t* composite_stream);
/* Make `stream` wait for everything rtxn_render_frame_async has enqueued on the internal streams. */
int rtxn_render_drain(rtxn_render* r, rtxn_stream_t stream);
/* Overflow report without polling the device.  Every frame updates four per-slot counters ON THE DEVICE (segments of the
 * frame, frames, frames over capacity, largest count) and copies them to pinned host memory (16 bytes, async); the counters
 * are cumulative, so frames replayed from a captured hipGraph are counted once per replay.  This call folds in the copies
 * that have arrived (wait != 0: synchronises the device first, so every enqueued or replayed frame is seen). */
int rtxn_render_status(rtxn_render* r, int wait, rtxn_render_stats* out);
/* Device buffers of a slot for inspection (tests, profiling tools); any out pointer may be NULL.  num_stored = segments
 * actually written per ray; t_vals: RTXN_RENDER_FLOAT4 only, segment_step: compact RTXN_VR_NERF only (else NULL). */
int rtxn_render_slot_buffers(rtxn_render* r, int slot, const int** num_hits, const int** num_stored, const int** indices,
                             const int** total_segments, const float** start_points, const float** end_points,
                             const float** seg_view, const void** radiance, const float** t_vals, const float** segment_step,
                             const float** viewing_direction);

/* ---- training path (tiny-cuda-nn surface of main.cu:721-787) ------------------------ */
/* Per-sample training tensors are FEATURE-MAJOR fp16: X[feature][S_pad] with
 * S_pad = rtxn_padded_samples(S) (S rounded up to 256), padding columns zero. */
long rtxn_padded_samples(long n_samples);

/* Encoders: input float[S][5] (x,y,z in [-1,1]; theta,phi) -> encT half[E][S_pad]. */
int rtxn_encode_frequency(const rtxn_mlp* m, const float* input, void* encT, long n_samples, rtxn_stream_t stream);

/* Multiresolution hash grid (Mueller et al. 2022, tcnn "HashGrid") for the position, composed
 * with Frequency(n_dir_freqs) for the view direction; width padded to 16 with ones. */
typedef struct rtxn_hashgrid_config {
  int n_levels;            /* <= 16 */
  int n_features;          /* per level: 1, 2, 4 or 8 */
  int log2_hashmap_size;   /* table entries per level = min(dense level size, 2^this) */
  int base_resolution;
  float per_level_scale;
} rtxn_hashgrid_config;
int rtxn_hashgrid_create(const rtxn_hashgrid_config* cfg, rtxn_hashgrid** out);
int rtxn_hashgrid_destroy(rtxn_hashgrid* g);
long rtxn_hashgrid_n_params(const rtxn_hashgrid* g);                 /* fp16 table entries */
int rtxn_hashgrid_encoded_width(const rtxn_hashgrid* g, int n_dir_freqs);
/* Table layout for callers that treat levels differently (the data-parallel gradient exchange): offset of a level in
 * PARAMETERS (entries x n_features; level == n_levels gives the total), and whether the level is hashed (its dense size
 * exceeds 2^log2_hashmap_size) or stored densely.  Dense levels come first. */
long rtxn_hashgrid_level_offset(const rtxn_hashgrid* g, int level);
int rtxn_hashgrid_level_is_hashed(const rtxn_hashgrid* g, int level);
int rtxn_hashgrid_encode(const rtxn_hashgrid* g, int n_dir_freqs, const void* table_fp16, const float* input,
                         void* encT, long n_samples, rtxn_stream_t stream);
/* dtable (fp32, table layout) += scatter of dencT; the caller zeroes dtable per step. */
int rtxn_hashgrid_backward(const rtxn_hashgrid* g, const float* input, const void* dencT, long n_samples,
                           float* dtable, rtxn_stream_t stream);

/* launchSampler folded into its consumers: the encoders and the hash-grid scatter take the packed SEGMENTS (start/end float3
 * per segment, seg_view float2 per segment as rtxn_trace_grid writes them) and form sample (segment g, i) themselves exactly
 * as rtxn_sample would -- sample_type RTXN_SAMPLING_REGULAR or RTXN_SAMPLING_MIDPOINT_WORLD (the deterministic modes) -- so the
 * 20-byte samples (sampler/sampler.h:19-30, `d_sampled_points`) never exist in memory.  n_samples = 32 n_segments.
 * t_vals (may be NULL): the sampler's t_vals, float[n_segments * 32], times t_scale (MIDPOINT_WORLD: world step x density
 * scale; REGULAR: (i + 1)/32, t_scale ignored).  backward_segments: dtable_hashed_half NULL = all levels fp32
 * (rtxn_hashgrid_backward), else the mixed form (rtxn_hashgrid_backward_mixed). */
int rtxn_encode_frequency_segments(const rtxn_mlp* m, const float* start_points, const float* end_points, const float* seg_view,
                                   long n_segments, int sample_type, float t_scale, void* encT, float* t_vals,
                                   rtxn_stream_t stream);
int rtxn_hashgrid_encode_segments(const rtxn_hashgrid* g, int n_dir_freqs, const void* table_fp16, const float* start_points,
                                  const float* end_points, const float* seg_view, long n_segments, int sample_type,
                                  float t_scale, void* encT, float* t_vals, rtxn_stream_t stream);
int rtxn_hashgrid_backward_segments(const rtxn_hashgrid* g, const float* start_points, const float* end_points, long n_segments,
                                    int sample_type, const void* dencT, float* dtable, void* dtable_hashed_half,
                                    rtxn_stream_t stream);

/* As rtxn_hashgrid_backward, with the HASHED levels' gradient accumulated in fp16 (n_features == 2: one
 * global_atomic_pk_add_f16 per corner instead of two fp32 atomics -- the scatter is bound by atomic instructions -- and what
 * tiny-cuda-nn does: its grid gradient is __half2).  dtable: fp32, whole-table layout, receives the densely stored levels;
 * dtable_hashed_half: fp16, the parameters from rtxn_hashgrid_level_offset(first hashed level) on.  Both accumulated into. */
int rtxn_hashgrid_backward_mixed(const rtxn_hashgrid* g, const float* input, const void* dencT, long n_samples,
                                 float* dtable, void* dtable_hashed_half, rtxn_stream_t stream);

/* network->forward with saved activations (main.cu:721).  workspace: at least
 * rtxn_mlp_train_workspace_bytes(m, S) bytes, shared with the backward call.
 * output_half: half[S][16]; radiance: float[S][4] or NULL (the glue of main.cu:723-728). */
size_t rtxn_mlp_train_workspace_bytes(const rtxn_mlp* m, long n_samples);
int rtxn_mlp_train_forward(const rtxn_mlp* m, const void* encT, long n_samples, void* workspace,
                           void* output_half, float* radiance, rtxn_stream_t stream);
/* network->backward (main.cu:781).  dout_half4: half[S][4], the layout
 * launch_volrender_backward_cuda writes (stride 4; output rows 4..15 carry no gradient).
 * dparams: float[n_params] in the tcnn parameter layout, ACCUMULATED into (zero it per step);
 * dencT: half[E][S_pad] gradient w.r.t. the encoded input, or NULL. */
int rtxn_mlp_train_backward(const rtxn_mlp* m, const void* encT, const void* output_half, const void* dout_half4,
                            long n_samples, void* workspace, float* dparams, void* dencT, rtxn_stream_t stream);

/* Recompute path for models whose whole gradient fits on the chip (64 wide, <= 4 hidden layers, encoded width <= 64:
 * BASELINE configs[2]).  network->forward (main.cu:721) WITHOUT saved activations, and network->backward (main.cu:781) as ONE
 * kernel that rebuilds the activations in registers from encT, runs the dgrad chain and accumulates every layer's weight
 * gradient on the chip (LDS-transposed operands, one pass of atomics per wave): no per-sample, per-layer tensor touches HBM
 * and no workspace is needed.  Same results as rtxn_mlp_train_forward + rtxn_mlp_train_backward up to fp32 summation order.
 * rtxn_mlp_train_recompute_supported: 1 if this model can use the pair, else 0 (backward_recompute then returns
 * RTXN_ERR_UNSUPPORTED; forward_outputs works for every trainable width). */
int rtxn_mlp_train_recompute_supported(const rtxn_mlp* m);
int rtxn_mlp_train_forward_outputs(const rtxn_mlp* m, const void* encT, long n_samples, void* output_half, float* radiance,
                                   rtxn_stream_t stream);
int rtxn_mlp_train_backward_recompute(const rtxn_mlp* m, const void* encT, const void* output_half, const void* dout_half4,
                                      long n_samples, float* dparams, void* dencT, rtxn_stream_t stream);

/* Lean path for the reference's own model (128 wide, 8 hidden layers, 112 encoded features: main.cu:35-69), where the whole
 * gradient does not fit on the chip.  The saved-activation pair above moves 8.8 KB per sample through device memory (2 KB of
 * activations out of the forward, 2 KB of dZ out of the backward chain, both back into the weight-gradient GEMM) and needs a
 * workspace of 4.3 KB per sample -- 40 GB at the reference's batch (main.cu:186).  Here the forward keeps only the 16-byte
 * sign masks per sample and layer (all the backward chain needs of the activations), the chain writes dZ as before, and the
 * weight gradient RECOMPUTES the activations from the encoded input in three passes (layers 0-2, 3-5, 6-7 + output) with the
 * gradients accumulated on chip: 5.4 KB moved (4.4 with the _segments pair below, which needs no encT) and 2.2 KB of workspace per sample.  Same results as the pair above up to the
 * summation order of the fp32 atomics.
 *   rtxn_mlp_train_lean_supported: 1 if the model has this path;
 *   workspace_lean: rtxn_mlp_train_lean_workspace_bytes(m, n_samples) bytes, written by _forward_lean, read by _backward_lean;
 *   live_ws (may be NULL): as rtxn_mlp_train_backward_live -- only the listed segments are visited. */
int rtxn_mlp_train_lean_supported(const rtxn_mlp* m);
size_t rtxn_mlp_train_lean_workspace_bytes(const rtxn_mlp* m, long n_samples);
int rtxn_mlp_train_forward_lean(const rtxn_mlp* m, const void* encT, long n_samples, void* workspace_lean, void* output_half,
                                float* radiance, rtxn_stream_t stream);
/* The same forward with the encoder folded in (the reference's model only: Composite-Frequency with 3 x 10 position and 2 x 12
 * direction frequencies, main.cu:35-69; _fused_supported says so): the samples are formed from the packed segments as
 * rtxn_encode_frequency_segments forms them and encoded straight into the first layer's operands -- bit for bit the values
 * _forward_lean reads out of encT; t_vals (may be NULL) as rtxn_encode_frequency_segments writes them.  _backward_lean_segments is the
 * matching backward: its weight gradient recomputes the encoding with the activations, a column tile's segment constants through
 * the scalar cache.  With the pair no encT exists at all (main.cu:721,781 are tcnn calls that encode and multiply in one):
 * 3.6 instead of 4.8 KB per sample through device memory (the pair also forms the last hidden layer's dZ on chip), and no encoder launch. */
int rtxn_mlp_train_forward_lean_fused_supported(const rtxn_mlp* m);
int rtxn_mlp_train_forward_lean_segments(const rtxn_mlp* m, const float* start_points, const float* end_points, const float* seg_view,
                                         long n_segments, int sample_type, float t_scale, float* t_vals, void* workspace_lean,
                                         void* output_half, float* radiance, rtxn_stream_t stream);
int rtxn_mlp_train_backward_lean_segments(const rtxn_mlp* m, const float* start_points, const float* end_points, const float* seg_view,
                                          long n_segments, int sample_type, const void* output_half, const void* dout_half4,
                                          void* workspace_lean, const void* live_ws, float* dparams, rtxn_stream_t stream);
int rtxn_mlp_train_backward_lean(const rtxn_mlp* m, const void* encT, const void* output_half, const void* dout_half4,
                                 long n_samples, void* workspace_lean, const void* live_ws, float* dparams, rtxn_stream_t stream);

/* loss->evaluate (tcnn "L2", main.cu:36-38,759): values[i] = d^2/n, grads[i] = loss_scale*2d/n (half),
 * *loss_sum (device float) = sum of values.  values/grads/loss_sum may each be NULL. */
int rtxn_l2_loss(const float* pred, const float* target, long n, float loss_scale, float* values, void* grads_half,
                 float* loss_sum, rtxn_stream_t stream);
/* optimizer->step (tcnn "Adam", main.cu:40-46,787): fp32 master weights + fp16 copy, fp32 gradients. */
int rtxn_adam_step(long n, float* master, void* params_fp16, const float* grads, float* m, float* v, int step,
                   float lr, float beta1, float beta2, float eps, float loss_scale, rtxn_stream_t stream);

/* rtxn_adam_step with the gradient in fp16 (the hashed levels' table gradient, rtxn_hashgrid_backward_mixed). */
int rtxn_adam_step_half_grads(long n, float* master, void* params_fp16, const void* grads_fp16, float* m, float* v, int step,
                              float lr, float beta1, float beta2, float eps, float loss_scale, rtxn_stream_t stream);

/* The two Adam entry points for a step that is replayed as a hipGraph: the bias-corrected rate lr*sqrt(1-beta2^t)/(1-beta1^t)
 * depends on the step number, which must not be baked into the captured launch, so it is read from DEVICE memory
 * (*effective_lr) -- the caller refreshes it before every replay (e.g. an H2D copy node from pinned memory holding
 * rtxn_adam_effective_lr(lr, beta1, beta2, t), which is exactly the value rtxn_adam_step computes on the host).
 * grad_flags: RTXN_ADAM_GRADS_FP16: `grads` is half[n] (as rtxn_adam_step_half_grads), else float[n];
 * RTXN_ADAM_ZERO_GRADS: the gradient is cleared as it is consumed, so the next step accumulates into zeros without a
 * separate fill pass over the (tens of MB of) table gradient. */
enum rtxn_adam_grad_flags { RTXN_ADAM_GRADS_FP16 = 1, RTXN_ADAM_ZERO_GRADS = 2 };
float rtxn_adam_effective_lr(float lr, float beta1, float beta2, int step);
int rtxn_adam_step_captured(long n, float* master, void* params_fp16, void* grads, int grad_flags, float* m, float* v,
                            const float* effective_lr, float beta1, float beta2, float eps, float loss_scale,
                            rtxn_stream_t stream);
/* tiny-cuda-nn's Adam for its "non-matrix" parameters, i.e. the hash table (optimizers/adam.h, adam_step -- [upstream], the
 * reference's optimizer, main.cu:36-46,787): an entry whose gradient is EXACTLY zero is skipped (moments and weight unchanged),
 * and the bias correction lr*sqrt(1-beta2^t)/(1-beta1^t) uses the entry's own update count t = ++param_steps[i] (uint32[n],
 * zero-initialised, part of the optimizer state).  No step number in the arguments: the call is capturable as it is.
 * grad_flags as rtxn_adam_step_captured.  HBM: the gradient everywhere, the 18 B of state per parameter only where it is
 * non-zero (a batch touches 0.2 .. 25 % of a hashed level). */
int rtxn_adam_step_sparse(long n, float* master, void* params_fp16, void* grads, int grad_flags, float* m, float* v,
                          unsigned* param_steps, float lr, float beta1, float beta2, float eps, float loss_scale,
                          rtxn_stream_t stream);

/* ---- deterministic training (debugging, tight comparisons) ------------------------------------------------------------
 * By default every gradient sum that crosses workgroups is a float atomic (the weight-gradient kernels' final flush; the hash
 * scatter, packed fp16 on the hashed levels as tiny-cuda-nn does): two runs of one step differ in the last bits, which Adam's
 * 1/sqrt(v) turns into +-lr on entries whose gradient is noise around zero.  With shadows registered here those sums are
 * accumulated in 64-bit fixed point (value x 2^40, integer atomics: order-independent) and folded into the gradient buffers
 * once per call, so identical inputs give identical bits, run after run and across data-parallel ranks.
 *   mlp_shadow:   rtxn_deterministic_workspace_bytes(rtxn_mlp_n_params) bytes, zeroed once by the caller, or NULL;
 *   table_shadow: rtxn_deterministic_workspace_bytes(rtxn_hashgrid_n_params) bytes, zeroed once, or NULL (no hash grid).
 * Process-wide and read when a backward / scatter entry point is CALLED (so: baked into a captured graph); (NULL, NULL)
 * restores the default.  The folds leave the shadows zero.  Costs: 8-byte atomics in the scatter (about 2x its time) and one
 * sweep over each shadow per call.  Not covered: the reported loss sum (a float atomic; it feeds nothing back). */
size_t rtxn_deterministic_workspace_bytes(long n_params);
int rtxn_set_deterministic_workspace(void* mlp_shadow, void* table_shadow);

/* ---- one training batch without a host round trip ------------------------------------------------------------------
 * The body of the reference's training loop between the traversal and the optimizer (main.cu:703-781: launchSampler ->
 * network->forward -> launch_volrender_cuda -> loss->evaluate -> launch_volrender_backward_cuda -> network->backward) as
 * ONE call whose kernels take the batch's segment count from the device (`total_segments`, as rtxn_scan_hits leaves it;
 * clamped to segment_capacity), where the reference -- and the per-stage entry points above -- need it on the host
 * (main.cu:632 synchronises for it).  Nothing in the call synchronises or allocates, so traversal + this call + the
 * optimizer can be captured into a hipGraph and replayed.  Same kernels and results as the per-stage sequence
 *   rtxn_hashgrid_encode_segments | rtxn_encode_frequency_segments -> rtxn_mlp_train_forward[_outputs] ->
 *   rtxn_volrender_l2_train (RTXN_VR_NERF) | rtxn_volrender_fwd + rtxn_l2_loss + rtxn_volrender_bwd (RTXN_VR_COMPAT) ->
 *   rtxn_mlp_train_backward[_recompute] -> rtxn_hashgrid_backward_segments
 * with one difference in layout only: the feature-major workspaces use the row stride of the CAPACITY,
 * rtxn_padded_samples(32 * segment_capacity), every step.  All buffers are the caller's, sized for segment_capacity
 * segments; the gradient outputs are ACCUMULATED into (zero them first). */
typedef struct rtxn_train_batch {
  const rtxn_mlp* mlp;
  const rtxn_hashgrid* grid;        /* NULL: the model's own Composite-Frequency encoding */
  int n_dir_freqs;                  /* hash grid only: Frequency(n) on the two direction dimensions */
  const void* table_fp16;           /* hash grid only: half[rtxn_hashgrid_n_params] */
  /* the batch's packed segments (rtxn_trace_grid write pass) */
  const float* start_points;        /* float[capacity][3] */
  const float* end_points;          /* float[capacity][3] */
  const float* seg_view;            /* float[capacity][2] */
  const int* num_stored;            /* int[n_rays]: segments per ray actually stored */
  const int* indices;               /* int[n_rays]: first segment of each ray */
  const int* total_segments;        /* DEVICE int: segments of this batch (rtxn_scan_hits' total) */
  long segment_capacity;
  int n_rays;
  int sample_type;                  /* RTXN_SAMPLING_REGULAR | RTXN_SAMPLING_MIDPOINT_WORLD */
  float t_scale;                    /* MIDPOINT_WORLD: factor on the step lengths written to t_vals */
  int vr_mode;                      /* RTXN_VR_COMPAT | RTXN_VR_NERF */
  const float* targets;             /* float[n_rays][3] */
  float loss_scale;
  /* workspaces */
  void* encT;                       /* half[E][Sp], Sp = rtxn_padded_samples(32 * capacity) */
  void* dencT;                      /* half[E][Sp]; hash grid only */
  void* workspace;                  /* rtxn_mlp_train_workspace_bytes(mlp, 32 * capacity), or NULL: recompute path
                                       (rtxn_mlp_train_recompute_supported) */
  void* output_half;                /* half[32 * capacity][16] */
  float* radiance;                  /* float[32 * capacity][4] */
  float* t_vals;                    /* float[32 * capacity] */
  void* radiance_gradients;         /* half[32 * capacity][4] */
  /* outputs */
  float* pixels;                    /* float[n_rays][3] */
  void* loss_gradients_half;        /* half[n_rays][3] */
  float* loss_sum;                  /* device float (may be NULL): mean squared error of the batch */
  float* dparams;                   /* float[rtxn_mlp_n_params], accumulated into */
  float* dtable;                    /* hash grid: float[n_params], accumulated into */
  void* dtable_hashed_half;         /* hash grid, optional: as rtxn_hashgrid_backward_segments */
  void* live_ws;                    /* optional: rtxn_live_segments_workspace_bytes(capacity); the backward then
                                       visits only the segments that carry a loss gradient (rtxn_live_segments) */
  int skip_table_backward;          /* hash grid: != 0 stops after network->backward (dparams and dencT complete): the caller
                                       runs rtxn_hashgrid_backward_segments[_live] itself -- data parallel: after handing the MLP
                                       gradient to its all-reduce, which then runs beside the scatter */
  int workspace_lean;               /* != 0: `workspace` is a LEAN workspace (rtxn_mlp_train_lean_workspace_bytes; models with
                                       rtxn_mlp_train_lean_supported): no activations are saved, the weight gradient recomputes them */
} rtxn_train_batch;
int rtxn_train_gradients(const rtxn_train_batch* batch, rtxn_stream_t stream);

/* ---- one optimisation step as one call -------------------------------------------------------------------------------
 * The body of the reference's training loop for one batch of rays (main.cu:619-805): traversal (count -> scan -> write, the
 * packed layout of main.cu:646-673 written by the device) -> rtxn_train_gradients -> optimizer->step -> training-weight
 * re-pack, enqueued on ONE stream with no host round trip (the reference synchronises for the segment count, main.cu:632, and
 * re-packs the segments on the host).  Nothing in it allocates or synchronises: captured once into a hipGraph it is replayed per
 * step with only the ray and target buffers refreshed.  The optimizer is tiny-cuda-nn's Adam: the MLP with the global step
 * count (`step`, advanced by the call on the device), the hash table by the non-matrix rule (rtxn_adam_step_sparse); every
 * gradient buffer is cleared as it is consumed, so they must be zero before the first call.
 * trace: the batch's rays (explicit or pinhole), grid and the per-ray outputs (num_hits, viewing_direction, sub_hits ...);
 *   its segment outputs are taken from `batch` (start_points, end_points, seg_view, num_stored, indices, segment_capacity).
 * batch: as rtxn_train_gradients; total_segments is written by the scan of this call. */
typedef struct rtxn_train_state {
  float* mlp_master;            /* float[rtxn_mlp_n_params]: fp32 master copy (main.cu:328-342) */
  void* mlp_params_fp16;        /* half[n]: the parameters the kernels read */
  float* mlp_m; float* mlp_v;   /* Adam moments */
  float* table_master; void* table_params_fp16; float* table_m; float* table_v;   /* hash grid only: the same for the table ... */
  unsigned* table_steps;        /* ... and its per-entry update counts (uint32[n], zero-initialised) */
  int* step;                    /* DEVICE int: optimisation steps taken so far; the call increments it */
  float* effective_lr;          /* DEVICE float scratch: the MLP's bias-corrected rate of this step */
  float lr, beta1, beta2, eps;  /* main.cu:36-46: 1e-3, 0.9, 0.999, 1e-8 */
  float table_lr, table_eps;    /* tcnn non_matrix_learning_rate_factor x lr; 1e-15 */
  float loss_scale_divisor;     /* 1, or the number of ranks whose gradients were summed into the buffers */
} rtxn_train_state;
typedef struct rtxn_train_step_args {
  rtxn_trace_params trace;
  void* scan_workspace;         /* rtxn_scan_workspace_bytes(batch.n_rays) */
  size_t scan_workspace_bytes;
  rtxn_train_batch batch;
  rtxn_train_state opt;
} rtxn_train_step_args;
int rtxn_train_step(const rtxn_train_step_args* args, rtxn_stream_t stream);

/* ---- segments that carry a loss gradient ---------------------------------------------------------------------------
 * In NeRF training most samples lie behind the first surface: their transmittance, and with it dL/d(radiance), is exactly
 * zero, and so is everything the backward pass would add for them.  rtxn_live_segments lists the 32-sample segments with a
 * non-zero radiance gradient (ascending segment indices + their count, in live_ws); the two _live entry points below are
 * rtxn_mlp_train_backward_recompute / rtxn_mlp_train_backward / rtxn_hashgrid_backward_segments visiting only those segments.  Same sums, fewer
 * terms: results equal to the unrestricted calls up to the order of fp32 / fp16 atomics.  d(encoding) is written for the
 * listed segments only (no reader of the other columns remains).  No counterpart in the reference (tiny-cuda-nn backpropagates
 * every sample). */
size_t rtxn_live_segments_workspace_bytes(long segment_capacity);
int rtxn_live_segments(const void* radiance_gradients_half4, long n_segments, long segment_capacity, void* live_ws,
                       rtxn_stream_t stream);
int rtxn_mlp_train_backward_recompute_live(const rtxn_mlp* m, const void* encT, const void* output_half,
                                           const void* dout_half4, long n_samples, const void* live_ws, float* dparams,
                                           void* dencT, rtxn_stream_t stream);
/* rtxn_mlp_train_backward (the saved-activation path, any built width) over the live segments: the chain visits only
 * them and writes their dZ compactly, the weight-gradient kernels contract over 32 * count samples. */
int rtxn_mlp_train_backward_live(const rtxn_mlp* m, const void* encT, const void* output_half, const void* dout_half4,
                                 long n_samples, void* workspace, const void* live_ws, float* dparams, void* dencT,
                                 rtxn_stream_t stream);
/* The saving half of network->forward over the live segments only: a step that has run rtxn_mlp_train_forward_outputs over
 * the whole batch (outputs, no activations), the compositor and rtxn_live_segments then saves the activations of just the
 * segments the backward will visit -- into the same workspace, at the same places rtxn_mlp_train_forward would have put them,
 * so rtxn_mlp_train_backward_live reads them unchanged.  (In NeRF training 70-90 % of a batch lies behind the first surface
 * and carries no gradient: the full saving forward is the largest kernel of the 8x128 step, and most of what it writes is
 * never read.) */
int rtxn_mlp_train_forward_live(const rtxn_mlp* m, const void* encT, long n_samples, void* workspace, const void* live_ws,
                                rtxn_stream_t stream);
int rtxn_hashgrid_backward_segments_live(const rtxn_hashgrid* g, const float* start_points, const float* end_points,
                                         long n_segments, int sample_type, const void* dencT, const void* live_ws,
                                         float* dtable, void* dtable_hashed_half, rtxn_stream_t stream);

/* fp32 <-> fp16 copies of a gradient block on the device (no counterpart in the reference, which is single-GPU): the
 * data-parallel exchange sends the hashed levels' gradient in fp16 -- tiny-cuda-nn holds that gradient in fp16 throughout. */
int rtxn_convert_f32_to_f16(const float* src, void* dst_half, long n, rtxn_stream_t stream);
int rtxn_convert_f16_to_f32(const void* src_half, float* dst, long n, rtxn_stream_t stream);

/* ---- sparse view of a half2 gradient (data-parallel exchange; no counterpart in the single-GPU reference) ----
 * `values` is half2[n_entries] (the hashed levels' gradient: one entry = both features of a grid corner), cut into blocks of
 * block_entries entries (one hash-grid level each).  A rank's batch touches 0.2 % .. 25 % of a 2^19-entry level, so a level
 * is exchanged as (index, half2) pairs where that is smaller than the level: count, let the ranks agree per block, pack the
 * chosen blocks, gather the lists, add every rank's list (the rank's own included, in rank order) into the cleared blocks.
 *   rtxn_half2_count_nonzero: workspace (rtxn_half2_workspace_bytes) begins with int counts[ceil(n_entries / block_entries)]
 *     = entries of each block with a non-zero half; behind them the per-wave counts the pack pass places its entries with.
 *   rtxn_half2_pack_nonzero:  pairs (device uint32[capacity][2]: entry index, half2 bits) of the non-zero entries of the blocks
 *     whose bit is set in block_mask (at most 64 blocks), in ascending index; `workspace` as rtxn_half2_count_nonzero left it
 *     for these same values (no global append counter: one address takes ~10 ns per atomic).  *count (device int) = entries
 *     needed, which may exceed capacity (the list is then cut off: size it from the counts).  clear != 0: the selected blocks
 *     are left zero.
 *   rtxn_half2_add_pairs:     values[index] += value for `count` pairs (indices unique within a list; out-of-range ones are
 *     ignored); fp16 adds. */
size_t rtxn_half2_workspace_bytes(long n_entries, long block_entries);
int rtxn_half2_count_nonzero(const void* values, long n_entries, long block_entries, void* workspace, rtxn_stream_t stream);
int rtxn_half2_pack_nonzero(void* values, long n_entries, long block_entries, const void* workspace, unsigned long long block_mask,
                            long capacity, void* pairs, int* count, int clear, rtxn_stream_t stream);
int rtxn_half2_add_pairs(void* values, long n_entries, const void* pairs, long count, rtxn_stream_t stream);

/* ---- dataset loader (host only) ------------------------------------------------------- */
/* Replaces load_images_json (loader/data_loader.cpp:34-94; jsoncpp + stb_image's stbi_loadf).
 * Reads <basename>/transforms_<split>.json and its PNG frames.  images: float[n][H][W][3],
 * poses: float[n][16] row-major 4x4 (both malloc'ed, host).  flags = 0 reproduces the reference
 * (alpha dropped without compositing, gamma-2.2 linearisation, focal = .5*800/tan(.5*camera_angle_x),
 * SURVEY Q11/Q12); bit 0: composite alpha over white; bit 1: keep v/255 (no gamma). */
typedef struct rtxn_image_dataset {
  int n_images;
  unsigned image_width, image_height, image_channels;
  float focal;
  float camera_angle_x;
  float* images;
  float* poses;
} rtxn_image_dataset;
int rtxn_load_images_json(const char* basename, const char* split, int flags, rtxn_image_dataset* out);
void rtxn_free_image_dataset(rtxn_image_dataset* d);
/* Fills the stub at loader/data_loader.cpp:140-142 (SceneType::LLFF: the reference sets the directory name and returns an
 * empty vector).  Reads <basedir>/poses_bounds.npy (float64[N][17]: row-major 3x5 [R | t | (H, W, focal)] in LLFF's
 * (down, right, back) axes + near/far bounds) and the PNG frames of <basedir>/images_<factor>/ (images/ if factor <= 1)
 * in name order.  poses: float[N][16] row-major camera-to-world, axes (right, up, back) like the synthetic loader's;
 * focal is in pixels of the loaded resolution; *bounds (may be NULL): malloc'ed float[N][2] near/far, release with
 * rtxn_free_llff_bounds.  flags as rtxn_load_images_json. */
int rtxn_load_llff(const char* basedir, int factor, int flags, rtxn_image_dataset* out, float** bounds);
void rtxn_free_llff_bounds(float* bounds);
/* stb_image_write's role (included, never called, main.cu:19-21): 8-bit RGB PNG of a rendered frame. */
int rtxn_write_png_rgb8(const char* path, const unsigned char* rgb, int width, int height);

#ifdef __cplusplus
}
#endif
#endif /* RTXN_H */
