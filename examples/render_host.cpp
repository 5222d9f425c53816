// C++ host over librtxn.so: the counterpart of the reference's main.cu render sequence, in two forms.
//
//   render_host frame  W H R out.ppm [pixels.f32 [occupancy.u32 [frames [look_at [focal]]]]]
//       One 800x800-class frame through the FRAME entry (include/rtxn.h: rtxn_render_create / rtxn_render_frame /
//       rtxn_render_frame_async): traversal (count, scan, packed write), fused sampler+encode+MLP, compositor, all enqueued by
//       one C call with nothing copied to the host in between -- what the reference does with a host copy and re-pack of every
//       traversal buffer per image (main.cu:510-543, :646-673).  The reference's 8x128 model (main.cu:35-69) with
//       rtxn_mlp_initialize_params(1337) weights; `occupancy.u32` = R^3 bits (raw little-endian words, e.g. written by
//       tests/test_gpu_host.py) or "-" for the reference's dense grid; `frames` > 1 additionally times that many
//       pipelined frames (rtxn_render_frame_async) and prints ms/frame; `look_at` = 16 comma-separated floats (row-major 4x4,
//       params.h:17), `focal` = Params::focal_length (defaults: a hemisphere pose, 1 / tan(camera_angle_x / 2)).
//
//   render_host stages W H out.ppm
//       The reference's own STAGE sequence over the drop-in headers (include/rtxn_dropin: sampler/sampler.h,
//       vol_render/vol_render.h, rtx/include/params.h), call for call in main.cu's order and on its constants (8^3 dense grid
//       main.cu:394, strided 3R slots per ray :486, host re-pack :646-673, REGULAR sampling :711): shows that a main.cu-style
//       host compiles against the drop-in headers with only cuda* -> hip* renames.
//
// Both write the rendered pixels as a binary PPM; `frame` optionally as raw float32 RGB as well.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sampler.h"
#include "vol_render.h"
#include "rtx/include/params.h"
#include "rtxn.h"

#define HIP_CHECK(x)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                  \
      std::exit(1);                                                                        \
    }                                                                                      \
  } while (0)
#define RTXN_CHECK(x)                                                                      \
  do {                                                                                     \
    int rc_ = (x);                                                                         \
    if (rc_ != RTXN_OK) {                                                                  \
      std::fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, rtxn_last_error());            \
      std::exit(1);                                                                        \
    }                                                                                      \
  } while (0)

static void write_ppm(const char* path, const std::vector<float>& pixels, unsigned width, unsigned height) {
  FILE* f = std::fopen(path, "wb");
  if (!f) return;
  std::fprintf(f, "P6\n%u %u\n255\n", width, height);
  for (float v : pixels) std::fputc((int)std::lround(255.0f * std::fmin(std::fmax(v, 0.0f), 1.0f)), f);
  std::fclose(f);
}

// the reference's model (main.cu:35-69, 325-352): 8 x 128, Composite-Frequency, seeded Xavier weights in fp16 on the device
static rtxn_mlp* make_reference_model(hipStream_t stream, __half** d_params_out) {
  rtxn_mlp_config cfg = {3, 10, 2, 12, 128, 8, 4, RTXN_ACT_SIGMOID};
  rtxn_mlp* net = nullptr;
  RTXN_CHECK(rtxn_mlp_create(&cfg, &net));
  const long n_params = rtxn_mlp_n_params(net);
  std::vector<float> params_fp(n_params);
  rtxn_mlp_initialize_params(net, 1337, params_fp.data());
  std::vector<__half> params_h(n_params);
  for (long i = 0; i < n_params; ++i) params_h[i] = __float2half(params_fp[i]);
  __half* d_params;
  HIP_CHECK(hipMalloc((void**)&d_params, n_params * sizeof(__half)));
  HIP_CHECK(hipMemcpy(d_params, params_h.data(), n_params * sizeof(__half), hipMemcpyHostToDevice));
  RTXN_CHECK(rtxn_mlp_set_params(net, d_params, stream));
  *d_params_out = d_params;
  return net;
}

// ------------------------------------------------------------------------------------------------ frame entry
static int run_frame(int argc, char** argv) {
  const unsigned width = argc > 2 ? std::atoi(argv[2]) : 800, height = argc > 3 ? std::atoi(argv[3]) : 800;
  const int R = argc > 4 ? std::atoi(argv[4]) : 128;
  const char* out_path = argc > 5 ? argv[5] : "render_host.ppm";
  const char* raw_path = argc > 6 ? argv[6] : nullptr;
  const char* occ_path = argc > 7 && std::strcmp(argv[7], "-") != 0 ? argv[7] : nullptr;
  const int frames = argc > 8 ? std::atoi(argv[8]) : 1;

  hipStream_t stream;
  HIP_CHECK(hipStreamCreate(&stream));
  __half* d_params;
  rtxn_mlp* net = make_reference_model(stream, &d_params);

  // occupancy bits of the R^3 grid (bit (x*R + y)*R + z), or none: the reference's dense grid (main.cu:393-399)
  uint32_t* d_occ = nullptr;
  if (occ_path) {
    const size_t words = ((size_t)R * R * R + 31) / 32;
    std::vector<uint32_t> occ(words);
    FILE* f = std::fopen(occ_path, "rb");
    if (!f || std::fread(occ.data(), 4, words, f) != words) { std::fprintf(stderr, "cannot read %zu occupancy words from %s\n", words, occ_path); return 1; }
    std::fclose(f);
    HIP_CHECK(hipMalloc((void**)&d_occ, words * 4));
    HIP_CHECK(hipMemcpy(d_occ, occ.data(), words * 4, hipMemcpyHostToDevice));
  }

  // pose: camera at radius 4.03 on the NeRF-synthetic hemisphere (azimuth 15, elevation -30 deg), translation pre-multiplied
  // by 10 to undo optixPrograms.cu:76-78; corrected focal from camera_angle_x (quirk Q1)
  const float fov_x = 0.6911112f, focal = argc > 10 ? std::strtof(argv[10], nullptr) : 1.0f / std::tan(0.5f * fov_x);
  const double az = 15.0 * M_PI / 180.0, el = -30.0 * M_PI / 180.0, radius = 4.031128874;
  const double cz[3] = {std::cos(el) * std::sin(az), -std::sin(el), std::cos(el) * std::cos(az)};   // camera back axis = position / radius
  double cx[3] = {cz[2], 0.0, -cz[0]};                                                               // right = up x back, up = +y
  const double nx = std::sqrt(cx[0] * cx[0] + cx[2] * cx[2]);
  cx[0] /= nx; cx[2] /= nx;
  const double cy[3] = {cz[1] * cx[2] - cz[2] * cx[1], cz[2] * cx[0] - cz[0] * cx[2], cz[0] * cx[1] - cz[1] * cx[0]};
  float look_at[16] = {(float)cx[0], (float)cy[0], (float)cz[0], (float)(10.0 * radius * cz[0]),
                       (float)cx[1], (float)cy[1], (float)cz[1], (float)(10.0 * radius * cz[1]),
                       (float)cx[2], (float)cy[2], (float)cz[2], (float)(10.0 * radius * cz[2]), 0, 0, 0, 1};
  if (argc > 9) {   // explicit pose: 16 floats, row-major, comma separated
    const char* p = argv[9];
    for (int i = 0; i < 16; ++i) { look_at[i] = std::strtof(p, const_cast<char**>(&p)); if (*p == ',') ++p; }
  }
  float* d_look_at;
  HIP_CHECK(hipMalloc((void**)&d_look_at, sizeof(look_at)));
  HIP_CHECK(hipMemcpy(d_look_at, look_at, sizeof(look_at), hipMemcpyHostToDevice));

  // the renderer: size the segment buffers from a counting pass, then one workspace allocation
  rtxn_render_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.mlp = net;
  cfg.width = width;
  cfg.height = height;
  cfg.focal_length = focal;
  cfg.aspect_ratio = (float)width / (float)height;
  cfg.grid_res = R;
  cfg.occupancy = d_occ;
  cfg.trace_mode = RTXN_TRACE_DDA;
  cfg.sub_rays = (size_t)width * height >= 300000 ? 2 : 8;
  cfg.vr_mode = RTXN_VR_COMPAT;
  cfg.sample_type = RTXN_SAMPLING_REGULAR;
  cfg.step_scale = 1.0f;
  cfg.n_slots = 3;
  cfg.max_segments = 1024;                         // enough for the counting pass (it stores no segments)
  rtxn_render* renderer = nullptr;
  void* d_ws = nullptr;
  size_t ws_bytes = rtxn_render_workspace_bytes(&cfg);
  HIP_CHECK(hipMalloc(&d_ws, ws_bytes));
  RTXN_CHECK(rtxn_render_create(&cfg, d_ws, ws_bytes, &renderer));
  long segments = 0;
  RTXN_CHECK(rtxn_render_count_segments(renderer, d_look_at, 0, 0, &segments, stream));
  RTXN_CHECK(rtxn_render_destroy(renderer));
  HIP_CHECK(hipFree(d_ws));
  cfg.max_segments = segments + segments / 10 + 1024;
  ws_bytes = rtxn_render_workspace_bytes(&cfg);
  HIP_CHECK(hipMalloc(&d_ws, ws_bytes));
  RTXN_CHECK(rtxn_render_create(&cfg, d_ws, ws_bytes, &renderer));

  const size_t n_rays = (size_t)width * height;
  float* d_pixels;
  HIP_CHECK(hipMalloc((void**)&d_pixels, n_rays * 3 * sizeof(float)));
  RTXN_CHECK(rtxn_render_frame(renderer, 0, d_look_at, 0, 0, d_pixels, stream));     // the frame: one call, no host in it
  HIP_CHECK(hipStreamSynchronize(stream));
  std::vector<float> pixels(n_rays * 3);
  HIP_CHECK(hipMemcpy(pixels.data(), d_pixels, n_rays * 3 * sizeof(float), hipMemcpyDeviceToHost));

  double ms_per_frame = 0.0;
  if (frames > 1) {
    for (int i = 0; i < 3; ++i) RTXN_CHECK(rtxn_render_frame_async(renderer, d_look_at, 0, 0, d_pixels, stream, nullptr));
    RTXN_CHECK(rtxn_render_drain(renderer, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < frames; ++i) RTXN_CHECK(rtxn_render_frame_async(renderer, d_look_at, 0, 0, d_pixels, stream, nullptr));
    RTXN_CHECK(rtxn_render_drain(renderer, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    ms_per_frame = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
  }
  rtxn_render_stats st;
  RTXN_CHECK(rtxn_render_status(renderer, 1, &st));

  double sum = 0;
  for (float v : pixels) sum += v;
  write_ppm(out_path, pixels, width, height);
  if (raw_path) {
    FILE* f = std::fopen(raw_path, "wb");
    if (f) { std::fwrite(pixels.data(), sizeof(float), pixels.size(), f); std::fclose(f); }
  }
  std::printf("render_host frame: %zu rays, %ld segments (capacity %ld, %ld overflowed frames), %ld samples, mean pixel %.6f -> %s\n", n_rays,
              st.last_segments, st.max_segments, st.overflow_frames, 32 * st.last_segments, sum / (double)pixels.size(), out_path);
  if (frames > 1) std::printf("render_host frame: %d pipelined frames, %.3f ms/frame, %.2f Mrays/s\n", frames, ms_per_frame, n_rays / ms_per_frame / 1e3);
  RTXN_CHECK(rtxn_render_destroy(renderer));
  rtxn_mlp_destroy(net);
  return (sum > 0 && st.overflow_frames == 0) ? 0 : 2;
}

// ------------------------------------------------------------------------------------------------ stage sequence (drop-in headers)
// The box of one grid cell: the reference's grid (main.cu:154-174) is R^3 equal cells over [-1,1]^3, numbered
// x-major / z-minor.  Written here as a function of the flat cell index; only OptixAabb's field names are the contract.
static OptixAabb cell_box(int cell, int R) {
  const float edge = 2.0f / (float)R;
  const int ijk[3] = {cell / (R * R), (cell / R) % R, cell % R};
  float lo[3], hi[3];
  for (int a = 0; a < 3; ++a) {
    lo[a] = -1.0f + (float)ijk[a] * edge;
    hi[a] = lo[a] + edge;
  }
  return OptixAabb{lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]};
}

static std::vector<OptixAabb> make_grid(int R) {
  std::vector<OptixAabb> boxes((size_t)R * R * R);
  for (size_t cell = 0; cell < boxes.size(); ++cell) boxes[cell] = cell_box((int)cell, R);
  return boxes;
}

// Launch parameters for one pose (the reference fills the same struct field by field before every optixLaunch,
// main.cu:481-501): grouped here by what they describe.
struct LaunchBuffers {
  float3 *start_points, *end_points, *ray_origins;
  float *t_start, *t_end;
  int* num_hits;
  float2* viewing_direction;
};
static Params make_params(unsigned width, unsigned height, int R, float fov_x, const float* d_look_at, OptixAabb* d_aabb,
                          OptixTraversableHandle gas, const LaunchBuffers& buf) {
  Params p{};
  // camera
  p.look_at = const_cast<float*>(d_look_at);
  p.width = width;
  p.height = height;
  p.aspect_ratio = (float)width / (float)height;
  p.focal_length = 1.0f / std::tan(0.5f * fov_x);   // corrected Q1: from camera_angle_x, not from the pixel focal
  // grid
  const float cell = 2.0f / (float)R;
  p.delta = make_float3(cell, cell, cell);
  p.min_point = make_float3(-1.0f, -1.0f, -1.0f);
  p.max_point = make_float3(1.0f, 1.0f, 1.0f);
  p.num_primitives = R * R * R;
  p.intersection_arr_size = 3 * R;                  // slots per ray, main.cu:486
  p.handle = gas;
  p.aabb = d_aabb;
  // outputs
  p.start_points = buf.start_points;
  p.end_points = buf.end_points;
  p.ray_origins = buf.ray_origins;
  p.t_start = buf.t_start;
  p.t_end = buf.t_end;
  p.num_hits = buf.num_hits;
  p.viewing_direction = buf.viewing_direction;
  return p;
}

static int run_stages(int argc, char** argv) {
  unsigned width = argc > 3 ? std::atoi(argv[2]) : 64, height = argc > 3 ? std::atoi(argv[3]) : 64;
  const char* out_path = argc > 4 ? argv[4] : "render_host.ppm";
  const int grid_resolution = 8;        // main.cu:394
  const int samples_per_intersect = 32; // main.cu:677

  hipStream_t inference_stream;
  HIP_CHECK(hipStreamCreate(&inference_stream));
  __half* d_params;
  rtxn_mlp* net = make_reference_model(inference_stream, &d_params);   // main.cu:325-352

  // "acceleration structure" (main.cu:381-399)
  RTXDataHolder* rtx_dataholder = new RTXDataHolder();
  rtx_dataholder->initContext();
  rtx_dataholder->createModule("unused.ptx");
  rtx_dataholder->createProgramGroups();
  rtx_dataholder->linkPipeline(false);
  rtx_dataholder->buildSBT();
  std::vector<OptixAabb> grid = make_grid(grid_resolution);
  OptixAabb* d_aabb = rtx_dataholder->initAccelerationStructure(grid);

  // pose: camera on the +z axis at distance 40 (origin/10 = 4, optixPrograms.cu:76-78), looking at the origin
  float look_at[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 40, 0, 0, 0, 1};
  float* d_look_at;
  HIP_CHECK(hipMalloc((void**)&d_look_at, sizeof(look_at)));
  HIP_CHECK(hipMemcpy(d_look_at, look_at, sizeof(look_at), hipMemcpyHostToDevice));

  // traversal buffers (main.cu:424-430)
  const size_t n_rays = (size_t)width * height, S = 3 * grid_resolution;
  float3 *d_start_points, *d_end_points, *d_ray_origins;
  float *d_t_start, *d_t_end;
  int* d_num_hits;
  float2* d_view_dir;
  HIP_CHECK(hipMalloc((void**)&d_ray_origins, n_rays * sizeof(float3)));
  HIP_CHECK(hipMalloc((void**)&d_t_start, n_rays * S * sizeof(float)));
  HIP_CHECK(hipMalloc((void**)&d_t_end, n_rays * S * sizeof(float)));
  HIP_CHECK(hipMalloc((void**)&d_start_points, n_rays * S * sizeof(float3)));
  HIP_CHECK(hipMalloc((void**)&d_end_points, n_rays * S * sizeof(float3)));
  HIP_CHECK(hipMalloc((void**)&d_num_hits, n_rays * sizeof(int)));
  HIP_CHECK(hipMalloc((void**)&d_view_dir, n_rays * sizeof(float2)));
  HIP_CHECK(hipMemsetAsync(d_num_hits, 0, n_rays * sizeof(int), inference_stream));

  const LaunchBuffers bufs{d_start_points, d_end_points, d_ray_origins, d_t_start, d_t_end, d_num_hits, d_view_dir};
  Params params = make_params(width, height, grid_resolution, 0.6911112f, d_look_at, d_aabb, rtx_dataholder->gas_handle, bufs);
  rtxnLaunch(params, inference_stream);  // optixLaunch, main.cu:506-508

  // compaction (main.cu:631-637) -- on the device, total included
  int *d_indices, *d_total;
  void* d_ws;
  size_t ws_bytes = rtxn_scan_workspace_bytes((int)n_rays);
  HIP_CHECK(hipMalloc((void**)&d_indices, n_rays * sizeof(int)));
  HIP_CHECK(hipMalloc((void**)&d_total, sizeof(int)));
  HIP_CHECK(hipMalloc(&d_ws, ws_bytes));
  rtxn_scan_hits(d_num_hits, d_indices, d_total, (int)n_rays, d_ws, ws_bytes, inference_stream);
  int num_points = 0;
  HIP_CHECK(hipMemcpyAsync(&num_points, d_total, sizeof(int), hipMemcpyDeviceToHost, inference_stream));
  HIP_CHECK(hipStreamSynchronize(inference_stream));

  // host re-pack of the strided segments into the packed layout, as main.cu:646-673 does (the `frame` form has none of this)
  std::vector<int> h_num_hits(n_rays);
  std::vector<float3> h_start(n_rays * S), h_end(n_rays * S), h_pstart(num_points), h_pend(num_points);
  HIP_CHECK(hipMemcpy(h_num_hits.data(), d_num_hits, n_rays * sizeof(int), hipMemcpyDeviceToHost));
  HIP_CHECK(hipMemcpy(h_start.data(), d_start_points, n_rays * S * sizeof(float3), hipMemcpyDeviceToHost));
  HIP_CHECK(hipMemcpy(h_end.data(), d_end_points, n_rays * S * sizeof(float3), hipMemcpyDeviceToHost));
  size_t offset = 0;
  for (size_t k = 0; k < n_rays; k++) {
    for (int l = 0; l < h_num_hits[k]; l++) { h_pstart[offset + l] = h_start[k * S + l]; h_pend[offset + l] = h_end[k * S + l]; }
    offset += h_num_hits[k];
  }
  float3 *d_pstart, *d_pend;
  HIP_CHECK(hipMalloc((void**)&d_pstart, (num_points + 1) * sizeof(float3)));
  HIP_CHECK(hipMalloc((void**)&d_pend, (num_points + 1) * sizeof(float3)));
  HIP_CHECK(hipMemcpy(d_pstart, h_pstart.data(), num_points * sizeof(float3), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(d_pend, h_pend.data(), num_points * sizeof(float3), hipMemcpyHostToDevice));

  long num_sampled_points = (long)samples_per_intersect * num_points;
  float *d_sampled_points, *d_sampled_points_radiance, *d_t_vals, *d_pixels;
  HIP_CHECK(hipMalloc((void**)&d_sampled_points, (num_sampled_points + 1) * 5 * sizeof(float)));
  HIP_CHECK(hipMalloc((void**)&d_sampled_points_radiance, (num_sampled_points + 1) * 4 * sizeof(float)));
  HIP_CHECK(hipMalloc((void**)&d_t_vals, (num_sampled_points + 1) * sizeof(float)));
  HIP_CHECK(hipMalloc((void**)&d_pixels, n_rays * 3 * sizeof(float)));

  launchSampler(d_pstart, d_pend, d_view_dir, d_t_vals, d_sampled_points, (int)n_rays, grid_resolution, d_num_hits,
                d_indices, SAMPLING_REGULAR, inference_stream);                                    // main.cu:704
  rtxn_mlp_forward_radiance(net, d_sampled_points, d_sampled_points_radiance, num_sampled_points,  // main.cu:721-728
                            inference_stream);
  HIP_CHECK(hipStreamSynchronize(inference_stream));  // the reference's vol_render runs on the default stream
  launch_volrender_cuda(d_sampled_points, d_sampled_points_radiance, d_num_hits, d_indices, d_t_vals, (int)n_rays,
                        samples_per_intersect, d_pixels);                                          // main.cu:737
  HIP_CHECK(hipDeviceSynchronize());

  std::vector<float> pixels(n_rays * 3);
  HIP_CHECK(hipMemcpy(pixels.data(), d_pixels, n_rays * 3 * sizeof(float), hipMemcpyDeviceToHost));
  double sum = 0;
  for (float v : pixels) sum += v;
  write_ppm(out_path, pixels, width, height);
  std::printf("render_host stages: %zu rays, %d segments, %ld samples, mean pixel %.6f -> %s\n", n_rays, num_points,
              num_sampled_points, sum / (double)pixels.size(), out_path);
  rtxn_mlp_destroy(net);
  return sum > 0 ? 0 : 2;
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "frame";
  if (mode == "frame") return run_frame(argc, argv);
  if (mode == "stages") return run_stages(argc, argv);
  std::fprintf(stderr, "usage: render_host frame W H R out.ppm [pixels.f32 [occupancy.u32|- [frames [look_at,16,floats [focal]]]]]\n"
                       "       render_host stages W H out.ppm\n");
  return 64;
}
