// C++ host that drives librtxn.so through the drop-in headers in the stage order
// of the reference's main.cu (traversal :463-543, compaction :631-637, host
// re-pack :646-673, sampler :704, network->forward :721, glue :728, compositing
// :737), on the reference's own workload constants (8^3 dense grid main.cu:394,
// 32 samples/segment, 8x128 model main.cu:35-69, REGULAR sampling :711) for one
// synthetic pose.  It exists to show that a main.cu-style host compiles against
// include/rtxn_dropin with only cuda* -> hip* renames; it writes the rendered
// pixels as a binary PPM.  Usage: render_host [width height out.ppm]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
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

int main(int argc, char** argv) {
  unsigned width = argc > 2 ? std::atoi(argv[1]) : 64, height = argc > 2 ? std::atoi(argv[2]) : 64;
  const char* out_path = argc > 3 ? argv[3] : "render_host.ppm";
  const int grid_resolution = 8;        // main.cu:394
  const int samples_per_intersect = 32; // main.cu:677

  // model (main.cu:325-352)
  rtxn_mlp_config cfg = {3, 10, 2, 12, 128, 8, 4, RTXN_ACT_SIGMOID};
  rtxn_mlp* net = nullptr;
  if (rtxn_mlp_create(&cfg, &net) != RTXN_OK) { std::fprintf(stderr, "%s\n", rtxn_last_error()); return 1; }
  long n_params = rtxn_mlp_n_params(net);
  std::vector<float> params_fp(n_params);
  rtxn_mlp_initialize_params(net, 1337, params_fp.data());
  std::vector<__half> params_h(n_params);
  for (long i = 0; i < n_params; ++i) params_h[i] = __float2half(params_fp[i]);
  __half* d_params;
  HIP_CHECK(hipMalloc((void**)&d_params, n_params * sizeof(__half)));
  HIP_CHECK(hipMemcpy(d_params, params_h.data(), n_params * sizeof(__half), hipMemcpyHostToDevice));
  hipStream_t inference_stream;
  HIP_CHECK(hipStreamCreate(&inference_stream));
  rtxn_mlp_set_params(net, d_params, inference_stream);

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

  // host re-pack of the strided segments into the packed layout, as main.cu:646-673 does
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
  FILE* f = std::fopen(out_path, "wb");
  if (f) {
    std::fprintf(f, "P6\n%u %u\n255\n", width, height);
    for (float v : pixels) std::fputc((int)std::lround(255.0f * std::fmin(std::fmax(v, 0.0f), 1.0f)), f);
    std::fclose(f);
  }
  std::printf("render_host: %zu rays, %d segments, %ld samples, mean pixel %.6f -> %s\n", n_rays, num_points,
              num_sampled_points, sum / (double)pixels.size(), out_path);
  rtxn_mlp_destroy(net);
  return sum > 0 ? 0 : 2;
}
