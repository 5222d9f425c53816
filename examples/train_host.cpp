// C++ host over librtxn.so: the reference's own TRAINING iteration (main.cu:612-805) on its own constants, with the whole loop
// body -- traversal, sampler, network forward, compositor, loss, compositor backward, network backward, optimizer -- as ONE C
// call per batch (include/rtxn.h: rtxn_train_step), captured once into a hipGraph and replayed.
//
//   train_host [steps [batch_rays [grid_res [out.f32|- [compat|nerf]]]]]     defaults: 200 steps, 4096 rays, 8^3 dense grid (main.cu:394), compat
//
// compat: the reference's arithmetic to the letter -- its compositor backward (vol_render.cu:75-143) is not the gradient of its
// forward, so the loss does not go down; nerf: the corrected quadrature with its exact gradient (midpoint samples,
// world-space steps, loss scale 128), which learns.
//
// Model: the reference's 8x128 ReLU MLP with Composite-Frequency(10, 12) encoding (main.cu:35-69), REGULAR sampler (:711), the
// reference compositor forward / backward (RTXN_VR_COMPAT, vol_render.cu:19-143), L2 loss, Adam lr 1e-3 (main.cu:36-46), loss
// scale 1.  Rays: a fixed pool of pinhole rays from eight hemisphere poses, a random batch of them per step drawn on the host
// ahead of time (the reference shuffles a host vector of RayPayload, main.cu:612-629); targets: a colour that is a smooth function
// of the ray's direction -- this program shows the call sequence and measures the step, it does not load a dataset.
// Prints the loss every 50 steps and ms per step; optionally writes the final fp32 master parameters.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

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

template <class T>
static T* dev_alloc(size_t n, bool zero = true) {
  T* p = nullptr;
  HIP_CHECK(hipMalloc(&p, n * sizeof(T) > 0 ? n * sizeof(T) : 16));
  if (zero) HIP_CHECK(hipMemset(p, 0, n * sizeof(T)));
  return p;
}

// ray of pixel (x, y) of a W x H pinhole image: optixPrograms.cu:43-116 with the corrected focal length and the
// camera on a radius-0.4 hemisphere looking at the origin
static void pool_rays(int n_poses, int side, std::vector<float>& o, std::vector<float>& d) {
  const float focal = 1.0f / std::tan(0.5f * 0.6911112f);
  for (int p = 0; p < n_poses; ++p) {
    const float az = (45.0f * p + 15.0f) * 3.14159265f / 180.0f, el = 30.0f * 3.14159265f / 180.0f;
    const float c[3] = {0.4f * std::cos(el) * std::cos(az), 0.4f * std::cos(el) * std::sin(az), 0.4f * std::sin(el)};
    float f[3] = {-c[0], -c[1], -c[2]};
    const float fn = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (float& v : f) v /= fn;
    float r[3] = {f[1] * 1.0f - f[2] * 0.0f, f[2] * 0.0f - f[0] * 1.0f, 0.0f};    // f x (0, 0, 1)
    const float rn = std::sqrt(r[0] * r[0] + r[1] * r[1]);
    r[0] /= rn; r[1] /= rn;
    const float u[3] = {r[1] * f[2] - r[2] * f[1], r[2] * f[0] - r[0] * f[2], r[0] * f[1] - r[1] * f[0]};
    for (int y = 0; y < side; ++y)
      for (int x = 0; x < side; ++x) {
        const float px = 2.0f * (x + 0.5f) / side - 1.0f, py = 2.0f * (y + 0.5f) / side - 1.0f;
        float dir[3];
        for (int k = 0; k < 3; ++k) dir[k] = px * r[k] + py * u[k] + focal * f[k];
        const float dn = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        for (int k = 0; k < 3; ++k) {
          o.push_back(c[k]);
          d.push_back(dir[k] / dn);
        }
      }
  }
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? std::atoi(argv[1]) : 200;
  const int B = argc > 2 ? std::atoi(argv[2]) : 4096;
  const int R = argc > 3 ? std::atoi(argv[3]) : 8;
  const char* out_path = argc > 4 && std::strcmp(argv[4], "-") != 0 ? argv[4] : nullptr;
  const bool nerf = argc > 5 && std::strcmp(argv[5], "nerf") == 0;
  hipStream_t stream;
  HIP_CHECK(hipStreamCreate(&stream));

  // ---- model (main.cu:325-352) ----
  rtxn_mlp_config cfg = {3, 10, 2, 12, 128, 8, 4, RTXN_ACT_SIGMOID};
  rtxn_mlp* net;
  RTXN_CHECK(rtxn_mlp_create(&cfg, &net));
  const long n_params = rtxn_mlp_n_params(net);
  std::vector<float> master_h(n_params);
  RTXN_CHECK(rtxn_mlp_initialize_params(net, 1337, master_h.data()));
  std::vector<__half> params_h(n_params);
  for (long i = 0; i < n_params; ++i) params_h[i] = __float2half(master_h[i]);
  float* master = dev_alloc<float>(n_params, false);
  __half* params = dev_alloc<__half>(n_params, false);
  HIP_CHECK(hipMemcpy(master, master_h.data(), n_params * sizeof(float), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(params, params_h.data(), n_params * sizeof(__half), hipMemcpyHostToDevice));
  RTXN_CHECK(rtxn_mlp_set_params(net, params, stream));

  // ---- the ray pool and every batch of the run, drawn up front (seed 42) ----
  std::vector<float> pool_o, pool_d;
  pool_rays(8, 128, pool_o, pool_d);
  const size_t n_pool = pool_o.size() / 3;
  std::mt19937 rng(42);
  std::vector<float> pool_t(3 * n_pool);
  for (size_t i = 0; i < 3 * n_pool; ++i) pool_t[i] = 0.5f + 0.5f * pool_d[i];      // a smooth function of the ray: learnable
  float *d_pool_o = dev_alloc<float>(3 * n_pool, false), *d_pool_d = dev_alloc<float>(3 * n_pool, false), *d_pool_t = dev_alloc<float>(3 * n_pool, false);
  HIP_CHECK(hipMemcpy(d_pool_o, pool_o.data(), pool_o.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(d_pool_d, pool_d.data(), pool_d.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(d_pool_t, pool_t.data(), pool_t.size() * sizeof(float), hipMemcpyHostToDevice));

  // ---- per-step buffers, allocated once at capacity (the reference allocates and frees them per batch, main.cu:667-694) ----
  // a ray crosses at most 3R cells of the grid (main.cu:486); a counting pass over the first batches would give a tighter bound
  const long capacity = (long)B * 3 * R;
  const long Sp = rtxn_padded_samples(32 * capacity);
  const int E = rtxn_mlp_encoded_width(net);
  float *rays_o = dev_alloc<float>(3 * B), *rays_d = dev_alloc<float>(3 * B), *targets = dev_alloc<float>(3 * B);
  float* view_dirs = dev_alloc<float>(2 * B);
  int *num_hits = dev_alloc<int>(B), *indices = dev_alloc<int>(B), *num_stored = dev_alloc<int>(B), *total = dev_alloc<int>(1);
  const int sub_rays = 16;
  int* sub_hits = dev_alloc<int>((size_t)B * sub_rays);
  const size_t scan_bytes = rtxn_scan_workspace_bytes(B);
  void* scan_ws = dev_alloc<char>(scan_bytes);
  float *start = dev_alloc<float>(3 * capacity), *end = dev_alloc<float>(3 * capacity), *seg_view = dev_alloc<float>(2 * capacity);
  __half* encT = dev_alloc<__half>((size_t)E * Sp);
  // the reference's 8 x 128 model takes the lean path: no saved activations, half the workspace (rtxn.h: rtxn_mlp_train_forward_lean)
  const bool lean = rtxn_mlp_train_lean_supported(net) && !(std::getenv("RTXN_TRAIN_LEAN") && std::atoi(std::getenv("RTXN_TRAIN_LEAN")) == 0);
  void* workspace = dev_alloc<char>(lean ? rtxn_mlp_train_lean_workspace_bytes(net, 32 * capacity) : rtxn_mlp_train_workspace_bytes(net, 32 * capacity));
  __half* out_half = dev_alloc<__half>(32 * capacity * 16);
  float *radiance = dev_alloc<float>(32 * capacity * 4), *t_vals = dev_alloc<float>(32 * capacity);
  __half* dout = dev_alloc<__half>(32 * capacity * 4);
  float* pixels = dev_alloc<float>(3 * B);
  __half* loss_grads = dev_alloc<__half>(3 * B);
  float* loss = dev_alloc<float>(1);
  float *dparams = dev_alloc<float>(n_params), *adam_m = dev_alloc<float>(n_params), *adam_v = dev_alloc<float>(n_params);
  void* live_ws = dev_alloc<char>(rtxn_live_segments_workspace_bytes(capacity));
  int* step_dev = dev_alloc<int>(1);
  float* lr_dev = dev_alloc<float>(1);

  // ---- one optimisation step = one call (main.cu:619-805) ----
  rtxn_train_step_args a;
  std::memset(&a, 0, sizeof(a));
  a.trace.rays_o = rays_o; a.trace.rays_d = rays_d; a.trace.width = (uint32_t)B; a.trace.height = 1;
  a.trace.ray_begin = 0; a.trace.ray_count = (uint32_t)B;
  a.trace.grid_res = R; a.trace.occupancy = nullptr /* dense, as the reference */; a.trace.mode = RTXN_TRACE_DDA;
  a.trace.viewing_direction = view_dirs; a.trace.num_hits = num_hits;
  a.trace.sub_rays = sub_rays; a.trace.sub_hits = sub_hits;
  a.scan_workspace = scan_ws; a.scan_workspace_bytes = scan_bytes;
  rtxn_train_batch& b = a.batch;
  b.mlp = net;
  b.start_points = start; b.end_points = end; b.seg_view = seg_view; b.num_stored = num_stored; b.indices = indices;
  b.total_segments = total; b.segment_capacity = capacity; b.n_rays = B;
  b.sample_type = nerf ? RTXN_SAMPLING_MIDPOINT_WORLD : RTXN_SAMPLING_REGULAR;
  b.t_scale = nerf ? 30.0f : 1.0f;               // sigma in (0, 1) (the model's sigmoid) x this = density per unit length
  b.vr_mode = nerf ? RTXN_VR_NERF : RTXN_VR_COMPAT;
  b.targets = targets; b.loss_scale = nerf ? 128.0f : 1.0f;
  b.encT = encT; b.workspace = workspace; b.output_half = out_half; b.radiance = radiance; b.t_vals = t_vals;
  b.radiance_gradients = dout; b.pixels = pixels; b.loss_gradients_half = loss_grads; b.loss_sum = loss; b.dparams = dparams;
  b.live_ws = live_ws;
  b.workspace_lean = lean ? 1 : 0;
  a.opt.mlp_master = master; a.opt.mlp_params_fp16 = params; a.opt.mlp_m = adam_m; a.opt.mlp_v = adam_v;
  a.opt.step = step_dev; a.opt.effective_lr = lr_dev;
  a.opt.lr = 1e-3f; a.opt.beta1 = 0.9f; a.opt.beta2 = 0.999f; a.opt.eps = 1e-8f;
  a.opt.table_lr = 1e-2f; a.opt.table_eps = 1e-15f; a.opt.loss_scale_divisor = 1.0f;

  // the batch gather is the host's business (main.cu:612-629 shuffles RayPayloads); here: index lists drawn up front, gathered
  // by three strided device copies per step so that nothing waits for the host
  std::vector<int> order(n_pool);
  for (size_t i = 0; i < n_pool; ++i) order[i] = (int)i;
  std::shuffle(order.begin(), order.end(), rng);
  // contiguous windows of a shuffled pool = random batches; the pool is permuted once on the device
  std::vector<float> so(3 * n_pool), sd(3 * n_pool), st(3 * n_pool);
  for (size_t i = 0; i < n_pool; ++i)
    for (int k = 0; k < 3; ++k) {
      so[3 * i + k] = pool_o[3 * (size_t)order[i] + k];
      sd[3 * i + k] = pool_d[3 * (size_t)order[i] + k];
      st[3 * i + k] = pool_t[3 * (size_t)order[i] + k];
    }
  HIP_CHECK(hipMemcpy(d_pool_o, so.data(), so.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(d_pool_d, sd.data(), sd.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(d_pool_t, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice));
  const size_t windows = n_pool / (size_t)B;
  if (windows == 0) { std::fprintf(stderr, "batch larger than the ray pool (%zu rays)\n", n_pool); return 1; }
  auto load_batch = [&](int s) {
    const size_t w = (size_t)s % windows;
    HIP_CHECK(hipMemcpyAsync(rays_o, d_pool_o + 3 * w * B, 3 * (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(rays_d, d_pool_d + 3 * w * B, 3 * (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(targets, d_pool_t + 3 * w * B, 3 * (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, stream));
  };

  // first step eagerly (kernel attributes, lazy module state), then the same call captured once and replayed
  load_batch(0);
  RTXN_CHECK(rtxn_train_step(&a, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  float loss_h = 0.0f;
  int total_h = 0;
  HIP_CHECK(hipMemcpy(&loss_h, loss, sizeof(float), hipMemcpyDeviceToHost));
  HIP_CHECK(hipMemcpy(&total_h, total, sizeof(int), hipMemcpyDeviceToHost));
  std::printf("step    0: loss %.6f, %d segments = %ld samples (capacity %ld)\n", loss_h, total_h, 32L * total_h, capacity);
  hipGraph_t graph;
  hipGraphExec_t exec;
  HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal));
  RTXN_CHECK(rtxn_train_step(&a, stream));
  HIP_CHECK(hipStreamEndCapture(stream, &graph));
  HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  const auto t0 = std::chrono::steady_clock::now();
  for (int s = 1; s < steps; ++s) {
    load_batch(s);
    HIP_CHECK(hipGraphLaunch(exec, stream));
    if (s % 50 == 0 || s == steps - 1) {
      HIP_CHECK(hipStreamSynchronize(stream));
      HIP_CHECK(hipMemcpy(&loss_h, loss, sizeof(float), hipMemcpyDeviceToHost));
      std::printf("step %4d: loss %.6f\n", s, loss_h);
    }
  }
  HIP_CHECK(hipStreamSynchronize(stream));
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  int step_h = 0;
  HIP_CHECK(hipMemcpy(&step_h, step_dev, sizeof(int), hipMemcpyDeviceToHost));
  std::printf("train_host: %d optimisation steps of %d rays (8x128 + Composite-Frequency, %d^3 dense grid, %s compositor), %.3f ms per step "
              "(graph replay, incl. the loss read-backs), device step counter %d\n", steps, B, R, nerf ? "NeRF" : "reference", steps > 1 ? ms / (steps - 1) : 0.0, step_h);
  if (out_path) {
    HIP_CHECK(hipMemcpy(master_h.data(), master, n_params * sizeof(float), hipMemcpyDeviceToHost));
    if (FILE* f = std::fopen(out_path, "wb")) {
      std::fwrite(master_h.data(), sizeof(float), (size_t)n_params, f);
      std::fclose(f);
    }
  }
  HIP_CHECK(hipGraphExecDestroy(exec));
  HIP_CHECK(hipGraphDestroy(graph));
  RTXN_CHECK(rtxn_mlp_destroy(net));
  return std::isfinite(loss_h) && step_h == steps ? 0 : 2;
}
