// C++ data-parallel TRAINING host over librtxn.so: the reference's own iteration (main.cu:612-805) with the ray batch split over the
// GPUs of a node, ONE process, no Python, no collective library.
//
//   train_host_mgpu steps batch_rays grid_res devices [out.f32|- [gradient.f32]]
//       devices = comma-separated HIP device ordinals, one shard per entry ("0,1,2,3,4,5,6,7"); an ordinal may repeat ("0,0"):
//       the shards then share that device -- the split, the gradient exchange and the replicated optimizer are exercised on ONE
//       GPU exactly as they run on several.  batch_rays must divide by the number of shards.
//
// The reference trains on one device.  Its step is data-parallel over rays with one exchange: every shard runs traversal ->
// sampler + encoder + network forward -> compositor -> loss -> backward over ITS rays (the stage entry points rtxn_train_step
// sequences, here with the optimizer taken out: rtxn_trace_grid / rtxn_scan_hits / rtxn_trace_grid, rtxn_train_gradients), then
// the MLP's gradient (131 k floats = 0.5 MB for the 8 x 128 model: latency-bound) is summed on the root device -- peer copies on
// the shards' own streams, one add per shard in shard order, so the sum is the same on every run that computes the same shard
// gradients -- copied back, and every shard steps ITS replica with the same sum (each shard scales its loss by 1 / N: the sum is
// the gradient of the mean over the whole batch).  Replicas therefore stay bit-identical to each other.
// Python counterpart: rtx_nerf_amd/dp.py over torch.distributed (which also exchanges the hash table's gradient as sparse lists;
// this host trains the reference's frequency-encoded model, whose only parameters are the MLP's).
// gradient.f32: the first step's summed gradient as it stands before the optimizer (for tests).
// Prints the loss of shard 0 every 20 steps, ms per step, and whether the replicas agree; optionally writes shard 0's parameters.
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

namespace {

template <class T>
T* dev_alloc(size_t n, bool zero = true) {
  T* p = nullptr;
  HIP_CHECK(hipMalloc(&p, n * sizeof(T) > 0 ? n * sizeof(T) : 16));
  if (zero) HIP_CHECK(hipMemset(p, 0, n * sizeof(T)));
  return p;
}

// the one kernel of this host: dst += src (the gradient sum on the root device; everything else is librtxn's)
__global__ void add_into(float* __restrict__ dst, const float* __restrict__ src, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

// as examples/train_host.cpp: pinhole rays of eight hemisphere poses (optixPrograms.cu:43-116, corrected focal length)
void pool_rays(int n_poses, int side, std::vector<float>& o, std::vector<float>& d) {
  const float focal = 1.0f / std::tan(0.5f * 0.6911112f);
  for (int p = 0; p < n_poses; ++p) {
    const float az = (45.0f * p + 15.0f) * 3.14159265f / 180.0f, el = 30.0f * 3.14159265f / 180.0f;
    const float c[3] = {0.4f * std::cos(el) * std::cos(az), 0.4f * std::cos(el) * std::sin(az), 0.4f * std::sin(el)};
    float f[3] = {-c[0], -c[1], -c[2]};
    const float fn = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (float& v : f) v /= fn;
    float r[3] = {f[1], -f[0], 0.0f};
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

struct Shard {
  int device;
  hipStream_t stream;
  hipEvent_t grads_ready, sum_here, copied;   // its gradient is on the root | (root) the sum is complete | it has fetched the sum
  rtxn_mlp* net;
  float *master, *adam_m, *adam_v, *dparams;
  __half* params;
  float *pool_o, *pool_d, *pool_t;          // the shard's own copy of the (shuffled) ray pool
  float *rays_o, *rays_d, *targets, *loss;
  float* staged;                            // this shard's gradient on the ROOT device (== dparams when the shard runs there)
  rtxn_trace_params trace;
  rtxn_train_batch batch;
  void* scan_ws;
  size_t scan_bytes;
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: train_host_mgpu steps batch_rays grid_res devices [out.f32|-]\n");
    return 1;
  }
  const int steps = std::atoi(argv[1]), B = std::atoi(argv[2]), R = std::atoi(argv[3]);
  std::vector<int> devices;
  for (const char* p = argv[4]; *p;) {
    devices.push_back(std::atoi(p));
    while (*p && *p != ',') ++p;
    if (*p == ',') ++p;
  }
  const char* out_path = argc > 5 && std::strcmp(argv[5], "-") != 0 ? argv[5] : nullptr;
  const char* grad_path = argc > 6 ? argv[6] : nullptr;
  const int N = (int)devices.size();
  if (steps < 1 || N < 1 || B <= 0 || B % N != 0) {
    std::fprintf(stderr, "need steps >= 1 and a batch (%d) that divides by the number of shards (%d)\n", B, N);
    return 1;
  }
  const int Bs = B / N;                     // rays per shard
  const int root = devices[0];

  // the ray pool, shuffled once (seed 42): step s takes window s of B rays, shard g rays [g Bs, (g + 1) Bs) of it
  std::vector<float> pool_o, pool_d;
  pool_rays(8, 128, pool_o, pool_d);
  const size_t n_pool = pool_o.size() / 3;
  std::vector<int> order(n_pool);
  for (size_t i = 0; i < n_pool; ++i) order[i] = (int)i;
  std::mt19937 rng(42);
  std::shuffle(order.begin(), order.end(), rng);
  std::vector<float> so(3 * n_pool), sd(3 * n_pool), st(3 * n_pool);
  for (size_t i = 0; i < n_pool; ++i)
    for (int k = 0; k < 3; ++k) {
      so[3 * i + k] = pool_o[3 * (size_t)order[i] + k];
      sd[3 * i + k] = pool_d[3 * (size_t)order[i] + k];
      st[3 * i + k] = 0.5f + 0.5f * sd[3 * i + k];          // a smooth function of the ray: learnable
    }
  const size_t windows = n_pool / (size_t)B;
  if (windows == 0) { std::fprintf(stderr, "batch larger than the ray pool (%zu rays)\n", n_pool); return 1; }

  // ---- per shard: a replica of the model and every per-step buffer at capacity ----
  rtxn_mlp_config cfg = {3, 10, 2, 12, 128, 8, 4, RTXN_ACT_SIGMOID};
  std::vector<Shard> shards((size_t)N);
  long n_params = 0;
  const long capacity = (long)Bs * 3 * R;
  for (int g = 0; g < N; ++g) {
    Shard& s = shards[(size_t)g];
    s.device = devices[(size_t)g];
    HIP_CHECK(hipSetDevice(s.device));
    if (s.device != root) {
      int can = 0;
      HIP_CHECK(hipDeviceCanAccessPeer(&can, s.device, root));
      if (can) (void)hipDeviceEnablePeerAccess(root, 0);      // already enabled by an earlier shard on this device: fine
      (void)hipGetLastError();
    }
    HIP_CHECK(hipStreamCreate(&s.stream));
    HIP_CHECK(hipEventCreateWithFlags(&s.grads_ready, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&s.sum_here, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
    RTXN_CHECK(rtxn_mlp_create(&cfg, &s.net));
    n_params = rtxn_mlp_n_params(s.net);
    std::vector<float> master_h((size_t)n_params);
    RTXN_CHECK(rtxn_mlp_initialize_params(s.net, 1337, master_h.data()));     // the same seed: identical replicas
    std::vector<__half> params_h((size_t)n_params);
    for (long i = 0; i < n_params; ++i) params_h[(size_t)i] = __float2half(master_h[(size_t)i]);
    s.master = dev_alloc<float>((size_t)n_params, false);
    s.params = dev_alloc<__half>((size_t)n_params, false);
    HIP_CHECK(hipMemcpy(s.master, master_h.data(), (size_t)n_params * sizeof(float), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(s.params, params_h.data(), (size_t)n_params * sizeof(__half), hipMemcpyHostToDevice));
    RTXN_CHECK(rtxn_mlp_set_params(s.net, s.params, s.stream));
    s.adam_m = dev_alloc<float>((size_t)n_params);
    s.adam_v = dev_alloc<float>((size_t)n_params);
    s.dparams = dev_alloc<float>((size_t)n_params);
    s.pool_o = dev_alloc<float>(3 * n_pool, false);
    s.pool_d = dev_alloc<float>(3 * n_pool, false);
    s.pool_t = dev_alloc<float>(3 * n_pool, false);
    HIP_CHECK(hipMemcpy(s.pool_o, so.data(), so.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(s.pool_d, sd.data(), sd.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(s.pool_t, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice));
    s.rays_o = dev_alloc<float>(3 * (size_t)Bs);
    s.rays_d = dev_alloc<float>(3 * (size_t)Bs);
    s.targets = dev_alloc<float>(3 * (size_t)Bs);
    s.loss = dev_alloc<float>(1);
    const int sub_rays = 16;
    const long Sp = rtxn_padded_samples(32 * capacity);
    const int E = rtxn_mlp_encoded_width(s.net);
    std::memset(&s.trace, 0, sizeof(s.trace));
    s.trace.rays_o = s.rays_o; s.trace.rays_d = s.rays_d; s.trace.width = (uint32_t)Bs; s.trace.height = 1;
    s.trace.ray_begin = 0; s.trace.ray_count = (uint32_t)Bs;
    s.trace.grid_res = R; s.trace.occupancy = nullptr; s.trace.mode = RTXN_TRACE_DDA;
    s.trace.viewing_direction = dev_alloc<float>(2 * (size_t)Bs);
    s.trace.num_hits = dev_alloc<int>((size_t)Bs);
    s.trace.sub_rays = sub_rays;
    s.trace.sub_hits = dev_alloc<int>((size_t)Bs * sub_rays);
    s.scan_bytes = rtxn_scan_workspace_bytes(Bs);
    s.scan_ws = dev_alloc<char>(s.scan_bytes);
    rtxn_train_batch& b = s.batch;
    std::memset(&b, 0, sizeof(b));
    b.mlp = s.net;
    b.start_points = dev_alloc<float>(3 * (size_t)capacity);
    b.end_points = dev_alloc<float>(3 * (size_t)capacity);
    b.seg_view = dev_alloc<float>(2 * (size_t)capacity);
    b.num_stored = dev_alloc<int>((size_t)Bs);
    b.indices = dev_alloc<int>((size_t)Bs);
    b.total_segments = dev_alloc<int>(1);
    b.segment_capacity = capacity; b.n_rays = Bs;
    b.sample_type = RTXN_SAMPLING_REGULAR; b.t_scale = 1.0f; b.vr_mode = RTXN_VR_COMPAT;
    // a shard's loss is the mean over ITS rays: scaled by 1 / N here, so that every ray's loss gradient is the number the one-device
    // step would round to fp16 (2 (p - t) / (3 B)) and the shards' gradients SUM to the batch mean's
    b.targets = s.targets; b.loss_scale = 1.0f / (float)N;
    b.encT = dev_alloc<__half>((size_t)E * (size_t)Sp);
    b.workspace = dev_alloc<char>(rtxn_mlp_train_lean_workspace_bytes(s.net, 32 * capacity));
    b.workspace_lean = 1;
    b.output_half = dev_alloc<__half>(32 * (size_t)capacity * 16);
    b.radiance = dev_alloc<float>(32 * (size_t)capacity * 4);
    b.t_vals = dev_alloc<float>(32 * (size_t)capacity);
    b.radiance_gradients = dev_alloc<__half>(32 * (size_t)capacity * 4);
    b.pixels = dev_alloc<float>(3 * (size_t)Bs);
    b.loss_gradients_half = dev_alloc<__half>(3 * (size_t)Bs);
    b.loss_sum = s.loss;
    b.dparams = s.dparams;
    b.live_ws = dev_alloc<char>(rtxn_live_segments_workspace_bytes(capacity));
    // where this shard's gradient lands on the root device
    if (s.device == root && g == 0) s.staged = s.dparams;
    else {
      HIP_CHECK(hipSetDevice(root));
      s.staged = dev_alloc<float>((size_t)n_params);
      HIP_CHECK(hipSetDevice(s.device));
    }
  }
  Shard& r0 = shards[0];

  // ---- one data-parallel step ----
  auto step = [&](int it) {
    const size_t w = (size_t)it % windows;
    for (int g = 0; g < N; ++g) {                          // every shard: its rays, traversal, gradients; then its gradient to the root
      Shard& s = shards[(size_t)g];
      HIP_CHECK(hipSetDevice(s.device));
      const size_t first = 3 * (w * (size_t)B + (size_t)g * (size_t)Bs), bytes = 3 * (size_t)Bs * sizeof(float);
      HIP_CHECK(hipMemcpyAsync(s.rays_o, s.pool_o + first, bytes, hipMemcpyDeviceToDevice, s.stream));
      HIP_CHECK(hipMemcpyAsync(s.rays_d, s.pool_d + first, bytes, hipMemcpyDeviceToDevice, s.stream));
      HIP_CHECK(hipMemcpyAsync(s.targets, s.pool_t + first, bytes, hipMemcpyDeviceToDevice, s.stream));
      rtxn_trace_params t = s.trace;                        // count -> scan -> write, as rtxn_train_step does (csrc/trainer.hip)
      RTXN_CHECK(rtxn_trace_grid(&t, s.stream));
      RTXN_CHECK(rtxn_scan_hits(s.trace.num_hits, const_cast<int*>(s.batch.indices), const_cast<int*>(s.batch.total_segments), Bs, s.scan_ws,
                                s.scan_bytes, s.stream));
      t.indices = s.batch.indices;
      t.start_points = const_cast<float*>(s.batch.start_points);
      t.end_points = const_cast<float*>(s.batch.end_points);
      t.seg_view = const_cast<float*>(s.batch.seg_view);
      t.num_stored = const_cast<int*>(s.batch.num_stored);
      t.segment_capacity = capacity;
      RTXN_CHECK(rtxn_trace_grid(&t, s.stream));
      RTXN_CHECK(rtxn_train_gradients(&s.batch, s.stream));
      if (s.staged != s.dparams)
        HIP_CHECK(hipMemcpyPeerAsync(s.staged, root, s.dparams, s.device, (size_t)n_params * sizeof(float), s.stream));
      HIP_CHECK(hipEventRecord(s.grads_ready, s.stream));
    }
    HIP_CHECK(hipSetDevice(root));                          // root: the sum, in shard order, into shard 0's gradient buffer
    for (int g = 1; g < N; ++g) {
      HIP_CHECK(hipStreamWaitEvent(r0.stream, shards[(size_t)g].grads_ready, 0));
      add_into<<<(unsigned)((n_params + 255) / 256), 256, 0, r0.stream>>>(r0.dparams, shards[(size_t)g].staged, n_params);
    }
    HIP_CHECK(hipEventRecord(r0.sum_here, r0.stream));
    if (it == 0 && grad_path) {                            // the summed gradient / N, before anything consumes it
      HIP_CHECK(hipStreamSynchronize(r0.stream));
      std::vector<float> gsum((size_t)n_params);
      HIP_CHECK(hipMemcpy(gsum.data(), r0.dparams, (size_t)n_params * sizeof(float), hipMemcpyDeviceToHost));
      if (FILE* f = std::fopen(grad_path, "wb")) {
        std::fwrite(gsum.data(), sizeof(float), (size_t)n_params, f);
        std::fclose(f);
      }
    }
    for (int g = N - 1; g >= 0; --g) {                     // every shard: the sum, then the same optimizer step on its replica
      Shard& s = shards[(size_t)g];                        // (shard 0 last: it clears the buffer the others fetch the sum from)
      HIP_CHECK(hipSetDevice(s.device));
      if (g > 0) {
        HIP_CHECK(hipStreamWaitEvent(s.stream, r0.sum_here, 0));
        HIP_CHECK(hipMemcpyPeerAsync(s.dparams, s.device, r0.dparams, root, (size_t)n_params * sizeof(float), s.stream));
        HIP_CHECK(hipEventRecord(s.copied, s.stream));
      }
      RTXN_CHECK(rtxn_adam_step(n_params, s.master, s.params, s.dparams, s.adam_m, s.adam_v, it + 1, 1e-3f, 0.9f, 0.999f, 1e-8f,
                                /*loss_scale=*/1.0f, s.stream));
      RTXN_CHECK(rtxn_mlp_set_params_training(s.net, s.params, s.stream));
      if (g == 0)
        for (int k = 1; k < N; ++k) HIP_CHECK(hipStreamWaitEvent(s.stream, shards[(size_t)k].copied, 0));
      HIP_CHECK(hipMemsetAsync(s.dparams, 0, (size_t)n_params * sizeof(float), s.stream));
    }
  };

  auto sync_all = [&]() {
    for (Shard& s : shards) {
      HIP_CHECK(hipSetDevice(s.device));
      HIP_CHECK(hipStreamSynchronize(s.stream));
    }
  };
  float loss_h = 0.0f;
  step(0);
  sync_all();
  HIP_CHECK(hipSetDevice(root));
  HIP_CHECK(hipMemcpy(&loss_h, r0.loss, sizeof(float), hipMemcpyDeviceToHost));
  std::printf("step    0: loss of shard 0 %.6f (%d shards x %d rays)\n", loss_h, N, Bs);
  const auto t0 = std::chrono::steady_clock::now();
  for (int it = 1; it < steps; ++it) {
    step(it);
    if (it % 20 == 0 || it == steps - 1) {
      sync_all();
      HIP_CHECK(hipSetDevice(root));
      HIP_CHECK(hipMemcpy(&loss_h, r0.loss, sizeof(float), hipMemcpyDeviceToHost));
      std::printf("step %4d: loss of shard 0 %.6f\n", it, loss_h);
    }
  }
  sync_all();
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

  // the replicas must agree bit for bit: the same sum, the same optimizer arithmetic
  std::vector<float> ref((size_t)n_params), other((size_t)n_params);
  HIP_CHECK(hipSetDevice(root));
  HIP_CHECK(hipMemcpy(ref.data(), r0.master, (size_t)n_params * sizeof(float), hipMemcpyDeviceToHost));
  bool identical = true;
  for (int g = 1; g < N; ++g) {
    HIP_CHECK(hipSetDevice(shards[(size_t)g].device));
    HIP_CHECK(hipMemcpy(other.data(), shards[(size_t)g].master, (size_t)n_params * sizeof(float), hipMemcpyDeviceToHost));
    identical = identical && std::memcmp(ref.data(), other.data(), (size_t)n_params * sizeof(float)) == 0;
  }
  std::printf("train_host_mgpu: %d steps of %d rays over %d shard(s), %.3f ms per step; replicas %s\n", steps, B, N, steps > 1 ? ms / (steps - 1) : 0.0,
              identical ? "bit-identical" : "DIFFER");
  if (out_path) {
    if (FILE* f = std::fopen(out_path, "wb")) {
      std::fwrite(ref.data(), sizeof(float), (size_t)n_params, f);
      std::fclose(f);
    }
  }
  for (Shard& s : shards) RTXN_CHECK(rtxn_mlp_destroy(s.net));
  return std::isfinite(loss_h) && identical ? 0 : 2;
}
