// C++ multi-GPU host over librtxn.so: one frame ray-sharded over the GPUs of a node, ONE process, no Python.
//
//   render_host_mgpu W H R out.ppm pixels.f32 occupancy.u32|- devices [frames [look_at [focal]]]
//       devices = comma-separated HIP device ordinals, one shard per entry: "0,1,2,3,4,5,6,7" on an 8-GPU node; an ordinal may
//       repeat ("0,0", "0,0,0"): the shards then share that device -- the sharding, the per-shard renderers and the collection
//       of the rows are exercised on ONE GPU exactly as they run on several.
//
// The reference renders on a single device (rtxFunctions.cpp:57); rays are independent (optixPrograms.cu:45,184,241), so a
// frame shards with no data-path collective: shard g of N renders image rows g, g + N, g + 2N, ... -- the strided
// ray window of rtxn_render_config (window_chunk = W, window_stride = N W, ray_begin = g W) -- through its own rtxn_render on
// its own device and stream; what remains is ONE copy of the shard's rows to the root device per frame (0.96 MB per GPU at
// 800 x 800 / 8: latency-bound, so a direct peer copy over the shard's own xGMI link rather than a ring collective), enqueued on
// the shard's COMPOSITOR stream behind the pixels (rtxn_render_frame_async hands that stream back), and N strided row copies
// on the root that interleave the shards into the image.  The same rows could travel by ncclGroupStart / ncclSend / ncclRecv
// (<rccl/rccl.h>) on the same streams; with one process driving every device the peer copy needs no communicator.
// Python counterpart: bench.py --gpus N (one process per GPU, torch.distributed gather; rtx_nerf_amd/shard.py).
// Prints the frame's statistics; exits non-zero if a shard overflowed its segment capacity.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

struct Shard {
  int device;
  unsigned rows;              // image rows g, g + N, ...
  hipStream_t stream;
  rtxn_mlp* net;
  uint32_t* occ;
  float* look_at;
  void* workspace;
  rtxn_render* renderer;
  float* pixels;              // float[rows * W][3] on `device`
  float* staged;              // the same rows on the ROOT device (== pixels when the shard runs there)
  hipEvent_t arrived;         // on the compositor stream: the rows are on the root device
};

// the reference's model (main.cu:35-69, 325-352) with seeded weights, on the current device
rtxn_mlp* make_model(hipStream_t stream) {
  rtxn_mlp_config cfg = {3, 10, 2, 12, 128, 8, 4, RTXN_ACT_SIGMOID};
  rtxn_mlp* net = nullptr;
  RTXN_CHECK(rtxn_mlp_create(&cfg, &net));
  const long n = rtxn_mlp_n_params(net);
  std::vector<float> w(n);
  rtxn_mlp_initialize_params(net, 1337, w.data());
  std::vector<__half> h(n);
  for (long i = 0; i < n; ++i) h[i] = __float2half(w[i]);
  __half* d;
  HIP_CHECK(hipMalloc((void**)&d, n * sizeof(__half)));
  HIP_CHECK(hipMemcpy(d, h.data(), n * sizeof(__half), hipMemcpyHostToDevice));
  RTXN_CHECK(rtxn_mlp_set_params(net, d, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  HIP_CHECK(hipFree(d));      // the model keeps its own packed copies
  return net;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 8) {
    std::fprintf(stderr, "usage: render_host_mgpu W H R out.ppm pixels.f32 occupancy.u32|- devices [frames [look_at [focal]]]\n");
    return 1;
  }
  const unsigned W = std::atoi(argv[1]), H = std::atoi(argv[2]);
  const int R = std::atoi(argv[3]);
  const char *out_path = argv[4], *raw_path = argv[5];
  const char* occ_path = std::strcmp(argv[6], "-") != 0 ? argv[6] : nullptr;
  std::vector<int> devices;
  for (const char* p = argv[7]; *p;) {
    devices.push_back((int)std::strtol(p, const_cast<char**>(&p), 10));
    if (*p == ',') ++p;
  }
  const int N = (int)devices.size();
  const int frames = argc > 8 ? std::atoi(argv[8]) : 1;
  int n_dev = 0;
  HIP_CHECK(hipGetDeviceCount(&n_dev));
  for (int d : devices)
    if (d < 0 || d >= n_dev) { std::fprintf(stderr, "device %d: this node has %d\n", d, n_dev); return 1; }
  if (N < 1 || (unsigned)N > H) { std::fprintf(stderr, "%d shards for %u rows\n", N, H); return 1; }
  const int root = devices[0];

  std::vector<uint32_t> occ_h;
  if (occ_path) {
    occ_h.resize(((size_t)R * R * R + 31) / 32);
    FILE* f = std::fopen(occ_path, "rb");
    if (!f || std::fread(occ_h.data(), 4, occ_h.size(), f) != occ_h.size()) { std::fprintf(stderr, "cannot read the occupancy words from %s\n", occ_path); return 1; }
    std::fclose(f);
  }
  // pose and focal as examples/render_host.cpp (hemisphere pose; corrected focal, quirk Q1), or given
  const float fov_x = 0.6911112f, focal = argc > 10 ? std::strtof(argv[10], nullptr) : 1.0f / std::tan(0.5f * fov_x);
  const double az = 15.0 * M_PI / 180.0, el = -30.0 * M_PI / 180.0, radius = 4.031128874;
  const double cz[3] = {std::cos(el) * std::sin(az), -std::sin(el), std::cos(el) * std::cos(az)};
  double cx[3] = {cz[2], 0.0, -cz[0]};
  const double nx = std::sqrt(cx[0] * cx[0] + cx[2] * cx[2]);
  cx[0] /= nx; cx[2] /= nx;
  const double cy[3] = {cz[1] * cx[2] - cz[2] * cx[1], cz[2] * cx[0] - cz[0] * cx[2], cz[0] * cx[1] - cz[1] * cx[0]};
  float look_at[16] = {(float)cx[0], (float)cy[0], (float)cz[0], (float)(10.0 * radius * cz[0]),
                       (float)cx[1], (float)cy[1], (float)cz[1], (float)(10.0 * radius * cz[1]),
                       (float)cx[2], (float)cy[2], (float)cz[2], (float)(10.0 * radius * cz[2]), 0, 0, 0, 1};
  if (argc > 9) {
    const char* p = argv[9];
    for (int i = 0; i < 16; ++i) { look_at[i] = std::strtof(p, const_cast<char**>(&p)); if (*p == ',') ++p; }
  }

  // ---- the root's image and what every shard needs on its own device ----
  HIP_CHECK(hipSetDevice(root));
  hipStream_t root_stream;
  HIP_CHECK(hipStreamCreate(&root_stream));
  float* image;                                            // float[H][W][3] on the root device
  HIP_CHECK(hipMalloc((void**)&image, (size_t)W * H * 3 * sizeof(float)));
  std::vector<Shard> shards(N);
  for (int g = 0; g < N; ++g) {
    Shard& s = shards[g];
    s.device = devices[g];
    s.rows = (H - g + N - 1) / N;
    HIP_CHECK(hipSetDevice(s.device));
    if (s.device != root) {
      int can = 0;
      HIP_CHECK(hipDeviceCanAccessPeer(&can, s.device, root));
      if (can) (void)hipDeviceEnablePeerAccess(root, 0);   // already enabled by an earlier shard of the same device: not an error here
      (void)hipGetLastError();
    }
    HIP_CHECK(hipStreamCreate(&s.stream));
    HIP_CHECK(hipEventCreateWithFlags(&s.arrived, hipEventDisableTiming));
    s.net = make_model(s.stream);
    s.occ = nullptr;
    if (occ_path) {
      HIP_CHECK(hipMalloc((void**)&s.occ, occ_h.size() * 4));
      HIP_CHECK(hipMemcpy(s.occ, occ_h.data(), occ_h.size() * 4, hipMemcpyHostToDevice));
    }
    HIP_CHECK(hipMalloc((void**)&s.look_at, sizeof(look_at)));
    HIP_CHECK(hipMemcpy(s.look_at, look_at, sizeof(look_at), hipMemcpyHostToDevice));
    const size_t n_local = (size_t)s.rows * W;
    HIP_CHECK(hipMalloc((void**)&s.pixels, n_local * 3 * sizeof(float)));
    s.staged = s.pixels;
    if (s.device != root) {
      HIP_CHECK(hipSetDevice(root));
      HIP_CHECK(hipMalloc((void**)&s.staged, n_local * 3 * sizeof(float)));
      HIP_CHECK(hipSetDevice(s.device));
    }
    // the shard's renderer: rows g, g + N, ... as a strided ray window; capacity from a counting pass
    rtxn_render_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.mlp = s.net;
    cfg.width = W;
    cfg.height = H;
    cfg.focal_length = focal;
    cfg.aspect_ratio = (float)W / (float)H;
    cfg.max_rays = (uint32_t)n_local;
    cfg.window_chunk = N > 1 ? W : 0;
    cfg.window_stride = N > 1 ? (uint32_t)N * W : 0;
    cfg.grid_res = R;
    cfg.occupancy = s.occ;
    cfg.trace_mode = RTXN_TRACE_DDA;
    cfg.sub_rays = n_local >= 300000 ? 2 : 8;
    cfg.vr_mode = RTXN_VR_COMPAT;
    cfg.sample_type = RTXN_SAMPLING_REGULAR;
    cfg.step_scale = 1.0f;
    cfg.n_slots = 3;
    cfg.flags = RTXN_RENDER_STABLE_INPUTS;                 // s.look_at is written once, above
    cfg.max_segments = 1024;
    size_t bytes = rtxn_render_workspace_bytes(&cfg);
    HIP_CHECK(hipMalloc(&s.workspace, bytes));
    RTXN_CHECK(rtxn_render_create(&cfg, s.workspace, bytes, &s.renderer));
    long segments = 0;
    RTXN_CHECK(rtxn_render_count_segments(s.renderer, s.look_at, (uint32_t)g * W, (uint32_t)n_local, &segments, s.stream));
    RTXN_CHECK(rtxn_render_destroy(s.renderer));
    HIP_CHECK(hipFree(s.workspace));
    cfg.max_segments = segments + segments / 10 + 1024;
    bytes = rtxn_render_workspace_bytes(&cfg);
    HIP_CHECK(hipMalloc(&s.workspace, bytes));
    RTXN_CHECK(rtxn_render_create(&cfg, s.workspace, bytes, &s.renderer));
  }

  // one frame: every shard enqueued (nothing waits for the host), rows sent to the root behind each shard's compositor, the
  // root interleaves them when they have arrived
  auto frame = [&]() {
    for (int g = 0; g < N; ++g) {
      Shard& s = shards[g];
      HIP_CHECK(hipSetDevice(s.device));
      rtxn_stream_t comp = nullptr;
      RTXN_CHECK(rtxn_render_frame_async(s.renderer, s.look_at, (uint32_t)g * W, (uint32_t)s.rows * W, s.pixels, s.stream, &comp));
      hipStream_t cs = reinterpret_cast<hipStream_t>(comp);
      if (s.staged != s.pixels)
        HIP_CHECK(hipMemcpyPeerAsync(s.staged, root, s.pixels, s.device, (size_t)s.rows * W * 3 * sizeof(float), cs));
      HIP_CHECK(hipEventRecord(s.arrived, cs));
    }
    HIP_CHECK(hipSetDevice(root));
    for (int g = 0; g < N; ++g) {
      Shard& s = shards[g];
      HIP_CHECK(hipStreamWaitEvent(root_stream, s.arrived, 0));
      // shard row k -> image row g + k N: one strided copy
      HIP_CHECK(hipMemcpy2DAsync(image + (size_t)g * W * 3, (size_t)N * W * 3 * sizeof(float), s.staged, (size_t)W * 3 * sizeof(float),
                                 (size_t)W * 3 * sizeof(float), s.rows, hipMemcpyDeviceToDevice, root_stream));
    }
    // the next frame of a shard may overwrite its pixels / staging only after the root has read them
    hipEvent_t done;
    HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(done, root_stream));
    for (int g = 0; g < N; ++g) {
      HIP_CHECK(hipSetDevice(shards[g].device));
      HIP_CHECK(hipStreamWaitEvent(shards[g].stream, done, 0));
    }
    HIP_CHECK(hipEventDestroy(done));
  };
  auto wait_all = [&]() {
    for (int g = 0; g < N; ++g) {
      HIP_CHECK(hipSetDevice(shards[g].device));
      RTXN_CHECK(rtxn_render_drain(shards[g].renderer, shards[g].stream));
      HIP_CHECK(hipStreamSynchronize(shards[g].stream));
    }
    HIP_CHECK(hipSetDevice(root));
    HIP_CHECK(hipStreamSynchronize(root_stream));
  };

  frame();
  wait_all();
  std::vector<float> pixels((size_t)W * H * 3);
  HIP_CHECK(hipMemcpy(pixels.data(), image, pixels.size() * sizeof(float), hipMemcpyDeviceToHost));
  double ms_per_frame = 0.0;
  if (frames > 1) {
    for (int i = 0; i < 2; ++i) frame();
    wait_all();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < frames; ++i) frame();
    wait_all();
    ms_per_frame = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
  }
  long segments = 0, overflowed = 0;
  for (int g = 0; g < N; ++g) {
    HIP_CHECK(hipSetDevice(shards[g].device));
    rtxn_render_stats st;
    RTXN_CHECK(rtxn_render_status(shards[g].renderer, 1, &st));
    segments += st.last_segments;
    overflowed += st.overflow_frames;
  }
  double sum = 0;
  for (float v : pixels) sum += v;
  if (FILE* f = std::fopen(out_path, "wb")) {
    std::fprintf(f, "P6\n%u %u\n255\n", W, H);
    for (float v : pixels) std::fputc((int)std::lround(255.0f * std::fmin(std::fmax(v, 0.0f), 1.0f)), f);
    std::fclose(f);
  }
  if (FILE* f = std::fopen(raw_path, "wb")) {
    std::fwrite(pixels.data(), sizeof(float), pixels.size(), f);
    std::fclose(f);
  }
  std::printf("render_host_mgpu: %u x %u rays over %d shards (devices %s), %ld segments, %ld overflowed frames, mean pixel %.6f -> %s\n", W, H, N, argv[7],
              segments, overflowed, sum / (double)pixels.size(), out_path);
  if (frames > 1) std::printf("render_host_mgpu: %d frames, %.3f ms/frame, %.2f Mrays/s\n", frames, ms_per_frame, (double)W * H / ms_per_frame / 1e3);
  for (int g = 0; g < N; ++g) {
    HIP_CHECK(hipSetDevice(shards[g].device));
    RTXN_CHECK(rtxn_render_destroy(shards[g].renderer));
    rtxn_mlp_destroy(shards[g].net);
  }
  return (sum > 0 && overflowed == 0) ? 0 : 2;
}
