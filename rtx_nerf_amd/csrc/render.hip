// One frame behind one call: rtxn_render_*.
// The counterpart of the reference's per-image host sequence -- Params fill + optixLaunch (main.cu:473-508), the host copy
// and re-pack of every traversal buffer (:510-543, :646-673), thrust compaction (:631-637), launchSampler (:704),
// network->forward (:721), convertHalfToFloat (:723-728), launch_volrender_cuda (:737) -- with the host taken out of it:
//   trace(count) -> scan -> trace(write packed CSR) -> sampler+encode+MLP (one kernel) -> composite
// are enqueued back to back; the segment count stays on the device (the MLP kernel bounds its persistent loop with it, the
// write pass clamps to the capacity), and the only thing that travels to the host is a 4-byte copy of that count into pinned
// memory which a LATER call looks at (overflow report without polling).  No kernels live here: this file only sequences the
// stage entry points of librtxn.so on caller-given streams, exactly as a C++ host would (examples/render_host.cpp), and is
// what rtx_nerf_amd/render.py calls.  (One single-thread bookkeeping kernel is the exception: frame_account_kernel keeps the
// per-slot frame / overflow counters on the device so that a replayed hipGraph is counted per replay.)
//
// rtxn_render_frame_async is the software-pipelined form: traversal of frame i+1 and compositing of frame
// i-1 run on two internal streams underneath the MLP kernel of frame i, over n_slots buffer slots.
#include "common.h"

#include <cstring>
#include <new>

#include "mlp_internal.h"

namespace {

constexpr int kMaxSlots = 4;
constexpr size_t kAlign = 256;
constexpr size_t kPinnedPerSlot = 4 + 16;     // 32-bit words: int[4] counters, float[16] pose staging

struct Slot {
  float* look_at;
  float* view_dirs;
  int* num_hits;
  int* num_stored;
  int* indices;
  int* total;
  int* acct;          // int[4]: {segments of the last frame, frames, frames over capacity, largest count}: frame_account_kernel
  void* scan_ws;      // per slot: two frames on two streams and slots must not share scan partials
  int* sub_hits;
  float* start;
  float* end;
  float* seg_view;
  void* radiance;     // half4[m*32] (compact) or float4[m*32]
  float* t_vals;      // float[m*32]: RTXN_RENDER_FLOAT4 only
  float* seg_step;    // float[m]: compact RTXN_VR_NERF only
  // host side
  int* acct_host;     // pinned int[4], copy of acct
  float* pose_host;   // pinned float[16]: staging of rtxn_render_frame_async_host's pose
  hipEvent_t total_ev, ev_geo, ev_mlp, ev_comp, ev_pose;
  bool total_pending, used, pose_pending;
  int seen_frames, seen_overflows;
};

size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

// carve the workspace; base == nullptr: only measure
struct Carver {
  uint8_t* base;
  size_t off;
  template <class T>
  T* take(size_t n) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += align_up(n * sizeof(T));
    return p;
  }
};

}  // namespace

struct rtxn_render {
  rtxn_render_config cfg;
  uint32_t max_rays;
  int sub_rays;
  bool compact, hash;
  Slot slots[kMaxSlots];
  int n_slots;
  uint32_t* coarse;
  uint32_t* super_mip;
  uint64_t* bricks;
  size_t scan_ws_bytes;
  hipStream_t geo, comp;
  hipEvent_t ev_tmp[2], ev_occ;
  bool occ_dep;         // rtxn_render_set_occupancy rebuilt the hierarchy on a caller stream: the next async traversal waits for ev_occ
  bool stable_inputs;   // RTXN_RENDER_STABLE_INPUTS
  int* pinned;          // n_slots x (int[4] counters + float[16] pose staging)
  long frame;           // async frames enqueued
  rtxn_render_stats st;
};

namespace {

int validate(const rtxn_render_config* c, const char* who) {
  RTXN_REQUIRE(c != nullptr, "%s: NULL config", who);
  RTXN_REQUIRE(c->mlp != nullptr, "%s: NULL model", who);
  RTXN_REQUIRE(c->width > 0 && c->height > 0 && (uint64_t)c->width * c->height <= 0x7fffffffull, "%s: launch %u x %u", who, c->width, c->height);
  RTXN_REQUIRE(c->grid_res >= 1 && c->grid_res <= 1024, "%s: grid_res = %d out of [1,1024]", who, c->grid_res);
  RTXN_REQUIRE(c->trace_mode == RTXN_TRACE_COMPAT || c->trace_mode == RTXN_TRACE_DDA, "%s: unknown trace_mode %d", who, c->trace_mode);
  RTXN_REQUIRE(c->vr_mode == RTXN_VR_COMPAT || c->vr_mode == RTXN_VR_NERF, "%s: unknown vr_mode %d", who, c->vr_mode);
  RTXN_REQUIRE(c->sample_type == RTXN_SAMPLING_REGULAR || c->sample_type == RTXN_SAMPLING_MIDPOINT_WORLD,
               "%s: sample_type %d (the deterministic modes only: REGULAR, MIDPOINT_WORLD)", who, c->sample_type);
  RTXN_REQUIRE(c->max_segments > 0 && c->max_segments <= (1L << 31) / 32 * 31, "%s: max_segments = %ld", who, c->max_segments);
  RTXN_REQUIRE(c->n_slots >= 1 && c->n_slots <= kMaxSlots, "%s: n_slots = %d out of [1,%d]", who, c->n_slots, kMaxSlots);
  RTXN_REQUIRE((c->flags & ~(RTXN_RENDER_FLOAT4 | RTXN_RENDER_STABLE_INPUTS)) == 0, "%s: unknown flags 0x%x", who, c->flags);
  RTXN_REQUIRE(c->sub_rays >= 0 && c->sub_rays <= 64 && (c->sub_rays & (c->sub_rays - 1)) == 0, "%s: sub_rays = %d must be 0 or a power of two up to 64", who, c->sub_rays);
  const uint64_t launch = (uint64_t)c->width * c->height;
  RTXN_REQUIRE(c->max_rays <= launch, "%s: max_rays = %u exceeds the %u x %u launch", who, c->max_rays, c->width, c->height);
  if (c->grid) {
    RTXN_REQUIRE(c->table_fp16 != nullptr, "%s: hash grid without a table", who);
    RTXN_REQUIRE(c->mlp->cfg.encoding == RTXN_ENC_EXTERNAL, "%s: a hash-grid renderer needs a pre-encoded (RTXN_ENC_EXTERNAL) model", who);
    RTXN_REQUIRE(rtxn_hashgrid_encoded_width(c->grid, c->n_dir_freqs) == c->mlp->enc_padded,
                 "%s: the grid encodes %d features, the model takes %d", who, rtxn_hashgrid_encoded_width(c->grid, c->n_dir_freqs), c->mlp->enc_padded);
    if (c->flags & RTXN_RENDER_FLOAT4) {
      rtxn::set_error("%s: RTXN_RENDER_FLOAT4 is the frequency model's reference-layout hand-over; the hash-grid kernel writes half4 only", who);
      return RTXN_ERR_UNSUPPORTED;
    }
  } else {
    RTXN_REQUIRE(c->mlp->cfg.encoding == RTXN_ENC_FREQUENCY && c->mlp->variant >= 0, "%s: without a grid the model must carry the Composite-Frequency encoding", who);
    RTXN_REQUIRE(c->sample_type == RTXN_SAMPLING_REGULAR, "%s: the frequency model's fused kernel samples REGULAR only", who);
    RTXN_REQUIRE(c->vr_mode == RTXN_VR_COMPAT || (c->flags & RTXN_RENDER_FLOAT4), "%s: RTXN_VR_NERF with the frequency model needs RTXN_RENDER_FLOAT4 (t_vals)", who);
  }
  return RTXN_OK;
}

uint32_t max_rays_of(const rtxn_render_config* c) { return c->max_rays ? c->max_rays : c->width * c->height; }

// Lay the renderer's buffers out in the workspace (r == nullptr: measure only).
size_t layout(const rtxn_render_config* c, uint8_t* base, rtxn_render* r) {
  Carver cv{base, 0};
  const size_t n = max_rays_of(c), m = (size_t)c->max_segments, K = RTXN_NUM_SAMPLES_PER_SEGMENT;
  const bool compact = !(c->flags & RTXN_RENDER_FLOAT4);
  const int Q = c->trace_mode == RTXN_TRACE_DDA && c->sub_rays > 1 ? c->sub_rays : 1;
  const int R = c->grid_res;
  const bool mips = c->occupancy && c->trace_mode == RTXN_TRACE_DDA && R % 4 == 0;
  uint32_t* coarse = nullptr;
  uint64_t* bricks = nullptr;
  uint32_t* super_mip = nullptr;
  if (mips) {
    const size_t rc = R / 4;
    coarse = cv.take<uint32_t>((rc * rc * rc + 31) / 32);
    bricks = cv.take<uint64_t>(rc * rc * rc);
    if (R % 16 == 0) super_mip = cv.take<uint32_t>(((rc / 4) * (rc / 4) * (rc / 4) + 31) / 32);
  }
  const size_t ws_bytes = rtxn_scan_workspace_bytes((int)n);
  if (r) { r->coarse = coarse; r->bricks = bricks; r->super_mip = super_mip; r->scan_ws_bytes = ws_bytes; }
  for (int i = 0; i < c->n_slots; ++i) {
    Slot s;
    memset(&s, 0, sizeof(s));
    s.look_at = cv.take<float>(16);
    s.view_dirs = cv.take<float>(2 * n);
    s.num_hits = cv.take<int>(n);
    s.num_stored = cv.take<int>(n);
    s.indices = cv.take<int>(n);
    s.total = cv.take<int>(1);
    s.acct = cv.take<int>(4);
    s.scan_ws = cv.take<uint8_t>(ws_bytes);
    s.sub_hits = Q > 1 ? cv.take<int>(n * Q) : nullptr;
    s.start = cv.take<float>(3 * m);
    s.end = cv.take<float>(3 * m);
    s.seg_view = cv.take<float>(2 * m);
    if (compact) {
      s.radiance = cv.take<uint8_t>(m * K * 8);
      s.seg_step = c->vr_mode == RTXN_VR_NERF ? cv.take<float>(m) : nullptr;
    } else {
      s.radiance = cv.take<uint8_t>(m * K * 16);
      s.t_vals = cv.take<float>(m * K);
    }
    if (r) r->slots[i] = s;
  }
  return cv.off;
}

int build_hierarchy(rtxn_render* r, hipStream_t s) {
  const rtxn_render_config& c = r->cfg;
  if (!r->coarse) return RTXN_OK;
  int rc = rtxn_build_occupancy_mip(c.occupancy, c.grid_res, r->coarse, s);
  if (rc != RTXN_OK) return rc;
  rc = rtxn_build_occupancy_bricks(c.occupancy, c.grid_res, r->bricks, s);
  if (rc != RTXN_OK) return rc;
  if (r->super_mip) rc = rtxn_build_occupancy_mip(r->coarse, c.grid_res / 4, r->super_mip, s);
  return rc;
}

int check_window(const rtxn_render* r, uint32_t ray_begin, uint32_t& ray_count, const char* who) {
  if (ray_count == 0) ray_count = r->max_rays;
  RTXN_REQUIRE(ray_count <= r->max_rays, "%s: ray_count = %u exceeds max_rays = %u", who, ray_count, r->max_rays);
  (void)ray_begin;   // range-checked against the launch by rtxn_trace_grid
  return RTXN_OK;
}

void trace_params(const rtxn_render* r, const Slot& g, uint32_t ray_begin, uint32_t ray_count, bool write, rtxn_trace_params& p) {
  const rtxn_render_config& c = r->cfg;
  memset(&p, 0, sizeof(p));
  p.look_at = g.look_at;
  p.focal_length = c.focal_length;
  p.aspect_ratio = c.aspect_ratio;
  p.width = c.width;
  p.height = c.height;
  p.ray_begin = ray_begin;
  p.ray_count = ray_count;
  p.window_chunk = c.window_chunk;
  p.window_stride = c.window_stride;
  p.grid_res = c.grid_res;
  p.occupancy = c.occupancy;
  p.occupancy_coarse = r->coarse;
  p.occupancy_bricks = r->bricks;
  p.occupancy_super = r->super_mip;
  p.mode = c.trace_mode;
  p.viewing_direction = g.view_dirs;
  p.num_hits = g.num_hits;
  p.sub_rays = r->sub_rays;
  p.sub_hits = g.sub_hits;
  if (write) {
    p.indices = g.indices;
    p.start_points = g.start;
    p.end_points = g.end;
    p.seg_view = g.seg_view;
    p.num_stored = g.num_stored;
    p.segment_capacity = c.max_segments;
  }
}

// Per-slot counters live on the device and are cumulative, so any copy of them that has reached the host is exact however many
// frames (or graph replays) ran since the last look: {segments of the last frame, frames, frames over capacity, largest count}.
__global__ void frame_account_kernel(const int* __restrict__ total, int capacity, int* __restrict__ acct) {
  const int t = *total;
  acct[0] = t;
  acct[1] += 1;
  acct[2] += t > capacity ? 1 : 0;
  acct[3] = t > acct[3] ? t : acct[3];
}

// Fold what slot g's counters say into the renderer's statistics.
void fold(rtxn_render* r, Slot& g) {
  const int* a = g.acct_host;
  r->st.frames_checked += a[1] - g.seen_frames;
  r->st.overflow_frames += a[2] - g.seen_overflows;
  if (a[1] != g.seen_frames) r->st.last_segments = a[0];
  g.seen_frames = a[1];
  g.seen_overflows = a[2];
  if (a[3] > r->st.max_segments_needed) r->st.max_segments_needed = a[3];
}

// Look at the counters slot g's LAST frame reported, if they have arrived (no synchronisation).
void harvest(rtxn_render* r, Slot& g, bool force) {
  if (!g.total_pending) return;
  if (!force && hipEventQuery(g.total_ev) != hipSuccess) return;
  g.total_pending = false;
  fold(r, g);
}

long c_max_segments(const rtxn_render* r) { return r->cfg.max_segments > 0x7fffffffL ? 0x7fffffffL : r->cfg.max_segments; }

bool capturing(hipStream_t s) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}

// count -> scan -> write of one frame into slot g
int geometry(rtxn_render* r, Slot& g, const float* look_at, bool pose_on_host, uint32_t ray_begin, uint32_t n, hipStream_t s) {
  RTXN_HIP(hipMemcpyAsync(g.look_at, look_at, 16 * sizeof(float), pose_on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
  if (pose_on_host) {
    RTXN_HIP(hipEventRecord(g.ev_pose, s));
    g.pose_pending = true;
  }
  rtxn_trace_params p;
  trace_params(r, g, ray_begin, n, false, p);
  int rc = rtxn_trace_grid(&p, s);
  if (rc != RTXN_OK) return rc;
  rc = rtxn_scan_hits(g.num_hits, g.indices, g.total, (int)n, g.scan_ws, r->scan_ws_bytes, s);
  if (rc != RTXN_OK) return rc;
  trace_params(r, g, ray_begin, n, true, p);
  rc = rtxn_trace_grid(&p, s);
  if (rc != RTXN_OK) return rc;
  // 16 bytes of counters to pinned memory: what a later call's overflow check reads.  Under stream capture the kernel and the
  // copy node are part of the graph (every replay counts itself) but there is no event to poll: rtxn_render_status(wait = 1)
  // reads them.
  frame_account_kernel<<<1, 1, 0, s>>>(g.total, (int)(c_max_segments(r)), g.acct);
  RTXN_LAUNCH_CHECK("frame_account_kernel");
  RTXN_HIP(hipMemcpyAsync(g.acct_host, g.acct, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  if (!capturing(s)) {
    RTXN_HIP(hipEventRecord(g.total_ev, s));
    g.total_pending = true;
  }
  r->st.frames++;
  return RTXN_OK;
}

// sampler + encode + MLP over slot g's packed segments
int shade(rtxn_render* r, Slot& g, hipStream_t s) {
  const rtxn_render_config& c = r->cfg;
  if (r->hash)
    return rtxn_hashmlp_forward_segments(c.mlp, c.grid, c.n_dir_freqs, c.table_fp16, g.start, g.end, g.seg_view, g.total, c.max_segments,
                                         c.sample_type, c.step_scale, g.radiance, g.seg_step, s);
  if (r->compact) return rtxn_mlp_forward_segments_compact(c.mlp, g.start, g.end, g.seg_view, g.total, c.max_segments, g.radiance, s);
  return rtxn_mlp_forward_segments(c.mlp, g.start, g.end, g.seg_view, g.total, c.max_segments, static_cast<float*>(g.radiance), g.t_vals, s);
}

// rays whose segments would overflow the capacity were truncated by the write pass: the compositor reads num_stored
int composite(rtxn_render* r, Slot& g, uint32_t n, float* pixels, hipStream_t s) {
  const rtxn_render_config& c = r->cfg;
  const int K = RTXN_NUM_SAMPLES_PER_SEGMENT;
  if (r->compact) {
    if (c.vr_mode == RTXN_VR_NERF) return rtxn_volrender_fwd_compact_nerf(g.radiance, g.seg_step, g.num_stored, g.indices, (int)n, K, pixels, s);
    return rtxn_volrender_fwd_compact(g.radiance, g.num_stored, g.indices, (int)n, K, pixels, s);
  }
  return rtxn_volrender_fwd(nullptr, static_cast<const float*>(g.radiance), g.num_stored, g.indices, g.t_vals, (int)n, K, pixels, c.vr_mode, s);
}

}  // namespace

extern "C" size_t rtxn_render_workspace_bytes(const rtxn_render_config* cfg) {
  if (validate(cfg, "rtxn_render_workspace_bytes") != RTXN_OK) return 0;
  return layout(cfg, nullptr, nullptr);
}

extern "C" int rtxn_render_create(const rtxn_render_config* cfg, void* workspace, size_t workspace_bytes, rtxn_render** out) {
  RTXN_REQUIRE(out != nullptr, "rtxn_render_create: NULL out");
  int rc = validate(cfg, "rtxn_render_create");
  if (rc != RTXN_OK) return rc;
  RTXN_DEVICE_OR_FAIL();
  const size_t need = layout(cfg, nullptr, nullptr);
  RTXN_REQUIRE(workspace && ((uintptr_t)workspace & (kAlign - 1)) == 0, "rtxn_render_create: workspace NULL or not %zu-byte aligned", kAlign);
  RTXN_REQUIRE(workspace_bytes >= need, "rtxn_render_create: workspace holds %zu bytes, rtxn_render_workspace_bytes says %zu", workspace_bytes, need);
  rtxn_render* r = new (std::nothrow) rtxn_render();
  RTXN_REQUIRE(r != nullptr, "rtxn_render_create: out of host memory");
  memset(static_cast<void*>(r), 0, sizeof(*r));
  r->cfg = *cfg;
  r->max_rays = max_rays_of(cfg);
  r->sub_rays = cfg->trace_mode == RTXN_TRACE_DDA && cfg->sub_rays > 1 ? cfg->sub_rays : 0;
  r->compact = !(cfg->flags & RTXN_RENDER_FLOAT4);
  r->hash = cfg->grid != nullptr;
  r->n_slots = cfg->n_slots;
  r->stable_inputs = (cfg->flags & RTXN_RENDER_STABLE_INPUTS) != 0;
  r->st.max_segments = cfg->max_segments;
  layout(cfg, static_cast<uint8_t*>(workspace), r);
  auto fail = [&](int code) { rtxn_render_destroy(r); return code; };
  hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&r->pinned), kPinnedPerSlot * sizeof(int) * kMaxSlots, hipHostMallocDefault);
  if (e != hipSuccess) return fail(rtxn::fail_hip(e, "hipHostMalloc(pinned segment counts)"));
  if ((e = hipStreamCreateWithFlags(&r->geo, hipStreamNonBlocking)) != hipSuccess) return fail(rtxn::fail_hip(e, "hipStreamCreate(geometry)"));
  if ((e = hipStreamCreateWithFlags(&r->comp, hipStreamNonBlocking)) != hipSuccess) return fail(rtxn::fail_hip(e, "hipStreamCreate(composite)"));
  for (int i = 0; i < 2; ++i)
    if ((e = hipEventCreateWithFlags(&r->ev_tmp[i], hipEventDisableTiming)) != hipSuccess) return fail(rtxn::fail_hip(e, "hipEventCreate"));
  if ((e = hipEventCreateWithFlags(&r->ev_occ, hipEventDisableTiming)) != hipSuccess) return fail(rtxn::fail_hip(e, "hipEventCreate"));
  for (int i = 0; i < r->n_slots; ++i) {
    Slot& g = r->slots[i];
    g.acct_host = r->pinned + kPinnedPerSlot * i;
    g.pose_host = reinterpret_cast<float*>(g.acct_host + 4);
    memset(g.acct_host, 0, kPinnedPerSlot * sizeof(int));
    hipEvent_t* evs[5] = {&g.total_ev, &g.ev_geo, &g.ev_mlp, &g.ev_comp, &g.ev_pose};
    for (hipEvent_t* ev : evs)
      if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return fail(rtxn::fail_hip(e, "hipEventCreate"));
  }
  // per-slot totals and counters start at 0 (a status query before the first frame reads them), and the hierarchy is built once here
  for (int i = 0; i < r->n_slots; ++i) {
    if ((e = rtxn::zero_words(r->slots[i].total, 1, nullptr)) != hipSuccess) return fail(rtxn::fail_hip(e, "zero_words(total)"));
    if ((e = rtxn::zero_words(r->slots[i].acct, 4, nullptr)) != hipSuccess) return fail(rtxn::fail_hip(e, "zero_words(acct)"));
  }
  rc = build_hierarchy(r, nullptr);
  if (rc != RTXN_OK) return fail(rc);
  if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) return fail(rtxn::fail_hip(e, "hipStreamSynchronize"));
  *out = r;
  return RTXN_OK;
}

extern "C" int rtxn_render_destroy(rtxn_render* r) {
  if (!r) return RTXN_OK;
  (void)hipDeviceSynchronize();      // frames in flight still use the slots' events
  for (int i = 0; i < r->n_slots; ++i) {
    Slot& g = r->slots[i];
    hipEvent_t evs[5] = {g.total_ev, g.ev_geo, g.ev_mlp, g.ev_comp, g.ev_pose};
    for (hipEvent_t ev : evs)
      if (ev) (void)hipEventDestroy(ev);
  }
  if (r->ev_occ) (void)hipEventDestroy(r->ev_occ);
  for (int i = 0; i < 2; ++i)
    if (r->ev_tmp[i]) (void)hipEventDestroy(r->ev_tmp[i]);
  if (r->geo) (void)hipStreamDestroy(r->geo);
  if (r->comp) (void)hipStreamDestroy(r->comp);
  if (r->pinned) (void)hipHostFree(r->pinned);
  delete r;
  return RTXN_OK;
}

extern "C" int rtxn_render_set_occupancy(rtxn_render* r, const uint32_t* occupancy, rtxn_stream_t stream) {
  RTXN_REQUIRE(r != nullptr, "rtxn_render_set_occupancy: NULL renderer");
  RTXN_REQUIRE((occupancy != nullptr) == (r->cfg.occupancy != nullptr),
               "rtxn_render_set_occupancy: a renderer created %s an occupancy grid cannot switch (the hierarchy's buffers are laid out at creation)",
               r->cfg.occupancy ? "with" : "without");
  RTXN_DEVICE_OR_FAIL();
  hipStream_t s = rtxn::as_stream(stream);
  if (!capturing(s)) {
    // traversals still in flight on the internal stream read the hierarchy this rebuilds: `stream` waits for them, and the
    // NEXT pipelined traversal waits for the rebuild (and for whatever the caller put on `stream` before it -- the copy of
    // the new bits, typically)
    RTXN_HIP(hipEventRecord(r->ev_tmp[0], r->geo));
    RTXN_HIP(hipStreamWaitEvent(s, r->ev_tmp[0], 0));
  }
  r->cfg.occupancy = occupancy;
  int rc = build_hierarchy(r, s);
  if (rc != RTXN_OK) return rc;
  if (!capturing(s)) {
    RTXN_HIP(hipEventRecord(r->ev_occ, s));
    r->occ_dep = true;
  }
  return RTXN_OK;
}

extern "C" int rtxn_render_count_segments(rtxn_render* r, const float* look_at, uint32_t ray_begin, uint32_t ray_count, long* segments,
                                          rtxn_stream_t stream) {
  RTXN_REQUIRE(r && look_at && segments, "rtxn_render_count_segments: NULL argument");
  int rc = check_window(r, ray_begin, ray_count, "rtxn_render_count_segments");
  if (rc != RTXN_OK) return rc;
  RTXN_DEVICE_OR_FAIL();
  hipStream_t s = rtxn::as_stream(stream);
  Slot& g = r->slots[0];
  RTXN_HIP(hipMemcpyAsync(g.look_at, look_at, 16 * sizeof(float), hipMemcpyDeviceToDevice, s));
  rtxn_trace_params p;
  trace_params(r, g, ray_begin, ray_count, false, p);
  rc = rtxn_trace_grid(&p, s);
  if (rc != RTXN_OK) return rc;
  rc = rtxn_scan_hits(g.num_hits, g.indices, g.total, (int)ray_count, g.scan_ws, r->scan_ws_bytes, s);
  if (rc != RTXN_OK) return rc;
  int total = 0;
  RTXN_HIP(hipMemcpyAsync(&total, g.total, sizeof(int), hipMemcpyDeviceToHost, s));
  RTXN_HIP(hipStreamSynchronize(s));
  *segments = total;
  return RTXN_OK;
}

extern "C" int rtxn_render_frame(rtxn_render* r, int slot, const float* look_at, uint32_t ray_begin, uint32_t ray_count, float* pixels,
                                 rtxn_stream_t stream) {
  RTXN_REQUIRE(r && look_at && pixels, "rtxn_render_frame: NULL argument");
  RTXN_REQUIRE(slot >= 0 && slot < r->n_slots, "rtxn_render_frame: slot %d out of [0,%d)", slot, r->n_slots);
  int rc = check_window(r, ray_begin, ray_count, "rtxn_render_frame");
  if (rc != RTXN_OK) return rc;
  RTXN_DEVICE_OR_FAIL();
  hipStream_t s = rtxn::as_stream(stream);
  Slot& g = r->slots[slot];
  if (!capturing(s)) harvest(r, g, false);
  rc = geometry(r, g, look_at, false, ray_begin, ray_count, s);
  if (rc != RTXN_OK) return rc;
  rc = shade(r, g, s);
  if (rc != RTXN_OK) return rc;
  return composite(r, g, ray_count, pixels, s);
}

namespace {

// What the traversal of an async frame has to wait for before it may read its inputs (look_at, the occupancy bits):
//  * the slot's previous frame (its MLP kernel and compositor still read the slot's buffers);
//  * by default EVERYTHING the caller enqueued on `stream` before this call: a host that rewrites one device pose buffer per
//    frame on `stream`, or copies new occupancy bits there, is then ordered correctly -- at the price that the traversal of
//    frame i+1 starts only when the MLP kernel of frame i (also on `stream`) has finished;
//  * with RTXN_RENDER_STABLE_INPUTS (the caller's promise that a frame's inputs are complete and stay untouched from the call
//    until the frame's traversal has run: pre-uploaded poses) or with a HOST pose (staged through pinned memory here): only on
//    a slot's first use and after rtxn_render_set_occupancy -- the fully overlapped pipeline.
int frame_async(rtxn_render* r, const float* look_at, bool pose_on_host, uint32_t ray_begin, uint32_t ray_count, float* pixels,
                hipStream_t main_s, rtxn_stream_t* composite_stream) {
  const int b = (int)(r->frame % r->n_slots);
  r->frame++;
  Slot& g = r->slots[b];
  harvest(r, g, false);                                   // the frame that used this slot n_slots frames ago
  if (g.used) RTXN_HIP(hipStreamWaitEvent(r->geo, g.ev_comp, 0));
  const bool ordered = !g.used || !(r->stable_inputs || pose_on_host);
  if (ordered) {
    RTXN_HIP(hipEventRecord(r->ev_tmp[0], main_s));
    RTXN_HIP(hipStreamWaitEvent(r->geo, r->ev_tmp[0], 0));
  }
  if (r->occ_dep) {                                       // one wait orders every later traversal: the stream is in order
    RTXN_HIP(hipStreamWaitEvent(r->geo, r->ev_occ, 0));
    r->occ_dep = false;
  }
  if (pose_on_host) {
    if (g.pose_pending) RTXN_HIP(hipEventSynchronize(g.ev_pose));   // the staging words' previous copy (n_slots frames ago)
    memcpy(g.pose_host, look_at, 16 * sizeof(float));
    look_at = g.pose_host;
  }
  int rc = geometry(r, g, look_at, pose_on_host, ray_begin, ray_count, r->geo);
  if (rc != RTXN_OK) return rc;
  RTXN_HIP(hipEventRecord(g.ev_geo, r->geo));
  RTXN_HIP(hipStreamWaitEvent(main_s, g.ev_geo, 0));
  if (g.used) RTXN_HIP(hipStreamWaitEvent(main_s, g.ev_comp, 0));   // this slot's radiance was last read by compositor i - n_slots
  rc = shade(r, g, main_s);
  if (rc != RTXN_OK) return rc;
  RTXN_HIP(hipEventRecord(g.ev_mlp, main_s));
  RTXN_HIP(hipStreamWaitEvent(r->comp, g.ev_mlp, 0));
  rc = composite(r, g, ray_count, pixels, r->comp);
  if (rc != RTXN_OK) return rc;
  RTXN_HIP(hipEventRecord(g.ev_comp, r->comp));
  g.used = true;
  if (composite_stream) *composite_stream = r->comp;
  return RTXN_OK;
}

}  // namespace

extern "C" int rtxn_render_frame_async(rtxn_render* r, const float* look_at, uint32_t ray_begin, uint32_t ray_count, float* pixels,
                                       rtxn_stream_t stream, rtxn_stream_t* composite_stream) {
  RTXN_REQUIRE(r && look_at && pixels, "rtxn_render_frame_async: NULL argument");
  int rc = check_window(r, ray_begin, ray_count, "rtxn_render_frame_async");
  if (rc != RTXN_OK) return rc;
  RTXN_DEVICE_OR_FAIL();
  hipStream_t main_s = rtxn::as_stream(stream);
  RTXN_REQUIRE(!capturing(main_s), "rtxn_render_frame_async: not capturable (internal streams); capture rtxn_render_frame instead");
  return frame_async(r, look_at, false, ray_begin, ray_count, pixels, main_s, composite_stream);
}

extern "C" int rtxn_render_frame_async_host(rtxn_render* r, const float* look_at_host, uint32_t ray_begin, uint32_t ray_count,
                                            float* pixels, rtxn_stream_t stream, rtxn_stream_t* composite_stream) {
  RTXN_REQUIRE(r && look_at_host && pixels, "rtxn_render_frame_async_host: NULL argument");
  int rc = check_window(r, ray_begin, ray_count, "rtxn_render_frame_async_host");
  if (rc != RTXN_OK) return rc;
  RTXN_DEVICE_OR_FAIL();
  hipStream_t main_s = rtxn::as_stream(stream);
  RTXN_REQUIRE(!capturing(main_s), "rtxn_render_frame_async_host: not capturable (internal streams, host staging)");
  return frame_async(r, look_at_host, true, ray_begin, ray_count, pixels, main_s, composite_stream);
}

extern "C" int rtxn_render_drain(rtxn_render* r, rtxn_stream_t stream) {
  RTXN_REQUIRE(r != nullptr, "rtxn_render_drain: NULL renderer");
  RTXN_DEVICE_OR_FAIL();
  hipStream_t s = rtxn::as_stream(stream);
  RTXN_HIP(hipEventRecord(r->ev_tmp[0], r->geo));
  RTXN_HIP(hipEventRecord(r->ev_tmp[1], r->comp));
  RTXN_HIP(hipStreamWaitEvent(s, r->ev_tmp[0], 0));
  RTXN_HIP(hipStreamWaitEvent(s, r->ev_tmp[1], 0));
  return RTXN_OK;
}

extern "C" int rtxn_render_status(rtxn_render* r, int wait, rtxn_render_stats* out) {
  RTXN_REQUIRE(r && out, "rtxn_render_status: NULL argument");
  if (wait) {
    RTXN_DEVICE_OR_FAIL();
    RTXN_HIP(hipDeviceSynchronize());
    for (int i = 0; i < r->n_slots; ++i) {
      Slot& g = r->slots[i];
      g.total_pending = false;
      fold(r, g);       // also the frames a captured graph replayed: they left their counters in pinned memory without an event
    }
  } else {
    for (int i = 0; i < r->n_slots; ++i) harvest(r, r->slots[i], false);
  }
  *out = r->st;
  return RTXN_OK;
}

extern "C" int rtxn_render_slot_buffers(rtxn_render* r, int slot, const int** num_hits, const int** num_stored, const int** indices,
                                        const int** total_segments, const float** start_points, const float** end_points,
                                        const float** seg_view, const void** radiance, const float** t_vals,
                                        const float** segment_step, const float** viewing_direction) {
  RTXN_REQUIRE(r != nullptr, "rtxn_render_slot_buffers: NULL renderer");
  RTXN_REQUIRE(slot >= 0 && slot < r->n_slots, "rtxn_render_slot_buffers: slot %d out of [0,%d)", slot, r->n_slots);
  const Slot& g = r->slots[slot];
  if (num_hits) *num_hits = g.num_hits;
  if (num_stored) *num_stored = g.num_stored;
  if (indices) *indices = g.indices;
  if (total_segments) *total_segments = g.total;
  if (start_points) *start_points = g.start;
  if (end_points) *end_points = g.end;
  if (seg_view) *seg_view = g.seg_view;
  if (radiance) *radiance = g.radiance;
  if (t_vals) *t_vals = g.t_vals;
  if (segment_step) *segment_step = g.seg_step;
  if (viewing_direction) *viewing_direction = g.view_dirs;
  return RTXN_OK;
}
