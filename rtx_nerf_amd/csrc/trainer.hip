// One optimisation step as one call (include/rtxn.h, rtxn_train_step): host-only sequencing of the stage entry points, as
// render.hip does for a frame.  Nothing here computes; the one kernel of its own advances the device step counter and forms
// the bias-corrected learning rate of tiny-cuda-nn's Adam for the MLP (main.cu:36-46, :787) from it, so that a replayed hipGraph
// of this call needs nothing refreshed from the host.
#include "common.h"

#include <cmath>

namespace {

// t = ++*step;  *lr_eff = lr sqrt(1 - beta2^t) / (1 - beta1^t)  -- the same single-precision expression rtxn_adam_effective_lr
// evaluates on the host (powf on the device library: within an ulp or two of glibc's; it multiplies a learning rate)
__global__ void advance_step_kernel(int* step, float lr, float beta1, float beta2, float* lr_eff) {
  const int t = *step + 1;
  *step = t;
  *lr_eff = lr * sqrtf(1.0f - powf(beta2, (float)t)) / (1.0f - powf(beta1, (float)t));
}

}  // namespace

extern "C" int rtxn_train_step(const rtxn_train_step_args* a, rtxn_stream_t stream) {
  RTXN_REQUIRE(a, "rtxn_train_step: NULL arguments");
  const rtxn_train_batch& b = a->batch;
  const rtxn_train_state& o = a->opt;
  RTXN_REQUIRE(b.mlp, "rtxn_train_step: batch.mlp is NULL");
  RTXN_REQUIRE(b.n_rays > 0 && (uint32_t)b.n_rays == a->trace.ray_count, "rtxn_train_step: batch.n_rays = %d, trace.ray_count = %u", b.n_rays,
               a->trace.ray_count);
  RTXN_REQUIRE(a->trace.num_hits && b.indices && b.num_stored && b.total_segments && a->scan_workspace,
               "rtxn_train_step: NULL num_hits / indices / num_stored / total_segments / scan workspace");
  RTXN_REQUIRE(b.start_points && b.end_points && b.seg_view && b.segment_capacity > 0, "rtxn_train_step: NULL segment buffers or capacity %ld",
               b.segment_capacity);
  RTXN_REQUIRE(o.mlp_master && o.mlp_params_fp16 && o.mlp_m && o.mlp_v && o.step && o.effective_lr, "rtxn_train_step: NULL optimizer state");
  RTXN_REQUIRE(o.loss_scale_divisor > 0.0f, "rtxn_train_step: loss_scale_divisor = %g", o.loss_scale_divisor);
  const bool hash = b.grid != nullptr;
  if (hash)
    RTXN_REQUIRE(o.table_master && o.table_params_fp16 && o.table_m && o.table_v && o.table_steps && b.dtable,
                 "rtxn_train_step: hash grid without table optimizer state / gradient");
  RTXN_DEVICE_OR_FAIL();

  // ---- traversal: count -> scan -> write (main.cu:506-508, 631-637; the packed layout of :646-673 written by the device) ----
  rtxn_trace_params t = a->trace;
  t.indices = nullptr;
  t.start_points = t.end_points = t.t_start = t.t_end = nullptr;
  t.seg_ray = nullptr;
  t.seg_view = nullptr;
  t.seg_first = nullptr;
  t.num_stored = nullptr;
  int rc = rtxn_trace_grid(&t, stream);
  if (rc != RTXN_OK) return rc;
  rc = rtxn_scan_hits(a->trace.num_hits, const_cast<int*>(b.indices), const_cast<int*>(b.total_segments), b.n_rays, a->scan_workspace,
                      a->scan_workspace_bytes, stream);
  if (rc != RTXN_OK) return rc;
  t.indices = b.indices;
  t.start_points = const_cast<float*>(b.start_points);
  t.end_points = const_cast<float*>(b.end_points);
  t.seg_view = const_cast<float*>(b.seg_view);
  t.num_stored = const_cast<int*>(b.num_stored);
  t.segment_capacity = b.segment_capacity;
  rc = rtxn_trace_grid(&t, stream);
  if (rc != RTXN_OK) return rc;

  // ---- sampler ... backward (main.cu:703-781), segment count read on the device ----
  rc = rtxn_train_gradients(&b, stream);
  if (rc != RTXN_OK) return rc;

  // ---- optimizer->step (main.cu:787): every gradient is cleared as it is consumed ----
  advance_step_kernel<<<1, 1, 0, rtxn::as_stream(stream)>>>(o.step, o.lr, o.beta1, o.beta2, o.effective_lr);
  RTXN_LAUNCH_CHECK("advance_step_kernel");
  const float ls = b.loss_scale * o.loss_scale_divisor;
  rc = rtxn_adam_step_captured(rtxn_mlp_n_params(b.mlp), o.mlp_master, o.mlp_params_fp16, b.dparams, RTXN_ADAM_ZERO_GRADS, o.mlp_m, o.mlp_v,
                               o.effective_lr, o.beta1, o.beta2, o.eps, ls, stream);
  if (rc != RTXN_OK) return rc;
  rc = rtxn_mlp_set_params_training(const_cast<rtxn_mlp*>(b.mlp), o.mlp_params_fp16, stream);
  if (rc != RTXN_OK) return rc;
  if (hash) {
    const long n = rtxn_hashgrid_n_params(b.grid);
    long lo = n;                                              // parameters before the first hashed level
    for (int l = 0; rtxn_hashgrid_level_offset(b.grid, l) < n; ++l)
      if (rtxn_hashgrid_level_is_hashed(b.grid, l) == 1) { lo = rtxn_hashgrid_level_offset(b.grid, l); break; }
    if (!b.dtable_hashed_half) lo = n;                        // everything in the fp32 gradient
    __half* p16 = static_cast<__half*>(o.table_params_fp16);
    if (lo > 0) {
      rc = rtxn_adam_step_sparse(lo, o.table_master, p16, b.dtable, RTXN_ADAM_ZERO_GRADS, o.table_m, o.table_v, o.table_steps, o.table_lr, o.beta1,
                                 o.beta2, o.table_eps, ls, stream);
      if (rc != RTXN_OK) return rc;
    }
    if (lo < n) {
      rc = rtxn_adam_step_sparse(n - lo, o.table_master + lo, p16 + lo, b.dtable_hashed_half, RTXN_ADAM_GRADS_FP16 | RTXN_ADAM_ZERO_GRADS,
                                 o.table_m + lo, o.table_v + lo, o.table_steps + lo, o.table_lr, o.beta1, o.beta2, o.table_eps, ls, stream);
      if (rc != RTXN_OK) return rc;
    }
  }
  return RTXN_OK;
}
