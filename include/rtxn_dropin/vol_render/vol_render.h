// Drop-in for the reference's vol_render/vol_render.h (lines 1-25): same two
// launcher signatures, forwarding to rtxn_volrender_fwd / rtxn_volrender_bwd in
// RTXN_VR_COMPAT mode on the null stream (the reference launches both kernels on
// the default stream, vol_render.cu:155,179).
#ifndef RTXN_DROPIN_VOL_RENDER_H
#define RTXN_DROPIN_VOL_RENDER_H
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include "rtxn.h"

inline void launch_volrender_cuda(
    float* network_inputs,
    float* network_outputs,
    int* num_hits,
    int* indices,
    float* ray_hit,
    int batch_size,
    int num_samples_per_hit,
    float* pixels) {
    int rc = rtxn_volrender_fwd(network_inputs, network_outputs, num_hits, indices, ray_hit, batch_size,
                                num_samples_per_hit, pixels, RTXN_VR_COMPAT, nullptr);
    if (rc != RTXN_OK) std::fprintf(stderr, "launch_volrender_cuda: %s\n", rtxn_last_error());
}

inline void launch_volrender_backward_cuda(
    float* loss_values,
    __half* loss_gradients,
    float* sampled_points_radiance,
    float* t_hit,
    int* num_hits,
    int* indices,
    int batch_size,
    int num_samples_per_hit,
    __half* radiance_gradients
) {
    int rc = rtxn_volrender_bwd(loss_values, loss_gradients, sampled_points_radiance, t_hit, num_hits, indices,
                                batch_size, num_samples_per_hit, radiance_gradients, RTXN_VR_COMPAT, nullptr);
    if (rc != RTXN_OK) std::fprintf(stderr, "launch_volrender_backward_cuda: %s\n", rtxn_last_error());
}
#endif
