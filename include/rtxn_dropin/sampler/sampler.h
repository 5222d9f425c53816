// Drop-in for the reference's sampler/sampler.h (lines 1-30): same macro, enum,
// struct and launchSampler signature (cudaStream_t& -> hipStream_t&), forwarding
// to the C ABI of librtxn.so (rtxn_sample).  Like the reference it returns void;
// a failure is printed to stderr and execution continues (common/common.h:38-50).
#ifndef RTXN_DROPIN_SAMPLER_H
#define RTXN_DROPIN_SAMPLER_H
#include <hip/hip_runtime.h>
#include <cstdio>
#include "rtxn.h"

#define NUM_SAMPLES_PER_SEGMENT 32
enum SAMPLING_TYPE {
    SAMPLING_REGULAR,
    SAMPLING_STRATIFIED_JITTERING,
    SAMPLING_UNIFORM,
};

struct float5 {
    float x;
    float y;
    float z;
    float theta;
    float phi;
};

inline void launchSampler(
    float3* d_start_points,
    float3* d_end_points,
    float2* d_view_dirs,
    float* d_t_vals,
    float* d_sampled_points,
    int batch_size,
    int grid_res,
    int* d_num_hits,
    int* d_indices,
    SAMPLING_TYPE sample_type,
    hipStream_t& stream) {
    int rc = rtxn_sample(reinterpret_cast<const float*>(d_start_points), reinterpret_cast<const float*>(d_end_points),
                         reinterpret_cast<const float*>(d_view_dirs), d_t_vals, d_sampled_points, batch_size, grid_res,
                         d_num_hits, d_indices, static_cast<int>(sample_type), static_cast<rtxn_stream_t>(stream));
    if (rc != RTXN_OK) std::fprintf(stderr, "launchSampler: %s\n", rtxn_last_error());
}
#endif
