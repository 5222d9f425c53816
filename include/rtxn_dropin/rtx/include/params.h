// Drop-in for the reference's rtx/include/params.h (lines 14-42) and for the
// slice of rtx/include/rtxFunctions.h (lines 56-100) main.cu uses.  `Params`
// keeps every field name of the reference so main.cu:481-501 fills it unchanged;
// OptiX types are reduced to what the hot path reads (OptixAabb = 6 floats,
// OptixTraversableHandle = an integer nobody dereferences).  rtxnLaunch replaces
//   optixLaunch(pipeline_ray_march, stream, d_param, sizeof(Params), &sbt, W, H, 1)
// (main.cu:506-508): no pipeline, SBT or device copy of Params is needed.
#ifndef RTXN_DROPIN_PARAMS_H
#define RTXN_DROPIN_PARAMS_H
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>
#include "rtxn.h"

struct OptixAabb { float minX, minY, minZ, maxX, maxY, maxZ; };
typedef unsigned long long OptixTraversableHandle;

struct RayGenData {};
struct HitGroupData {};
struct MissData {};

struct Params {
    OptixTraversableHandle handle;   // unused: the grid is walked analytically
    float* look_at;                  // 4x4 matrix (device)
    int num_primitives;              // unused
    int intersection_arr_size;
    OptixAabb* aabb;                 // unused: cells are make_grid's, recomputed in-kernel
    float3* start_points;
    float3* end_points;
    float3* ray_origins;
    float* t_start;
    float* t_end;
    int* num_hits;
    float2* viewing_direction;
    float focal_length;
    float aspect_ratio;
    float3 delta;
    float3 min_point;
    float3 max_point;
    unsigned int width, height;
};

// Counterpart of RTXDataHolder (rtxFunctions.h:56-100).  initContext /
// createModule / createProgramGroups / linkPipeline / buildSBT have nothing to
// do on this platform and are kept as no-ops so main.cu:381-391 compiles.
struct RTXDataHolder {
    Params params{};
    OptixTraversableHandle gas_handle = 0;
    hipStream_t stream = nullptr;
    void initContext() {}
    void createModule(const std::string&) {}
    void createProgramGroups() {}
    void linkPipeline(bool) {}
    void buildSBT() {}
    // The reference uploads the AABBs and builds a GAS over them
    // (rtxFunctions.cpp:293-351).  Here the boxes are only uploaded so that
    // params.aabb stays a valid device pointer; the caller owns and frees it.
    OptixAabb* initAccelerationStructure(const std::vector<OptixAabb>& grid) {
        OptixAabb* d = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&d), grid.size() * sizeof(OptixAabb)) != hipSuccess) return nullptr;
        (void)hipMemcpy(d, grid.data(), grid.size() * sizeof(OptixAabb), hipMemcpyHostToDevice);
        return d;
    }
    void setStream(const hipStream_t& s) { stream = s; }
};

// mode: RTXN_TRACE_COMPAT reproduces the reference arithmetic.
inline int rtxnLaunch(const Params& p, hipStream_t stream, int mode = RTXN_TRACE_COMPAT,
                      const uint32_t* occupancy = nullptr, const uint32_t* occupancy_coarse = nullptr) {
    rtxn_trace_params t{};
    t.look_at = p.look_at;
    t.focal_length = p.focal_length;
    t.aspect_ratio = p.aspect_ratio;
    t.width = p.width;
    t.height = p.height;
    t.ray_begin = 0;
    t.ray_count = p.width * p.height;
    t.grid_res = (int)std::lround((p.max_point.x - p.min_point.x) / p.delta.x);
    t.occupancy = occupancy;
    t.occupancy_coarse = occupancy_coarse;
    t.mode = mode;
    t.ray_origins = reinterpret_cast<float*>(p.ray_origins);
    t.viewing_direction = reinterpret_cast<float*>(p.viewing_direction);
    t.num_hits = p.num_hits;
    t.intersection_arr_size = p.intersection_arr_size;
    t.start_points = reinterpret_cast<float*>(p.start_points);
    t.end_points = reinterpret_cast<float*>(p.end_points);
    t.t_start = p.t_start;
    t.t_end = p.t_end;
    int rc = rtxn_trace_grid(&t, static_cast<rtxn_stream_t>(stream));
    if (rc != RTXN_OK) std::fprintf(stderr, "rtxnLaunch: %s\n", rtxn_last_error());
    return rc;
}
#endif
