// kernels.h -- host-callable launchers of the HIP kernels (product code).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "rtc_tables.h"

struct RenderParams {
    int width, height;
    int s_begin, s_end;
    int max_depth, integrator;
    uint32_t seed_lo, seed_hi;
    float light_weight;
    int tiles_x, tiles_y, n_tiles;   // 8x8-pixel tiles, one wave each
    int use_time;                    // camera shutter open (time0 != time1)
};

hipError_t launch_first_hit_f64(const SceneT<double> &sc, const double *rays7, size_t n, double tmin, double tmax,
                                int32_t *prim_id, double *t, double *rec12, hipStream_t stream);
hipError_t launch_primary_first_hit_f64(const SceneT<double> &sc, const CameraT<double> &cam, const double *st2, const double *lens3,
                                        size_t n, double tmin, double tmax, int32_t *prim_id, double *t, double *rays7_out, hipStream_t stream);

// Persistent-thread path tracer: accum (W*H float4) += radiance sums; *ray_counter += closest-hit queries.
hipError_t launch_pathtrace_mega(const SceneT<float> &sc, const CameraT<float> &cam, const RenderParams &p,
                                 float4 *accum, unsigned long long *ray_counter, hipStream_t stream);
// Small-scene persistent path tracer (kernels_small.hip): counters[0] += rays, counters[1] += wave bounce-iterations.
hipError_t launch_pathtrace_small(const SmallScene &ss, const SceneT<float> &sc, const CameraT<float> &cam, const RenderParams &p,
                                  float4 *accum, unsigned long long *counters, hipStream_t stream);
// Production FP32 traversal alone (brute-force loop or flattened BVH), one lane per ray; media skipped.
hipError_t launch_first_hit_f32(const SceneT<float> &sc, const float *rays7, size_t n, float tmin, float tmax, int32_t *prim_id, float *t, hipStream_t stream);
hipError_t launch_tonemap_u8(const float4 *accum, int n_pixels, unsigned char *rgb, hipStream_t stream);
