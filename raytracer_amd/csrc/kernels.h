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
    float light_weight_each, cosine_weight;   // light_weight / number of lights (0 without lights), 1 - light_weight: launch constants of pdf.c:119-121
    int tiles_x, tiles_y, n_tiles;   // 8x8-pixel tiles, one wave each
    int use_time;                    // camera shutter open (time0 != time1)
    // pinhole camera of the small-scene kernel, folded on the host: direction(px + jx, py + jy) = base + hs (px + jx) + vs (py + jy)
    // with base = lower_left_corner - origin, hs = horizontal / (W - 1), vs = vertical / (H - 1)   (camera.c:46-53, main.c:150-151)
    float cam_base[3], cam_hs[3], cam_vs[3];
};

hipError_t launch_first_hit_f64(const SceneT<double> &sc, const double *rays7, size_t n, double tmin, double tmax,
                                int32_t *prim_id, double *t, double *rec12, hipStream_t stream);
hipError_t launch_primary_first_hit_f64(const SceneT<double> &sc, const CameraT<double> &cam, const double *st2, const double *lens3,
                                        size_t n, double tmin, double tmax, int32_t *prim_id, double *t, double *rays7_out, hipStream_t stream);

// Persistent-thread path tracer: accum (W*H float4) += radiance sums; *ray_counter += closest-hit queries.
hipError_t launch_pathtrace_mega(const SceneT<float> &sc, const CameraT<float> &cam, const RenderParams &p,
                                 float4 *accum, unsigned long long *ray_counter, hipStream_t stream);
// Small-scene persistent path tracer (kernels_small.hip): counters[0] += rays, counters[1] += wave bounce-iterations.
// tile_scratch: 2 * p.n_tiles ints of device memory for the heaviest-first tile order, or NULL (tiles in image order).
// *ordered: in = tile_scratch already holds the order for this camera and image (skip the probe); out = it holds one now.
// partial: 2^log2_chunks planes of W*H float4 when a tile's samples are split over 2^log2_chunks waves, or NULL.
hipError_t launch_pathtrace_small(const SmallScene &ss, const SceneT<float> &sc, const CameraT<float> &cam, const RenderParams &p,
                                  float4 *accum, unsigned long long *counters, int *tile_scratch, int *ordered, float4 *partial, int log2_chunks, hipStream_t stream);
// Production FP32 traversal alone (brute-force loop or flattened BVH), one lane per ray; media skipped.
hipError_t launch_first_hit_f32(const SceneT<float> &sc, const float *rays7, size_t n, float tmin, float tmax, int32_t *prim_id, float *t, hipStream_t stream);
hipError_t launch_tonemap_u8(const float4 *accum, int n_pixels, unsigned char *rgb, hipStream_t stream);
hipError_t launch_accum_add(float4 *dst, const float4 *const *src, int n_src, size_t n, hipStream_t stream);   // dst += src[0] + ... + src[n_src - 1], in that order (n_src <= 63)

// ---- wavefront path tracer for BVH scenes (kernels_wf.hip) ----------------------------------------------------------
// Path-slot state in HBM, SoA.  Slot s of layer k = s / slots_per_layer renders samples s_begin + k, + layers, ... of ONE
// pixel (8x8 tiles, 64 consecutive slots per tile).
// floor(x / d) for 0 <= x < 2^30 as (x * mul) >> shift: mul = ceil(2^shift / d), shift = 30 + ceil(log2 d)  (d >= 1; exact because
// mul * d - 2^shift < d <= 2^(shift - 30))
struct WfDiv { uint32_t mul, shift; };
inline WfDiv wf_make_div(uint32_t d) {
    uint32_t s = 0; while ((1ull << s) < d) ++s;
    WfDiv r; r.shift = 30 + s; r.mul = (uint32_t)((((unsigned long long)1 << r.shift) + d - 1) / d);
    return r;
}
enum { WF_LIVE = 0, WF_CURSOR = 1, WF_NLIST = 2, WF_CUR = 3, WF_LIVE0 = 4 };   // LIVE: as of the last compaction point (never too low); LIVE0: at the start of the render
struct WfBuffers {
    float4 *ro, *rd, *tm, *sum;       // origin | time, direction | throughput.x, throughput.yz | sample + (depth << 24) | leaf slot
                                      // the ray starts on (int bits), radiance sum of the slot; the pixel follows from the slot
    float2 *hit;                      // t, leaf slot (int bits; -1 = miss) written by extend
    unsigned char *key;               // per LIST position, written by extend: regrouping key of a hit parked for k_wf_shade (material type * 4 + texture type), RTC_WF_KEY_DONE otherwise
    float2 *phit;                     // (phit_k x n_slots) closest hits of the camera rays of every slot's first phit_k samples, traced ahead by k_wf_primary
    int *active[2];                   // slot lists (ping-pong across compactions); -1 = hole
    unsigned *counters;               // 16 words, device-side control of the render (WF_*): live slots, compaction cursor, length and
                                      // buffer index of the current slot list -- the host enqueues a fixed schedule and never reads them
    unsigned long long *stats;        // wf_stats_bytes(): striped partial counters (rays, extend waves, dead slots) behind one unused line; scene-wide like the pool
    unsigned long long *stats_out;    // [0] rays, [1] extend waves of the calling stream: k_wf_finish folds the stripes into them
    int n_slots, slots_per_layer, layers;
    int quad;                         // slot <-> (pixel, layer) mapping of kernels_wf.hip slot_pixel
    WfDiv div_group, div_tiles_x;     // divisions of that mapping: by the slots of a layer group (4 layers when quad, else 1), by tiles_x
    int phit_k;                       // 0: no camera rays traced ahead (every sample's first ray goes through the slot list)
    WfDiv div_layers;                 // (sample - s_begin) / layers = which of the slot's samples
    int compact_div;                  // k_wf_compact squeezes the list when more than 1 / compact_div of it is holes
    int noise_lds_slots;              // k_wf_shade keeps the tables of up to this many noise textures in LDS (2; 0 = read them from memory: RTC_WF_NOISE_LDS=0, an A/B knob)
};
size_t wf_bytes_per_slot();
size_t wf_stats_bytes();
hipError_t launch_wf_init(const CameraT<float> &cam, const RenderParams &p, const WfBuffers &b, hipStream_t stream);
hipError_t launch_wf_primary(const SceneT<float> &sc, const CameraT<float> &cam, const RenderParams &p, const WfBuffers &b, hipStream_t stream);
// one bounce: extend (+ shade of the parked hits) over the current slot list, whose length and buffer index live in b.counters
bool wf_lds_tables(const SceneT<float> &sc, size_t *lds_bytes);   // extend from LDS-resident node / leaf tables (small trees)?
hipError_t launch_wf_bounce(const SceneT<float> &sc, const CameraT<float> &cam, const RenderParams &p, const WfBuffers &b, hipStream_t stream);
// current list -> the other list without holes, IF more than 1 / b.compact_div of it is holes (decided on the device)
hipError_t launch_wf_compact(const WfBuffers &b, hipStream_t stream);
hipError_t launch_wf_finish(const RenderParams &p, const WfBuffers &b, float4 *accum, hipStream_t stream);

// ---- single rays and known-answer probes (kernels_probe.hip, kernels_small.hip) --------------------------------------
// rgb_sum[3 i ..] += the estimator of ray i summed over samples [p.s_begin, p.s_end), `chunks` lanes per ray
hipError_t launch_ray_colors(const SceneT<float> &sc, const RenderParams &p, const float *rays7, unsigned n, unsigned chunks, double *rgb_sum,
                             unsigned long long *ray_counter, hipStream_t stream);
// rtc_kat (include/rtc.h): `what` selects the device function; doubles in, doubles out
enum { RTC_KAT_PHILOX = 0, RTC_KAT_LIGHT_PDF = 1, RTC_KAT_LIGHT_PDF_SMALL = 2, RTC_KAT_COSINE_PDF = 3, RTC_KAT_SCHLICK = 4, RTC_KAT_REFRACT = 5,
       RTC_KAT_LIGHT_SAMPLE = 6, RTC_KAT_LIGHT_SAMPLE_SMALL = 7, RTC_KAT_DIELECTRIC = 8, RTC_KAT_DIELECTRIC_SMALL = 9, RTC_KAT_COUNT = 10 };
__host__ __device__ inline int rtc_kat_in_stride(int what) { const int s[RTC_KAT_COUNT] = {5, 6, 6, 6, 2, 7, 6, 6, 9, 9}; return what >= 0 && what < RTC_KAT_COUNT ? s[what] : 0; }
__host__ __device__ inline int rtc_kat_out_stride(int what) { const int s[RTC_KAT_COUNT] = {4, 1, 1, 1, 1, 3, 3, 3, 3, 3}; return what >= 0 && what < RTC_KAT_COUNT ? s[what] : 0; }
hipError_t launch_kat(const SceneT<float> &sc, int what, const double *in, unsigned n, double *out, hipStream_t stream);
hipError_t launch_small_kat(const SceneT<float> &sc, const SmallScene &ss, int what, const double *in, unsigned n, double *out, hipStream_t stream);
