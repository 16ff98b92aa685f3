/*
 * include/rtc.h -- the C-ABI boundary between the C host and the HIP (gfx950) path tracer.
 *
 * The reference (elkiwy/raytraCer) has no plugin/FFI layer: its main() calls camera_get_ray +
 * ray_color + write_color directly inside an OpenMP loop (03_FullTracer/main.c:137-167).  These entry
 * points are what a maintainer binds IN PLACE OF that loop; each one names the reference code it
 * replaces.  Plain C symbols, opaque handles, int status (0 ok, <0 error; text via rtc_last_error()),
 * no C++/torch types; the caller owns every buffer it passes.  INTEGRATION.md shows the call sequence.
 * (What only tests call -- table dumps, inspection, known-answer probes -- is declared in rtc_test.h, which is not installed.)
 *
 * Device selection: every call acts on the CURRENT HIP device of the calling thread (one process per
 * GPU; under torch.distributed that is LOCAL_RANK's device).  There is no CPU fallback: with no usable
 * GPU every compute entry point returns RTC_ERR_DEVICE.
 */
#ifndef RAYTRACER_AMD_RTC_H
#define RAYTRACER_AMD_RTC_H

#include <stddef.h>
#include <stdint.h>
#include "raytracer.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { RTC_OK = 0, RTC_ERR_ARG = -1, RTC_ERR_DEVICE = -2, RTC_ERR_SCENE = -3, RTC_ERR_NOMEM = -4 };

typedef struct rtc_scene rtc_scene;

/* Render settings; the defaults of the reference's main() are 1024,1024,256,10 (03 main.c:88-92). */
typedef struct rtc_settings {
    int width, height;
    int max_depth;               /* recursion depth of ray_color: 10 (03 main.c:92), 8 (02 main.c:393) */
    int sample_begin, sample_end;/* this call renders samples [begin,end) of every pixel: the multi-GPU shard */
    int integrator;              /* 0 = 03 mixture PDF (ray.c:50-84), 1 = 02 attenuation (02 main.c:44-63), 2 = 01 sky (01 main.c:39-65) */
    uint64_t seed;               /* Philox key; the reference uses unseeded rand() (utils.h:23) */
    double mixture_light_weight; /* 0.1 (ray.c:72) */
    int kernel;                  /* 0 = auto, 1 = general megakernel (BVH, media, textures), 2 = small-scene megakernel
                                    (<= 64 rect/sphere leaves, solid textures), 3 = wavefront (BVH scenes: more than 64 leaves);
                                    RTC_ERR_SCENE when the scene is not eligible for the kernel asked for; the wavefront kernel takes
                                    max_depth <= 255, sample_end <= 2^24 and images of at most 32 M pixels (auto falls back to 1
                                    beyond that, asking for 3 is RTC_ERR_ARG) */
    int reserved;
} rtc_settings;

/* Counters of the last rtc_render* call on this thread's device. */
typedef struct rtc_stats {
    uint64_t samples;            /* W*H*(sample_end-sample_begin) */
    uint64_t rays;               /* closest-hit queries issued by the integrator (ray_color bodies past the depth check) */
    double   kernel_ms;          /* HIP-event time of the path-tracing kernel(s) on the launch stream */
    double   total_ms;           /* whole call incl. uploads/downloads */
    int      kernel_launches;
    int      kernel_used;        /* 1 general megakernel, 2 small-scene megakernel, 3 wavefront */
    uint64_t wave_iterations;    /* bounce iterations summed over waves (wavefront: extend waves that held a live slot):
                                    rays / (64 * wave_iterations) = fraction of lane slots that traced a ray */
} rtc_stats;

/* Probe the current device.  0 when a gfx950-class GPU is usable.  No reference counterpart. */
int rtc_init(void);
const char *rtc_last_error(void);
const char *rtc_device_name(void);
/* HIP devices this process sees (0 without a usable runtime); rtc_scene_clone_to_device takes an index below it. */
int rtc_device_count(void);

/* Flatten the pointer graph (world list, lights list) into SoA device tables + linear BVH and upload them to the
 * current device.  Replaces what the reference does implicitly by keeping `world`/`lights` as pointer graphs that
 * hittable_list_hit walks (hittable_list.c:125-143, hittable.c:595-615).  `background` as main.c:119;
 * time0/time1 = the camera shutter used for moving-sphere bounds (hittable.c:646-654). */
rtc_scene *rtc_scene_build(const hittable_list *world, const hittable_list *lights, const color *background,
                           double time0, double time1);
/* Host half of rtc_scene_build only (no device, no upload): for inspecting the flattened tables on a machine
 * without a GPU.  Compute entry points return RTC_ERR_DEVICE for such a handle. */
rtc_scene *rtc_scene_flatten_only(const hittable_list *world, const hittable_list *lights, const color *background,
                                  double time0, double time1);
/* A flattened scene on disk.  No reference counterpart: the reference's scenes are C functions that draw rand() while
 * they build (03 main.c:28-78, 02 main.c:76-289, 01 main.c:81-114), so a scene exists only inside the process that built
 * it; a scene file makes it a fixture and skips construction, flatten and BVH build (1M spheres: 1.1 s to
 * load and verify a 373 MB file against 4.9 s to build, same CPU).
 * rtc_scene_save writes the tables of any handle.  rtc_scene_load reads them back, verifies the checksum, the table
 * layouts and every index a kernel follows, and uploads them to the current device; rtc_scene_load_host stops before
 * the upload (a handle like rtc_scene_flatten_only's).  NULL + rtc_last_error() on any failure; a file that loads renders
 * bit-identically to the scene it was saved from. */
int        rtc_scene_save(const rtc_scene *scene, const char *path);
rtc_scene *rtc_scene_load(const char *path);
rtc_scene *rtc_scene_load_host(const char *path);
void rtc_scene_free(rtc_scene *scene);
/* Number of flattened leaf primitives (visit order of the reference's traversal = primitive id). */
int  rtc_scene_num_prims(const rtc_scene *scene);
/* FP64, no-FMA closest hit for n host rays (7 doubles each: orig, dir, time): bit-exact replacement of
 * hittable_list_hit(world, r, tmin, tmax, &rec) (hittable_list.c:125-143).  prim_id[i] = -1 on miss.
 * The kernel replays the reference's own traversal (list order; bvh_node_hit's left-then-right recursion over the tree
 * bvh_node_constructor builds, culled by the float-reciprocal aabb_hit of aabb.c:33-63), so even leaves that share a t
 * resolve to the leaf the reference keeps.  Constant media are stochastic in the reference (hittable.c:574) and are skipped.
 * rec12 (optional): the hit_record of hittable.h:26-33 as 12 doubles: hit, t, p(3), normal(3), u, v, front_face,
 * material type (-1 on miss). */
int rtc_first_hit_f64(const rtc_scene *scene, const double *rays7, size_t n, double tmin, double tmax,
                      int32_t *prim_id, double *t, double *rec12);
/* Same, but the rays are generated ON DEVICE from (s,t) screen coordinates exactly as camera_get_ray does
 * (camera.c:46-53) with the supplied lens sample and time: lens3 = unit-disk point (x,y) and ray time per ray
 * (NULL = no lens offset, time 0).  rays7_out (optional) returns the generated rays. */
int rtc_primary_first_hit_f64(const rtc_scene *scene, const camera *cam, const double *st2, const double *lens3,
                              size_t n, double tmin, double tmax, int32_t *prim_id, double *t, double *rays7_out);

/* The hot path: replaces the pixel/sample loop of 03 main.c:137-160 (camera_get_ray + ray_color + sum).
 * accum_rgba is W*H float4 = (sum R, sum G, sum B, sample count), row j=0 at the BOTTOM like the reference's
 * pixel index (main.c:157); the call ADDS to it, so successive calls with disjoint sample ranges accumulate
 * progressively.  rtc_render_device takes a DEVICE pointer (e.g. a torch tensor's data_ptr) and enqueues on
 * `stream` (a hipStream_t, NULL = default stream) without synchronising; rtc_render takes a HOST buffer.
 * Every kernel only enqueues: the wavefront pipeline (kernel 3, the automatic choice for every scene flattened into a BVH) puts a
 * fixed schedule of bounces on `stream` and on up to two more streams the scene owns (the samples are split into parts that run
 * side by side; they fork from `stream` and join it again before the call's last kernel), and whatever the host used to read back
 * -- the length of the path list, whether to compact it, whether anything is alive -- is decided on the device.  What follows
 * the call on `stream` (the next frame, a tonemap, a collective) is ordered behind the whole render. */
int rtc_render_device(const rtc_scene *scene, const camera *cam, const rtc_settings *settings,
                      float *accum_rgba_device, void *stream);
int rtc_render(const rtc_scene *scene, const camera *cam, const rtc_settings *settings, float *accum_rgba_host);
/* Multi-GPU from a C host (BASELINE config 4; the reference's only fan-out is the OpenMP loop over 16 row chunks of ONE process,
 * 03 main.c:132-138).  rtc_scene_clone_to_device uploads a copy of a scene's flattened tables to another HIP device (the
 * flattening and BVH build are not repeated); NULL + rtc_last_error() on failure.  rtc_render_multi renders samples
 * [sample_begin, sample_end) of every pixel split over the n handles -- one host thread per handle, each on its handle's
 * device, contiguous disjoint shards, earlier handles take the remainder -- brings the accumulators to scenes[0]'s device with
 * peer copies (xGMI), adds them there in handle order and ADDS the sum to accum_rgba_host (W*H float4).  Handles must be
 * distinct; two handles may live on one device (that is how a one-GPU box tests the path).  Float sums are reassociated
 * across shards: 1e-5 relative against the one-device render, sample counts exact. */
rtc_scene *rtc_scene_clone_to_device(const rtc_scene *scene, int device);
int rtc_render_multi(const rtc_scene *const *scenes, int n_scenes, const camera *cam, const rtc_settings *settings, float *accum_rgba_host);

/* NO COUNTERPART IN THE REFERENCE (its CPU loop allocates freely: two mallocs per diffuse bounce, pdf.c:21-25).  For callers
 * that cannot allocate -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc: rtc_render_device
 * with the same arguments and the same kernels, but its scratch (ray counters, tile costs and order, per-chunk partial sums)
 * lives in a file-scope device array of the library (17 MiB + 256 B per GPU, held until the process ends) instead of
 * hipMalloc'ed per-stream buffers.  Calls no hipMalloc / hipFree / stream or event creation and touches no host heap.
 * Limits: small scenes only (<= 64 rect/sphere leaves, solid textures, no media: kernel 0 or 2) -> RTC_ERR_SCENE otherwise;
 * at most 131,072 tiles of 8x8 pixels (e.g. 3840 x 2160) -> RTC_ERR_ARG "image too large"; when W * H * chunks exceeds 2^20
 * float4 of partial sums a tile's samples are run by fewer waves (float sums reassociated: relative 1e-5 against
 * rtc_render_device, bit-identical otherwise).  No per-stream state is kept: rtc_render_stats does not see these calls and
 * the tile order is probed on every call.  All calls on one GPU share the array: calls that OVERLAP on one GPU (two streams)
 * race; calls on one stream, or ordered by events, are safe. */
int rtc_render_device_noalloc(const rtc_scene *scene, const camera *cam, const rtc_settings *settings,
                              float *accum_rgba_device, void *stream);
/* Blocks until the last rtc_render_device on `stream` has finished and fills the counters. */
int rtc_render_stats(const rtc_scene *scene, void *stream, rtc_stats *out);

/* ray_color (ray.c:50-84; integrator 1: 02 main.c:44-63; 2: 01 main.c:39-65) for n rays the CALLER supplies (7 doubles
 * each: orig, dir, time) instead of the camera's: rgb_sum[3 i ..] += the estimator of ray i summed over samples
 * [sample_begin, sample_end) -- divide by the sample count for the mean.  Of `settings` only max_depth, the sample range,
 * integrator, seed and mixture_light_weight are read; the Philox stream is keyed on (ray index, sample, bounce).  Runs the
 * general bounce loop (the one kernel = 1 renders with) on the current device; the host ray_color of raytracer.h is this
 * call with n = 1 and one sample. */
int rtc_ray_colors(const rtc_scene *scene, const double *rays7, size_t n, const rtc_settings *settings, double *rgb_sum);

/* write_color (ray.c:35-47): rgb/count -> sqrt -> clamp [0,0.999] -> *256 -> u8, for every pixel, on device.
 * rgb_u8 is W*H*3, same row order as accum (flip at PNG time like main.c:170).  NaN and negative sums map to 0
 * (the reference casts NaN to unsigned char: undefined, F16); +inf clamps to 255 like any large value. */
int rtc_tonemap_u8_device(const float *accum_rgba_device, int width, int height, unsigned char *rgb_u8_device, void *stream);
int rtc_tonemap_u8(const float *accum_rgba_host, int width, int height, unsigned char *rgb_u8_host);

/* ---- library facade: the three calls of the reference's library build (src/renderer.h:56-63, `make lib`
 * Makefile:55-62) over the entry points above.  scene_settings is renderer.h:31-36; a render is `iterations` passes
 * of `samples_per_pixel` samples (renderer.c:372-375) accumulated into one image and written as a PNG
 * (renderer.c:429).  make_scene: `size` is the reference's table capacity, unused here; NULL on failure
 * (rtc_last_error()).  Scene conventions of 03 main.c: black background, shutter 0..1, depth 10, mixture PDF 0.1.
 * render_scene_accumulate has no counterpart in the reference: passes [first, first + n) added to a caller-owned host
 * accumulator (W*H float4), so that a render can be resumed or previewed between passes. */
typedef struct scene_settings { int res_width, res_height, samples_per_pixel, iterations; } scene_settings;
typedef rtc_scene scene;
scene *make_scene(const hittable_list *objects, const hittable_list *lights, int size);
void   free_scene(scene *s);
void   render_scene(const scene *s, const camera *cam, const scene_settings *settings, const char *filename);
int    render_scene_accumulate(const scene *s, const camera *cam, const scene_settings *settings, int first_iteration,
                               int n_iterations, float *accum_rgba_host);

#ifdef __cplusplus
}
#endif
#endif
