/*
 * include/rtc_test.h -- entry points of librtc.so that only TESTS and tools call: dumps of the flattened tables, inspection of the
 * kernel choice, the production traversal on its own, known-answer probes of device functions.  Not part of the drop-in boundary
 * (include/rtc.h) and not installed by `make install`; same conventions (plain C symbols, int status, rtc_last_error()).
 */
#ifndef RAYTRACER_AMD_RTC_TEST_H
#define RAYTRACER_AMD_RTC_TEST_H

#include "rtc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Flattened leaf table for inspection: per leaf 16 doubles = reference hittable_type (hittable.h:42-52), material type,
 * geometry (sphere: c,r | moving sphere: c0,c1,t0,t1,r | rect: x0,x1,y0,y1,z0,z1,k,axis | medium: -1/density) from slot 2,
 * slot 12 = number of translate/rotate wrappers, slot 13 = flip_face parity. */
int  rtc_scene_dump_leaves(const rtc_scene *scene, double *out16, int cap);
/* Shape of the production BVH (inspection; no reference counterpart: the reference's tree is bvh_node_constructor's,
 * hittable.c:163-215): out4 = inner nodes, leaf slots, levels, bytes of nodes + traversal leaves.  All 0 for a scene of
 * at most 64 leaves (brute-force loop). */
int  rtc_scene_bvh_info(const rtc_scene *scene, long long *out4);
/* The small-scene pack (scenes of <= 64 rect / sphere leaves): out8 = eligible, pairs, object spaces, slab-tested boxes, spheres,
 * light pdfs fused into the intersection (every light-list entry is a world-space leaf of the pack), rects and spheres that
 * carry a light's pdf.  Inspection only; no reference counterpart. */
int  rtc_scene_small_info(const rtc_scene *scene, int *out8);
/* How the wavefront path intersects this scene's rays (BVH scenes; inspection, no reference counterpart): out4 = [0] 1 if the node and
 * leaf tables are staged in LDS (small trees without media, at most 80 KB with the stacks; RTC_WF_LDS_TABLES=0 in the environment
 * turns it off), [1] bytes of LDS per workgroup in that case, [2] inner nodes, [3] leaf records.  Needs device tables. */
int  rtc_scene_extend_info(const rtc_scene *scene, int *out4);
/* The production BVH itself, for structural checks on a machine without a GPU: nodes16 = per inner node 16 words (float
 * lo0[3], int child0, float hi0[3], int count0, the same for child 1; child < 0 encodes a leaf run ~((first << 3) | (count-1))),
 * leaves16 = per leaf slot 16 words (15 floats of world-space geometry + meta), prim_of_slot = primitive id per leaf slot.
 * Any pointer may be NULL; the caps are in entries.  Returns the number of leaf slots. */
int  rtc_scene_dump_bvh(const rtc_scene *scene, void *nodes16, int node_cap, void *leaves16, int *prim_of_slot, int leaf_cap);

/* The PRODUCTION traversal on its own: hittable_list_hit (hittable_list.c:125-143) as the FP32 path tracer executes it
 * (brute-force leaf loop up to 64 leaves, flattened SoA BVH with a per-lane LDS stack above), rays converted to float.
 * Ties between leaves go to the highest primitive id; media are skipped.  Agreement with rtc_first_hit_f64 is a
 * tolerance statement (tests/test_gpu_first_hit.py), not a bit-exact one. */
int rtc_first_hit_f32(const rtc_scene *scene, const double *rays7, size_t n, double tmin, double tmax,
                      int32_t *prim_id, float *t);

/* TEST-ONLY known-answer probe: evaluates, on the device and for n supplied inputs, the functions the production kernels
 * call, so that they can be compared with the reference's outputs for the same inputs (tests/golden/unit_vectors.npz).
 * Doubles in and out (converted to the kernels' FP32 on the device).  `what`:
 *   0 Philox4x32-7 block of (pixel, sample, tag, seed lo, seed hi) -> 4 raw words: counter = (pixel, sample, tag, seed lo),
 *     key = ("RTC1", seed hi)                                                              (replaces rand(), utils.h:23)
 *   1 light-list pdf, general kernels: origin(3) dir(3) -> pdf       (hittable_list.c:165-172, hittable.c:760-788)
 *   2 the same as the small-scene kernel computes it (unit direction, folded hit tests)
 *   3 cosine pdf = lambertian scattering pdf: normal(3) dir(3) -> pdf (pdf.c:100-103, material.c:205-208)
 *   4 reflectance: cosine, ref_idx -> r                               (material.c:142)
 *   5 refract: uv(3) n(3) eta -> dir(3)                               (vec3.c:44-50)
 *   6 light sample, general kernels: origin(3) pick r1 r2 -> dir(3)   (hittable.c:821-848; the reference draws r1, r2 from rand())
 *   7 the same, small-scene kernel (a basis of its own around the cone axis: same law, other azimuth)
 *   8 glass direction, general kernels: dir(3) n(3) front_face ir xi -> dir(3)   (material.c:144-157; xi replaces random_double())
 *   9 the same, small-scene kernel (takes the direction normalised) */
int rtc_kat(const rtc_scene *scene, int what, const double *in, size_t n, double *out);

#ifdef __cplusplus
}
#endif
#endif
