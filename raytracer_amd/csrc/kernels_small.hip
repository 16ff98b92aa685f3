// kernels_small.hip -- the headline kernel: persistent-thread path tracer for SMALL scenes on gfx950 (product code).
//
// "Small" = at most 64 rect / sphere leaves with solid-colour textures (the 03_FullTracer Cornell box: 13 leaves).  Such
// a scene never leaves the chip: the bounce loop of ray.c:50-84 runs out of SGPRs, VGPRs and 3 KB of LDS, and HBM only
// sees the float4 accumulator once per launch.  Design, in the order the profile asked for it:
//   * one wave64 = one 8x8 pixel tile x a chunk of its samples; the 64 * S paths of that work item are handed to whichever
//     lanes fall free (lanes are not pixels), two primary rays are kept ready per lane, and a finished path adds its
//     radiance to the tile's sums in LDS: 98.6 % of the lane slots trace a ray until the tile runs dry; tiles are taken
//     heaviest first (one-sample probe + counting sort, launch_pathtrace_small);
//   * intersection = wave-uniform loops over the records of one object space: a slab test for the rects that are faces of
//     one box (SmallBox), ONE 64-byte scalar load feeding two rect tests for the others (SmallPair; the padding of an odd
//     run is skipped by a scalar branch), spheres in closest-approach form; wrapper stacks are pre-collapsed affine maps
//     applied once per space, not once per leaf (a y-rotation passes y through);
//   * the winner's shading record (world normal or centre, resolved colour, flags) is ONE 48-byte LDS gather -- the
//     generic path chases prim -> material -> texture through three dependent global loads;
//   * material / pdf dispatch is a switch on the record's type; lambertian sampling uses n + uniform-sphere (the same
//     cosine law as onb.c + random_cosine_direction, without building the basis), the mixture pdf of pdf.c:119-121;
//   * Philox4x32-7 on (pixel, sample, bounce); 2 blocks per path start are not needed here (no lens, no moving leaf).
// 72 VGPRs -> 7 waves per SIMD (amdgpu_waves_per_eu), no VGPR spills, 6 KB of LDS (shading records + tile sums); the SIMDs issue VALU instructions all the time, so what
// is left is the instruction count and the divergence of the material switch (DESIGN.md 3.1 has the cycle budget).
#include <hip/hip_runtime.h>

#include "device/rt_shade.hpp"
#include "kernels.h"


// waves per SIMD the kernel is compiled for (registers per lane = 512 / waves, rounded down to a multiple of 8)
// (measured on the r02 kernel, 1080p / 1024^2: 5 waves 17.66 / 14.33 ms, 6: 17.69 / 14.35, 7: 17.37 / 14.04 -- 72 VGPRs, no VGPR spills --
// 8: 17.67 / 14.40 with 4 spilled registers)
#ifndef RT_SMALL_WAVES
#define RT_SMALL_WAVES 7
#endif

namespace {

struct SmallHit { float t; int leaf; };

// (t, leaf) <- (nt, nleaf) where `take`.  (hipcc emits the pair as two v_cndmask_b32_e32 ..., vcc in a row, which a micro-benchmark shows
// stalling a lone wave for ~40 cycles; issuing the second one in VOP3 form with the mask in an SGPR pair -- inline assembly -- made the
// kernel 2 % SLOWER: with 7 waves resident the bubble is hidden and the extra scalar instruction is not.)
RT_DEV void take_hit(bool take, float nt, int nleaf, SmallHit &h) { h.t = take ? nt : h.t; h.leaf = take ? nleaf : h.leaf; }

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
RT_DEV float recf(const u32x16 &R, int k) { return __uint_as_float(R[k]); }
RT_DEV f3 rcp3(f3 d) { return f3mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)); }

// rect_hit (hittable.c:494-506) for one rect of a pair record, the ray already split into the run's normal (on, in = 1/dn) and
// in-plane (oa, da, ob, db) parts; the rect is centre (ca, cb) and half extent (ha, hb).  All range tests fold into ONE compare:
// the largest violation  max(|pa - ca| - ha, |pb - cb| - hb, tmin - t, t - tbest) <= 0, so no lane masks are combined on the
// scalar unit.  tbest starts at a large FINITE value: a t of +inf (dn == 0) then violates t - tbest, NaNs fail the compare.
// A ray cannot hit the plane it leaves again: the leaf it starts on (src) is excluded -- in FP32 the start point is off the
// plane by up to |k| 2^-24 (3e-5 at k = 555) and a grazing ray (|d_n| < 0.03) would meet it again at t >= tmin.
// A rect that is also an entry of the lights list (`light`: a bit of the record's head, a scalar branch) adds its pdf for THIS ray
// to `pdf` on the way: hittable_pdf_value (hittable.c:772-788) is this very plane test with tmin 1e-4 and no upper bound, value
// t^2 |1/d_n| / area for a unit direction (small_light_pdf_one below is the stand-alone form).
// (Round 1 did the two in-plane axes in packed FP32 instructions; the plain form measured 1-2 % faster, 17.95 -> 17.57 ms, because
// every run paid six moves to build the aligned register pairs; the packed one left the sources in round 3.)
template <bool PDF>
RT_DEV void small_rect(float k, float ca, float cb, float ha, float hb, int id, bool light, float inv_area, float on, float in, float oa, float da, float ob, float db,
                       float tmin, int src, SmallHit &h, float &pdf) {
    const float t = (k - on) * in;
    const float ea = fabsf(__builtin_fmaf(da, t, oa) - ca) - ha, eb = fabsf(__builtin_fmaf(db, t, ob) - cb) - hb;
    const float outside = fmaxf(ea, eb);
    const float viol = fmaxf(outside, fmaxf(tmin - t, t - h.t));
    take_hit((viol <= 0.0f) & (id != src), t, id, h);
    if (PDF && light) pdf += ((outside <= 0.0f) & (t >= 0.0001f)) ? (t * t) * (fabsf(in) * inv_area) : 0.0f;
}
// the two rects of a pair record; the caller has split the ray by the run's axis
template <bool PDF>
RT_DEV void small_pair(const u32x16 &R, float on, float in, float oa, float da, float ob, float db, float tmin, int src, SmallHit &h, float &pdf) {
    const uint32_t head = R[0];
    small_rect<PDF>(recf(R, 1), recf(R, 3), recf(R, 4), recf(R, 5), recf(R, 6), (int)((head >> 16) & 0xffu), (head & (1u << 11)) != 0, recf(R, 11), on, in, oa, da, ob, db, tmin, src, h, pdf);
    if (head & (1u << 10))
        small_rect<PDF>(recf(R, 2), recf(R, 7), recf(R, 8), recf(R, 9), recf(R, 10), (int)(head >> 24), (head & (1u << 12)) != 0, recf(R, 12), on, in, oa, da, ob, db, tmin, src, h, pdf);
}
template <bool PDF>
RT_DEV void small_pair_any(const u32x16 &R, f3 o, f3 d, f3 inv, float tmin, int src, SmallHit &h, float &pdf) {
    const uint32_t axis = (R[0] >> 8) & 3u;                            // in-plane axes: XY -> x, y   XZ -> x, z   YZ -> y, z
    if (axis == RTC_AXIS_XY) small_pair<PDF>(R, o.z, inv.z, o.x, d.x, o.y, d.y, tmin, src, h, pdf);
    else if (axis == RTC_AXIS_XZ) small_pair<PDF>(R, o.y, inv.y, o.x, d.x, o.z, d.z, tmin, src, h, pdf);
    else small_pair<PDF>(R, o.x, inv.x, o.y, d.y, o.z, d.z, tmin, src, h, pdf);
}
// box_hit (hittable.c:521-523 = the list scan over the sides) as one slab test, also for a box with openings: the surface is entered
// at the largest near-plane parameter and left at the smallest far-plane parameter; a ray that starts inside (or on) the box hits
// the exit face.  The face gives the leaf id, so shading sees the same record as after six rect tests.
// A ray that starts on a face (src) cannot hit that face again (rt_core.hpp box_leaf_t): not as the entry, not as the exit.
// Centre / half-extent form: near, far = (c - o) / d -+ h / |d| -- no per-axis min / max.  Faces: the axis that set the parameter
// and, on it, the side the ray comes from -- code 2 axis + (d < 0) for the entry (lo face when d > 0), that code ^ 1 for the exit;
// the code picks the face's id byte out of the record's two id words with one v_perm_b32.
RT_DEV void small_box(const u32x16 &R, f3 o, f3 inv, float tmin, int src, SmallHit &h) {
    const float tcx = __builtin_fmaf(recf(R, 3), inv.x, -o.x * inv.x), tcy = __builtin_fmaf(recf(R, 4), inv.y, -o.y * inv.y), tcz = __builtin_fmaf(recf(R, 5), inv.z, -o.z * inv.z);
    const float hx = recf(R, 6) * fabsf(inv.x), hy = recf(R, 7) * fabsf(inv.y), hz = recf(R, 8) * fabsf(inv.z);
    const float nx = tcx - hx, ny = tcy - hy, nz = tcz - hz, fx = tcx + hx, fy = tcy + hy, fz = tcz + hz;
    const float t_in = fmaxf(fmaxf(nx, ny), nz), t_out = fminf(fminf(fx, fy), fz);
    const uint32_t kx = (__float_as_uint(inv.x) >> 31) | 0x0c0c0c00u, ky = (__float_as_uint(inv.y) >> 31) | 0x0c0c0c02u, kz = (__float_as_uint(inv.z) >> 31) | 0x0c0c0c04u;
    const uint32_t k_in = nx == t_in ? kx : (ny == t_in ? ky : kz), k_out = (fx == t_out ? kx : (fy == t_out ? ky : kz)) ^ 1u;
    const uint32_t ids_hi = R[2];                                     // (a VGPR copy: v_perm_b32 may read one scalar operand)
    const uint32_t id_in = __builtin_amdgcn_perm(ids_hi, R[1], k_in), id_out = __builtin_amdgcn_perm(ids_hi, R[1], k_out);
    // the entry point counts when it is in range and that face exists; otherwise (ray starts inside, or enters through an
    // opening) the exit point
    const bool enter = (t_in >= tmin) & (id_in != 0xffu) & (id_in != (uint32_t)src);
    const float t = enter ? t_in : t_out;
    const uint32_t id = enter ? id_in : id_out;
    const bool hit = (t_in <= t_out) & (t >= tmin) & (t <= h.t) & (id != 0xffu) & (id != (uint32_t)src);
    take_hit(hit, t, (int)id, h);
}
// world space, closest-approach form (rt_core.hpp sphere_t<float>); |dw| = 1 (every ray of this kernel): the quadratic's a is 1.
// `src` = leaf the ray starts on: a sphere left behind has the roots {0, 2 tc} (rt_core.hpp sphere_leave_t).
template <bool PDF>
RT_DEV void small_sphere(const u32x16 &R, f3 ow, f3 dw, float tmin, int src, SmallHit &h, float &pdf) {
    const uint32_t head = R[0];
    const int id = (int)((head >> 8) & 0xffu);
    const float r2 = recf(R, 4);
    const f3 oc = ow - f3mk(recf(R, 1), recf(R, 2), recf(R, 3));
    const float tc = -dot(oc, dw);
    const f3 l = oc + dw * tc;
    const float q = r2 - len2(l);
    const float sq = __builtin_amdgcn_sqrtf(fmaxf(q, 0.0f));
    const bool from = id == src;                                 // leaving this sphere: roots are {0, 2 tc}
    const float t0 = from ? 2.0f * tc : tc - sq, t1 = tc + sq;
    const bool ok0 = (t0 >= tmin) & (t0 <= h.t), ok1 = (t1 >= tmin) & (t1 <= h.t) & !from;
    const float t = ok0 ? t0 : t1;
    take_hit((from | (q >= 0.0f)) & (ok0 | ok1), t, id, h);
    if (PDF && (head & (1u << 16))) {                            // wave-uniform: this sphere is a light-list entry (hittable.c:760-768)
        const float need = 0.001f - tc;                          // the far root tc + sqrt(q) decides alone: small_light_pdf_one
        const bool seen = !(q < 0.0f) & (!(need > 0.0f) | !(q < need * need));
        // 1 / (2 pi (1 - cos_max)), cos_max = sqrt(1 - r^2 / d^2):  1 / (1 - c) = (1 + c) d^2 / r^2, so the value is
        // (d^2 + sqrt((d^2 - r^2) d^2)) / (2 pi r^2) -- one square root, no reciprocal; d^2 = |l|^2 + tc^2 for a unit
        // direction.  An origin inside the sphere gives NaN like the reference's sqrt of a negative number.
        const float d2 = __builtin_fmaf(tc, tc, len2(l));
        pdf += seen ? (d2 + __builtin_amdgcn_sqrtf((d2 - r2) * d2)) * recf(R, 5) : 0.0f;
    }
}
// hittable_list_hit (hittable_list.c:125-143) over the record stream of the pack (rtc_tables.h SmallRec): one s_load_dwordx16 per
// record by a wave-uniform cursor.  The records that run on the world ray are sorted by type, so each type is one counted loop with
// one arm and the hit (t, leaf) stays in its registers -- a single loop with a type switch cost ten scalar instructions per record
// for the dispatch and four moves to bring each arm's result back to the loop's registers.  A box alone behind a y-rotation is one
// self-contained record among them; what is left -- general rotations, spaces with loose rects -- follows in a loop that carries its
// current space and does look at the type.  Wrapper stacks are pre-collapsed affine maps.
// `pdf` (out) = sum over the light-list entries that are leaves of this pack of their pdf for this ray (fused scenes, flatten.cpp
// mark_small_lights; 0 otherwise): what small_light_pdf_sum would return for (ow, dw), taken from the tests that run here anyway
// (lights are matched to world-space leaves only).
template <bool PDF>
RT_DEV SmallHit small_closest_hit(const SmallScene &ss, f3 ow, f3 dw, float tmin, int src, float &pdf) {
    pdf = 0.0f;
    SmallHit h; h.t = 3.0e38f; h.leaf = -1;
    const f3 invw = rcp3(dw);
    const u32x16 *prog = reinterpret_cast<const u32x16 *>(ss.prog);
    for (int k = 0; k < ss.n_box; ++k) { const u32x16 R = *prog++; small_box(R, ow, invw, tmin, src, h); }
    for (int k = 0; k < ss.n_xbox; ++k) {   // a rotation about y and a translation (every rotated object of the reference's scenes): y passes through
        const u32x16 R = *prog++;
        const float m0 = recf(R, 9), m2 = recf(R, 10), m6 = recf(R, 11), m8 = recf(R, 12);
        const f3 o = f3mk(m0 * ow.x + m2 * ow.z + recf(R, 13), ow.y + recf(R, 14), m6 * ow.x + m8 * ow.z + recf(R, 15));
        const float dx = m0 * dw.x + m2 * dw.z, dz = m6 * dw.x + m8 * dw.z;
        small_box(R, o, f3mk(__builtin_amdgcn_rcpf(dx), invw.y, __builtin_amdgcn_rcpf(dz)), tmin, src, h);
    }
    for (int k = 0; k < ss.n_xy; ++k) { const u32x16 R = *prog++; small_pair<PDF>(R, ow.z, invw.z, ow.x, dw.x, ow.y, dw.y, tmin, src, h, pdf); }
    for (int k = 0; k < ss.n_xz; ++k) { const u32x16 R = *prog++; small_pair<PDF>(R, ow.y, invw.y, ow.x, dw.x, ow.z, dw.z, tmin, src, h, pdf); }
    for (int k = 0; k < ss.n_yz; ++k) { const u32x16 R = *prog++; small_pair<PDF>(R, ow.x, invw.x, ow.y, dw.y, ow.z, dw.z, tmin, src, h, pdf); }
    for (int k = 0; k < ss.n_sphere; ++k) { const u32x16 R = *prog++; small_sphere<PDF>(R, ow, dw, tmin, src, h, pdf); }
    const u32x16 *const end = reinterpret_cast<const u32x16 *>(ss.prog) + ss.n_recs;
    if (prog < end) {
        f3 o = ow, d = dw, inv = invw;
        float no_pdf = 0.0f;
        for (; prog < end; ++prog) {
            const u32x16 R = *prog;
            const uint32_t type = R[0] & 0xffu;
            if (type == RTC_SR_XFORM) {
                if (R[0] & (1u << 8)) {
                    o = f3mk(recf(R, 1) * ow.x + recf(R, 3) * ow.z + recf(R, 10), ow.y + recf(R, 11), recf(R, 7) * ow.x + recf(R, 9) * ow.z + recf(R, 12));
                    d = f3mk(recf(R, 1) * dw.x + recf(R, 3) * dw.z, dw.y, recf(R, 7) * dw.x + recf(R, 9) * dw.z);
                    inv = f3mk(__builtin_amdgcn_rcpf(d.x), invw.y, __builtin_amdgcn_rcpf(d.z));
                } else {
                    o = f3mk(recf(R, 1) * ow.x + recf(R, 2) * ow.y + recf(R, 3) * ow.z + recf(R, 10), recf(R, 4) * ow.x + recf(R, 5) * ow.y + recf(R, 6) * ow.z + recf(R, 11),
                             recf(R, 7) * ow.x + recf(R, 8) * ow.y + recf(R, 9) * ow.z + recf(R, 12));
                    d = f3mk(recf(R, 1) * dw.x + recf(R, 2) * dw.y + recf(R, 3) * dw.z, recf(R, 4) * dw.x + recf(R, 5) * dw.y + recf(R, 6) * dw.z, recf(R, 7) * dw.x + recf(R, 8) * dw.y + recf(R, 9) * dw.z);
                    inv = rcp3(d);
                }
            } else if (type == RTC_SR_BOX) small_box(R, o, inv, tmin, src, h);
            else small_pair_any<false>(R, o, d, inv, tmin, src, h, no_pdf);
        }
    }
    return h;
}

// hittable_list_pdf_value over the lights list (hittable_list.c:165-172, hittable.c:760-788): wave-uniform loop.  The maths
// of rt_shade.hpp light_pdf_one for a UNIT direction v (every ray of this kernel is normalised where it is made), with
// what does not depend on the ray taken from the record (1/area, r^2):
//   rect  : d^2 / (cos * area) = t^2 / (|vn| area) = t^2 |1/vn| (1/area) for a unit v   -- one rcp, no second one
//   sphere: the far root decides alone (it is the larger one): q >= 0 and (tc >= tmin or q >= (tmin - tc)^2) -- no sqrt
RT_DEV float small_light_pdf_one(const LightT<float> &l, f3 o, f3 v) {
    if (l.kind == RTC_LIGHT_RECT) {                                                   // :772-788, tmin 1e-4
        const bool xy = l.axis == RTC_AXIS_XY, xz = l.axis == RTC_AXIS_XZ, yz = l.axis == RTC_AXIS_YZ;
        const float on = xy ? o.z : (xz ? o.y : o.x), vn = xy ? v.z : (xz ? v.y : v.x);
        const float oa = yz ? o.y : o.x, va = yz ? v.y : v.x, ob = xy ? o.y : o.z, vb = xy ? v.y : v.z;
        const float a0 = yz ? l.d[2] : l.d[0], a1 = yz ? l.d[3] : l.d[1], b0 = xy ? l.d[2] : l.d[4], b1 = xy ? l.d[3] : l.d[5];
        const float inv_vn = __builtin_amdgcn_rcpf(vn);
        const float t = (l.d[6] - on) * inv_vn;                                       // rect_hit :494-517
        const float pa = oa + t * va, pb = ob + t * vb;
        const bool hit = !(t < 0.0001f) & !(pa < a0) & !(pa > a1) & !(pb < b0) & !(pb > b1);   // a NaN t passes (F20)
        return hit ? (t * t) * (fabsf(inv_vn) * l.d[7]) : 0.0f;
    }
    if (l.kind == RTC_LIGHT_SPHERE) {                                                 // :760-768, tmin 1e-3
        const f3 oc = o - f3mk(l.d[0], l.d[1], l.d[2]);
        const float tc = -dot(oc, v);                                                 // closest-approach form, rt_core.hpp sphere_t<float>
        const f3 lv = oc + v * tc;
        const float q = l.d[4] - len2(lv);
        const float need = 0.001f - tc;                                               // far root tc + sqrt(q) >= tmin
        const bool hit = !(q < 0.0f) & (!(need > 0.0f) | !(q < need * need));           // NaN passes as in sphere_t
        const float cos_theta_max = __builtin_amdgcn_sqrtf(1.0f - l.d[4] * __builtin_amdgcn_rcpf(len2(oc)));
        return hit ? __builtin_amdgcn_rcpf(2.0f * RT_PI * (1.0f - cos_theta_max)) : 0.0f;
    }
    return 0.0f;
}
// sum over the list; the caller multiplies by (mixture weight / number of lights), a launch constant
RT_DEV float small_light_pdf_sum(const SceneT<float> &sc, f3 o, f3 v) {
    float sum = 0.0f;
    for (int i = 0; i < sc.n_lights; ++i) sum += small_light_pdf_one(sc.lights[i], o, v);
    return sum;
}
// material.c:144-157 for a unit incoming direction (rt_shade.hpp dielectric_dir without its normalisation); the result is a unit vector
// (1 / ir and Schlick's r0 come from the shading record: both are constants of the material)
RT_DEV f3 small_dielectric_dir(float ir, float inv_ir, float r0, f3 ud, f3 n, bool front, float xi) {
    // Both outcomes are a ud + b n -- reflect (vec3.c:26): (1, 2 c), c = -ud.n; refract (vec3.c:44-50): (eta, eta cos + k), k = -sqrt|1 - |perp|^2|,
    // |perp|^2 = eta^2 |ud + n cos|^2 = eta^2 (1 - 2 cos c + cos^2) for unit ud, n -- so the choice is two selects, not a divergent branch
    // around two vector expressions (the branch cost twenty scalar instructions; both sides ran for a mixed wave anyway).
    const float ratio = front ? inv_ir : ir;
    const float c = -dot(ud, n), cos_theta = fminf(c, 1.0f);
    const float sin_theta = __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));
    const float m = 1.0f - cos_theta, m2 = m * m;
    const bool refl = (ratio * sin_theta > 1.0f) | (r0 + (1.0f - r0) * (m2 * m2 * m) > xi);   // material.c:142,152
    const float perp2 = (ratio * ratio) * (1.0f - 2.0f * cos_theta * c + cos_theta * cos_theta);
    const float k = -__builtin_amdgcn_sqrtf(fabsf(1.0f - perp2));
    const float a = refl ? 1.0f : ratio, b = refl ? 2.0f * c : ratio * cos_theta + k;
    return ud * a + n * b;
}
// Duff et al. 2017 branch-free orthonormal basis (any basis around w gives the same uniform azimuth as onb.c:7-12)
RT_DEV void basis_around(f3 w, f3 &u, f3 &v) {
    const float sign = __builtin_copysignf(1.0f, w.z);
    const float a = -__builtin_amdgcn_rcpf(sign + w.z), b = w.x * w.y * a;
    u = f3mk(1.0f + sign * w.x * w.x * a, sign * b, -sign * w.x);
    v = f3mk(b, sign + w.y * w.y * a, -w.y);
}
// hittable_list_random (hittable_list.c:174-180) + hittable_random (hittable.c:821-848) over the light records of the pack (rtc_tables.h):
// a wave-uniform loop with one s_load_dwordx16 per light, the picked light a per-lane mask.  (Reading LightT -- kind, axis and eight
// reals chosen by axis -- cost thirty scalar instructions and a dozen moves per rect light: every `axis ==` became a scalar select.)
RT_DEV f3 small_light_sample(const SmallScene &ss, int pick, f3 o, float r1, float r2) {
    f3 dir = f3mk(1, 0, 0);                                                                // hittable.c:861-863
    const u32x16 *const lights = reinterpret_cast<const u32x16 *>(ss.lights);
    for (int i = 0; i < ss.n_lights; ++i) {
        const u32x16 R = lights[i];
        if (pick != i) continue;
        if (R[0] == RTC_LIGHT_RECT) {                                                      // hittable.c:838-848: a uniform point of the rect
            dir = f3mk(__builtin_fmaf(recf(R, 7), r2, __builtin_fmaf(recf(R, 4), r1, recf(R, 1) - o.x)), __builtin_fmaf(recf(R, 8), r2, __builtin_fmaf(recf(R, 5), r1, recf(R, 2) - o.y)),
                       __builtin_fmaf(recf(R, 9), r2, __builtin_fmaf(recf(R, 6), r1, recf(R, 3) - o.z)));
        } else if (R[0] == RTC_LIGHT_SPHERE) {                                             // hittable.c:821-835, cone towards the sphere
            const f3 axis = f3mk(recf(R, 1), recf(R, 2), recf(R, 3)) - o;
            const float d2 = len2(axis);
            const float inv_d = __builtin_amdgcn_rsqf(d2);
            const f3 w = axis * inv_d;
            f3 u, v; basis_around(w, u, v);
            const float z = 1.0f + r2 * (__builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - recf(R, 4) * (inv_d * inv_d))) - 1.0f);
            const float s = __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - z * z));
            dir = u * (cos2pi(r1) * s) + (v * (sin2pi(r1) * s) + w * z);
        }
    }
    return dir;
}

// PROBE = true: the one-sample cost probe of launch_pathtrace_small (its own kernel name in profiles): writes the rays of
// each tile to tile_cost and nothing else.
// LENS = false: pinhole camera (aperture 0, every reference scene but Book 1 / Book 2 random): no lens sample in the loop.
// FUSED = true (03 integrator only): every light of the scene is a leaf of the pack and its pdf comes out of the next ray's
// intersection (flatten.cpp mark_small_lights); the stand-alone light-pdf loop is then not even compiled in.
// (Two more variants were built, measured and taken out again -- the paths of a workgroup regrouped by material through LDS, +32 %;
// one camera Philox block per sample pair, +2.5 %: DESIGN.md 3.1, profiles/r02a_regroup; the code is in the history up to 9791647.)
// The kernel's arguments are ONE struct so that the parts of the kernel that run rarely -- the refill of the primary-ray buffers,
// the epilogue -- can read theirs again where they stand: small_args_again() hands out the kernel-argument segment through a
// pointer the compiler cannot see through (s_load where it is used), instead of twenty camera and image constants held in SGPRs
// across the bounce loop next to the table pointers, the Philox key and the mixture weights.  The headline variant spilled 14
// SGPRs to vector lanes before (LENS variants 21-24), none now.
struct SmallArgs {
    SmallScene ss; SceneT<float> sc; CameraT<float> cam; RenderParams P;
    float4 *accum; unsigned long long *counters; const int *tile_perm; int *tile_cost; float4 *partial; int log2_chunks;
};
// (`when` = a wave-uniform value that changes from use to use: the asm statement is NOT volatile -- a volatile one counts as a write
// to memory, after which hipcc fetches the scene records of the bounce loop with vector loads instead of s_load -- so what keeps it
// from being hoisted out of the loop, and the loads behind it with it, is that its result depends on `when`.)
RT_DEV const SmallArgs &small_args_again(int when) {
    const SmallArgs __attribute__((address_space(4))) *p = (const SmallArgs __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    asm("" : "+s"(p) : "s"(when));
    return *(const SmallArgs *)p;
}
template <int INTEGRATOR, bool LENS, bool PROBE, bool FUSED>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RT_SMALL_WAVES))) k_pathtrace_small(SmallArgs a) {
    const SmallScene &ss = a.ss; const SceneT<float> &sc = a.sc; const RenderParams &P = a.P; const int log2_chunks = a.log2_chunks;
    __shared__ SmallShade lds_shade[RTC_SMALL_MAX_LEAVES];
    {   // 48 B x n_leaves, copied dword by dword
        const uint32_t *src = (const uint32_t *)ss.shade;
        uint32_t *dst = (uint32_t *)lds_shade;
        for (int i = threadIdx.x; i < ss.n_leaves * (int)(sizeof(SmallShade) / 4); i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // work item = (tile, chunk of the sample range): 2^log2_chunks waves share a tile, each with its own samples
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (P.n_tiles << log2_chunks)) return;              // wave-uniform
    int tile = item >> log2_chunks;
    const int chunk = item & ((1 << log2_chunks) - 1);
    if (!PROBE && a.tile_perm) tile = a.tile_perm[tile];   // heaviest tiles first (launch_pathtrace_small)
    const int spp_all = P.s_end - P.s_begin, q = spp_all >> log2_chunks, r = spp_all & ((1 << log2_chunks) - 1);
    const int s_lo = P.s_begin + chunk * q + (chunk < r ? chunk : r), s_hi = s_lo + q + (chunk < r ? 1 : 0);
    // Lanes are not pixels.  The 64 * S paths of the work item (64 pixels x S samples) are numbered q = sample * 64 + pixel
    // and handed out in that order to whichever lanes need a path (a wave-uniform counter and a prefix count: no atomics).
    // With a lane per pixel, the lane of a border pixel (one ray per sample) ran out of work long before the lane of an
    // interior pixel (five) and 9.5 % of the lane-iterations idled; now all lanes finish within a path of each other
    // (fill 0.986).  A finished path is added to the tile's sums in LDS (ds_add_f32; one wave, lock-step: the order is fixed,
    // the result reproducible; consecutive paths are different pixels, so the lanes that finish together rarely meet on an
    // address -- numbering the paths pixel-major put whole waves on one address and cost 10-16 %); lane l writes pixel l of
    // the tile at the end.
    __shared__ float lds_sum[4][64 * 3];
    float *const tile_sum = lds_sum[threadIdx.x >> 6];
    tile_sum[lane] = 0.0f; tile_sum[64 + lane] = 0.0f; tile_sum[128 + lane] = 0.0f;
    const int tx0 = (tile % P.tiles_x) * 8, ty0 = (tile / P.tiles_x) * 8;
    const int S = s_hi - s_lo, n_paths = 64 * S;
    const f3 bg = f3mk(sc.background[0], sc.background[1], sc.background[2]);

    f3 T = f3mk(1, 1, 1), ro = f3mk(0, 0, 0), rd = f3mk(0, 0, 1);
    unsigned nrays = 0, iters = 0;
    int q_next = 0;                                                       // wave-uniform: paths handed out so far
    int depth = 0, src = -1, cur_pix = 0;
    uint32_t s_cur = 0, pixel = 0;
    bool alive = false;
    // Fused light pdfs (03 integrator, scenes whose lights are leaves of the pack): a diffuse bounce leaves the scattering pdf of
    // its new direction in `pend` and multiplies the throughput by the albedo only; the mixture pdf pdf.c:119-121 needs the
    // light-list pdf of that direction, which the NEXT iteration's intersection of that very ray delivers (`pdf_ray`), and the
    // throughput is completed there -- before anything reads it.  pend = 0: nothing pending.
    constexpr bool fused = FUSED && INTEGRATOR == 0;
    float pend = 0.0f;
    constexpr int kVoid = 1 << 30;   // flag on a buffered path number: its pixel is outside the image (ragged right / top edge), nothing to trace
    auto locate = [&](int q, int &pix, uint32_t &smp) { pix = q & 63; smp = (uint32_t)(s_lo + ((q & (kVoid - 1)) >> 6)); };   // path number q = sample * 64 + pixel
    // Primary rays are generated ahead, two per lane (b0 is used first): a third of the lanes ends a path in every
    // iteration, so generating on demand runs the camera code (a Philox block and the ray set-up) in EVERY iteration
    // for a third of the wave; filled only when some lane has run dry, it runs every few iterations for most of it
    // (a third buffered ray changes nothing; a wave-wide batch of 64 rays through an LDS ring ran the camera code at full
    // width but made every path start wait for an LDS read: 5.6 % slower).  A pixel outside the image (ragged right / top
    // edge) is buffered with the kVoid flag on its path number.
    f3 b0d = rd, b1d = rd, b0o = ro, b1o = ro;
    int b0q = 0, b1q = 0, n_buf = 0;                                      // buffered paths and their numbers
    auto primary = [&](int &q, f3 &o, f3 &d) {                            // main.c:150-152 + camera.c:46-53
        int pix; uint32_t smp; locate(q, pix, smp);
        const int px = tx0 + (pix & 7), py = ty0 + (pix >> 3);
        const Rng4 c = rng_block((uint32_t)(py * P.width + px), smp, RT_TAG_CAMERA, P.seed_lo, P.seed_hi);
        const SmallArgs &t = small_args_again(q_next);                            // the camera's constants: read here, not held across the loop
        f3 dir;
        if (LENS) {
            const float u = ((float)px + c.x) / (float)(t.P.width - 1), v = ((float)py + c.y) / (float)(t.P.height - 1);
            const float rr = __builtin_amdgcn_sqrtf(c.z);
            const RayT<float> r = camera_ray<float>(t.cam, u, v, rr * cos2pi(c.w), rr * sin2pi(c.w), 0.0f);   // no leaf moves: the ray time is never read
            o = r.o; dir = r.d;
        } else {   // pinhole: direction = base + hs (px + jx) + vs (py + jy), folded on the host (rtc_api.cpp; camera.c:46-53, main.c:150-151)
            const float fx = (float)px + c.x, fy = (float)py + c.y;
            dir = f3mk(__builtin_fmaf(t.P.cam_vs[0], fy, __builtin_fmaf(t.P.cam_hs[0], fx, t.P.cam_base[0])), __builtin_fmaf(t.P.cam_vs[1], fy, __builtin_fmaf(t.P.cam_hs[1], fx, t.P.cam_base[1])),
                       __builtin_fmaf(t.P.cam_vs[2], fy, __builtin_fmaf(t.P.cam_hs[2], fx, t.P.cam_base[2])));
            o = f3mk(0, 0, 0);                                              // (not buffered: the bounce loop takes the camera's origin)
        }
        d = normalize(dir);                                                 // unit directions throughout the bounce loop
        q |= ((px < t.P.width) & (py < t.P.height)) ? 0 : kVoid;            // (no branch around the camera code: such pixels are rare)
    };
    // the next paths go to the lanes of `want`, in lane order
    auto hand_out = [&](bool want, int &q) -> bool {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(want);
        q = q_next + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        q_next += (int)__builtin_popcountll(m);
        return want && q < n_paths;
    };

    for (;;) {
        if (P.max_depth <= 0) break;                                       // ray.c:52
        const bool starving = !alive && n_buf == 0;
        if (q_next < n_paths && __builtin_amdgcn_ballot_w64(starving) != 0) {   // (waiting until 4 or 8 lanes starve: no gain, 0 to +2 %)
            int q;
            if (hand_out(n_buf == 0, q)) { primary(q, b0o, b0d); b0q = q; n_buf = 1; }
            if (hand_out(n_buf == 1, q)) { primary(q, b1o, b1d); b1q = q; n_buf = 2; }
        }
        if (!alive && n_buf > 0) {
            const int path_q = b0q;
            locate(path_q, cur_pix, s_cur); --n_buf;
            pixel = (uint32_t)((ty0 + (cur_pix >> 3)) * P.width + tx0 + (cur_pix & 7));
            rd = b0d; b0d = b1d; b0q = b1q;
            if (LENS) { ro = b0o; b0o = b1o; } else ro = f3mk(a.cam.origin[0], a.cam.origin[1], a.cam.origin[2]);
            T = f3mk(1, 1, 1);
            depth = 0; src = -1; pend = 0.0f;
            alive = path_q < kVoid;                                         // a pixel outside the image: nothing to trace
        }
        // (a buffered ray may be void -- a pixel outside the image: a wave whose lanes all drew one goes on while rays are buffered)
        if (__builtin_amdgcn_ballot_w64(alive || n_buf > 0) == 0) break;
        ++iters;
        SmallHit h; h.t = 0.0f; h.leaf = -1;
        if (alive) {
            float pdf_ray;
            h = small_closest_hit<fused>(ss, ro, rd, 0.001f, src, pdf_ray);  // ray.c:56
            if (fused) {                                              // complete the previous bounce: T *= pdf_cos / pdf_val (ray.c:80-83)
                // (pend > 0 and a cosine weight > 0 make pdf_val > 0.  A pdf_val of 0 or NaN -- an underflowed mixture, a ray from inside a
                // light sphere -- is the reference's 0/0: nothing more arrives from this path.  The stand-alone light loop ends such a path at
                // the bounce that made the direction; here the direction's ray has just been traced to learn its pdf, so the path ends now,
                // carrying nothing, and that one ray is not counted: both forms report the same paths and the same rays.  No branch: selects.)
                const float pdf_val = P.light_weight_each * pdf_ray + P.cosine_weight * pend;
                const bool pending = pend > 0.0f, dead = pending & !(pdf_val > 0.0f);
                const float f = pending ? pend * __builtin_amdgcn_rcpf(pdf_val) : 1.0f;
                T = T * f;
                pend = 0.0f;
                nrays += dead ? 0u : 1u;
                alive = !dead;
            } else ++nrays;
        }
        if (alive) {
            // `done` is an integer in a VGPR on purpose: a bool that is set in divergent arms lives in an SGPR mask, and every assignment to
            // it costs three scalar instructions to merge -- scalar instructions take issue slots like vector ones.  The
            // material arms are separate ifs (they exclude each other by construction), not an else-if chain: no else-mask bookkeeping.
            int done = 0;
            f3 L = f3mk(0, 0, 0);   // a path collects radiance once, where it ends (emitter or background): nothing to carry between bounces
            if (h.leaf < 0) {
                L = T * (INTEGRATOR == 2 ? sky_color(rd) : bg);
                done = 1;
            } else {
                const SmallShade sh = lds_shade[h.leaf];
                const uint32_t mtype = sh.meta & 0xff;
                const f3 p = ro + rd * h.t;
                // outward normal: the record holds (n, 0) for a rect and (-c / r, 1 / r) for a sphere (rtc_api.cpp upload_scene), so that
                // n_out = p * g3 + g is the rect's n or the sphere's (p - c) / r without a branch or a select
                const f3 n_out = f3mk(__builtin_fmaf(p.x, sh.g[3], sh.g[0]), __builtin_fmaf(p.y, sh.g[3], sh.g[1]), __builtin_fmaf(p.z, sh.g[3], sh.g[2]));
                bool front = dot(rd, n_out) < 0.0f;                        // hit_record_set_facenormal, hittable.c:13-16
                const f3 n = front ? n_out : -n_out;
                if (sh.meta & RTC_SF_FORCED_FRONT) front = true;            // translate/rotate re-face an already faced normal (F11)
                if (sh.meta & RTC_SF_FLIP) front = !front;                  // flip_face_hit, hittable.c:587-591
                const f3 col = f3mk(sh.rgb[0], sh.rgb[1], sh.rgb[2]);
                Rng4 xi = rng_block(pixel, s_cur, (uint32_t)depth, P.seed_lo, P.seed_hi);
                f3 nd = rd;
                if (mtype == RTC_MAT_LIGHT) {                               // material.c:229-235 (03: front face only, F15); no scatter
                    if (INTEGRATOR == 1 || (INTEGRATOR == 0 && front)) L = T * col;
                    done = 1;
                }
                if (mtype == RTC_MAT_METAL) {                               // material.c:130-138 / 02 material.c:132-141
                    nd = reflect(rd, n);                                     // |rd| = 1
                    if (ss.any_fuzz) nd = normalize(nd + uniform_ball(xi.x, xi.y, xi.z) * sh.param);
                    if (INTEGRATOR != 0 && !(dot(nd, n) > 0.0f)) done = 1;
                    T = T * col;
                }
                if (mtype == RTC_MAT_DIELECTRIC)                            // material.c:144-162
                    nd = small_dielectric_dir(sh.param, sh.pad[0], sh.pad[1], rd, n, front, xi.x);
                if (mtype == RTC_MAT_ISOTROPIC) done = 1;                   // only reachable through a medium; none in a small scene
                if (mtype == RTC_MAT_LAMBERTIAN) {
                if (INTEGRATOR != 0) {                                      // 02 material.c:120-128: normal + random unit vector
                    nd = n + uniform_sphere(xi.x, xi.y);
                    if (fabsf(nd.x) < 1e-8f && fabsf(nd.y) < 1e-8f && fabsf(nd.z) < 1e-8f) nd = n;
                    nd = normalize(nd);
                    T = T * col;
                } else {                                                    // 03: ray.c:66-83, mixture of light-list and cosine pdfs
                    const float w = P.light_weight;
                    if (ss.n_lights > 0 && xi.x < w) {                       // pdf.c:175-181
                        int pick = (int)((float)ss.n_lights * xi.y);         // random_int(0, n-1), utils.h:26
                        pick = pick < ss.n_lights ? pick : ss.n_lights - 1;
                        nd = small_light_sample(ss, pick, p, xi.z, xi.w);
                    } else {                                                 // cosine law about n (vec3.c:97-107 + onb.c:25-30)
                        nd = n + uniform_sphere(xi.z, xi.w);
                        if (len2(nd) < 1e-12f) nd = n;
                    }
                    nd = nd * __builtin_amdgcn_rsqf(len2(nd));              // unit from here on: the pdfs and the next bounce's tests drop |v|
                    const float pdf_cos = lambert_pdf(n, nd);                         // pdf.c:100-103 == material.c:205-208
                    if (fused) {                                            // the light-list pdf of nd arrives with the next intersection
                        if (!(pdf_cos > 0.0f)) done = 1;
                        else { T = T * col; pend = pdf_cos; }
                    } else {
                    const float pdf_light = sc.n_lights > 0 ? small_light_pdf_sum(sc, p, nd) : 0.0f;   // x 1 / n_lights below
                    const float pdf_val = P.light_weight_each * pdf_light + P.cosine_weight * pdf_cos;  // pdf.c:119-121
                    // scattering pdf 0: the reference recurses and multiplies by 0; pdf_val 0: 0/0 (scrubbed) -- nothing more arrives
                    if (!(pdf_cos > 0.0f) || !(pdf_val > 0.0f)) done = 1;
                    else T = T * (col * (pdf_cos * __builtin_amdgcn_rcpf(pdf_val)));
                    }
                }
                }
                ro = p; rd = nd; src = h.leaf;
            }
            ++depth;
            if ((done != 0) | (depth >= P.max_depth)) {
                alive = false;
                // finite and not all zero in two compares: |x| + |y| + |z| is NaN or inf when a component is, 0 when all are
                // (most paths of a scene with a black background end with nothing to add: no LDS traffic for those)
                const float mag = fabsf(L.x) + fabsf(L.y) + fabsf(L.z);
                if ((mag > 0.0f) & (mag < 3.0e38f)) { atomicAdd(&tile_sum[cur_pix], L.x); atomicAdd(&tile_sum[64 + cur_pix], L.y); atomicAdd(&tile_sum[128 + cur_pix], L.z); }
            }
        }
    }
    unsigned total = nrays;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
    const SmallArgs &e = small_args_again(q_next);
    if (PROBE) { if (lane == 0) e.tile_cost[tile] = (int)total; return; }   // cost probe: rays of this tile, nothing else is written
    const int px = tx0 + (lane & 7), py = ty0 + (lane >> 3);
    if (px < e.P.width && py < e.P.height) {                 // lane l writes pixel l of the tile
        const size_t pix_id = (size_t)py * e.P.width + px;
        const f3 sum = f3mk(tile_sum[lane], tile_sum[64 + lane], tile_sum[128 + lane]);
        if (e.log2_chunks == 0) {                                          // the only wave of this pixel: straight into the accumulator
            float4 acc = e.accum[pix_id];
            acc.x += sum.x; acc.y += sum.y; acc.z += sum.z; acc.w += (float)(s_hi - s_lo);
            e.accum[pix_id] = acc;
        } else e.partial[(size_t)chunk * ((size_t)e.P.width * e.P.height) + pix_id] = make_float4(sum.x, sum.y, sum.z, (float)(s_hi - s_lo));
    }
    if (lane == 0) { atomicAdd(e.counters, (unsigned long long)total); atomicAdd(e.counters + 1, (unsigned long long)iters); }
}

// accum += the partial sums of a pixel's chunks, in chunk order: the result does not depend on which wave finished first
__global__ void __launch_bounds__(256) k_small_combine(const float4 *__restrict__ partial, int n_pixels, int chunks, float4 *__restrict__ accum) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    float4 a = accum[i];
    for (int c = 0; c < chunks; ++c) { const float4 s = partial[(size_t)c * n_pixels + i]; a.x += s.x; a.y += s.y; a.z += s.z; a.w += s.w; }
    accum[i] = a;
}

// Tiles sorted by descending cost (rays of a one-sample probe): counting sort in one workgroup.  The order inside a cost
// bin depends on scheduling, which only changes WHEN a tile runs, never what it computes.
__global__ void __launch_bounds__(1024) k_tile_order(const int *__restrict__ cost, int n, int *__restrict__ perm) {
    __shared__ unsigned bin_start[1024];
    for (int b = threadIdx.x; b < 1024; b += 1024) bin_start[b] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += 1024) atomicAdd(&bin_start[1023 - min(max(cost[t], 0), 1023)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) { unsigned acc = 0; for (int b = 0; b < 1024; ++b) { const unsigned c = bin_start[b]; bin_start[b] = acc; acc += c; } }
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += 1024) perm[atomicAdd(&bin_start[1023 - min(max(cost[t], 0), 1023)], 1u)] = t;
}

// Known-answer probe of this file's own shading functions (they differ from rt_shade.hpp's: unit directions, folded hit
// tests, basis-free samplers), on supplied inputs; kernels_probe.hip has the general ones.
__global__ void __launch_bounds__(256) k_small_kat(SceneT<float> sc, SmallScene ss, int what, const double *__restrict__ in, unsigned n, double *__restrict__ out) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int si = rtc_kat_in_stride(what), so = rtc_kat_out_stride(what);
    const double *a = in + (size_t)si * i;
    double *o = out + (size_t)so * i;
    auto v3at = [&](int k) { return f3mk((float)a[k], (float)a[k + 1], (float)a[k + 2]); };
    if (what == RTC_KAT_LIGHT_PDF_SMALL) {            // the list pdf as the bounce loop computes it: sum over the lights for a UNIT direction, / n
        o[0] = sc.n_lights > 0 ? small_light_pdf_sum(sc, v3at(0), normalize(v3at(3))) / (float)sc.n_lights : 0.0f;
    } else if (what == RTC_KAT_LIGHT_SAMPLE_SMALL) {
        const f3 d = small_light_sample(ss, (int)a[3], v3at(0), (float)a[4], (float)a[5]);
        o[0] = d.x; o[1] = d.y; o[2] = d.z;
    } else if (what == RTC_KAT_DIELECTRIC_SMALL) {
        const float ir = (float)a[7], rr = (1.0f - ir) / (1.0f + ir);
        const f3 d = small_dielectric_dir(ir, 1.0f / ir, rr * rr, normalize(v3at(0)), v3at(3), a[6] != 0.0, (float)a[8]);
        o[0] = d.x; o[1] = d.y; o[2] = d.z;
    } else for (int k = 0; k < so; ++k) o[k] = 0.0;
}

}  // namespace

hipError_t launch_small_kat(const SceneT<float> &sc, const SmallScene &ss, int what, const double *in, unsigned n, double *out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_small_kat, dim3((n + 255) / 256), dim3(256), 0, stream, sc, ss, what, in, n, out);
    return hipGetLastError();
}

template <bool PROBE>
static void launch_small(const SmallScene &ss, const SceneT<float> &sc, const CameraT<float> &cam, const RenderParams &p, float4 *accum,
                         unsigned long long *counters, const int *perm, int *cost, float4 *partial, int log2_chunks, hipStream_t stream) {
    const unsigned blocks = (unsigned)((((long long)p.n_tiles << log2_chunks) + 3) / 4);
    const SmallArgs args{ss, sc, cam, p, accum, counters, perm, cost, partial, log2_chunks};
#define RT_SMALL_LAUNCH(I, L) hipLaunchKernelGGL((k_pathtrace_small<I, L, PROBE, false>), dim3(blocks), dim3(256), 0, stream, args)
    const bool lens = cam.lens_radius > 0.0f;
    if (p.integrator == 0 && ss.fused_lights) {
        if (lens) hipLaunchKernelGGL((k_pathtrace_small<0, true, PROBE, true>), dim3(blocks), dim3(256), 0, stream, args);
        else hipLaunchKernelGGL((k_pathtrace_small<0, false, PROBE, true>), dim3(blocks), dim3(256), 0, stream, args);
    }
    else if (p.integrator == 0) { if (lens) RT_SMALL_LAUNCH(0, true); else RT_SMALL_LAUNCH(0, false); }
    else if (p.integrator == 1) { if (lens) RT_SMALL_LAUNCH(1, true); else RT_SMALL_LAUNCH(1, false); }
    else { if (lens) RT_SMALL_LAUNCH(2, true); else RT_SMALL_LAUNCH(2, false); }
#undef RT_SMALL_LAUNCH
}

// tile_scratch: 2 * n_tiles ints (cost, order) or NULL.  A wave owns a tile for its samples of the launch, and an interior
// Cornell tile costs 5x a border tile: with tiles in image order the last heavy tiles run on a nearly empty GPU.
//  * A one-sample probe measures every tile's rays, the tiles are sorted, and the main launch takes them heaviest first, so
//    the tail is made of the cheapest tiles (1080p x 256 spp: 37.5 -> 33.5 ms).
//  * partial (2^log2_chunks * W * H float4, or NULL): the sample range of a tile is split over 2^log2_chunks waves whose
//    sums go to their own plane and are added in a fixed order afterwards -- four times finer work items for 0.1 ms of
//    extra HBM traffic, and still no atomics on radiance.
hipError_t launch_pathtrace_small(const SmallScene &ss, const SceneT<float> &sc, const CameraT<float> &cam, const RenderParams &p,
                                  float4 *accum, unsigned long long *counters, int *tile_scratch, int *ordered, float4 *partial, int log2_chunks, hipStream_t stream) {
    const bool have_order = ordered && *ordered;
    if (ordered) *ordered = have_order;
    if (p.n_tiles <= 0 || p.s_end <= p.s_begin) return hipSuccess;
    if (ss.n_leaves <= 0 || ss.n_leaves > RTC_SMALL_MAX_LEAVES) return hipErrorInvalidValue;
    if (!partial) log2_chunks = 0;
    const int *perm = nullptr;
    if (tile_scratch && have_order) perm = tile_scratch + p.n_tiles;
    else if (tile_scratch && p.s_end - p.s_begin >= 32 && p.n_tiles >= 2048 && p.max_depth > 0) {
        RenderParams probe = p; probe.s_end = probe.s_begin + 1;
        int *cost = tile_scratch, *order = tile_scratch + p.n_tiles;
        launch_small<true>(ss, sc, cam, probe, accum, counters, nullptr, cost, nullptr, 0, stream);
        hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, stream, cost, p.n_tiles, order);
        perm = order;
        if (ordered) *ordered = 1;
    }
    launch_small<false>(ss, sc, cam, p, accum, counters, perm, nullptr, partial, log2_chunks, stream);
    if (log2_chunks > 0) {
        const int n_pixels = p.width * p.height;
        hipLaunchKernelGGL(k_small_combine, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, stream, partial, n_pixels, 1 << log2_chunks, accum);
    }
    return hipGetLastError();
}
