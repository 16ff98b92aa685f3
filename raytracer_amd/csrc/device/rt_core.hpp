// rt_core.hpp -- geometry half of the hot path for gfx950 (product code, device only).
//
// Templated on the real type: R = double instantiates the bit-exact check path (built with -ffp-contract=off so
// no mul+add pair is fused; IEEE / and sqrt), R = float the production path (contraction on, v_rcp/v_rsq/v_sqrt).
// The FP64 instantiation reproduces the operation order of the reference's hit functions; citations are
// file:line under /root/reference/03_FullTracer/.
#pragma once
#include <hip/hip_runtime.h>

#include "../rtc_tables.h"

#define RT_DEV __device__ __forceinline__

template <typename R> struct V3 { R x, y, z; };
template <typename R> RT_DEV V3<R> mk3(R x, R y, R z) { V3<R> v; v.x = x; v.y = y; v.z = z; return v; }
template <typename R> RT_DEV V3<R> operator+(V3<R> a, V3<R> b) { return mk3<R>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename R> RT_DEV V3<R> operator-(V3<R> a, V3<R> b) { return mk3<R>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename R> RT_DEV V3<R> operator*(V3<R> a, V3<R> b) { return mk3<R>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <typename R> RT_DEV V3<R> operator*(V3<R> a, R k) { return mk3<R>(a.x * k, a.y * k, a.z * k); }
template <typename R> RT_DEV V3<R> operator-(V3<R> a) { return mk3<R>(-a.x, -a.y, -a.z); }
template <typename R> RT_DEV R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // vec3.c:22, left to right
template <typename R> RT_DEV R len2(V3<R> a) { return (a.x * a.x) + (a.y * a.y) + (a.z * a.z); }          // vec3.c:21
template <typename R> RT_DEV V3<R> cross(V3<R> a, V3<R> b) { return mk3<R>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

// Math traits: exact for double, fast hardware approximations (<= 1 ulp) for float.
template <typename R> struct Mth;
template <> struct Mth<double> {
    static RT_DEV double div(double a, double b) { return a / b; }
    static RT_DEV double sqrt(double a) { return ::sqrt(a); }
    static RT_DEV double inf() { return __builtin_huge_val(); }
};
template <> struct Mth<float> {
    static RT_DEV float div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
    static RT_DEV float sqrt(float a) { return __builtin_amdgcn_sqrtf(a); }
    static RT_DEV float inf() { return __builtin_huge_valf(); }
};
template <typename R> RT_DEV V3<R> divk(V3<R> a, R k) { return mk3<R>(Mth<R>::div(a.x, k), Mth<R>::div(a.y, k), Mth<R>::div(a.z, k)); }  // vec3.c:19 (true division, F12)
template <typename R> RT_DEV R length(V3<R> a) { return Mth<R>::sqrt(len2(a)); }
template <typename R> RT_DEV V3<R> unit(V3<R> a) { return divk(a, length(a)); }

template <typename R> struct RayT { V3<R> o, d; R time; };

// rotatePoint (hittable.c:37-48): writes two components, keeps the third (axis naming quirk F10)
template <typename R> RT_DEV V3<R> rot(V3<R> p, R s, R c, int axis) {
    V3<R> q = p;
    if (axis == RTC_ROT_X) { q.x = c * p.x - s * p.y; q.y = s * p.x + c * p.y; }
    else if (axis == RTC_ROT_Y) { q.x = c * p.x + s * p.z; q.z = -s * p.x + c * p.z; }
    else { q.y = c * p.y - s * p.z; q.z = s * p.y + c * p.z; }
    return q;
}

// Enter the first `upto` wrappers of a chain, outermost first: translate_hit (:529) moves the origin,
// rotate_hit (:542-545) rotates origin and direction with (-sin, cos); flip_face leaves the ray alone.
template <typename R> RT_DEV RayT<R> chain_enter(const SceneT<R> &sc, int chain, RayT<R> r, int upto = 1 << 30) {
    const ChainRange cr = sc.chains[chain];
    const int n = cr.count < upto ? cr.count : upto;
    for (int i = 0; i < n; ++i) {
        const OpT<R> op = sc.ops[cr.first + i];
        if (op.type == RTC_OP_TRANSLATE) r.o = r.o - mk3<R>(op.a, op.b, op.c);
        else if (op.type == RTC_OP_ROTATE) { r.o = rot(r.o, -op.a, op.b, op.axis); r.d = rot(r.d, -op.a, op.b, op.axis); }
    }
    return r;
}

// Candidate parameter of one leaf for a ray already in the leaf's space: the first root inside [tmin, tmax]
// exactly as line_sphere_intersection (:406-429) / rect_hit (:494-506) accept it.
template <typename R> RT_DEV bool sphere_t(V3<R> center, R rad, const RayT<R> &r, R tmin, R tmax, R &t_out) {
    const V3<R> oc = r.o - center;
    const R a = len2(r.d), half_b = dot(oc, r.d), c = len2(oc) - rad * rad;
    const R disc = half_b * half_b - a * c;
    if (disc < 0) return false;
    const R sq = Mth<R>::sqrt(disc);
    R t = Mth<R>::div(-half_b - sq, a);
    if (t < tmin || t > tmax) { t = Mth<R>::div(-half_b + sq, a); if (t < tmin || t > tmax) return false; }
    t_out = t; return true;
}
// FP32 production form of the same test.  The textbook discriminant half_b^2 - a*c cancels catastrophically in float
// once the radius is large against the feature size (ground sphere r = 1000, fog boundary r = 5000: c = |oc|^2 - r^2
// carries an absolute error of ~r^2 * 2^-23).  Same roots, evaluated from the point of closest approach instead
// (Haines et al., "Precision improvements for ray/sphere intersection", Ray Tracing Gems 2019, ch. 7):
//   tc = -half_b/a,  l = oc + tc*d (perpendicular offset),  roots = tc -+ sqrt((r^2 - |l|^2)/a).
template <> RT_DEV bool sphere_t<float>(V3<float> center, float rad, const RayT<float> &r, float tmin, float tmax, float &t_out) {
    const V3<float> oc = r.o - center;
    const float inv_a = __builtin_amdgcn_rcpf(len2(r.d));
    const float tc = -dot(oc, r.d) * inv_a;
    const V3<float> l = oc + r.d * tc;
    const float q = (rad * rad - len2(l)) * inv_a;
    if (q < 0.0f) return false;
    const float sq = __builtin_amdgcn_sqrtf(q);
    float t = tc - sq;
    if (t < tmin || t > tmax) { t = tc + sq; if (t < tmin || t > tmax) return false; }
    t_out = t; return true;
}
template <typename R> RT_DEV V3<R> msphere_center(const R *d, R time) {  // moving_sphere_center :455-458
    const R k = Mth<R>::div(time - d[6], d[7] - d[6]);
    return mk3<R>((d[3] - d[0]) * k + d[0], (d[4] - d[1]) * k + d[1], (d[5] - d[2]) * k + d[2]);
}
template <typename R> RT_DEV bool rect_t(const R *d, int axis, const RayT<R> &r, R tmin, R tmax, R &t_out) {
    R t;
    if (axis == RTC_AXIS_XY) t = Mth<R>::div(d[6] - r.o.z, r.d.z);
    else if (axis == RTC_AXIS_XZ) t = Mth<R>::div(d[6] - r.o.y, r.d.y);
    else t = Mth<R>::div(d[6] - r.o.x, r.d.x);
    if (t < tmin || t > tmax) return false;   // a NaN t passes, like the reference (F20)
    const R x = r.o.x + t * r.d.x, y = r.o.y + t * r.d.y, z = r.o.z + t * r.d.z;
    if (axis == RTC_AXIS_XZ) { if (x < d[0] || x > d[1] || z < d[4] || z > d[5]) return false; }
    else if (axis == RTC_AXIS_XY) { if (x < d[0] || x > d[1] || y < d[2] || y > d[3]) return false; }
    else { if (z < d[4] || z > d[5] || y < d[2] || y > d[3]) return false; }
    t_out = t; return true;
}
// A ray that STARTS on a sphere (the path tracer's scattered rays): in exact arithmetic the roots are {0, -2*half_b/a};
// tmin rejects the first, so only the far one can be a hit.  In FP32 the "0" root comes out as +-1e-4/cos(theta) for
// r = 1000 -- often past tmin = 0.001 -- and the ray would hit the surface it is leaving.  The FP64 check path never
// passes from_here: the reference's doubles keep that root ~1e-13 and its tmin test does the job.
template <typename R> RT_DEV bool sphere_leave_t(V3<R> center, const RayT<R> &r, R tmin, R tmax, R &t_out) {
    const V3<R> oc = r.o - center;
    const R t = (R)-2 * Mth<R>::div(dot(oc, r.d), len2(r.d));
    if (t < tmin || t > tmax) return false;
    t_out = t; return true;
}
template <typename R> RT_DEV bool leaf_t(const PrimT<R> &p, const RayT<R> &rl, R tmin, R tmax, R &t, bool from_here = false) {
    if (p.type == RTC_PRIM_RECT) return !from_here && rect_t(p.d, p.axis, rl, tmin, tmax, t);   // a ray cannot re-hit the plane it leaves
    if (p.type == RTC_PRIM_SPHERE) {
        const V3<R> c = mk3<R>(p.d[0], p.d[1], p.d[2]);
        return from_here ? sphere_leave_t(c, rl, tmin, tmax, t) : sphere_t(c, p.d[3], rl, tmin, tmax, t);
    }
    if (p.type == RTC_PRIM_MSPHERE) {
        const V3<R> c = msphere_center(p.d, rl.time);
        return from_here ? sphere_leave_t(c, rl, tmin, tmax, t) : sphere_t(c, p.d[8], rl, tmin, tmax, t);
    }
    return false;
}

template <typename R> struct HitT { R t; int prim; };

// hittable_list_hit (hittable_list.c:125-143) over the flattened leaves in visit order: `t > closest` rejects,
// so an equal t on a later leaf replaces the earlier one.  `prim` is wave-uniform, so the table reads are scalar.
// Media are handled by the caller-supplied functor (stochastic in the reference, hittable.c:563-584).
template <typename R, typename MediumFn>
RT_DEV HitT<R> closest_hit_range(const SceneT<R> &sc, const PrimT<R> *prims, int first, int count, const RayT<R> &r, R tmin, R tmax, MediumFn &&medium, int src = -1) {
    HitT<R> h; h.t = tmax; h.prim = -1;
    int cur_chain = 0; RayT<R> rl = r;
    for (int i = first; i < first + count; ++i) {
        const PrimT<R> &p = prims[i];
        R t;
        if (p.type == RTC_PRIM_MEDIUM) { if (medium(i, h.t, t)) { h.t = t; h.prim = i; } continue; }
        if (p.chain != cur_chain) { cur_chain = p.chain; rl = chain_enter(sc, cur_chain, r); }
        if (leaf_t(p, rl, tmin, h.t, t, i == src)) { h.t = t; h.prim = i; }
    }
    return h;
}
// src = the leaf the ray starts on (FP32 path tracer only; -1 = none): see sphere_leave_t.
template <typename R, typename MediumFn>
RT_DEV HitT<R> closest_hit_linear(const SceneT<R> &sc, const RayT<R> &r, R tmin, R tmax, MediumFn &&medium, int src = -1) {
    return closest_hit_range<R>(sc, sc.prims, 0, sc.n_prims, r, tmin, tmax, medium, src);
}

// aabb_hit (aabb.c:33-63) with the reference's mixed precision (F9): the reciprocal is a FLOAT division, the slab
// distances are double x (double)float products; swap on a negative reciprocal; strict tmax <= tmin rejects.
template <typename R> RT_DEV bool aabb_hit_ref(const double *b, const RayT<R> &r, R tmin, R tmax) {
    const R o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float invD = 1.0f / (float)d[a];
        R t0 = ((R)b[a] - o[a]) * (R)invD, t1 = ((R)b[3 + a] - o[a]) * (R)invD;
        if (invD < 0.0f) { const R sw = t0; t0 = t1; t1 = sw; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return false;
    }
    return true;
}
// The reference's own traversal, step by step (RefStep, rtc_tables.h): list scans in insertion order, BVH nodes
// left subtree then right subtree, each node culled by aabb_hit against the closest t found so far
// (bvh_node_hit, hittable.c:485-490).  Bit-exact primitive ids even where two leaves share a t.
template <typename R, typename MediumFn>
RT_DEV HitT<R> closest_hit_ref(const SceneT<R> &sc, const RayT<R> &r, R tmin, R tmax, MediumFn &&medium) {
    HitT<R> h; h.t = tmax; h.prim = -1;
    int i = 0;
    while (i < sc.n_ref_steps) {
        const RefStep s = sc.ref_steps[i];
        if (s.kind == RTC_STEP_NODE) {
            const RayT<R> rl = s.chain ? chain_enter(sc, s.chain, r) : r;
            i += aabb_hit_ref<R>(sc.ref_boxes + 6 * (size_t)s.b, rl, tmin, h.t) ? 1 : s.a;
            continue;
        }
        const HitT<R> g = closest_hit_range<R>(sc, sc.prims, s.a, s.b, r, tmin, h.t, medium);
        if (g.prim >= 0) h = g;
        ++i;
    }
    return h;
}

// ---- FP32 production BVH ------------------------------------------------------------------------------------
// Same closest hit from the flattened BVH (rtc_tables.h BvhNode / BLeaf).  Any visiting order gives the same t; ties are
// broken explicitly -- the leaf with the highest visit-order index (primitive id) wins, which is what the reference's
// left-then-right scan produces.
// Slab test of one child box given as centre c and half-extent h (BvhNodeCH): entry distance in tn.  Per axis m = c * inv + noi
// (noi = -o * inv, bvh_begin) is the parameter of the box centre, the planes lie h |inv| before and behind it -- nine FMAs and no
// per-axis min / max (4 SIMD-cycles each on this part against 2.2 for an FMA).  Boxes are padded outward at build and upload time
// and the extra slack covers the rounding of this test itself; against (lo - o) * inv the planes move by up to |o| * 2^-24, the
// resolution of the FP32 ray origin itself; inv is finite (bvh_begin).
RT_DEV bool slab_f32(const float *c, const float *h, V3<float> noi, V3<float> inv, float tmin, float tmax, float &tn) {
    const float mx = __builtin_fmaf(c[0], inv.x, noi.x), my = __builtin_fmaf(c[1], inv.y, noi.y), mz = __builtin_fmaf(c[2], inv.z, noi.z);
    const float ax = fabsf(inv.x), ay = fabsf(inv.y), az = fabsf(inv.z);                 // (source modifiers: no instructions)
    const float x0 = __builtin_fmaf(-h[0], ax, mx), y0 = __builtin_fmaf(-h[1], ay, my), z0 = __builtin_fmaf(-h[2], az, mz);
    const float x1 = __builtin_fmaf(h[0], ax, mx), y1 = __builtin_fmaf(h[1], ay, my), z1 = __builtin_fmaf(h[2], az, mz);
    tn = fmaxf(fmaxf(x0, y0), fmaxf(z0, tmin));
    const float tf = fminf(fminf(x1, y1), fminf(z1, tmax));
    return !(tn > tf * 1.0000001f + 1e-7f);
}
// Traversal stack: an LDS column per lane (entry e of lane l lives at base[e * stride + l]: consecutive lanes hit
// consecutive banks).  A push beyond the capacity is dropped: kBvhStackCap bounds the depth the builder may produce.
struct LdsStack {
    int *base; int stride, cap; int sp = 0;
    RT_DEV LdsStack(int *b, int s, int c) : base(b), stride(s), cap(c) {}
    RT_DEV void push(int x) { if (sp < cap) { base[sp * stride] = x; ++sp; } }
    RT_DEV int pop() { if (sp == 0) return RTC_BVH_EMPTY; --sp; return base[sp * stride]; }
};
// The same with 16-bit entries, for trees whose node indices and leaf references fit (k_wf_extend_lds: half the LDS per lane).
struct LdsStack16 {
    short *base; int stride, cap; int sp = 0;
    RT_DEV LdsStack16(short *b, int s, int c) : base(b), stride(s), cap(c) {}
    RT_DEV void push(int x) { if (sp < cap) { base[sp * stride] = (short)x; ++sp; } }
    RT_DEV int pop() { if (sp == 0) return RTC_BVH_EMPTY; --sp; return (int)base[sp * stride]; }
};
struct HitS { float t; int prim, slot; };
// One BLeaf against the ray: candidate parameter in [tmin, tmax] as sphere_t / rect_t accept it.  Written without early
// returns: the leaf type is the only branch (usually wave-uniform), everything below it is selects, so the traversal
// loop carries no nested exec masks.
// Dot product with its contraction spelled out.  Under -ffp-contract=fast the compiler may fuse either product of x x' + y y' into the
// FMA, and it does choose differently from one kernel to the next (seen in the listings: y first in one, x first in the other) -- the leaf
// tests below must give the SAME t bit for bit in every kernel that inlines them (k_wf_extend, k_wf_extend_lds, k_wf_primary per lane
// and per wave), so nothing in them is left to that choice: explicit FMAs, and no product that feeds a plain add.
RT_DEV float dot_f(V3<float> a, V3<float> b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
RT_DEV float len2_f(V3<float> a) { return dot_f(a, a); }
RT_DEV V3<float> madd_f(V3<float> a, float k, V3<float> b) { return mk3<float>(__builtin_fmaf(a.x, k, b.x), __builtin_fmaf(a.y, k, b.y), __builtin_fmaf(a.z, k, b.z)); }   // a k + b
RT_DEV bool sphere_roots_f32(V3<float> c, float rad, const RayT<float> &r, float tmin, float tmax, bool from_here, float &t) {
    // closest-approach form of line_sphere_intersection (see sphere_t<float>); a ray that starts on the sphere has the
    // roots {0, 2 tc} (sphere_leave_t)
    const V3<float> oc = r.o - c;
    const float inv_a = __builtin_amdgcn_rcpf(len2_f(r.d));
    const float tc = -dot_f(oc, r.d) * inv_a;
    const V3<float> l = madd_f(r.d, tc, oc);
    const float q = __builtin_fmaf(rad, rad, -len2_f(l)) * inv_a;
    const float sq = __builtin_amdgcn_sqrtf(fmaxf(q, 0.0f));
    const float t0 = from_here ? 2.0f * tc : tc - sq, t1 = tc + sq;
    const bool ok0 = (t0 >= tmin) & (t0 <= tmax), ok1 = (t1 >= tmin) & (t1 <= tmax) & !from_here;
    t = ok0 ? t0 : t1;
    return (from_here | (q >= 0.0f)) & (ok0 | ok1);
}
// KINDS: leaf types the scene's BVH holds (bit 0 rects, 1 spheres, 2 moving spheres, 3 boxes; SceneT::leaf_kinds) -- a kernel for a
// spheres-only scene carries no rect / transform code and fewer registers.
enum { RTC_KIND_RECT = 1, RTC_KIND_SPHERE = 2, RTC_KIND_MSPHERE = 4, RTC_KIND_BOX = 8, RTC_KIND_ALL = 15 };
// Box record (rtc_tables.h): the ray crosses a closed box through an entry and an exit face; the list scan over its six
// two-sided rects (hittable.c:521-523) returns the entry point if it lies in [tmin, tmax] and the exit point otherwise.
// Each t is rect_hit's (k - o_n) / d_n on the face's own axis; `face` = 2 * axis + (hi side).  A ray that starts on a
// face (src_face) cannot hit that face again.
RT_DEV bool box_leaf_t(const BLeaf &lf, const RayT<float> &r, V3<float> inv, float tmin, float tmax, int src_face, float &t, int &face) {
    const float x0 = (lf.d[0] - r.o.x) * inv.x, x1 = (lf.d[3] - r.o.x) * inv.x;
    const float y0 = (lf.d[1] - r.o.y) * inv.y, y1 = (lf.d[4] - r.o.y) * inv.y;
    const float z0 = (lf.d[2] - r.o.z) * inv.z, z1 = (lf.d[5] - r.o.z) * inv.z;
    const float nx = fminf(x0, x1), ny = fminf(y0, y1), nz = fminf(z0, z1);
    const float fx = fmaxf(x0, x1), fy = fmaxf(y0, y1), fz = fmaxf(z0, z1);
    const float t_in = fmaxf(fmaxf(nx, ny), nz), t_out = fminf(fminf(fx, fy), fz);
    const int ax_in = nx == t_in ? 0 : (ny == t_in ? 1 : 2), ax_out = fx == t_out ? 0 : (fy == t_out ? 1 : 2);
    const float inv_in = ax_in == 0 ? inv.x : (ax_in == 1 ? inv.y : inv.z), inv_out = ax_out == 0 ? inv.x : (ax_out == 1 ? inv.y : inv.z);
    const int f_in = 2 * ax_in + (inv_in > 0.0f ? 0 : 1), f_out = 2 * ax_out + (inv_out > 0.0f ? 1 : 0);
    const bool enter = (t_in >= tmin) & (t_in <= tmax) & (f_in != src_face);
    t = enter ? t_in : t_out;
    face = enter ? f_in : f_out;
    return (t_in <= t_out) & (t >= tmin) & (t <= tmax) & (face != src_face);
}
template <int KINDS = RTC_KIND_ALL>
RT_DEV bool bleaf_t(const BLeaf &lf, const RayT<float> &r, float tmin, float tmax, bool from_here, float &t) {
    const uint32_t type = lf.meta & 3u;
    if ((KINDS & RTC_KIND_RECT) && type == RTC_PRIM_RECT) {
        const V3<float> N = mk3<float>(lf.d[0], lf.d[1], lf.d[2]), U = mk3<float>(lf.d[4], lf.d[5], lf.d[6]), V = mk3<float>(lf.d[8], lf.d[9], lf.d[10]);
        const float tt = (lf.d[3] - dot_f(N, r.o)) * __builtin_amdgcn_rcpf(dot_f(N, r.d));
        const V3<float> p = madd_f(r.d, tt, r.o);
        const float pa = dot_f(U, p), pb = dot_f(V, p);
        t = tt;
        // a ray cannot re-hit the plane it leaves; a NaN t passes the range test like in the reference (F20)
        return !from_here & !(tt < tmin) & !(tt > tmax) & !(pa < lf.d[7]) & !(pa > lf.d[12]) & !(pb < lf.d[11]) & !(pb > lf.d[13]);
    }
    if ((KINDS & RTC_KIND_SPHERE) && (type == RTC_PRIM_SPHERE || KINDS == RTC_KIND_SPHERE))
        return sphere_roots_f32(mk3<float>(lf.d[0], lf.d[1], lf.d[2]), lf.d[3], r, tmin, tmax, from_here, t);
    if ((KINDS & RTC_KIND_MSPHERE) && type == RTC_PRIM_MSPHERE)
        return sphere_roots_f32(mk3<float>(__builtin_fmaf(lf.d[3], r.time, lf.d[0]), __builtin_fmaf(lf.d[4], r.time, lf.d[1]), __builtin_fmaf(lf.d[5], r.time, lf.d[2])), lf.d[6], r, tmin, tmax, from_here, t);
    return false;
}
// The traversal loop.  A lane stands on an inner node (cur >= 0) or on a leaf record (cur < 0); the nearer child is entered first,
// the farther one waits on the stack.
struct BvhTrav {
    RayT<float> r; V3<float> inv, noi; HitS h; int cur, src_slot;   // src_slot = leaf slot the ray starts on (-1: none)
};
RT_DEV void bvh_begin(BvhTrav &s, const RayT<float> &r, float tmax, int src_slot, int root) {
    // 1/d clamped to +-1e30: a zero component must not turn lo * inv - o * inv into inf - inf (slab_f32)
    s.r = r; s.inv = mk3<float>(fminf(fmaxf(__builtin_amdgcn_rcpf(r.d.x), -1e30f), 1e30f), fminf(fmaxf(__builtin_amdgcn_rcpf(r.d.y), -1e30f), 1e30f),
                                fminf(fmaxf(__builtin_amdgcn_rcpf(r.d.z), -1e30f), 1e30f));
    s.noi = mk3<float>(-r.o.x * s.inv.x, -r.o.y * s.inv.y, -r.o.z * s.inv.z);
    s.h.t = tmax; s.h.prim = -1; s.h.slot = -1; s.cur = root; s.src_slot = src_slot;
}
template <typename Stack>
RT_DEV void bvh_node_step(const BvhNodeCH &nd, BvhTrav &s, float tmin, Stack &stack) {
    float n0, n1;
    const bool h0 = slab_f32(nd.c0, nd.h0, s.noi, s.inv, tmin, s.h.t, n0), h1 = slab_f32(nd.c1, nd.h1, s.noi, s.inv, tmin, s.h.t, n1);
    const bool both = h0 & h1, first0 = h0 & ((n0 <= n1) | !h1);   // the only hit child, or the nearer of two
    if (both) stack.push(first0 ? nd.child1 : nd.child0);
    s.cur = first0 ? nd.child0 : nd.child1;
    if (!(h0 | h1)) s.cur = stack.pop();
}
// One leaf record (the one of slot `slot`) against the lane's ray: the closest hit so far is updated; returns the slots the record owns.
template <int KINDS = RTC_KIND_ALL, typename MediumFn>
RT_DEV int bvh_leaf_test(const SceneT<float> &sc, const BLeaf &lf, int slot, BvhTrav &s, float tmin, MediumFn &&medium) {
    int hit_slot = slot, width = 1;                              // a box record answers for six slots
    float t; bool hit;
    if ((KINDS & RTC_KIND_SPHERE) && (lf.meta & RTC_LEAF_PAIR)) {
        // A leaf run of two plain spheres packed into ONE record at upload time (rtc_api.cpp upload_tables: d[0..3] the sphere of `slot`, d[4..7] the
        // sphere of slot + 1): one fetch and one trip instead of two.  Tested in slot order with the tie rule below, exactly as two trips would.
        hit = sphere_roots_f32(mk3<float>(lf.d[0], lf.d[1], lf.d[2]), lf.d[3], s.r, tmin, s.h.t, slot == s.src_slot, t);
        if (hit && t == s.h.t && s.h.slot >= 0) hit = sc.leaf_prims[slot] >= sc.leaf_prims[s.h.slot];
        if (hit) { s.h.t = t; s.h.slot = slot; }
        hit = sphere_roots_f32(mk3<float>(lf.d[4], lf.d[5], lf.d[6]), lf.d[7], s.r, tmin, s.h.t, slot + 1 == s.src_slot, t);
        hit_slot = slot + 1; width = 2;
    }
    else if ((lf.meta & 3u) == RTC_PRIM_MEDIUM) {
        if (lf.meta & 0x80u) hit = medium.sphere_bounded(lf.d, s.r, s.h.t, t, (uint32_t)slot);
        else hit = medium(sc.leaf_prims[slot], s.r, s.h.t, t, (uint32_t)slot);
    }
    else if ((KINDS & RTC_KIND_BOX) && (lf.meta & RTC_LEAF_BOX)) {
        const unsigned sf = (unsigned)(s.src_slot - slot);
        int face;
        hit = box_leaf_t(lf, s.r, s.inv, tmin, s.h.t, sf < 6u ? (int)sf : -1, t, face);
        hit_slot = slot + face; width = 6;
    }
    else hit = bleaf_t<KINDS>(lf, s.r, tmin, s.h.t, slot == s.src_slot, t);
    // an equal t is the only case that needs the primitive ids (the later leaf in the reference's visit order wins), so
    // the slot -> id table is read there and once at the end, not on every accepted hit
    if (hit && t == s.h.t && s.h.slot >= 0) hit = sc.leaf_prims[hit_slot] >= sc.leaf_prims[s.h.slot];
    if (hit) { s.h.t = t; s.h.slot = hit_slot; }
    return width;
}
template <int KINDS = RTC_KIND_ALL, typename Stack, typename MediumFn>
RT_DEV void bvh_leaf_step(const SceneT<float> &sc, const BLeaf &lf, BvhTrav &s, float tmin, Stack &stack, MediumFn &&medium) {
    const int v = ~s.cur, slot = v >> 3, rem = v & 7;
    const int width = bvh_leaf_test<KINDS>(sc, lf, slot, s, tmin, medium);
    s.cur = rem ? ~(((slot + width) << 3) | (rem - 1)) : stack.pop();
}
// One loop, one step per trip: a lane either tests the two child boxes of an inner node or one leaf record, so lanes of a wave that
// are in different phases of their traversal never wait for each other's inner loops.
// One trip = one memory round trip for the whole wave: every lane requests the 64 bytes it stands on at the top of the
// trip (a node or a leaf, through one address), and only then do the two arms run.  Written as
// `if (node) { load; test } else { load; test }` the compiler waits for each arm's own load inside the arm (and moves
// shared loads into the arms if they are not pinned), so a wave with lanes in both phases pays two dependent round
// trips per trip.
// (Round 3, measured and dropped: TWO steps per trip -- a lane on the last record of a leaf run pops its stack at the top of the trip
// and requests the node that comes off together with the leaf, so the node arm of the same trip takes it one step further; same
// visiting order, bit-identical hits, fewer trips -- and 9-17 % SLOWER on every BVH scene: eight dwordx4 requests per trip through an
// address unit that is already busy two thirds of the time, and 16 more registers.  tools/patches/bvh_two_steps_per_trip.patch.)
// node_tab / leaf_tab: the node and leaf tables as float4[4] records -- sc.nodes / sc.bleaves themselves, or a workgroup's copy of them
// in LDS (k_wf_extend_lds; the compiler sees the address space after inlining).
template <int KINDS = RTC_KIND_ALL, typename Stack, typename MediumFn>
RT_DEV void bvh_trip_tab(const SceneT<float> &sc, const float4 *node_tab, const float4 *leaf_tab, BvhTrav &s, float tmin, Stack &stack, MediumFn &&medium) {
    const bool node = s.cur >= 0;
    const float4 *ptr = node ? node_tab + 4 * s.cur : leaf_tab + 4 * ((~s.cur) >> 3);
    float4 q0 = ptr[0], q1 = ptr[1], q2 = ptr[2], q3 = ptr[3];
    asm volatile("" : "+v"(q0.x), "+v"(q1.x), "+v"(q2.x), "+v"(q3.x));   // pins the four loads here, ahead of both arms
    if (node) {
        BvhNodeCH nd;
        nd.c0[0] = q0.x; nd.c0[1] = q0.y; nd.c0[2] = q0.z; nd.child0 = __float_as_int(q0.w); nd.h0[0] = q1.x; nd.h0[1] = q1.y; nd.h0[2] = q1.z;
        nd.c1[0] = q2.x; nd.c1[1] = q2.y; nd.c1[2] = q2.z; nd.child1 = __float_as_int(q2.w); nd.h1[0] = q3.x; nd.h1[1] = q3.y; nd.h1[2] = q3.z;
        bvh_node_step(nd, s, tmin, stack);
    } else {
        BLeaf lf;
        lf.d[0] = q0.x; lf.d[1] = q0.y; lf.d[2] = q0.z; lf.d[3] = q0.w; lf.d[4] = q1.x; lf.d[5] = q1.y; lf.d[6] = q1.z; lf.d[7] = q1.w;
        lf.d[8] = q2.x; lf.d[9] = q2.y; lf.d[10] = q2.z; lf.d[11] = q2.w; lf.d[12] = q3.x; lf.d[13] = q3.y; lf.d[14] = q3.z; lf.meta = __float_as_uint(q3.w);
        bvh_leaf_step<KINDS>(sc, lf, s, tmin, stack, medium);
    }
}
template <int KINDS = RTC_KIND_ALL, typename Stack, typename MediumFn>
RT_DEV HitS bvh_closest_hit_tab(const SceneT<float> &sc, const float4 *node_tab, const float4 *leaf_tab, const RayT<float> &r, float tmin, float tmax,
                                Stack &stack, MediumFn &&medium, int src_slot = -1, bool skip = false) {
    BvhTrav s; bvh_begin(s, r, tmax, src_slot, skip ? RTC_BVH_EMPTY : sc.bvh_root);   // skip: this lane has nothing to trace (its hit is known)
    // the medium every ray meets (SceneT::pre_slot) is tested ahead of the loop: one trip and one node step fewer per ray, and the closest hit is the
    // same whatever the order (a medium's free-flight draw is keyed by ray and medium, bvh_leaf_test)
    if (sc.pre_slot >= 0 && !skip) { const BLeaf lf = sc.bleaves[sc.pre_slot]; bvh_leaf_test<KINDS>(sc, lf, sc.pre_slot, s, tmin, medium); }
    while (s.cur != RTC_BVH_EMPTY) bvh_trip_tab<KINDS>(sc, node_tab, leaf_tab, s, tmin, stack, medium);
    s.h.prim = s.h.slot >= 0 ? sc.leaf_prims[s.h.slot] : -1;
    return s.h;
}
template <int KINDS = RTC_KIND_ALL, typename Stack, typename MediumFn>
RT_DEV HitS bvh_closest_hit(const SceneT<float> &sc, const RayT<float> &r, float tmin, float tmax, Stack &stack, MediumFn &&medium, int src_slot = -1, bool skip = false) {
    return bvh_closest_hit_tab<KINDS>(sc, reinterpret_cast<const float4 *>(sc.nodes), reinterpret_cast<const float4 *>(sc.bleaves), r, tmin, tmax, stack, medium, src_slot, skip);
}

// The same closest hit with ONE traversal for the whole wave (camera rays: the 64 rays of a wave start at the eye and leave through a 4x4-pixel
// window, k_wf_primary): the current node is wave-uniform, so a node or a leaf record is ONE scalar load (s_load_dwordx16) instead of 64
// lanes fetching 64 bytes each, a trip runs the node arm OR one leaf kind (scalar branches: no arm runs for lanes that are elsewhere), and
// the stack is one column of LDS per wave.  A child is entered when ANY lane's slab test passes (each lane tests against its own closest hit
// so far), the child more lanes meet first goes first, and every lane tests every leaf record on the way: each lane ends with ITS closest
// hit whatever the order (equal t: the later primitive wins, bvh_leaf_test; media draw by medium, not by visiting order) -- bit-identical
// to bvh_closest_hit.  `on` = this lane has a ray (the others vote for nothing and test nothing).  wstack: sc.bvh_stack ints of LDS per wave.
template <int KINDS = RTC_KIND_ALL, typename MediumFn>
RT_DEV HitS bvh_closest_hit_wave(const SceneT<float> &sc, const RayT<float> &r, bool on, float tmin, float tmax, int *wstack, MediumFn &&medium) {
    BvhTrav s; bvh_begin(s, r, tmax, -1, sc.bvh_root);
    if (sc.pre_slot >= 0) { const BLeaf lf = sc.bleaves[sc.pre_slot]; if (on) bvh_leaf_test<KINDS>(sc, lf, sc.pre_slot, s, tmin, medium); }   // (bvh_closest_hit_tab)
    int cur = sc.bvh_root, sp = 0;                                   // wave-uniform
    while (cur != RTC_BVH_EMPTY) {
        if (cur >= 0) {
            const BvhNodeCH nd = sc.nodes[cur];
            float n0, n1;
            const bool h0 = on & slab_f32(nd.c0, nd.h0, s.noi, s.inv, tmin, s.h.t, n0), h1 = on & slab_f32(nd.c1, nd.h1, s.noi, s.inv, tmin, s.h.t, n1);
            const unsigned long long b0 = __builtin_amdgcn_ballot_w64(h0), b1 = __builtin_amdgcn_ballot_w64(h1);
            if (b0 != 0 && b1 != 0) {
                const unsigned long long near0 = __builtin_amdgcn_ballot_w64(h0 & (!h1 | (n0 <= n1)));   // lanes that would enter child 0 first
                const bool first0 = 2 * __builtin_popcountll(near0) >= __builtin_popcountll(b0 | b1);
                wstack[sp] = first0 ? nd.child1 : nd.child0; ++sp;
                cur = first0 ? nd.child0 : nd.child1;
            } else if (b0 != 0) cur = nd.child0;
            else if (b1 != 0) cur = nd.child1;
            else if (sp > 0) { --sp; cur = __builtin_amdgcn_readfirstlane(wstack[sp]); }
            else cur = RTC_BVH_EMPTY;
        } else {
            const int v = ~cur;
            int slot = v >> 3;
            for (int rem = v & 7;; --rem) {
                const BLeaf lf = sc.bleaves[slot];
                int width = 1;
                if (on) width = bvh_leaf_test<KINDS>(sc, lf, slot, s, tmin, medium);
                width = (KINDS & RTC_KIND_BOX) && (lf.meta & RTC_LEAF_BOX) && (lf.meta & 3u) != RTC_PRIM_MEDIUM ? 6 : 1;   // (wave-uniform: from the record, not from a lane)
                if (rem == 0) break;
                slot += width;
            }
            if (sp > 0) { --sp; cur = __builtin_amdgcn_readfirstlane(wstack[sp]); }
            else cur = RTC_BVH_EMPTY;
        }
    }
    s.h.prim = -1;
    return s.h;
}

// Full hit record of the winning leaf: sphere_hit (:440-445) / rect_hit (:509-515), then the wrappers inside-out:
// rotate_hit (:552-557) turns p and normal back with (+sin, cos) and re-faces the normal against the OBJECT-space
// direction (F11), translate_hit (:533-534) shifts p and re-faces against its own ray, flip_face (:589) toggles
// front_face only.
template <typename R> struct RecT { V3<R> p, n; R u, v; int front, mat; };
template <typename R> RT_DEV void face(RecT<R> &rec, V3<R> dir, V3<R> outward) {
    rec.front = dot(dir, outward) < 0;
    rec.n = rec.front ? outward : -outward;
}
template <typename R> RT_DEV void sphere_uv(V3<R> n, R &u, R &v);
template <> RT_DEV void sphere_uv<double>(V3<double> n, double &u, double &v) {   // :381-386
    const double PI_ = 3.1415926535897932385;
    const double theta = ::acos(-n.y), phi = ::atan2(-n.z, n.x) + PI_;
    u = phi / (2 * PI_); v = theta / PI_;
}
template <> RT_DEV void sphere_uv<float>(V3<float> n, float &u, float &v) {
    const float PI_ = 3.14159265358979f;
    const float theta = acosf(fminf(fmaxf(-n.y, -1.f), 1.f)), phi = atan2f(-n.z, n.x) + PI_;
    u = phi * (0.5f / PI_); v = theta * (1.f / PI_);
}
template <typename R> RT_DEV RecT<R> resolve_hit(const SceneT<R> &sc, const RayT<R> &r, const PrimT<R> &p, R t, bool want_uv) {
    RecT<R> rec; rec.mat = p.mat; rec.u = 0; rec.v = 0;
    const RayT<R> rl = chain_enter(sc, p.chain, r);
    rec.p = rl.o + rl.d * t;                                   // ray_at, ray.c:8-11
    if (p.type == RTC_PRIM_RECT) {
        if (want_uv) {
            const R x = rec.p.x, y = rec.p.y, z = rec.p.z; const R *d = p.d;
            rec.u = p.axis == RTC_AXIS_YZ ? Mth<R>::div(y - d[2], d[3] - d[2]) : Mth<R>::div(x - d[0], d[1] - d[0]);
            rec.v = p.axis == RTC_AXIS_XY ? Mth<R>::div(y - d[2], d[3] - d[2]) : Mth<R>::div(z - d[4], d[5] - d[4]);
        }
        face(rec, rl.d, mk3<R>(p.axis == RTC_AXIS_YZ ? 1 : 0, p.axis == RTC_AXIS_XZ ? 1 : 0, p.axis == RTC_AXIS_XY ? 1 : 0));
    } else if (p.type == RTC_PRIM_MEDIUM) {                    // constant_medium_hit :578-582
        rec.n = mk3<R>(1, 0, 0); rec.front = 1;
    } else {
        const V3<R> c = p.type == RTC_PRIM_SPHERE ? mk3<R>(p.d[0], p.d[1], p.d[2]) : msphere_center(p.d, rl.time);
        const R rad = p.type == RTC_PRIM_SPHERE ? p.d[3] : p.d[8];
        const V3<R> outward = divk(rec.p - c, rad);
        face(rec, rl.d, outward);
        if (want_uv) sphere_uv<R>(outward, rec.u, rec.v);
    }
    const ChainRange cr = sc.chains[p.chain];
    for (int i = cr.count - 1; i >= 0; --i) {
        const OpT<R> op = sc.ops[cr.first + i];
        if (op.type == RTC_OP_FLIP) { rec.front = !rec.front; continue; }
        if (op.type == RTC_OP_ROTATE) {
            const RayT<R> ri = chain_enter(sc, p.chain, r, i + 1);   // the rotated_ray of this wrapper, bit for bit
            rec.p = rot(rec.p, op.a, op.b, op.axis);
            face(rec, ri.d, rot(rec.n, op.a, op.b, op.axis));
        } else {
            const RayT<R> ri = chain_enter(sc, p.chain, r, i);       // direction seen by this translate
            rec.p = rec.p + mk3<R>(op.a, op.b, op.c);
            face(rec, ri.d, rec.n);
        }
    }
    return rec;
}

// camera_get_ray (camera.c:46-53) for a supplied unit-disk sample and time.
template <typename R> RT_DEV RayT<R> camera_ray(const CameraT<R> &c, R s, R t, R disk_x, R disk_y, R time) {
    auto L = [](const R *a) { return mk3<R>(a[0], a[1], a[2]); };
    const V3<R> rd = mk3<R>(disk_x, disk_y, (R)0) * c.lens_radius;
    const V3<R> offset = L(c.u) * rd.x + L(c.v) * rd.y;
    V3<R> dir = ((L(c.llc) + L(c.horizontal) * s) + L(c.vertical) * t) - L(c.origin);
    dir = dir - offset;
    RayT<R> r; r.o = L(c.origin) + offset; r.d = dir; r.time = time;
    return r;
}
