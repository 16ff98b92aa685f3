// rt_shade.hpp -- shading half of the hot path, FP32, gfx950 (product code, device only).
//
// Switch-dispatched restatement of the reference's material / pdf / onb / texture maths
// (material.c, pdf.c, onb.c, texture.c, perlin.c under /root/reference/03_FullTracer and
// 02_AdvancedTracer/src/material.c for integrator 1/2).  rand() is replaced by Philox4x32-7 keyed on
// (pixel, sample, bounce); rejection loops (vec3.c:66-72,87-93) are replaced by analytic samplers with the same
// distribution (uniform ball / uniform disk), which is what image-level parity needs.
#pragma once
#include "rt_core.hpp"

typedef V3<float> f3;
#define RT_PI 3.14159265358979323846f
#define RT_INV_PI 0.31830988618379067154f

// ---- Philox4x32 (Salmon et al., SC'11), 7 rounds --------------------------------------------------
// Seven rounds is the count the paper reports as Crush-resistant for Philox4x32 (ten is its default, with a safety margin).
// The generator is a fifth of the small-scene kernel: a round is two v_mad_u64_u32 (quarter rate, 7.4 cycles each,
// tools/ubench/valu_rates) and four xors, and a wave runs ~1.8 blocks per bounce iteration (one per diffuse bounce, one in
// the glass arm, the camera's) -- ten rounds cost 5.8 % of the headline render (22.2 -> 20.9 ms).
struct U4 { uint32_t x, y, z, w; };
#ifndef RT_PHILOX_ROUNDS
#define RT_PHILOX_ROUNDS 7
#endif
RT_DEV U4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < RT_PHILOX_ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    U4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3; return o;
}
RT_DEV float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }   // [0,1), 24 bits
struct Rng4 { float x, y, z, w; };
// stream tags (counter word 2): bounce b -> b ; camera -> 0xFFFFFFFF ; shutter time -> 0xFFFFFFFE ; media -> 0x80000000|...
#define RT_TAG_CAMERA 0xFFFFFFFFu
#define RT_TAG_TIME 0xFFFFFFFEu
#define RT_TAG_MEDIA 0x80000000u
// Counter = (pixel, sample, tag, seed low word), key = ("RTC1", seed high word).  The seed's low word rides in the COUNTER so
// that key word 0 is a compile-time constant: its seven round keys become literals instead of seven more SGPRs held through
// the bounce loop (the small-scene kernel spilled 11 SGPRs to lanes with the seed in the key, 2 without: -2 %, r02a).
RT_DEV Rng4 rng_block(uint32_t pixel, uint32_t sample, uint32_t tag, uint32_t seed_lo, uint32_t seed_hi) {
    const U4 b = philox4x32(pixel, sample, tag, seed_lo, 0x52544331u /* "RTC1" */, seed_hi);
    Rng4 r; r.x = u01(b.x); r.y = u01(b.y); r.z = u01(b.z); r.w = u01(b.w); return r;
}

// ---- small maths ---------------------------------------------------------------------------------
RT_DEV f3 f3mk(float x, float y, float z) { return mk3<float>(x, y, z); }
RT_DEV f3 normalize(f3 a) { const float k = __builtin_amdgcn_rsqf(len2(a)); return a * k; }
RT_DEV f3 reflect(f3 v, f3 n) { const float a = dot(v, n) * 2.0f; return v - n * a; }                    // vec3.c:26
RT_DEV f3 refract(f3 uv, f3 n, float eta) {                                                              // vec3.c:44-50
    const float cos_theta = fminf(dot(-uv, n), 1.0f);
    const f3 perp = (uv + n * cos_theta) * eta;
    const float k = -__builtin_amdgcn_sqrtf(fabsf(1.0f - len2(perp)));
    return perp + n * k;
}
// v_sin_f32 / v_cos_f32 take revolutions: sin(2*pi*x) for x in [0,1) is one instruction
RT_DEV float sin2pi(float x) { return __builtin_amdgcn_sinf(x); }
RT_DEV float cos2pi(float x) { return __builtin_amdgcn_cosf(x); }
RT_DEV f3 uniform_sphere(float a, float b) {             // same law as vec3_random_unit_vector_in_unit_sphere
    const float z = 1.0f - 2.0f * a, s = __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - z * z));
    return f3mk(s * cos2pi(b), s * sin2pi(b), z);
}
// radius of a uniform point in the unit ball: c^(1/3) as exp2(log2(c) / 3) -- v_log, v_mul, v_exp (c = 0 -> 0) instead of libm's
// cbrtf (~40 instructions); the 1-2 ulp of the hardware log / exp are far below what a sampled radius needs
RT_DEV float ball_radius(float c) { return __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(c) * (1.0f / 3.0f)); }
RT_DEV f3 uniform_ball(float a, float b, float c) {      // same law as vec3_random_in_unit_sphere (vec3.c:66-72)
    return uniform_sphere(a, b) * ball_radius(c);
}

// pdf_cosine_value (pdf.c:100-103) == material_lambertian_scattering_pdf (material.c:205-208) for UNIT n and direction
RT_DEV float lambert_pdf(f3 n, f3 udir) { const float c = dot(n, udir); return c > 0.0f ? c * RT_INV_PI : 0.0f; }

struct Onb { f3 u, v, w; };
RT_DEV Onb onb_from_w(f3 n) {                            // onb.c:7-12
    Onb o; o.w = normalize(n);
    const f3 a = fabsf(o.w.x) > 0.9f ? f3mk(0, 1, 0) : f3mk(1, 0, 0);
    o.v = normalize(cross(o.w, a));
    o.u = cross(o.w, o.v);
    return o;
}
RT_DEV f3 onb_local(const Onb &o, f3 a) { return o.u * a.x + (o.v * a.y + o.w * a.z); }                 // onb.c:25-30

// ---- textures (texture.c:115-166, perlin.c:33-50,108-148) ------------------------------------------
RT_DEV float perlin_noise(const float *__restrict__ ranvec, const int *__restrict__ perm, f3 p) {
    const float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    const float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const int i = (int)fx, j = (int)fy, k = (int)fz;
    const float uu = u * u * (3 - 2 * u), vv = v * v * (3 - 2 * v), ww = w * w * (3 - 2 * w);
    float accum = 0.0f;
#pragma unroll
    for (int di = 0; di < 2; di++)
#pragma unroll
        for (int dj = 0; dj < 2; dj++)
#pragma unroll
            for (int dk = 0; dk < 2; dk++) {
                const int h = perm[(i + di) & 255] ^ perm[256 + ((j + dj) & 255)] ^ perm[512 + ((k + dk) & 255)];
                const f3 c = f3mk(ranvec[3 * h], ranvec[3 * h + 1], ranvec[3 * h + 2]);
                const f3 wv = f3mk(u - di, v - dj, w - dk);
                accum += (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) * (dk * ww + (1 - dk) * (1 - ww)) * dot(c, wv);
            }
    return accum;
}
RT_DEV float perlin_turb(const float *ranvec, const int *perm, f3 p) {
    float accum = 0.0f, weight = 1.0f;
    for (int i = 0; i < 7; i++) { accum += weight * perlin_noise(ranvec, perm, p); weight *= 0.5f; p = p * 2.0f; }
    return fabsf(accum);
}
// NOISE = false: for scenes without a perlin texture (SceneT::tex_kinds) -- the turbulence loop is what sets the register
// count of the shading kernels
template <bool NOISE = true>
RT_DEV f3 texture_value(const SceneT<float> &sc, int tex, float u, float v, f3 p) {
    TextureT<float> t = sc.textures[tex];
    for (int guard = 0; guard < 8 && t.type == RTC_TEX_CHECKER; ++guard) {           // texture.c:121-124
        const float sines = sinf(10 * p.x) * sinf(10 * p.y) * sinf(10 * p.z);
        t = sc.textures[sines < 0 ? t.a : t.b];
    }
    if (t.type == RTC_TEX_SOLID) return f3mk(t.v[0], t.v[1], t.v[2]);
    if (NOISE && t.type == RTC_TEX_NOISE) {                                           // texture.c:128-132
        const float perl = perlin_turb(sc.perlin_vec + 768 * t.a, sc.perlin_perm + 768 * t.a, p);
        const float val = 0.5f * (1.0f + sinf(t.v[0] * p.z + 10 * perl));
        return f3mk(val, val, val);
    }
    if (t.type == RTC_TEX_IMAGE) {                                                    // texture.c:136-152, nearest texel, v flipped
        const ImageDesc im = sc.images[t.a];
        u = fminf(fmaxf(u, 0.0f), 1.0f); v = 1.0f - fminf(fmaxf(v, 0.0f), 1.0f);
        int i = (int)(u * im.width), j = (int)(v * im.height);
        if (i >= im.width) i = im.width - 1;
        if (j >= im.height) j = im.height - 1;
        const unsigned char *px = sc.image_data + im.offset + ((size_t)j * im.width + i) * 3;
        const float s = 1.0f / 255.0f;
        return f3mk(s * px[0], s * px[1], s * px[2]);
    }
    return f3mk(0, 0, 0);
}
RT_DEV bool material_needs_uv(const SceneT<float> &sc, const MaterialT<float> &m) {
    if (m.tex < 0) return false;
    const int ty = sc.textures[m.tex].type;
    return ty == RTC_TEX_IMAGE || ty == RTC_TEX_CHECKER;   // a checker may select an image texture
}

// ---- constant media in the closest-hit query (hittable.c:563-584) ----------------------------------------------------
// The reference's constant_medium_hit is stochastic (it draws rand() for the free-flight distance), so the traversal calls
// back into this functor for a medium leaf.  The free-flight draw of medium `id` (its BVH leaf slot, or its primitive index in the list
// scan) comes from the Philox block (pixel, sample, MEDIA | bounce | id): a function of the ray and the medium alone, NOT of the order in
// which a traversal meets the media of a scene -- so two traversals that visit the tree in different orders (per-lane stacks in extend,
// one wave-wide path for camera rays) find the same hit bit for bit.
struct NoMedium32 {
    __device__ bool operator()(int, float, float &) const { return false; }
    __device__ bool operator()(int, const RayT<float> &, float, float &, uint32_t) const { return false; }
    __device__ bool sphere_bounded(const float *, const RayT<float> &, float, float &, uint32_t) const { return false; }
};
// GENERAL = false: only sphere-bounded media (their BLeaf carries the boundary); no boundary-list queries.
template <bool GENERAL> struct MediumSamplerT {
    const SceneT<float> *sc; const RayT<float> *ray; float tmin; uint32_t pixel, sample, bounce, k0, k1;
    __device__ float draw(uint32_t id) const { return rng_block(pixel, sample, RT_TAG_MEDIA | ((bounce & 0x7fffu) << 16) | (id & 0xffffu), k0, k1).x; }
    // hittable.c:568-584 once the two boundary parameters are known; ray_length = |direction| in the medium's space
    __device__ bool free_flight(float t1, float t2, float tmax, float ray_length, float neg_inv_density, float &t_out, uint32_t id) const {
        if (t1 < tmin) t1 = tmin;
        if (t2 > tmax) t2 = tmax;
        if (t1 >= t2) return false;
        if (t1 < 0) t1 = 0;
        const float inside = (t2 - t1) * ray_length;
        const float hit_distance = neg_inv_density * __logf(draw(id));
        if (hit_distance > inside) return false;
        t_out = __builtin_fmaf(hit_distance, __builtin_amdgcn_rcpf(ray_length), t1);   // (explicit: k_wf_primary and k_wf_extend must agree bit for bit)
        return true;
    }
    __device__ bool operator()(int prim, float tmax, float &t_out) const { return (*this)(prim, *ray, tmax, t_out, (uint32_t)prim); }
    __device__ bool operator()(int prim, const RayT<float> &r, float tmax, float &t_out, uint32_t id) const {
        if (!GENERAL) return false;
        const PrimT<float> &p = sc->prims[prim];
        NoMedium32 nv;
        const float inf = Mth<float>::inf();
        const HitT<float> a = closest_hit_range<float>(*sc, sc->bprims, p.bnd_first, p.bnd_count, r, -inf, inf, nv);
        if (a.prim < 0) return false;
        // rec1.t + 0.0001 (hittable.c:566): in float the 1e-4 is absorbed once |t| > ~800 and the second query would
        // return the first root again; keep the step at least 4 ulp so it stays a step
        const HitT<float> b = closest_hit_range<float>(*sc, sc->bprims, p.bnd_first, p.bnd_count, r, a.t + fmaxf(0.0001f, fabsf(a.t) * 4.8e-7f), inf, nv);
        if (b.prim < 0) return false;
        const RayT<float> rm = chain_enter(*sc, p.chain, r);
        return free_flight(a.t, b.t, tmax, length(rm.d), p.d[0], t_out, id);
    }
    // boundary = one sphere, d = centre, radius, neg_inv_density in world space (rtc_tables.h BLeaf): the first boundary
    // query over (-inf, inf) returns the near root, the second one over (near + 0.0001, inf) the far root (sphere_t)
    __device__ bool sphere_bounded(const float *d, const RayT<float> &r, float tmax, float &t_out, uint32_t id) const {
        const f3 oc = r.o - f3mk(d[0], d[1], d[2]);
        const float a = len2_f(r.d), inv_a = __builtin_amdgcn_rcpf(a);       // (explicit FMAs: rt_core.hpp dot_f)
        const float tc = -dot_f(oc, r.d) * inv_a;
        const f3 l = madd_f(r.d, tc, oc);
        const float q = __builtin_fmaf(d[3], d[3], -len2_f(l)) * inv_a;
        if (q < 0.0f) return false;
        const float sq = __builtin_amdgcn_sqrtf(q);
        const float t1 = tc - sq, t2 = tc + sq;
        if (t2 < t1 + fmaxf(0.0001f, fabsf(t1) * 4.8e-7f)) return false;
        return free_flight(t1, t2, tmax, __builtin_amdgcn_sqrtf(a), d[4], t_out, id);
    }
};
using MediumSampler = MediumSamplerT<true>;

// ---- light list pdf / sampling (hittable_list.c:165-180, hittable.c:760-864) -------------------------
// The light record is wave-uniform (the callers loop over the list), so kind and axis are scalar branches; below them
// there are no early returns -- the hit tests of rect_t / sphere_t are folded into one select each.
RT_DEV float light_pdf_one(const LightT<float> &l, f3 o, f3 v) {
    if (l.kind == RTC_LIGHT_RECT) {                                                   // :772-788, tmin 1e-4
        const bool xy = l.axis == RTC_AXIS_XY, xz = l.axis == RTC_AXIS_XZ, yz = l.axis == RTC_AXIS_YZ;
        const float on = xy ? o.z : (xz ? o.y : o.x), vn = xy ? v.z : (xz ? v.y : v.x);
        const float oa = yz ? o.y : o.x, va = yz ? v.y : v.x, ob = xy ? o.y : o.z, vb = xy ? v.y : v.z;
        const float a0 = yz ? l.d[2] : l.d[0], a1 = yz ? l.d[3] : l.d[1], b0 = xy ? l.d[2] : l.d[4], b1 = xy ? l.d[3] : l.d[5];
        const float t = (l.d[6] - on) * __builtin_amdgcn_rcpf(vn);                    // rect_hit :494-517
        const float pa = oa + t * va, pb = ob + t * vb;
        const bool hit = !(t < 0.0001f) & !(pa < a0) & !(pa > a1) & !(pb < b0) & !(pb > b1);   // a NaN t passes (F20)
        const float vv = len2(v);
        const float distance_squared = t * t * vv;
        const float area = (a1 - a0) * (b1 - b0);
        const float cosine = fabsf(vn) * __builtin_amdgcn_rsqf(vv);
        return hit ? distance_squared * __builtin_amdgcn_rcpf(cosine * area) : 0.0f;
    }
    if (l.kind == RTC_LIGHT_SPHERE) {                                                 // :760-768, tmin 1e-3
        const f3 c = f3mk(l.d[0], l.d[1], l.d[2]);
        const f3 oc = o - c;
        const float inv_a = __builtin_amdgcn_rcpf(len2(v));                           // closest-approach form, rt_core.hpp sphere_t<float>
        const float tc = -dot(oc, v) * inv_a;
        const f3 lv = oc + v * tc;
        const float q = (l.d[3] * l.d[3] - len2(lv)) * inv_a;
        const float sq = __builtin_amdgcn_sqrtf(fmaxf(q, 0.0f));
        const bool hit = !(q < 0.0f) & (!(tc - sq < 0.001f) | !(tc + sq < 0.001f));    // NaN passes as in sphere_t
        const float cos_theta_max = __builtin_amdgcn_sqrtf(1.0f - l.d[3] * l.d[3] * __builtin_amdgcn_rcpf(len2(oc)));
        return hit ? __builtin_amdgcn_rcpf(2.0f * RT_PI * (1.0f - cos_theta_max)) : 0.0f;
    }
    return 0.0f;
}
RT_DEV float light_list_pdf(const SceneT<float> &sc, f3 o, f3 v) {
    const float weight = 1.0f / (float)sc.n_lights;
    float sum = 0.0f;
    for (int i = 0; i < sc.n_lights; ++i) sum += weight * light_pdf_one(sc.lights[i], o, v);
    return sum;
}
RT_DEV f3 light_random_one(const LightT<float> &l, f3 o, float r1, float r2) {
    if (l.kind == RTC_LIGHT_RECT) {                                                   // :838-848
        const float a0 = l.axis == RTC_AXIS_YZ ? l.d[2] : l.d[0], a1 = l.axis == RTC_AXIS_YZ ? l.d[3] : l.d[1];
        const float b0 = l.axis == RTC_AXIS_XY ? l.d[2] : l.d[4], b1 = l.axis == RTC_AXIS_XY ? l.d[3] : l.d[5];
        const float a = a0 + (a1 - a0) * r1, b = b0 + (b1 - b0) * r2;
        const f3 p = l.axis == RTC_AXIS_XZ ? f3mk(a, l.d[6], b) : (l.axis == RTC_AXIS_XY ? f3mk(a, b, l.d[6]) : f3mk(l.d[6], a, b));
        return p - o;
    }
    if (l.kind == RTC_LIGHT_SPHERE) {                                                 // :821-835
        const f3 direction = f3mk(l.d[0], l.d[1], l.d[2]) - o;
        const float distance_squared = len2(direction);
        const Onb uvw = onb_from_w(direction);
        const float z = 1.0f + r2 * (__builtin_amdgcn_sqrtf(1.0f - l.d[3] * l.d[3] * __builtin_amdgcn_rcpf(distance_squared)) - 1.0f);
        const float s = __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - z * z));
        return onb_local(uvw, f3mk(cos2pi(r1) * s, sin2pi(r1) * s, z));
    }
    return f3mk(1, 0, 0);                                                             // :861-863
}

// ---- one bounce -----------------------------------------------------------------------------------
struct PathState {
    RayT<float> ray;
    f3 T;        // throughput: product of the per-level weights of ray.c:80-83
    f3 L;        // radiance collected by this sample
};
enum { RT_CONTINUE = 0, RT_TERMINATE = 1 };

RT_DEV float schlick(float cosine, float ref_idx) {                                   // material.c:142
    float r0 = (1.0f - ref_idx) * __builtin_amdgcn_rcpf(1.0f + ref_idx); r0 = r0 * r0;
    const float m = 1.0f - cosine, m2 = m * m;
    return r0 + (1.0f - r0) * (m2 * m2 * m);
}
RT_DEV f3 dielectric_dir(float ir, f3 dir, const RecT<float> &rec, float xi) {          // material.c:144-157
    // Both outcomes are a ud + b n -- reflect (vec3.c:26): (1, 2 c) with c = -ud.n; refract (vec3.c:44-50): (eta, eta cos + k),
    // k = -sqrt|1 - |perp|^2|, |perp|^2 = eta^2 |ud + n cos|^2 = eta^2 (1 - 2 cos c + cos^2) for unit ud and n -- so the choice is two
    // selects instead of a divergent branch around two vector expressions (both sides ran for a mixed wave anyway, and the branch's
    // mask bookkeeping is scalar instructions that take issue slots like vector ones).
    const float ratio = rec.front ? __builtin_amdgcn_rcpf(ir) : ir;
    const f3 ud = normalize(dir);
    const float c = -dot(ud, rec.n), cos_theta = fminf(c, 1.0f);
    const float sin_theta = __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));
    const bool refl = (ratio * sin_theta > 1.0f) | (schlick(cos_theta, ratio) > xi);
    const float perp2 = (ratio * ratio) * (1.0f - 2.0f * cos_theta * c + cos_theta * cos_theta);
    const float k = -__builtin_amdgcn_sqrtf(fabsf(1.0f - perp2));
    const float a = refl ? 1.0f : ratio, b = refl ? 2.0f * c : ratio * cos_theta + k;
    return ud * a + rec.n * b;
}

// Shade one hit.  integrator 0: ray.c:58-83 in loop form (L += T*emitted; T *= att*scattering_pdf/pdf_val).
// integrator 1/2: 02 main.c:54-62 / 01 main.c:45-56 (T *= attenuation).  xi = this bounce's Philox block.
// `m` = the material record; resolved = its texture is the constant m.albedo (a leaf_shade record, rtc_tables.h): no table is
// read at all then.
// One exit: the material arms are separate ifs (they exclude each other) that leave the next direction, the throughput factor and an
// integer status in registers -- with early returns out of an else-if chain the compiler keeps "which lanes have returned" in scalar
// masks and spends three scalar instructions on every merge (the small-scene kernel lost 5 % of its time to that).
// After RT_TERMINATE the callers read ps.L only.
template <bool NOISE = true>
RT_DEV int shade_hit_mat(const SceneT<float> &sc, PathState &ps, const RecT<float> &rec, const MaterialT<float> &m, bool resolved, int integrator,
                         float light_weight, const Rng4 &xi) {
    auto tex = [&]() { return resolved ? f3mk(m.albedo[0], m.albedo[1], m.albedo[2]) : texture_value<NOISE>(sc, m.tex, rec.u, rec.v, rec.p); };
    const f3 d = ps.ray.d;
    int status = RT_CONTINUE;
    f3 nd = d, tm = f3mk(1, 1, 1);                                                      // next direction, throughput factor
    // lambertian (02 / 01), fuzzy metal and the isotropic phase function all start from ONE uniform point on the unit sphere: drawn
    // here, ahead of the arms, so that a wave whose lanes stand on different materials runs the sqrt / sin / cos sequence once
    const bool ball = (m.type == RTC_MAT_METAL && m.param > 0.0f) || m.type == RTC_MAT_ISOTROPIC;
    f3 us = f3mk(0, 0, 0);
    if (ball || (m.type == RTC_MAT_LAMBERTIAN && integrator != 0)) us = uniform_sphere(xi.x, xi.y);
    if (ball) us = us * ball_radius(xi.z);
    if (m.type == RTC_MAT_LIGHT) {                                                     // material.c:229-235 (03: front face only, F15)
        if (integrator != 2 && (integrator == 1 || rec.front)) ps.L = ps.L + ps.T * tex();
        status = RT_TERMINATE;                                                         // material_light_scatter returns 0
    }
    if (m.type == RTC_MAT_METAL) {                                                     // material.c:130-138 / 02 :132-141
        const f3 reflected = reflect(normalize(d), rec.n);
        nd = m.param > 0.0f ? reflected + us * m.param : reflected;
        if (integrator != 0 && !(dot(nd, rec.n) > 0.0f)) status = RT_TERMINATE;        // 02/01: absorbed
        tm = f3mk(m.albedo[0], m.albedo[1], m.albedo[2]);
    }
    if (m.type == RTC_MAT_DIELECTRIC) nd = dielectric_dir(m.param, d, rec, xi.x);      // attenuation (1,1,1)
    if (m.type == RTC_MAT_ISOTROPIC) {                                                 // 02 material.c:173-177; 03's is UB (F14) -> absorb
        if (integrator == 0) status = RT_TERMINATE;
        nd = us;
        tm = tex();
    }
    if (m.type == RTC_MAT_LAMBERTIAN) {
        const f3 att = tex();
        if (integrator != 0) {                                                         // 02 material.c:120-128
            f3 dir = rec.n + us;
            if (fabsf(dir.x) < 1e-8f && fabsf(dir.y) < 1e-8f && fabsf(dir.z) < 1e-8f) dir = rec.n;
            nd = dir; tm = att;
        } else {
            // 03: mixture of the light-list pdf and the cosine pdf (ray.c:66-83, pdf.c:119-121,175-181)
            const Onb uvw = onb_from_w(rec.n);
            f3 dir;
            if (sc.n_lights > 0 && xi.x < light_weight) {
                int pick = (int)((float)sc.n_lights * xi.y);                           // random_int(0,n-1), utils.h:26
                pick = pick < sc.n_lights ? pick : sc.n_lights - 1;
                dir = light_random_one(sc.lights[pick], rec.p, xi.z, xi.w);
            } else {                                                                   // random_cosine_direction, vec3.c:97-107
                const float r2 = xi.w, sr = __builtin_amdgcn_sqrtf(r2);
                dir = onb_local(uvw, f3mk(cos2pi(xi.z) * sr, sin2pi(xi.z) * sr, __builtin_amdgcn_sqrtf(1.0f - r2)));
            }
            const f3 udir = normalize(dir);
            const float pdf_cos = lambert_pdf(uvw.w, udir);                             // pdf.c:100-103
            const float pdf_light = sc.n_lights > 0 ? light_list_pdf(sc, rec.p, dir) : 0.0f;
            const float pdf_val = light_weight * pdf_light + (1.0f - light_weight) * pdf_cos;   // pdf.c:119-121
            const float spdf = lambert_pdf(rec.n, udir);                                // material.c:205-208
            // spdf == 0: the reference keeps recursing but multiplies the child by 0; pdf_val == 0: 0/0 = NaN in the
            // reference (scrubbed to a black sample).  Either way nothing more can reach the pixel.
            if (!(spdf > 0.0f) | !(pdf_val > 0.0f)) status = RT_TERMINATE;
            tm = att * (spdf * __builtin_amdgcn_rcpf(pdf_val));
            nd = dir;
        }
    }
    ps.T = ps.T * tm;
    ps.ray.o = rec.p; ps.ray.d = nd;
    return status;
}

template <bool NOISE = true>
RT_DEV int shade_hit(const SceneT<float> &sc, PathState &ps, const RecT<float> &rec, int integrator, float light_weight, const Rng4 &xi) {
    const MaterialT<float> m = sc.materials[rec.mat];
    return shade_hit_mat<NOISE>(sc, ps, rec, m, false, integrator, light_weight, xi);
}

// main.c:150-152 + camera.c:46-53 for sample s of a pixel: jitter from words 0, 1 of the camera block; a thin lens takes words 2, 3
// for its disk sample and the shutter time a block of its own; a pinhole camera (every reference scene) leaves words 2, 3 unused, so
// the shutter time is word 2 and a primary ray costs ONE Philox block.  Shared by the general megakernel and the wavefront kernels.
// Every multiply-add below is an explicit fma and no product feeds a plain add: the compiler has no contraction left to decide, so every
// kernel that inlines this function (init, the camera-ray pass, the tails that start a slot's next sample, the megakernel) makes the
// SAME ray bit for bit -- the wavefront pipeline traces a camera ray in one kernel and shades its hit in another.
RT_DEV RayT<float> primary_ray_f32(const CameraT<float> &cam, int width, int height, bool use_time, uint32_t seed_lo, uint32_t seed_hi,
                                   uint32_t pixel, int px, int py, uint32_t s) {
    const Rng4 c = rng_block(pixel, s, RT_TAG_CAMERA, seed_lo, seed_hi);
    const float u = ((float)px + c.x) / (float)(width - 1), v = ((float)py + c.y) / (float)(height - 1);
    float dx = 0.0f, dy = 0.0f, tm = cam.time0;
    const float shutter = cam.time1 - cam.time0;
    if (cam.lens_radius > 0.0f) {
        const float rr = __builtin_amdgcn_sqrtf(c.z) * cam.lens_radius; dx = rr * cos2pi(c.w); dy = rr * sin2pi(c.w);
        if (use_time) tm = __builtin_fmaf(shutter, rng_block(pixel, s, RT_TAG_TIME, seed_lo, seed_hi).x, cam.time0);
    } else if (use_time) tm = __builtin_fmaf(shutter, c.z, cam.time0);
    RayT<float> r;                                                   // camera.c:46-53
    const float ox = __builtin_fmaf(cam.v[0], dy, cam.u[0] * dx), oy = __builtin_fmaf(cam.v[1], dy, cam.u[1] * dx), oz = __builtin_fmaf(cam.v[2], dy, cam.u[2] * dx);   // lens offset
    r.o = f3mk(cam.origin[0] + ox, cam.origin[1] + oy, cam.origin[2] + oz);
    r.d = f3mk(__builtin_fmaf(cam.vertical[0], v, __builtin_fmaf(cam.horizontal[0], u, cam.llc[0])) - r.o.x,
               __builtin_fmaf(cam.vertical[1], v, __builtin_fmaf(cam.horizontal[1], u, cam.llc[1])) - r.o.y,
               __builtin_fmaf(cam.vertical[2], v, __builtin_fmaf(cam.horizontal[2], u, cam.llc[2])) - r.o.z);
    r.time = tm;
    return r;
}

RT_DEV f3 sky_color(f3 dir) {                                                          // 01 main.c:59-64
    const float t = 0.5f * (normalize(dir).y + 1.0f);
    return f3mk(1, 1, 1) * (1.0f - t) + f3mk(0.5f, 0.7f, 1.0f) * t;
}
