// kernels_probe.hip -- the integrator and its shading functions addressed one ray / one input at a time (product code).
//
//  * k_ray_colors: ray_color (ray.c:50-84; 02 main.c:44-63; 01 main.c:39-65) for rays the CALLER supplies -- the same bounce
//    loop, traversal and shade_hit as the general megakernel (kernels_f32.hip), without its camera and tile mapping.
//  * k_kat: the device functions the production kernels call (Philox block, light-list pdf and sampler, cosine pdf,
//    Schlick reflectance, refraction, glass direction) evaluated on supplied inputs, for known-answer tests against the
//    reference's outputs (tests/golden/unit_vectors.npz).  The small-scene kernel's own variants live in kernels_small.hip.
#include <hip/hip_runtime.h>

#include "device/rt_shade.hpp"
#include "kernels.h"

namespace {

// lane = (ray i, sample residue c): samples s_begin + c, + chunks, ... of ray i; Philox keyed on (ray index, sample, bounce).
// Sums go to rgb_sum with double atomics (a test / single-ray entry point: the order of the float additions is not fixed).
template <bool BVH>
__global__ void __launch_bounds__(256) k_ray_colors(SceneT<float> sc, RenderParams P, const float *__restrict__ rays7, unsigned n, unsigned chunks,
                                                     double *__restrict__ rgb_sum, unsigned long long *__restrict__ ray_counter) {
    extern __shared__ int lds_stack[];                                     // BVH: sc.bvh_stack entries per lane
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned i = gid / chunks, c = gid % chunks;
    const bool valid = i < n;
    const float *a = rays7 + 7 * (size_t)(valid ? i : 0);
    RayT<float> r0; r0.o = f3mk(a[0], a[1], a[2]); r0.d = f3mk(a[3], a[4], a[5]); r0.time = a[6];
    f3 sum = f3mk(0, 0, 0);
    unsigned nrays = 0;
    int s_next = P.s_begin + (int)c, depth = 0, src = -1;
    uint32_t s_cur = 0;
    bool alive = false;
    PathState ps;
    ps.T = f3mk(1, 1, 1); ps.L = f3mk(0, 0, 0); ps.ray = r0;
    for (;;) {
        if (P.max_depth <= 0) break;                                       // ray.c:52
        if (!alive && valid && s_next < P.s_end) {
            s_cur = (uint32_t)s_next; s_next += (int)chunks;
            ps.ray = r0; ps.T = f3mk(1, 1, 1); ps.L = f3mk(0, 0, 0);
            depth = 0; alive = true; src = -1;
        }
        if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
        if (alive) {
            MediumSampler medium{&sc, &ps.ray, 0.001f, i, s_cur, (uint32_t)depth, P.seed_lo, P.seed_hi};
            HitT<float> h; int hit_src;
            if (BVH) {
                LdsStack stack(lds_stack + threadIdx.x, 256, sc.bvh_stack);
                const HitS hs = bvh_closest_hit(sc, ps.ray, 0.001f, Mth<float>::inf(), stack, medium, src);   // ray.c:56
                h.t = hs.t; h.prim = hs.prim; hit_src = hs.slot;
            } else {
                h = closest_hit_linear<float>(sc, ps.ray, 0.001f, Mth<float>::inf(), medium, src);
                hit_src = h.prim;
            }
            ++nrays;
            int status;
            if (h.prim < 0) {
                const f3 bg = P.integrator == 2 ? sky_color(ps.ray.d) : f3mk(sc.background[0], sc.background[1], sc.background[2]);
                ps.L = ps.L + ps.T * bg;
                status = RT_TERMINATE;
            } else {
                const PrimT<float> p = sc.prims[h.prim];
                const MaterialT<float> &m = sc.materials[p.mat];
                const RecT<float> rec = resolve_hit<float>(sc, ps.ray, p, h.t, material_needs_uv(sc, m));
                const Rng4 xi = rng_block(i, s_cur, (uint32_t)depth, P.seed_lo, P.seed_hi);
                status = shade_hit(sc, ps, rec, P.integrator, P.light_weight, xi);
                src = p.type == RTC_PRIM_MEDIUM ? -1 : hit_src;
            }
            ++depth;
            if (status == RT_TERMINATE || depth >= P.max_depth) {
                alive = false;
                const bool finite = (fabsf(ps.L.x) < 3.0e38f) && (fabsf(ps.L.y) < 3.0e38f) && (fabsf(ps.L.z) < 3.0e38f);
                if (finite) sum = sum + ps.L;
            }
        }
    }
    if (valid && (sum.x != 0.0f || sum.y != 0.0f || sum.z != 0.0f)) {
        atomicAdd(rgb_sum + 3 * (size_t)i, (double)sum.x); atomicAdd(rgb_sum + 3 * (size_t)i + 1, (double)sum.y); atomicAdd(rgb_sum + 3 * (size_t)i + 2, (double)sum.z);
    }
    unsigned total = nrays;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
    if ((threadIdx.x & 63) == 0 && total) atomicAdd(ray_counter, (unsigned long long)total);
}

__global__ void __launch_bounds__(256) k_kat(SceneT<float> sc, int what, const double *__restrict__ in, unsigned n, double *__restrict__ out) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int si = rtc_kat_in_stride(what), so = rtc_kat_out_stride(what);
    const double *a = in + (size_t)si * i;
    double *o = out + (size_t)so * i;
    auto v3at = [&](int k) { return f3mk((float)a[k], (float)a[k + 1], (float)a[k + 2]); };
    switch (what) {
    case RTC_KAT_PHILOX: {       // the raw block behind rng_block(pixel, sample, tag, k0, k1)
        const U4 b = philox4x32((uint32_t)a[0], (uint32_t)a[1], (uint32_t)a[2], (uint32_t)a[3], 0x52544331u, (uint32_t)a[4]);   // = rng_block's block
        o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; break;
    }
    case RTC_KAT_LIGHT_PDF: o[0] = sc.n_lights > 0 ? light_list_pdf(sc, v3at(0), v3at(3)) : 0.0; break;   // hittable_list.c:165-172
    case RTC_KAT_COSINE_PDF: o[0] = lambert_pdf(normalize(v3at(0)), normalize(v3at(3))); break;         // pdf.c:100-103 == material.c:205-208
    case RTC_KAT_SCHLICK: o[0] = schlick((float)a[0], (float)a[1]); break;                              // material.c:142
    case RTC_KAT_REFRACT: { const f3 r = refract(v3at(0), v3at(3), (float)a[6]); o[0] = r.x; o[1] = r.y; o[2] = r.z; break; }   // vec3.c:44-50
    case RTC_KAT_LIGHT_SAMPLE: {                                                                         // hittable.c:821-848
        const int pick = (int)a[3];
        const f3 d = (pick >= 0 && pick < sc.n_lights) ? light_random_one(sc.lights[pick], v3at(0), (float)a[4], (float)a[5]) : f3mk(1, 0, 0);
        o[0] = d.x; o[1] = d.y; o[2] = d.z; break;
    }
    case RTC_KAT_DIELECTRIC: {                                                                           // material.c:144-157
        RecT<float> rec; rec.n = v3at(3); rec.front = a[6] != 0.0;
        const f3 d = dielectric_dir((float)a[7], v3at(0), rec, (float)a[8]);
        o[0] = d.x; o[1] = d.y; o[2] = d.z; break;
    }
    default: for (int k = 0; k < so; ++k) o[k] = 0.0; break;
    }
}

}  // namespace

hipError_t launch_ray_colors(const SceneT<float> &sc, const RenderParams &p, const float *rays7, unsigned n, unsigned chunks, double *rgb_sum,
                             unsigned long long *ray_counter, hipStream_t stream) {
    if (n == 0 || p.s_end <= p.s_begin) return hipSuccess;
    const unsigned long long lanes = (unsigned long long)n * chunks;
    if (chunks == 0 || lanes > (1ull << 31)) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((lanes + 255) / 256);
    if (sc.n_nodes > 0) hipLaunchKernelGGL(k_ray_colors<true>, dim3(blocks), dim3(256), (size_t)sc.bvh_stack * 256 * sizeof(int), stream, sc, p, rays7, n, chunks, rgb_sum, ray_counter);
    else hipLaunchKernelGGL(k_ray_colors<false>, dim3(blocks), dim3(256), 0, stream, sc, p, rays7, n, chunks, rgb_sum, ray_counter);
    return hipGetLastError();
}
hipError_t launch_kat(const SceneT<float> &sc, int what, const double *in, unsigned n, double *out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_kat, dim3((n + 255) / 256), dim3(256), 0, stream, sc, what, in, n, out);
    return hipGetLastError();
}
