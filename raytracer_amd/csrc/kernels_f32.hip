// kernels_f32.hip -- production path tracer for gfx950 (product code): persistent-thread megakernel + tonemap.
//
// Replaces the pixel/sample loop of 03_FullTracer/main.c:137-160 and the recursion of ray.c:50-84.
//  * one wave64 = one 8x8 pixel tile; a lane owns one pixel for all samples of this call and keeps the pixel sum
//    in registers, so the float4 accumulator in HBM is read and written once per call (8 x 128-byte rows per wave);
//  * the recursion is a loop: every iteration traces ONE bounce for every live lane; a lane whose path ended
//    regenerates its next sample in the same iteration, so the wave stays full until its tile runs out of samples;
//  * the primitive loop index is wave-uniform: the scene tables of a small scene (Cornell: 13 leaves) are fetched
//    with scalar loads into SGPRs -- no VGPRs, no vector memory traffic, nothing to miss in HBM;
//  * material / pdf dispatch is a switch (rt_shade.hpp); random numbers are Philox4x32-7 on (pixel, sample, bounce).
#include <hip/hip_runtime.h>

#include "device/rt_shade.hpp"
#include "kernels.h"

namespace {


template <bool BVH>
__global__ void __launch_bounds__(256) k_pathtrace_mega(SceneT<float> sc, CameraT<float> cam, RenderParams P,
                                                         float4 *__restrict__ accum, unsigned long long *__restrict__ ray_counter) {
    extern __shared__ int lds_stack[];                                     // BVH: sc.bvh_stack entries per lane
    const int lane = threadIdx.x & 63;
    int block = blockIdx.x;
    if (BVH) {   // blocks with equal (id % 8) share an XCD and its L2: give each XCD one contiguous band of tiles
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = block & 7;
        block = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (block >> 3);
    }
    const int tile = block * 4 + (threadIdx.x >> 6);
    if (tile >= P.n_tiles) return;   // wave-uniform
    const int px = (tile % P.tiles_x) * 8 + (lane & 7), py = (tile / P.tiles_x) * 8 + (lane >> 3);
    const bool valid = px < P.width && py < P.height;
    const uint32_t pixel = (uint32_t)(py * P.width + px);

    f3 sum = f3mk(0, 0, 0);
    unsigned nrays = 0, iters = 0;
    int s_next = P.s_begin, depth = 0;
    uint32_t s_cur = 0;
    bool alive = false;
    int src = -1;                                                          // leaf the current ray starts on (BVH: its slot, else its id; -1: camera / inside a medium)
    PathState ps;
    ps.T = f3mk(1, 1, 1); ps.L = f3mk(0, 0, 0); ps.ray.o = f3mk(0, 0, 0); ps.ray.d = f3mk(0, 0, 1); ps.ray.time = 0;

    for (;;) {
        if (P.max_depth <= 0) break;                                       // ray.c:52: depth 0 -> every sample is black (wave-uniform)
        if (!alive && valid && s_next < P.s_end) {                         // main.c:150-152 + camera.c:46-53
            s_cur = (uint32_t)s_next++;
            ps.ray = primary_ray_f32(cam, P.width, P.height, P.use_time != 0, P.seed_lo, P.seed_hi, pixel, px, py, s_cur);
            ps.T = f3mk(1, 1, 1); ps.L = f3mk(0, 0, 0);
            depth = 0; alive = true; src = -1;
        }
        if (__builtin_amdgcn_ballot_w64(alive) == 0) break;                // every lane of the wave is out of samples
        ++iters;                                                           // wave-uniform: bounce iterations of this wave
        if (alive) {
            MediumSampler medium{&sc, &ps.ray, 0.001f, pixel, s_cur, (uint32_t)depth, P.seed_lo, P.seed_hi};
            HitT<float> h; int hit_src;                                    // hit_src: what `src` must hold for the next ray
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
            if (h.prim < 0) {                                              // ray.c:56: background (01: sky gradient)
                const f3 bg = P.integrator == 2 ? sky_color(ps.ray.d) : f3mk(sc.background[0], sc.background[1], sc.background[2]);
                ps.L = ps.L + ps.T * bg;
                status = RT_TERMINATE;
            } else {
                const PrimT<float> p = sc.prims[h.prim];
                const MaterialT<float> &m = sc.materials[p.mat];
                const RecT<float> rec = resolve_hit<float>(sc, ps.ray, p, h.t, material_needs_uv(sc, m));
                const Rng4 xi = rng_block(pixel, s_cur, (uint32_t)depth, P.seed_lo, P.seed_hi);
                status = shade_hit(sc, ps, rec, P.integrator, P.light_weight, xi);
                src = p.type == RTC_PRIM_MEDIUM ? -1 : hit_src;
            }
            ++depth;
            if (status == RT_TERMINATE || depth >= P.max_depth) {          // ray.c:52: depth exhausted adds nothing
                alive = false;
                const bool finite = (fabsf(ps.L.x) < 3.0e38f) && (fabsf(ps.L.y) < 3.0e38f) && (fabsf(ps.L.z) < 3.0e38f);
                if (finite) sum = sum + ps.L;                              // non-finite samples are dropped (reference: NaN pixel, F16)
            }
        }
    }
    if (valid) {
        float4 a = accum[pixel];
        a.x += sum.x; a.y += sum.y; a.z += sum.z; a.w += (float)(P.s_end - P.s_begin);
        accum[pixel] = a;
    }
    // one atomic per wave for the ray counter
    unsigned total = nrays;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
    if (lane == 0) { atomicAdd(ray_counter, (unsigned long long)total); atomicAdd(ray_counter + 1, (unsigned long long)iters); }
}

// The production traversal on its own (same code path the megakernel uses: brute-force scalar loop or the flattened
// BVH with its LDS stack), one lane per ray.  Media are skipped.  For validating the FP32 path against the FP64 one.
template <bool BVH>
__global__ void __launch_bounds__(256) k_first_hit_f32(SceneT<float> sc, const float *__restrict__ rays7, size_t n, float tmin, float tmax,
                                                        int32_t *__restrict__ prim_id, float *__restrict__ t_out) {
    extern __shared__ int lds_stack[];                                     // BVH: sc.bvh_stack entries per lane
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const float *a = rays7 + 7 * (live ? i : 0);
    RayT<float> r; r.o = f3mk(a[0], a[1], a[2]); r.d = f3mk(a[3], a[4], a[5]); r.time = a[6];
    NoMedium32 nv;
    HitT<float> h;
    if (BVH) { LdsStack stack(lds_stack + threadIdx.x, 256, sc.bvh_stack); const HitS hs = bvh_closest_hit(sc, r, tmin, tmax, stack, nv); h.t = hs.t; h.prim = hs.prim; }
    else h = closest_hit_linear<float>(sc, r, tmin, tmax, nv);
    if (live) { prim_id[i] = h.prim; t_out[i] = h.prim >= 0 ? h.t : 0.0f; }
}

// write_color (ray.c:35-47): mean -> sqrt -> clamp [0, 0.999] -> *256 -> u8.  One lane per pixel.
__global__ void __launch_bounds__(256) k_tonemap_u8(const float4 *__restrict__ accum, int n_pixels, unsigned char *__restrict__ rgb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const float4 a = accum[i];
    // double arithmetic like the reference (scale = 1.0/spp; sqrt(scale*c)): the byte must not depend on FP32 rounding
    const double scale = a.w > 0.0f ? 1.0 / (double)a.w : 0.0;
    const float c[3] = {a.x, a.y, a.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double v = sqrt(scale * (double)c[k]);
        v = (v < 0.0) ? 0.0 : ((v > 0.999) ? 0.999 : v);
        if (!(v == v)) v = 0.0;                        // NaN -> 0 (the reference casts NaN to unsigned char: UB, F16)
        rgb[3 * (size_t)i + k] = (unsigned char)(256 * v);
    }
}

// dst += src[0] + src[1] + ... over n float4, added in that order (rtc_render_multi: the accumulators of the other handles, staged on the
// root device, in handle order -- the sum does not depend on which device finished first)
struct AccumSources { const float4 *src[63]; int count; };
__global__ void __launch_bounds__(256) k_accum_add(float4 *__restrict__ dst, AccumSources s, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = dst[i];
    for (int k = 0; k < s.count; ++k) { const float4 b = s.src[k][i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    dst[i] = a;
}

}  // namespace

hipError_t launch_accum_add(float4 *dst, const float4 *const *src, int n_src, size_t n, hipStream_t stream) {
    if (n == 0 || n_src <= 0) return hipSuccess;
    if (n_src > 63) return hipErrorInvalidValue;
    AccumSources s; s.count = n_src;
    for (int k = 0; k < 63; ++k) s.src[k] = k < n_src ? src[k] : nullptr;
    hipLaunchKernelGGL(k_accum_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dst, s, n);
    return hipGetLastError();
}
hipError_t launch_pathtrace_mega(const SceneT<float> &sc, const CameraT<float> &cam, const RenderParams &p,
                                 float4 *accum, unsigned long long *ray_counter, hipStream_t stream) {
    if (p.n_tiles <= 0 || p.s_end <= p.s_begin) return hipSuccess;
    const unsigned blocks = (unsigned)((p.n_tiles + 3) / 4);
    if (sc.n_nodes > 0) hipLaunchKernelGGL(k_pathtrace_mega<true>, dim3(blocks), dim3(256), (size_t)sc.bvh_stack * 256 * sizeof(int), stream, sc, cam, p, accum, ray_counter);
    else hipLaunchKernelGGL(k_pathtrace_mega<false>, dim3(blocks), dim3(256), 0, stream, sc, cam, p, accum, ray_counter);
    return hipGetLastError();
}
hipError_t launch_first_hit_f32(const SceneT<float> &sc, const float *rays7, size_t n, float tmin, float tmax, int32_t *prim_id, float *t, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (sc.n_nodes > 0) hipLaunchKernelGGL(k_first_hit_f32<true>, dim3(blocks), dim3(256), (size_t)sc.bvh_stack * 256 * sizeof(int), stream, sc, rays7, n, tmin, tmax, prim_id, t);
    else hipLaunchKernelGGL(k_first_hit_f32<false>, dim3(blocks), dim3(256), 0, stream, sc, rays7, n, tmin, tmax, prim_id, t);
    return hipGetLastError();
}
hipError_t launch_tonemap_u8(const float4 *accum, int n_pixels, unsigned char *rgb, hipStream_t stream) {
    if (n_pixels <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_tonemap_u8, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, stream, accum, n_pixels, rgb);
    return hipGetLastError();
}
