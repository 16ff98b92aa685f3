// kernels_wf.hip -- wavefront path tracer for BVH scenes on gfx950 (product code).
//
// The persistent megakernel (kernels_f32.hip) carries the whole shading code (perlin turbulence, media, image textures)
// in one register allocation: 154 VGPRs -> 3 waves per SIMD, and a wave waits for its slowest lane in BOTH the traversal
// and the material switch (rocprofv3: active-lane fraction 0.20 on the Book-2 final scene).  For scenes that live in a
// BVH the bounce loop of ray.c:50-84 is therefore split into stages that run over queues of path slots in HBM:
//
//   k_wf_init     every slot: first camera ray of its pixel (main.c:150-152, camera.c:46-53)
//   k_wf_extend   closest hit for every live slot (hittable_list_hit, ray.c:56): traversal only
//   k_wf_shade    material / pdf switch of rt_shade.hpp; ends the path or writes the scattered ray; a finished path
//                 regenerates the slot's next sample, so a slot stays live until its pixel is out of samples
//   k_wf_compact  every other bounce, once a sixteenth of the list is holes: squeezes the holes out of the slot list
//   k_wf_finish   per pixel: accumulator += the sums of its slots (coalesced float4 read-modify-write)
//
// A slot owns ONE pixel and walks through its samples in a fixed order, so no atomics touch radiance and the image does
// not depend on scheduling.  The slot list keeps its order (tile order), which keeps the state accesses coalesced.
// Same-address atomics cost ~11 ns each on this part (measured: 32 K wave-level appends = 0.37 ms), so nothing on the
// per-bounce path appends to a shared queue: slots die in place, and the compaction uses one atomic per 4 K entries.
// Philox keys are the megakernel's: (pixel, sample, bounce) -- both kernels render the same estimator.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>

#include "device/rt_shade.hpp"
#include "kernels.h"

namespace {

// (slot sums: no-return atomics instead of the read-modify-write measured 1-3 % slower, round 3)
// Live slots = the count k_wf_init wrote (WF_LIVE0) minus the deaths reported since, and the deaths are a STRIPED statistic like the ray counts
// (B.stats, kWfStripes cache lines; one fire-and-forget atomic per wave that has something to report): a device-scope atomic costs ~10 ns
// on its address whether or not anything waits for its result -- k_wf_init spent 200 of its 350 us on 21,600 of them -- and in the second
// half of a render nearly every wave of an extend launch reports a death.  Who needs the count sums the stripes: the compaction kernels
// (every wave for itself, one stripe per lane), and k_wf_flip folds it into counters[WF_LIVE] for the early exits of extend and shade,
// which may lag behind (a stale count is only ever too high).
constexpr int kWfStripes = 64;   // cache lines each statistic is striped over
// What a wave of the bounce schedule reports -- rays traced, 1 if it held a live lane, slots that died -- goes out as ONE atomic on a packed word of
// its stripe: rays (24 bits) | waves (20) | dead slots (20).  k_wf_flip, alone on its part's stream at every compaction point, unpacks the words into the
// wide per-stripe sums and clears them, so a field holds at most the counts of the rounds between two compaction points (<= 16: render_wavefront) of
// 1/64 of <= 32 M slots; k_wf_finish adds what the last rounds left.  (Three atomics per wave instead of one: final 67.7 -> 65.8 ms without any.)
constexpr int kWfPackWaves = 24, kWfPackDeaths = 44, kWfPackPrimaryWaves = 40;
__device__ __forceinline__ unsigned long long *wf_packed(const WfBuffers &B, unsigned stripe) { return B.stats + 8 * (1 + 2 * kWfStripes + (stripe & (kWfStripes - 1))); }   // [0] packed, [1] dead slots folded so far, [2] k_wf_primary's rays | waves << 40
__device__ __forceinline__ void wave_report(const WfBuffers &B, unsigned stripe, unsigned long long rays_mask, unsigned long long died_mask) {   // one lane calls
    const unsigned long long v = (unsigned long long)__builtin_popcountll(rays_mask) | ((unsigned long long)(rays_mask != 0) << kWfPackWaves)
                               | ((unsigned long long)__builtin_popcountll(died_mask) << kWfPackDeaths);
    if (v) atomicAdd(wf_packed(B, stripe), v);
}
__device__ __forceinline__ unsigned wf_live_now(const WfBuffers &B) {          // called by whole waves
    const unsigned long long *st = wf_packed(B, threadIdx.x & 63);
    unsigned d = (unsigned)(st[0] >> kWfPackDeaths) + (unsigned)st[1];
#pragma unroll
    for (int off = 32; off; off >>= 1) d += __shfl_xor(d, off);
    return B.counters[WF_LIVE0] - d;
}

// Slot <-> (pixel, layer).  A wave is 64 consecutive slots.  quad = 0: one 8x8-pixel tile, one layer.  quad = 1 (layers a
// multiple of 4): a 4x4-pixel quadrant of the tile in 4 consecutive layers -- the rays of a wave then start closer
// together, and traversal length inside a wave diverges less (regrouping the list by material, which mixes 8 tiles
// into a wave, costs 25 %: spatial coherence is worth more than a uniform material).
// x / d for x < 2^30 through the multiplier the host has prepared (WfDiv, kernels.h): one 32 x 32 -> 64 multiply and a shift
// instead of the ~25 instructions of a general unsigned division
__device__ __forceinline__ int wf_div(int x, WfDiv d) { return (int)(((unsigned long long)(unsigned)x * d.mul) >> d.shift); }
__device__ __forceinline__ void slot_pixel(const RenderParams &P, const WfBuffers &B, int slot, int &px, int &py, int &layer) {
    const int ns = B.slots_per_layer;
    int tile, x, y;
    if (B.quad) {
        const int g = wf_div(slot, B.div_group), r = slot - g * 4 * ns;
        tile = r >> 8;
        const int qd = (r >> 6) & 3, pq = r & 15;
        layer = 4 * g + ((r >> 4) & 3);
        x = (qd & 1) * 4 + (pq & 3); y = (qd >> 1) * 4 + (pq >> 2);
    } else {
        layer = wf_div(slot, B.div_group);
        const int q = slot - layer * ns, lane = q & 63;
        tile = q >> 6; x = lane & 7; y = lane >> 3;
    }
    const int ty = wf_div(tile, B.div_tiles_x);
    px = (tile - ty * P.tiles_x) * 8 + x; py = ty * 8 + y;
}
__device__ __forceinline__ int pixel_slot(const WfBuffers &B, int tile, int x, int y, int layer) {
    if (B.quad) return (layer >> 2) * 4 * B.slots_per_layer + tile * 256 + ((y >> 2) * 2 + (x >> 2)) * 64 + (layer & 3) * 16 + (y & 3) * 4 + (x & 3);
    return layer * B.slots_per_layer + tile * 64 + y * 8 + x;
}

__device__ __forceinline__ RayT<float> primary_ray(const CameraT<float> &cam, const RenderParams &P, uint32_t pixel, int px, int py, uint32_t s) {
    return primary_ray_f32(cam, P.width, P.height, P.use_time != 0, P.seed_lo, P.seed_hi, pixel, px, py, s);   // same Philox blocks as the megakernel
}

// The closest hit of the camera ray of `sample` of this slot, if k_wf_primary has traced it ahead (the slot's samples 0 .. phit_k - 1)
__device__ __forceinline__ bool wf_camera_hit(const RenderParams &P, const WfBuffers &B, int slot, int sample, float2 &ph) {
    const int k = wf_div(sample - P.s_begin, B.div_layers);          // the slot's k-th sample
    if (k >= B.phit_k) return false;
    ph = B.phit[(size_t)k * (size_t)B.n_slots + (size_t)slot];
    return true;
}
// ray.c:58-83 for ONE list entry whose closest hit is known, then the slot's bookkeeping: a finished path adds its radiance to the
// slot's sum and the slot starts its next sample (or dies); otherwise the scattered ray is written back.  Returns true when the slot died.
//   GENERAL = false: the hit's 48-byte shading record answers everything (leaf_shade: normal or centre, resolved colour, face flags) --
//                    the form extend runs in its own tail, for every hit that is not marked RTC_SF_GENERAL;
//   GENERAL = true : the leaf needs prims / chains / materials / textures (a texture to evaluate, a wrapper chain to replay): k_wf_shade.
// A record k_wf_shade must shade in full: marked so, or an albedo-pending record (RTC_SF_TEXTURED) under integrator 0, whose weights read the albedo
__device__ __forceinline__ bool wf_is_general(const RenderParams &P, uint32_t meta) {
    return (meta & RTC_SF_GENERAL) || ((meta & RTC_SF_TEXTURED) && P.integrator == 0);
}
template <bool NOISE, bool GENERAL>
__device__ __forceinline__ bool wf_shade_entry(const SceneT<float> &sc, const CameraT<float> &cam, const RenderParams &P, const WfBuffers &B, int slot,
                                               int px, int py, const float4 &o, const float4 &d, const float4 &tm, float hit_t, int hslot, SmallShade sh, unsigned char &key) {
    // sample, depth, leaf slot the ray starts on, pixel
    int4 m = make_int4((int)(__float_as_uint(tm.z) & 0xffffffu), (int)(__float_as_uint(tm.z) >> 24), __float_as_int(tm.w), py * P.width + px);
    PathState ps;
    ps.ray.o = f3mk(o.x, o.y, o.z); ps.ray.d = f3mk(d.x, d.y, d.z); ps.ray.time = o.w;
    ps.T = f3mk(d.w, tm.x, tm.y); ps.L = f3mk(0, 0, 0);
    const uint32_t pixel = (uint32_t)m.w;
    bool died = false, have_sum = false;
    bool albedo_pending = false;                                     // the hit shaded last was a RTC_SF_TEXTURED record and its path goes on
    float4 s4 = make_float4(0, 0, 0, 0);
    // One trip = one hit shaded.  A path that ends here starts the slot's next sample, and the closest hit of that sample's camera ray is
    // already known (B.phit, k_wf_primary: the camera rays of a render are traced up front, in coherent waves of their own) -- so the
    // loop shades it at once, and the slot goes back to the list with the SECOND ray of its new path: camera rays, a third of all rays,
    // never enter the slot lists, where they would share waves with rays of every depth.
    for (;;) {
        int status;
        ps.L = f3mk(0, 0, 0);
        if (hslot < 0) {                                             // ray.c:56: background (01: sky gradient)
            const f3 bg = P.integrator == 2 ? sky_color(ps.ray.d) : f3mk(sc.background[0], sc.background[1], sc.background[2]);
            ps.L = ps.T * bg;
            status = RT_TERMINATE;
        } else {
            const Rng4 xi = rng_block(pixel, (uint32_t)m.x, (uint32_t)m.y, P.seed_lo, P.seed_hi);
            if (!GENERAL) {
                RecT<float> rec; rec.u = 0; rec.v = 0; rec.mat = 0;
                rec.p = ps.ray.o + ps.ray.d * hit_t;                  // ray_at, ray.c:8-11
                f3 gv = f3mk(sh.g[0], sh.g[1], sh.g[2]);
                if (sh.meta & RTC_SF_MOVING) gv = gv + f3mk(sh.pad[0], sh.pad[1], sh.pad[2]) * ps.ray.time;   // moving_sphere_center, hittable.c:455-458
                const f3 n_out = (sh.meta & RTC_SF_SPHERE) ? (rec.p - gv) * sh.g[3] : gv;
                bool front = dot(ps.ray.d, n_out) < 0.0f;             // hit_record_set_facenormal, hittable.c:13-16
                rec.n = front ? n_out : -n_out;
                if (sh.meta & RTC_SF_FORCED_FRONT) front = true;      // translate/rotate re-face an already faced normal (F11)
                if (sh.meta & RTC_SF_FLIP) front = !front;            // flip_face_hit, hittable.c:587-591
                if (sh.meta & RTC_SF_MEDIUM) { rec.n = f3mk(1, 0, 0); front = true; }   // constant_medium_hit, hittable.c:578-582
                rec.front = front;
                MaterialT<float> mat; mat.type = (int)(sh.meta & 0xffu); mat.tex = -1; mat.albedo[0] = sh.rgb[0]; mat.albedo[1] = sh.rgb[1]; mat.albedo[2] = sh.rgb[2]; mat.param = sh.param;
                status = shade_hit_mat<false>(sc, ps, rec, mat, true, P.integrator, P.light_weight, xi);
                m.z = (sh.meta & RTC_SF_MEDIUM) ? -1 : hslot;
                albedo_pending = (sh.meta & RTC_SF_TEXTURED) != 0;
            } else {
                const PrimT<float> p = sc.prims[sc.leaf_prims[hslot]];
                const MaterialT<float> &mat = sc.materials[p.mat];
                const RecT<float> rec = resolve_hit<float>(sc, ps.ray, p, hit_t, material_needs_uv(sc, mat));
                status = shade_hit<NOISE>(sc, ps, rec, P.integrator, P.light_weight, xi);
                m.z = p.type == RTC_PRIM_MEDIUM ? -1 : hslot;
            }
        }
        ++m.y;
        if (!(status == RT_TERMINATE || m.y >= P.max_depth)) break;  // the path goes on with the scattered ray (ray.c:52: depth exhausted adds nothing)
        const bool finite = (fabsf(ps.L.x) < 3.0e38f) && (fabsf(ps.L.y) < 3.0e38f) && (fabsf(ps.L.z) < 3.0e38f);
        if (finite && (ps.L.x != 0.0f || ps.L.y != 0.0f || ps.L.z != 0.0f)) {
            // a slot is touched by ONE lane per launch and launches are ordered, and its samples are added in sample order: the sum is the
            // same from run to run (read once, written once per call, whatever the number of paths that end in it)
            if (!have_sum) { s4 = B.sum[slot]; have_sum = true; }
            s4.x += ps.L.x; s4.y += ps.L.y; s4.z += ps.L.z;
        }
        albedo_pending = false;                                      // (that path is over: nothing reads its throughput any more)
        m.x += B.layers;                                             // this slot's next sample
        if (m.x >= P.s_end) { died = true; break; }
        ps.ray = primary_ray(cam, P, pixel, px, py, (uint32_t)m.x);
        ps.T = f3mk(1, 1, 1); m.y = 0; m.z = -1;
        float2 ph;
        if (!wf_camera_hit(P, B, slot, m.x, ph)) break;              // not traced ahead: the camera ray takes the ordinary route through extend
        hit_t = ph.x; hslot = __float_as_int(ph.y);
        if (!GENERAL) {
            sh.meta = 0;
            if (hslot >= 0) sh = sc.leaf_shade[hslot];
            if (wf_is_general(P, sh.meta)) {                         // k_wf_shade's: park the hit and its regrouping key next to the new path's state
                B.hit[slot] = ph;
                key = (unsigned char)((sh.meta >> 16) & 31u);
                break;
            }
        } else if (hslot >= 0 && !wf_is_general(P, sc.leaf_shade[hslot].meta)) break;   // a leaf that extend's tail shades from its record: left to the
                                                                     // next extend launch (which takes the known hit and traces nothing), so that a hit is
                                                                     // shaded by the same code whichever kernel started its path
    }
    if (have_sum) B.sum[slot] = s4;
    if (!GENERAL && albedo_pending && !died) key = (unsigned char)(RTC_WF_KEY_ALBEDO | ((sh.meta >> 16) & 31u));   // k_wf_shade multiplies the throughput written below
    if (!died) {
        B.ro[slot] = make_float4(ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, ps.ray.time);
        B.rd[slot] = make_float4(ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, ps.T.x);
        B.tm[slot] = make_float4(ps.T.y, ps.T.z, __uint_as_float((uint32_t)m.x | ((uint32_t)m.y << 24)), __int_as_float(m.z));
    }
    return died;
}
// The extend kernels take their arguments as ONE struct, so that the tail can read them a second time: the bounce loop needs a
// dozen scalars of them, the tail (camera, render parameters, buffer pointers) fifty more, and with everything live across the loop
// the kernels spilled up to 86 SGPRs to vector lanes.  wf_args_again() hands out the kernel-argument segment through a pointer the
// compiler cannot see through, so what is read through it is (re)loaded by s_load where the tail stands -- behind the loop.
struct WfArgs { SceneT<float> sc; CameraT<float> cam; RenderParams P; WfBuffers B; };
// (`when` = a wave-uniform value that differs from use to use.  The asm statement is not volatile -- a volatile one counts as a write
// to memory, and hipcc then stops using scalar loads for uniform reads behind it -- so what keeps it, and the loads behind it, from
// being hoisted to the top of the kernel is that its result depends on `when`.)
__device__ __forceinline__ const WfArgs &wf_args_again(unsigned when) {
    const WfArgs __attribute__((address_space(4))) *p = (const WfArgs __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    asm("" : "+s"(p) : "s"(when));
    return *(const WfArgs *)p;
}
// The tail of both extend kernels: a hit whose shading record is complete is shaded where it was found -- the ray state is still in
// registers, and 97-100 % of the hits of the reference's scenes are of that kind -- so the path state makes ONE trip through HBM per
// bounce instead of two and k_wf_shade is left with the leaves that need a texture evaluated or a wrapper chain replayed (a scene
// without such leaves does not launch it at all).  Measured: final 126 -> ... ms, 1M spheres 54 -> ... ms.
__device__ __forceinline__ bool wf_extend_tail(const SceneT<float> &sc, const CameraT<float> &cam, const RenderParams &P, const WfBuffers &B, int slot,
                                               int px, int py, const float4 &o, const float4 &d, const float4 &tm, const HitS &h, unsigned char &key) {
    SmallShade sh; sh.meta = 0;
    if (h.slot >= 0) sh = sc.leaf_shade[h.slot];
    if (wf_is_general(P, sh.meta)) {                                 // k_wf_shade's: park the hit and its regrouping key
        B.hit[slot] = make_float2(h.t, __int_as_float(h.slot));
        key = (unsigned char)((sh.meta >> 16) & 31u);
        return false;
    }
    return wf_shade_entry<false, false>(sc, cam, P, B, slot, px, py, o, d, tm, h.t, h.slot, sh, key);
}

__global__ void __launch_bounds__(256) k_wf_init(CameraT<float> cam, RenderParams P, WfBuffers B) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = false;
    if (slot < B.n_slots) {
        int px, py, layer; slot_pixel(P, B, slot, px, py, layer);
        B.sum[slot] = make_float4(0, 0, 0, 0);
        const int s = P.s_begin + layer;
        if (px < P.width && py < P.height && s < P.s_end && P.max_depth > 0) {
            const uint32_t pixel = (uint32_t)(py * P.width + px);
            const RayT<float> r = primary_ray(cam, P, pixel, px, py, (uint32_t)s);
            B.ro[slot] = make_float4(r.o.x, r.o.y, r.o.z, r.time);
            B.rd[slot] = make_float4(r.d.x, r.d.y, r.d.z, 1.0f);
            B.tm[slot] = make_float4(1.0f, 1.0f, __int_as_float(s), __int_as_float(-1));
            live = true;
        }
        B.active[0][slot] = live ? slot : -1;
    }
    // the number of live slots needs no counting: every pixel of the image has one slot per layer, and a part has at least as many samples
    // as layers (render_wavefront) -- one atomic per block on one address, 21,600 of them for 5.5 M slots, was what this kernel waited for
    if (slot == 0) {
        B.counters[WF_LIVE] = B.counters[WF_LIVE0] = P.max_depth > 0 ? (unsigned)P.width * (unsigned)P.height * (unsigned)min(B.layers, P.s_end - P.s_begin) : 0u;
        B.counters[WF_NLIST] = (unsigned)B.n_slots; B.counters[WF_CUR] = 0u; B.counters[WF_CURSOR] = 0u;
    }
}

// counters: [0] live slots, [1] cursor of the compaction; stats (64-bit): [0] rays, [1] extend waves with a live lane
//
// One lane per list entry, one ray per lane.  Two persistent variants were measured and dropped (wavefront):
// lanes refilled from an LDS cursor once 16 of a wave idle, and lanes walking a strided private sequence with the next
// ray prefetched into registers.  Both raised the fraction of busy lanes per traversal step (0.42 -> ~0.7 on 1M spheres)
// and both were 35-40 % SLOWER: a refill puts a dependent slot-list -> ray-state load chain into the step loop and the
// whole wave waits for it, while plain one-ray waves leave that latency to the other resident waves.
// 6 waves per SIMD (80 VGPRs, 4 spilled in the medium path): +3 % over the uncapped 88; 8 waves (24 spilled) is slower
// MEDIA (SceneT::media_kind) keeps the boundary-list loops of a general constant medium, and the Philox draws of any
// medium, out of the kernels of scenes that have none.
using MediumSphereOnly = MediumSamplerT<false>;
template <int MEDIA> struct WfMedium;
template <> struct WfMedium<0> { using type = NoMedium32; };
template <> struct WfMedium<1> { using type = MediumSphereOnly; };
template <> struct WfMedium<2> { using type = MediumSampler; };
// (Round 3, measured and dropped: an XCD-aware block -> chunk mapping -- block b on chunk (b % 8) * (n / 8) + b / 8, so that each XCD's L2 sees one
// contiguous eighth of the slot list, a band of the image -- was 1-4 % SLOWER on final, book1 and the 1M spheres: the bands differ in cost
// (sky against ground) and the XCDs then finish at different times; the round-robin deal balances them.)
#ifndef RT_WF_EXTEND_BLOCK
#define RT_WF_EXTEND_BLOCK 64
#endif
constexpr int kWfExtendBlock = RT_WF_EXTEND_BLOCK;
// Resident waves per SIMD the register budget of extend and of the camera-ray pass is set for (512 / waves registers per lane).  Since extend
// shades the hits it finds (round 3) the kernels want 104-110 registers; at 8 waves (64 registers, 24 of them spilled) the spills cost more
// than the eighth wave hides.  One GPU run, whole renders, interleaved: 1M spheres 8 / 7 / 6 / 5 / 4 waves = 40.4 / 37.5 / 36.9 / 36.7 / 36.5 ms,
// final scene 91.1 / 88.4 / 90.5 / 89.4 / 89.4 ms, final + earth 159.4 / 155.0 / 157.4 / 157.8 / 158.4 ms (profiles/r03e/extend_waves_ab.txt) --
// so the spheres-only kernels are built for 5 waves (96 registers) and the others for 7 (72).  -DRT_WF_EXTEND_WAVES=n forces one value (A/B).
constexpr int wf_extend_waves(int media, int kinds) {
#ifdef RT_WF_EXTEND_WAVES
    return media == 2 ? 6 : RT_WF_EXTEND_WAVES;
#else
    return media == 2 ? 6 : (kinds == RTC_KIND_SPHERE ? 5 : 7);
#endif
}
template <int MEDIA, int KINDS>
__global__ void __launch_bounds__(kWfExtendBlock) __attribute__((amdgpu_waves_per_eu(wf_extend_waves(MEDIA, KINDS)))) k_wf_extend(WfArgs a) {
    const SceneT<float> &sc = a.sc; const RenderParams &P = a.P; const WfBuffers &B = a.B;
    if (B.counters[WF_LIVE] == 0) return;                            // the render is over: the rest of the fixed bounce schedule drains at once
    const unsigned n_list = B.counters[WF_NLIST]; const int cur = (int)B.counters[WF_CUR];   // the list as the last compaction left it
    if (blockIdx.x * blockDim.x >= n_list) return;
    extern __shared__ int lds_stack[];                               // sc.bvh_stack entries per lane
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const int slot = i < n_list ? B.active[cur][i] : -1;
    const bool live = slot >= 0;
    bool died = false, known = false;
    unsigned char key = (unsigned char)RTC_WF_KEY_DONE;
    if (live) {
        const float4 o = B.ro[slot], d = B.rd[slot];
        const float4 tm = B.tm[slot];                                // throughput y z, sample | depth << 24, src slot
        const int src_slot = __float_as_int(tm.w);
        RayT<float> r; r.o = f3mk(o.x, o.y, o.z); r.d = f3mk(d.x, d.y, d.z); r.time = o.w;
        int px, py, layer; slot_pixel(P, B, slot, px, py, layer);
        typename WfMedium<MEDIA>::type medium;
        if constexpr (MEDIA != 0) {
            const uint32_t sd = __float_as_uint(tm.z);
            medium.sc = &sc; medium.ray = &r; medium.tmin = 0.001f; medium.pixel = (uint32_t)(py * P.width + px); medium.sample = sd & 0xffffffu; medium.bounce = sd >> 24;
            medium.k0 = P.seed_lo; medium.k1 = P.seed_hi;
        }
        LdsStack stack(lds_stack + threadIdx.x, kWfExtendBlock, sc.bvh_stack);
        float2 ph;                                                   // a camera ray (depth 0) whose hit k_wf_primary has found already: nothing to trace
        known = (__float_as_uint(tm.z) >> 24) == 0u && wf_camera_hit(P, B, slot, (int)(__float_as_uint(tm.z) & 0xffffffu), ph);
        HitS h = bvh_closest_hit<KINDS>(sc, r, 0.001f, Mth<float>::inf(), stack, medium, src_slot, known);   // ray.c:56
        if (known) { h.t = ph.x; h.slot = __float_as_int(ph.y); }
        const WfArgs &t = wf_args_again(n_list);
        died = wf_extend_tail(t.sc, t.cam, t.P, t.B, slot, px, py, o, d, tm, h, key);
        if (died) t.B.active[cur][i] = -1;                           // stays in the list as a hole until the next compaction
    }
    if (sc.n_general && i < n_list) B.key[i] = key;                  // by LIST position: k_wf_shade scans them with coalesced reads
    // statistics and the slots that died: one fire-and-forget atomic per wave (a known hit was counted by k_wf_primary)
    const unsigned long long rays_mask = __builtin_amdgcn_ballot_w64(live && !known), died_mask = __builtin_amdgcn_ballot_w64(died);
    if ((threadIdx.x & 63) == 0) wave_report(B, blockIdx.x * (kWfExtendBlock / 64) + (threadIdx.x >> 6), rays_mask, died_mask);
}

// The camera rays of a render, traced ahead of its bounce schedule: entry (k, slot) = the closest hit of the slot's k-th sample, written
// to B.phit as (t, leaf slot); an extend lane that finds a camera ray in its list (sample 0 from k_wf_init) takes the hit from there.  A wave = 64 consecutive slots = a 4x4-pixel
// quadrant in four sample layers (slot_pixel), all on camera rays: its lanes walk nearly the same nodes, where the waves of the bounce
// schedule mix rays of every depth (final scene: 14 G rays/s here against 6.6 G for the rays of the lists).  Same ray (primary_ray),
// same traversal, same medium draws as extend would make for it at depth 0 -- the images are bit-identical to tracing them from the list.
// WAVE = true: one traversal for the whole wave (rt_core.hpp bvh_closest_hit_wave: the node in SGPRs, one scalar load per node or leaf record).
#ifndef RT_WF_PRIMARY_WAVES
#define RT_WF_PRIMARY_WAVES(media, kinds) wf_extend_waves(media, kinds)
#endif
template <int MEDIA, int KINDS, bool WAVE>
__global__ void __launch_bounds__(kWfExtendBlock) __attribute__((amdgpu_waves_per_eu(RT_WF_PRIMARY_WAVES(MEDIA, KINDS)))) k_wf_primary(WfArgs a) {
    const SceneT<float> &sc = a.sc; const RenderParams &P = a.P; const WfBuffers &B = a.B;
    extern __shared__ int lds_stack[];                               // sc.bvh_stack entries per lane
    const int slot = blockIdx.x * blockDim.x + threadIdx.x, k = (int)blockIdx.y;
    bool live = false;
    if (slot < B.n_slots) {
        int px, py, layer; slot_pixel(P, B, slot, px, py, layer);
        const int s = P.s_begin + layer + k * B.layers;
        if (px < P.width && py < P.height && s < P.s_end) {
            live = true;
            if constexpr (!WAVE) {
                const uint32_t pixel = (uint32_t)(py * P.width + px);
                RayT<float> r = primary_ray(a.cam, P, pixel, px, py, (uint32_t)s);
                typename WfMedium<MEDIA>::type medium;
                if constexpr (MEDIA != 0) {
                    medium.sc = &sc; medium.ray = &r; medium.tmin = 0.001f; medium.pixel = pixel; medium.sample = (uint32_t)s; medium.bounce = 0;
                    medium.k0 = P.seed_lo; medium.k1 = P.seed_hi;
                }
                LdsStack stack(lds_stack + threadIdx.x, kWfExtendBlock, sc.bvh_stack);
                const HitS h = bvh_closest_hit<KINDS>(sc, r, 0.001f, Mth<float>::inf(), stack, medium, -1);   // ray.c:56
                B.phit[(size_t)k * (size_t)B.n_slots + (size_t)slot] = make_float2(h.t, __int_as_float(h.slot));
            }
        }
    }
    if constexpr (WAVE) {                                            // every lane takes part in the wave's traversal; lanes without a ray only follow
        static_assert(kWfExtendBlock == 64, "one wave per workgroup");
        int px = 0, py = 0, layer = 0;
        if (slot < B.n_slots) slot_pixel(P, B, slot, px, py, layer);
        const int s = P.s_begin + layer + k * B.layers;
        const uint32_t pixel = (uint32_t)(py * P.width + px);
        RayT<float> r = primary_ray(a.cam, P, pixel, px, py, (uint32_t)s);
        typename WfMedium<MEDIA>::type medium;
        if constexpr (MEDIA != 0) {
            medium.sc = &sc; medium.ray = &r; medium.tmin = 0.001f; medium.pixel = pixel; medium.sample = (uint32_t)s; medium.bounce = 0;
            medium.k0 = P.seed_lo; medium.k1 = P.seed_hi;
        }
        const HitS h = bvh_closest_hit_wave<KINDS>(sc, r, live, 0.001f, Mth<float>::inf(), lds_stack, medium);   // ray.c:56
        if (live) B.phit[(size_t)k * (size_t)B.n_slots + (size_t)slot] = make_float2(h.t, __int_as_float(h.slot));
    }
    // statistics: one atomic per wave on a word of its own in the stripe's line, rays (40 bits) | waves (24) -- a launch is at most 32 M slots x 32
    // samples over 64 stripes; k_wf_finish unpacks
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(live);
    if ((threadIdx.x & 63) == 0 && mask)
        atomicAdd(wf_packed(B, (blockIdx.x + blockIdx.y) * (kWfExtendBlock / 64) + (threadIdx.x >> 6)) + 2, (unsigned long long)__builtin_popcountll(mask) | (1ull << kWfPackPrimaryWaves));
}

// Extend for SMALL trees (book1: 281 nodes + 486 leaf records = 49 KB): a workgroup of 768 threads copies the node and the leaf table
// into LDS, and every trip's 64-byte fetch is four ds_read_b128 instead of a trip through the address unit, L1 and L2 -- which is
// what the plain kernel waits for on such a scene (its lanes are coherent enough to outrun the L1's 64-byte-record rate).
// Two workgroups per CU (6 waves per SIMD) need tables + stacks <= 80 KB each: the stacks hold 16-bit entries (wf_lds_tables()
// checks that every node index and leaf reference fits).  Hits are bit-identical to k_wf_extend's: same traversal, same order.
// Threads per workgroup; two workgroups fit a CU, i.e. 6 waves per SIMD with 80 registers each at 768.  Round 2 ran 1024 (8 waves, 64 registers);
// since the kernel shades its hits (round 3) that meant 36 spilled registers.  book1 1080p x 64 spp, one GPU run, interleaved:
// 1024 / 896 / 768 / 640 / 512 threads = 23.6 / 22.7 / 21.5 / 23.7 / 22.4 ms (profiles/r03e/extend_waves_ab.txt).  -DRT_WF_LDS_BLOCK=n: A/B knob.
#ifndef RT_WF_LDS_BLOCK
#define RT_WF_LDS_BLOCK 768
#endif
constexpr int kWfLdsBlock = RT_WF_LDS_BLOCK;
template <int KINDS>
__global__ void __launch_bounds__(kWfLdsBlock) __attribute__((amdgpu_waves_per_eu(kWfLdsBlock / 128))) k_wf_extend_lds(WfArgs a) {
    const SceneT<float> &sc = a.sc; const RenderParams &P = a.P; const WfBuffers &B = a.B;
    if (B.counters[WF_LIVE] == 0) return;
    const unsigned n_list = B.counters[WF_NLIST]; const int cur = (int)B.counters[WF_CUR];
    const int n_leaf_recs = sc.n_bleaves;
    extern __shared__ float4 lds_tab[];                              // nodes | leaves | stacks
    float4 *const node_tab = lds_tab, *const leaf_tab = lds_tab + 4 * sc.n_nodes;
    short *const stacks = reinterpret_cast<short *>(leaf_tab + 4 * n_leaf_recs);
    {
        const float4 *gn = reinterpret_cast<const float4 *>(sc.nodes), *gl = reinterpret_cast<const float4 *>(sc.bleaves);
        for (int k = threadIdx.x; k < 4 * sc.n_nodes; k += kWfLdsBlock) node_tab[k] = gn[k];
        for (int k = threadIdx.x; k < 4 * n_leaf_recs; k += kWfLdsBlock) leaf_tab[k] = gl[k];
    }
    __shared__ unsigned wg_next;
    if (threadIdx.x == 0) wg_next = 0u;
    __syncthreads();
    // persistent: a workgroup owns a contiguous slice of the list's 64-entry chunks and its 12 waves claim them one at a time from a counter
    // in LDS (a same-address atomic in HBM costs ~11 ns and serialises: one per chunk was 12 % slower than the static partition it replaced)
    // -- no barrier after the copy, so a workgroup's waves do not wait for its slowest one, and the table is copied once per workgroup
    unsigned long long n_rays = 0, n_waves = 0, n_died = 0;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned n_chunks = (n_list + 63u) / 64u, per_wg = (n_chunks + gridDim.x - 1) / gridDim.x;
    const unsigned wg_first = blockIdx.x * per_wg, wg_end = wg_first + per_wg < n_chunks ? wg_first + per_wg : n_chunks;
    for (;;) {
        unsigned chunk = 0;
        if (lane == 0) chunk = atomicAdd(&wg_next, 1u);
        chunk = wg_first + (unsigned)__builtin_amdgcn_readfirstlane((int)chunk);
        if (chunk >= wg_end) break;
        const unsigned i = chunk * 64u + lane;
        const int slot = i < n_list ? B.active[cur][i] : -1;
        const bool live = slot >= 0;
        bool died = false, known = false;
        unsigned char key = (unsigned char)RTC_WF_KEY_DONE;
        if (live) {
            const float4 o = B.ro[slot], d = B.rd[slot];
            const float4 tm = B.tm[slot];                            // throughput y z, sample | depth << 24, src slot
            const int src_slot = __float_as_int(tm.w);
            RayT<float> r; r.o = f3mk(o.x, o.y, o.z); r.d = f3mk(d.x, d.y, d.z); r.time = o.w;
            NoMedium32 medium;
            LdsStack16 stack(stacks + threadIdx.x, kWfLdsBlock, sc.bvh_stack);
            float2 ph;                                               // a camera ray whose hit is known (k_wf_extend)
            known = (__float_as_uint(tm.z) >> 24) == 0u && wf_camera_hit(P, B, slot, (int)(__float_as_uint(tm.z) & 0xffffffu), ph);
            HitS h = bvh_closest_hit_tab<KINDS>(sc, node_tab, leaf_tab, r, 0.001f, Mth<float>::inf(), stack, medium, src_slot, known);   // ray.c:56
            if (known) { h.t = ph.x; h.slot = __float_as_int(ph.y); }
            const WfArgs &t = wf_args_again(chunk);
            int px, py, layer; slot_pixel(t.P, t.B, slot, px, py, layer);
            died = wf_extend_tail(t.sc, t.cam, t.P, t.B, slot, px, py, o, d, tm, h, key);
            if (died) t.B.active[cur][i] = -1;
        }
        if (sc.n_general && i < n_list) B.key[i] = key;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(live && !known);
        n_rays += (unsigned long long)__builtin_popcountll(mask); n_waves += mask != 0;
        n_died += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(died));
    }
    // (a persistent wave sees n_list / (waves of the grid) entries per launch: far inside the packed fields)
    const unsigned long long v = n_rays | (n_waves << kWfPackWaves) | (n_died << kWfPackDeaths);
    if (lane == 0 && v) atomicAdd(wf_packed(B, blockIdx.x * (kWfLdsBlock / 64) + wave), v);
}

// The hits extend has parked for this stage (leaves marked RTC_SF_GENERAL: a perlin / image / checker texture to evaluate, a wrapper
// chain to replay).  They may be few -- the final scene's perlin sphere is 2-3 % of the hits -- and a wave that shades 6 of them
// among 58 idle lanes issues the 200 table loads of the turbulence at full price: the address unit takes a load instruction as
// long whatever the number of lanes (a launch of the one-block-per-1024-entries form: 530 us for 16.7 M entries).  So a persistent
// workgroup walks its share of the list 1024 entries at a time, queues what is parked in LDS, and shades only when 1024 entries
// wait (and once more at the end): dense waves whatever the density of the list.  The queued entries are counting-sorted in LDS
// by (material type, texture type) of what they hit, so a wave runs one arm of the material switch instead of all of them.
// Only slot ids move; no global atomics; the order of the work is a function of the list alone.
// 256 threads per workgroup, four workgroups per CU, and NO register ceiling below the hardware's (waves_per_eu 3: 142 / 12x VGPRs, no
// spills).  Until the last session of round 3 this was 1024 threads at 5 waves per SIMD (96 VGPRs, spills in both variants) -- tuned when the
// kernel shaded every hit.  It now shades the few per cent extend parks, it is latency-bound (dependent table loads), and it runs beside
// the other parts' extend launches: what counts is how long its waves stay resident, not how many fit.  One GPU run, 1024 x 5 / 256 x 3:
// final 74.1 / 72.9 ms, final + earth 134.0 / 129.0, Book-2 random (checker ground: a third of the hits parked) 28.7 / 26.5; 256 x 2, 256 x 4,
// 128 x 3, 512 x 3 within 1 % of 256 x 3 (tools/ab_configs.sh).
#ifndef RT_WF_SHADE_BLOCK
#define RT_WF_SHADE_BLOCK(noise) 256
#endif
#ifndef RT_WF_SHADE_WAVES
#define RT_WF_SHADE_WAVES(noise) 3
#endif
constexpr int wf_shade_block(bool noise) { return RT_WF_SHADE_BLOCK(noise); }
constexpr int wf_shade_waves(bool noise) { return RT_WF_SHADE_WAVES(noise); }
// The turbulence of a noise texture is 7 octaves x 8 lattice corners x (3 permutation + 3 vector words): 210 dependent-pair table loads per hit
// from 6 KB of tables.  The workgroup keeps the tables of up to kWfShadeNoiseSlots noise textures in LDS and hands the shading code
// generic pointers to them (a scene with more noise textures reads them from memory as before).
constexpr int kWfShadeNoiseSlots = 2;
// The scan takes FOUR keys per thread and trip (one dword load, one barrier per 4096 list entries; the next trip's keys are requested
// before this trip's are looked at): with one key per thread and two barriers per 1024 entries the scan of a sparse list -- 2-3 % parked --
// was most of this kernel's time.  The queue is a ring of 8192 entries (list position + key) in LDS.
template <bool NOISE>
__global__ void __launch_bounds__(wf_shade_block(NOISE)) __attribute__((amdgpu_waves_per_eu(wf_shade_waves(NOISE)))) k_wf_shade(WfArgs a) {
    constexpr unsigned N = wf_shade_block(NOISE), KPT = 4, QCAP = 8 * N;
    const WfBuffers &B = a.B;
    if (B.counters[WF_LIVE] == 0) return;
    const unsigned n_list = B.counters[WF_NLIST]; const int cur = (int)B.counters[WF_CUR];
    __shared__ unsigned q_index[QCAP];
    __shared__ unsigned char q_key[QCAP];
    __shared__ unsigned wave_count[N / 64], key_count[64], key_start[64];
    __shared__ unsigned sorted_index[N];
    __shared__ float noise_vec[NOISE ? 768 * kWfShadeNoiseSlots : 1];
    __shared__ int noise_perm[NOISE ? 768 * kWfShadeNoiseSlots : 1];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const bool noise_in_lds = NOISE && a.sc.n_perlin <= (a.B.noise_lds_slots < kWfShadeNoiseSlots ? a.B.noise_lds_slots : kWfShadeNoiseSlots);
    if (noise_in_lds) {                                              // (visible to every wave behind the first barrier below)
        for (unsigned t = tid; t < 768u * (unsigned)a.sc.n_perlin; t += N) { noise_vec[t] = a.sc.perlin_vec[t]; noise_perm[t] = a.sc.perlin_perm[t]; }
    }
    const unsigned n_chunks = (n_list + KPT * N - 1) / (KPT * N);
    const unsigned *const key_words = reinterpret_cast<const unsigned *>(B.key);
    const unsigned done4 = 0x01010101u * (unsigned)RTC_WF_KEY_DONE;
    unsigned q_n = 0, q_head = 0;                                    // block-uniform: every thread keeps the same counts
    auto keys_of = [&](unsigned chunk) { const unsigned w = chunk * N + tid; return w * KPT < n_list ? key_words[w] : done4; };   // (n_slots is a multiple of 64: a word is inside the array or not at all)
    unsigned next = keys_of(blockIdx.x);
    for (unsigned c = blockIdx.x;; c += gridDim.x) {
        const bool flush = c >= n_chunks;                            // one trip past the end: shade what is still queued
        if (!flush) {
            unsigned keys = next;
            next = keys_of(c + gridDim.x);
            const unsigned i0 = (c * N + tid) * KPT;
            unsigned pos[KPT], n_wave = 0;
            bool parked[KPT];
#pragma unroll
            for (unsigned j = 0; j < KPT; ++j) {                      // DONE: shaded in extend's tail, or a hole of the list; behind the list's end: stale
                parked[j] = ((keys >> (8 * j)) & 0xffu) != (unsigned)RTC_WF_KEY_DONE && i0 + j < n_list;
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(parked[j]);
                pos[j] = n_wave + (unsigned)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
                n_wave += (unsigned)__builtin_popcountll(mask);
            }
            if (lane == 0) wave_count[wave] = n_wave;
            __syncthreads();
            unsigned before = 0, total = 0;
            for (unsigned w = 0; w < N / 64; ++w) { const unsigned n = wave_count[w]; before += w < wave ? n : 0u; total += n; }
#pragma unroll
            for (unsigned j = 0; j < KPT; ++j)
                if (parked[j]) { const unsigned at = (q_head + q_n + before + pos[j]) & (QCAP - 1); q_index[at] = i0 + j; q_key[at] = (unsigned char)((keys >> (8 * j)) & 63u); }
            q_n += total;                                            // <= (N - 1) + 4 N < QCAP
        }
        while (q_n >= N || (flush && q_n > 0)) {
            const unsigned take = q_n < N ? q_n : N;
            if (tid < 64) key_count[tid] = 0;
            __syncthreads();                                         // (also: the queue entries written above are visible)
            const unsigned at = (q_head + tid) & (QCAP - 1);
            const unsigned key = tid < take ? q_key[at] : 0u;
            const unsigned index = tid < take ? q_index[at] : 0u;
            const unsigned rank = tid < take ? atomicAdd(&key_count[key], 1u) : 0u;
            __syncthreads();
            if (tid == 0) { unsigned acc = 0; for (int k = 0; k < 64; ++k) { key_start[k] = acc; acc += key_count[k]; } }
            __syncthreads();
            if (tid < take) sorted_index[key_start[key] + rank] = index | ((key & (unsigned)RTC_WF_KEY_ALBEDO) << 26);   // (list positions are < 2^25; bit 31 = albedo only)
            q_head = (q_head + take) & (QCAP - 1); q_n -= take;
            __syncthreads();
            bool died = false;
            if (tid < take) {
                const WfArgs &t = wf_args_again(c);                  // scene, camera and render parameters: read here, not held across the scan
                const unsigned i = sorted_index[tid] & 0x7fffffffu;
                const bool albedo_only = (sorted_index[tid] >> 31) != 0;
                const int slot = t.B.active[cur][i];
                const float4 o = t.B.ro[slot], d = t.B.rd[slot], tm = t.B.tm[slot];
                if (albedo_only) {                                   // RTC_SF_TEXTURED: extend's tail has shaded the hit with albedo 1; the new ray starts at the hit point, on the textured leaf
                    const SmallShade sh = t.sc.leaf_shade[__float_as_int(tm.w)];
                    const f3 p = f3mk(o.x, o.y, o.z);
                    float u, v; sphere_uv<float>((p - f3mk(sh.g[0], sh.g[1], sh.g[2])) * sh.g[3], u, v);   // hittable.c:381-386 on the outward normal (an image texture reads them)
                    f3 val;
                    if (noise_in_lds) { SceneT<float> sc = t.sc; sc.perlin_vec = noise_vec; sc.perlin_perm = noise_perm; val = texture_value<NOISE>(sc, __float_as_int(sh.param), u, v, p); }
                    else val = texture_value<NOISE>(t.sc, __float_as_int(sh.param), u, v, p);
                    t.B.rd[slot] = make_float4(d.x, d.y, d.z, d.w * val.x);
                    t.B.tm[slot] = make_float4(tm.x * val.y, tm.y * val.z, tm.z, tm.w);
                } else {
                int px, py, layer; slot_pixel(t.P, t.B, slot, px, py, layer);
                const float2 hh = t.B.hit[slot];
                SmallShade none; none.meta = 0;
                unsigned char no_key = 0;
                if (noise_in_lds) {                                  // (its own copy of the shading code: the compiler then knows the tables' address space and reads them with ds_read)
                    SceneT<float> sc = t.sc; sc.perlin_vec = noise_vec; sc.perlin_perm = noise_perm;
                    died = wf_shade_entry<NOISE, true>(sc, t.cam, t.P, t.B, slot, px, py, o, d, tm, hh.x, __float_as_int(hh.y), none, no_key);
                }
                else died = wf_shade_entry<NOISE, true>(t.sc, t.cam, t.P, t.B, slot, px, py, o, d, tm, hh.x, __float_as_int(hh.y), none, no_key);
                if (died) t.B.active[cur][i] = -1;                   // stays in the list as a hole until the next compaction
                }
            }
            { const unsigned long long dm = __builtin_amdgcn_ballot_w64(died); if (lane == 0) wave_report(B, blockIdx.x * (N / 64) + wave, 0ull, dm); }
            __syncthreads();                                         // sorted_index is free again
        }
        if (flush) break;
        __syncthreads();                                             // wave_count is free again
    }
}

// active[cur][0..n_list) -> active[cur^1][0..live): order kept inside a block's 4 K entries, one atomic per block.  Launched at fixed points of the
// bounce schedule; whether the list is worth squeezing -- 1 / B.compact_div of it holes -- is decided HERE, from the device's own counters,
// so the host never waits for a count: every block takes the same decision, and k_wf_flip, the next kernel in the stream, takes it
// once more and makes the new list current.
__device__ __forceinline__ bool wf_worth_compacting(const WfBuffers &B, unsigned live) {
    const unsigned n_list = B.counters[WF_NLIST];
    return live != 0 && live < n_list - n_list / (unsigned)B.compact_div;
}
// A workgroup squeezes kCompactK chunks of kCompactBlock consecutive entries (held in registers) and claims its output range with ONE atomic: a
// same-address RETURNED atomic costs ~11 ns and serialises, and with one per 1024 entries a compaction of 11 M entries waited 118 us for its
// 10,700 atomics -- harmless at a compaction every 8 rounds, 14 % of a render's kernel time at one every 2 (profiles/r03e/final/kernel_stats.csv).
// Small workgroups on purpose: this kernel runs beside the extend launches of the other parts, and a 1024-thread workgroup waits until one CU
// has 16 wave slots free at once (threads x chunks, one GPU run: 1024 x 16 / 256 x 32 / 256 x 64 / 128 x 32 / 64 x 64 -- 1M spheres 35.3 / 34.7 /
// 35.3 / 33.9 / 34.3 ms, book1 16.9 / 16.4 / 17.1 / 16.8 / 17.1, final 81.9 / 82.0 / 82.0 / 82.4 / 82.3: profiles/r03e/compaction_sweep.txt).
#ifndef RT_WF_COMPACT_BLOCK
#define RT_WF_COMPACT_BLOCK 128
#endif
#ifndef RT_WF_COMPACT_K
#define RT_WF_COMPACT_K 32
#endif
constexpr int kCompactK = RT_WF_COMPACT_K, kCompactBlock = RT_WF_COMPACT_BLOCK, kCompactWaves = kCompactBlock / 64;
__global__ void __launch_bounds__(kCompactBlock) k_wf_compact(WfBuffers B) {
    __shared__ unsigned counts[kCompactK * kCompactWaves], offs[kCompactK * kCompactWaves];   // live entries per (chunk, wave), and their exclusive prefix sums
    __shared__ unsigned block_base;
    if (!wf_worth_compacting(B, wf_live_now(B))) return;
    const unsigned n_list = B.counters[WF_NLIST]; const int cur = (int)B.counters[WF_CUR];
    const unsigned first = blockIdx.x * (unsigned)(kCompactK * kCompactBlock);
    if (first >= n_list) return;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    int slot[kCompactK]; unsigned pos[kCompactK];
#pragma unroll
    for (int k = 0; k < kCompactK; ++k) {
        const unsigned i = first + (unsigned)(k * kCompactBlock) + tid;
        slot[k] = i < n_list ? B.active[cur][i] : -1;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(slot[k] >= 0);
        pos[k] = (unsigned)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) counts[k * kCompactWaves + wave] = (unsigned)__builtin_popcountll(mask);
    }
    __syncthreads();
    for (unsigned t = tid; t < kCompactK * kCompactWaves; t += kCompactBlock) { unsigned acc = 0; for (unsigned u = 0; u < t; ++u) acc += counts[u]; offs[t] = acc; }
    __syncthreads();
    if (tid == 0) { const unsigned total = offs[kCompactK * kCompactWaves - 1] + counts[kCompactK * kCompactWaves - 1]; block_base = total ? atomicAdd(B.counters + WF_CURSOR, total) : 0u; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kCompactK; ++k)
        if (slot[k] >= 0) B.active[cur ^ 1][block_base + offs[k * kCompactWaves + wave] + pos[k]] = slot[k];
}
__global__ void __launch_bounds__(64) k_wf_flip(WfBuffers B) {      // one wave
    const unsigned live = wf_live_now(B);
    {   // unpack this stripe's word into the wide sums (nothing else of this part runs beside this kernel)
        unsigned long long *const st = wf_packed(B, threadIdx.x);
        const unsigned long long p = st[0];
        if (p) {
            B.stats[8 * (1 + threadIdx.x)] += p & ((1ull << kWfPackWaves) - 1);
            B.stats[8 * (1 + kWfStripes + threadIdx.x)] += (p >> kWfPackWaves) & ((1ull << (kWfPackDeaths - kWfPackWaves)) - 1);
            st[1] += p >> kWfPackDeaths; st[0] = 0;
        }
    }
    if (threadIdx.x != 0) return;
    if (wf_worth_compacting(B, live)) { B.counters[WF_NLIST] = B.counters[WF_CURSOR]; B.counters[WF_CUR] ^= 1u; }
    B.counters[WF_CURSOR] = 0;
    B.counters[WF_LIVE] = live;                                      // what the early exits of extend and shade read
}

__global__ void __launch_bounds__(256) k_wf_finish(RenderParams P, WfBuffers B, float4 *__restrict__ accum, const unsigned long long *__restrict__ stats,
                                                   unsigned long long *__restrict__ stats_out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;            // slot of layer 0
    if (q == 0) {                                                    // fold the striped statistics into the calling stream's counters
        unsigned long long rays = 0, waves = 0;
        for (int k = 0; k < kWfStripes; ++k) {
            const unsigned long long p = stats[8 * (1 + 2 * kWfStripes + k)];   // what the rounds behind the last compaction point reported
            const unsigned long long q = stats[8 * (1 + 2 * kWfStripes + k) + 2];   // the camera pass
            rays += stats[8 * (1 + k)] + (p & ((1ull << kWfPackWaves) - 1)) + (q & ((1ull << kWfPackPrimaryWaves) - 1));
            waves += stats[8 * (1 + kWfStripes + k)] + ((p >> kWfPackWaves) & ((1ull << (kWfPackDeaths - kWfPackWaves)) - 1)) + (q >> kWfPackPrimaryWaves);
        }
        stats_out[0] += rays; stats_out[1] += waves;
    }
    if (q >= B.slots_per_layer) return;
    const int tile = q >> 6, x = q & 7, y = (q >> 3) & 7;
    const int px = (tile % P.tiles_x) * 8 + x, py = (tile / P.tiles_x) * 8 + y;
    if (px >= P.width || py >= P.height) return;
    const size_t pixel = (size_t)py * P.width + px;
    float4 a = accum[pixel];
    for (int k = 0; k < B.layers; ++k) { const float4 s = B.sum[pixel_slot(B, tile, x, y, k)]; a.x += s.x; a.y += s.y; a.z += s.z; }
    a.w += (float)(P.s_end - P.s_begin);
    accum[pixel] = a;
}

}  // namespace

size_t wf_bytes_per_slot() { return 4 * sizeof(float4) + sizeof(float2) + 2 * sizeof(int) + 1; }
size_t wf_stats_bytes() { return (1 + 3 * kWfStripes) * 8 * sizeof(unsigned long long); }   // cache lines of 64 bytes

// Does the extend stage of this scene run from LDS tables (k_wf_extend_lds)?  No media, every node index and leaf reference in 16
// bits, and tables + 16-bit stacks of a workgroup's 768 lanes within half a CU's LDS.  RTC_WF_LDS_TABLES=0 keeps the plain kernel (A/B, tests).
bool wf_lds_tables(const SceneT<float> &sc, size_t *lds_bytes) {
    if (const char *e = std::getenv("RTC_WF_LDS_TABLES")) if (std::atoi(e) == 0) return false;
    if (sc.media_kind != 0 || sc.n_nodes <= 0 || sc.n_bleaves <= 0 || sc.n_nodes >= 32768 || sc.n_bleaves >= 4096) return false;   // ~(slot << 3 | count - 1) >= -32768
    const size_t bytes = ((size_t)sc.n_nodes + (size_t)sc.n_bleaves) * 64 + (size_t)sc.bvh_stack * kWfLdsBlock * sizeof(short);
    if (bytes + 64 > 80 * 1024) return false;
    if (lds_bytes) *lds_bytes = bytes;
    return true;
}

hipError_t launch_wf_init(const CameraT<float> &cam, const RenderParams &p, const WfBuffers &b, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(b.counters, 0, 16 * sizeof(unsigned), stream);
    if (e == hipSuccess) e = hipMemsetAsync(b.stats + 8, 0, 3 * kWfStripes * 8 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_wf_init, dim3((unsigned)((b.n_slots + 255) / 256)), dim3(256), 0, stream, cam, p, b);
    return hipGetLastError();
}
hipError_t launch_wf_bounce(const SceneT<float> &sc, const CameraT<float> &cam, const RenderParams &p, const WfBuffers &b, hipStream_t stream) {
    // the grid covers every slot: the list only shrinks, by how much is known on the device alone (counters[WF_NLIST]); workgroups
    // behind its end return at once
    const unsigned n_max = (unsigned)b.n_slots;
    if (n_max == 0) return hipSuccess;
    const unsigned blocks = (n_max + kWfExtendBlock - 1) / kWfExtendBlock;
    const size_t lds = (size_t)sc.bvh_stack * kWfExtendBlock * sizeof(int);
    size_t lds_tab = 0;
    const WfArgs args{sc, cam, p, b};
    static const unsigned n_cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return (unsigned)(n > 0 ? n : 256); }();
    if (wf_lds_tables(sc, &lds_tab)) {
        const unsigned lblocks = std::min((n_max + kWfLdsBlock - 1) / kWfLdsBlock, 2u * n_cus);   // persistent: two workgroups per CU
        if (sc.leaf_kinds == RTC_KIND_SPHERE) hipLaunchKernelGGL((k_wf_extend_lds<RTC_KIND_SPHERE>), dim3(lblocks), dim3(kWfLdsBlock), lds_tab, stream, args);
        else hipLaunchKernelGGL((k_wf_extend_lds<RTC_KIND_ALL>), dim3(lblocks), dim3(kWfLdsBlock), lds_tab, stream, args);
    }
    else if (sc.media_kind == 0 && sc.leaf_kinds == RTC_KIND_SPHERE) hipLaunchKernelGGL((k_wf_extend<0, RTC_KIND_SPHERE>), dim3(blocks), dim3(kWfExtendBlock), lds, stream, args);
    else if (sc.media_kind == 0) hipLaunchKernelGGL((k_wf_extend<0, RTC_KIND_ALL>), dim3(blocks), dim3(kWfExtendBlock), lds, stream, args);
    else if (sc.media_kind == 1) hipLaunchKernelGGL((k_wf_extend<1, RTC_KIND_ALL>), dim3(blocks), dim3(kWfExtendBlock), lds, stream, args);
    else hipLaunchKernelGGL((k_wf_extend<2, RTC_KIND_ALL>), dim3(blocks), dim3(kWfExtendBlock), lds, stream, args);
    if (sc.n_general == 0) return hipGetLastError();               // every hit was shaded in extend's tail
    static const unsigned shade_wgs = [] { const char *e = std::getenv("RTC_WF_SHADE_WGS"); const int v = e ? std::atoi(e) : 0; return (unsigned)(v > 0 ? v : 0); }();
    const bool noise = (sc.tex_kinds & (1 << RTC_TEX_NOISE)) != 0;
    const unsigned sblock = (unsigned)wf_shade_block(noise);
    const dim3 sgrid(std::min((n_max + sblock - 1) / sblock, shade_wgs ? shade_wgs : n_cus * (1024u / sblock)));   // persistent: 1024 threads per CU
    if (noise) hipLaunchKernelGGL(k_wf_shade<true>, sgrid, dim3(sblock), 0, stream, args);
    else hipLaunchKernelGGL(k_wf_shade<false>, sgrid, dim3(sblock), 0, stream, args);
    return hipGetLastError();
}
// the camera rays of the first b.phit_k samples of every slot (b.phit); behind launch_wf_init on the part's stream, ahead of its first bounce
hipError_t launch_wf_primary(const SceneT<float> &sc, const CameraT<float> &cam, const RenderParams &p, const WfBuffers &b, hipStream_t stream) {
    if (b.phit_k <= 0 || b.n_slots <= 0 || p.max_depth <= 0) return hipSuccess;
    const dim3 grid(((unsigned)b.n_slots + kWfExtendBlock - 1) / kWfExtendBlock, (unsigned)b.phit_k);
    const WfArgs args{sc, cam, p, b};
    bool wave = true;                                                // RTC_WF_PRIMARY_WAVE=0: per-lane traversal, as in extend (A/B, tests)
    if (const char *e = std::getenv("RTC_WF_PRIMARY_WAVE")) wave = std::atoi(e) != 0;
    const size_t lds = wave ? (size_t)sc.bvh_stack * sizeof(int) : (size_t)sc.bvh_stack * kWfExtendBlock * sizeof(int);
#define RT_WF_PRIMARY(M, K) do { if (wave) hipLaunchKernelGGL((k_wf_primary<M, K, true>), grid, dim3(kWfExtendBlock), lds, stream, args); \
                                 else hipLaunchKernelGGL((k_wf_primary<M, K, false>), grid, dim3(kWfExtendBlock), lds, stream, args); } while (0)
    if (sc.media_kind == 0 && sc.leaf_kinds == RTC_KIND_SPHERE) RT_WF_PRIMARY(0, RTC_KIND_SPHERE);
    else if (sc.media_kind == 0) RT_WF_PRIMARY(0, RTC_KIND_ALL);
    else if (sc.media_kind == 1) RT_WF_PRIMARY(1, RTC_KIND_ALL);
    else RT_WF_PRIMARY(2, RTC_KIND_ALL);
#undef RT_WF_PRIMARY
    return hipGetLastError();
}
hipError_t launch_wf_compact(const WfBuffers &b, hipStream_t stream) {
    hipLaunchKernelGGL(k_wf_compact, dim3(((unsigned)b.n_slots + kCompactK * kCompactBlock - 1) / (kCompactK * kCompactBlock)), dim3(kCompactBlock), 0, stream, b);
    hipLaunchKernelGGL(k_wf_flip, dim3(1), dim3(64), 0, stream, b);
    return hipGetLastError();
}
hipError_t launch_wf_finish(const RenderParams &p, const WfBuffers &b, float4 *accum, hipStream_t stream) {
    hipLaunchKernelGGL(k_wf_finish, dim3((unsigned)((b.slots_per_layer + 255) / 256)), dim3(256), 0, stream, p, b, accum, b.stats, b.stats_out);
    return hipGetLastError();
}
