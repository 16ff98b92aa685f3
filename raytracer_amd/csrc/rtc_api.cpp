// rtc_api.cpp -- the C-ABI of include/rtc.h (product code; host side, compiled by hipcc).
// Owns device memory for the flattened scene, launches the kernels of kernels_f64.hip / kernels_f32.hip.
// There is deliberately no CPU path here: without a usable HIP device every compute entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <limits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
// hip_runtime.h declares a class template named `texture` (legacy texture references); the host API's opaque
// `struct texture` (reference name, texture.h:20) is renamed inside this translation unit only.
#define texture rtc_host_texture
#include "host/flatten.h"
#include "host/scene_io.h"
#include "rtc.h"
#include "rtc_test.h"

namespace {

thread_local std::string g_error;
thread_local const char *g_error_literal = nullptr;   // set by fail_literal: an error path that must not touch the heap
// every error goes through fail / fail_scene / fail_literal: the first two clear the literal, so that a refusal of the allocation-free
// entry point never outlives a later error on the same thread
int fail(int code, const std::string &msg) { g_error = msg; g_error_literal = nullptr; return code; }

int fail_literal(int code, const char *msg) { g_error_literal = msg; return code; }
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(RTC_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t upload(const void *src, size_t n) {
        if (p) { (void)hipFree(p); p = nullptr; }
        bytes = n;
        if (n == 0) return hipSuccess;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) return e;
        return hipMemcpy(p, src, n, hipMemcpyHostToDevice);
    }
};

template <typename R> struct DeviceTables {
    DevBuf prims, bprims, ops, chains, materials, textures, lights, perlin_vec, perlin_perm, images, image_data, nodes, leaf_prims, bleaves, leaf_shade, ref_steps, ref_boxes;
    SceneT<R> view;
    int pre_slot = -1, pre_root = 0;   // SceneT::pre_slot and the root's other child, where the traversals then start
    int root_child0 = 0;   // child0 of node 0 as uploaded (a one-leaf tree starts its traversals at that run, and the device copy may have packed it)
    bool ready = false;
};

template <typename R, typename Src> std::vector<R> convert_vec(const std::vector<Src> &s) { std::vector<R> d(s.size()); for (size_t i = 0; i < s.size(); i++) d[i] = (R)s[i]; return d; }
template <typename R> PrimT<R> conv(const PrimT<double> &s) { PrimT<R> d; std::memset(&d, 0, sizeof d); d.type = s.type; d.axis = s.axis; d.mat = s.mat; d.chain = s.chain; d.bnd_first = s.bnd_first; d.bnd_count = s.bnd_count; for (int i = 0; i < 12; i++) d.d[i] = (R)s.d[i]; return d; }
template <typename R> OpT<R> conv(const OpT<double> &s) { OpT<R> d; d.type = s.type; d.axis = s.axis; d.a = (R)s.a; d.b = (R)s.b; d.c = (R)s.c; return d; }
template <typename R> MaterialT<R> conv(const MaterialT<double> &s) { MaterialT<R> d; d.type = s.type; d.tex = s.tex; for (int i = 0; i < 3; i++) d.albedo[i] = (R)s.albedo[i]; d.param = (R)s.param; return d; }
template <typename R> TextureT<R> conv(const TextureT<double> &s) { TextureT<R> d; d.type = s.type; d.a = s.a; d.b = s.b; d.pad = 0; for (int i = 0; i < 4; i++) d.v[i] = (R)s.v[i]; return d; }
template <typename R> LightT<R> conv(const LightT<double> &s) { LightT<R> d; d.kind = s.kind; d.axis = s.axis; d.pad0 = d.pad1 = 0; for (int i = 0; i < 8; i++) d.d[i] = (R)s.d[i]; return d; }

template <typename R> hipError_t upload_tables(const FlatScene &fs, DeviceTables<R> &dt, bool with_ref_steps) {
    hipError_t e;
#define UP(buf, vec) do { e = dt.buf.upload((vec).data(), (vec).size() * sizeof((vec)[0])); if (e != hipSuccess) return e; } while (0)
    std::vector<PrimT<R>> prims, bprims; for (const auto &x : fs.prims) prims.push_back(conv<R>(x)); for (const auto &x : fs.bprims) bprims.push_back(conv<R>(x));
    std::vector<OpT<R>> ops; for (const auto &x : fs.ops) ops.push_back(conv<R>(x));
    std::vector<MaterialT<R>> mats; for (const auto &x : fs.materials) mats.push_back(conv<R>(x));
    std::vector<TextureT<R>> texs; for (const auto &x : fs.textures) texs.push_back(conv<R>(x));
    std::vector<LightT<R>> lights; for (const auto &x : fs.lights) lights.push_back(conv<R>(x));
    std::vector<R> pvec = convert_vec<R>(fs.perlin_vec);
    UP(prims, prims); UP(bprims, bprims); UP(ops, ops); UP(chains, fs.chains); UP(materials, mats); UP(textures, texs); UP(lights, lights);
    UP(perlin_vec, pvec); UP(perlin_perm, fs.perlin_perm); UP(images, fs.images); UP(image_data, fs.image_data); UP(leaf_prims, fs.leaf_prims);
    if (with_ref_steps) { UP(ref_steps, fs.ref_steps); UP(ref_boxes, fs.ref_boxes); }
    else {
        {   // lo / hi -> centre / half-extent (BvhNodeCH), rounded so that [c - h, c + h] still encloses the builder's padded box
            std::vector<BvhNodeCH> ch(fs.nodes.size());
            for (size_t i = 0; i < fs.nodes.size(); i++) {
                const BvhNode &n = fs.nodes[i]; BvhNodeCH &o = ch[i];
                o.child0 = n.child0; o.count0 = n.count0; o.child1 = n.child1; o.count1 = n.count1;
                auto conv1 = [](float lo, float hi, float &c, float &h) {
                    if (!(lo <= hi)) { c = 0.0f; h = -1.0f; return; }                      // the empty child of a one-leaf tree: never entered
                    c = (float)(0.5 * ((double)lo + (double)hi));
                    const double need = std::max((double)hi - (double)c, (double)c - (double)lo);
                    h = (float)need; if ((double)h < need) h = std::nextafter(h, std::numeric_limits<float>::infinity());
                };
                for (int a = 0; a < 3; a++) { conv1(n.lo0[a], n.hi0[a], o.c0[a], o.h0[a]); conv1(n.lo1[a], n.hi1[a], o.c1[a], o.h1[a]); }
            }
            // Leaf runs of two plain spheres become ONE record on the device (RTC_LEAF_PAIR: the second sphere rides in d[4..7] of the first, the run's
            // reference shrinks to one record): one fetch and one trip of the traversal instead of two.  The host tables -- scene files, dumps, the
            // validator -- keep one record per slot; slot numbers, leaf_prims and leaf_shade are untouched.  RTC_LEAF_PAIRS=0: off (A/B, tests).
            std::vector<BLeaf> bl = fs.bleaves;
            bool pairs = true;
            if (const char *e = std::getenv("RTC_LEAF_PAIRS")) pairs = std::atoi(e) != 0;
            auto plain_sphere = [&](int s) { return s >= 0 && s < (int)bl.size() && (bl[(size_t)s].meta & 3u) == RTC_PRIM_SPHERE && !(bl[(size_t)s].meta & (RTC_LEAF_BOX | RTC_LEAF_PAIR)); };
            auto pack = [&](int &child, int &count) {
                if (!pairs || count != 2 || child >= 0) return;
                const int first = (~child) >> 3;
                if (((~child) & 7) != 1 || !plain_sphere(first) || !plain_sphere(first + 1)) return;
                BLeaf &a = bl[(size_t)first]; const BLeaf &b = fs.bleaves[(size_t)first + 1];
                for (int k = 0; k < 4; k++) a.d[4 + k] = b.d[k];
                a.meta |= RTC_LEAF_PAIR;
                child = rtc_leaf_ref(first, 1); count = 1;
            };
            for (BvhNodeCH &o : ch) { pack(o.child0, o.count0); pack(o.child1, o.count1); }
            UP(nodes, ch);
            UP(bleaves, bl);
            dt.root_child0 = ch.empty() ? 0 : ch[0].child0;
            // A one-record leaf run that hangs off the ROOT: flatten.cpp puts an item there whose box is (nearly) the whole scene's -- the global fog of the
            // Book-2 final scene (r = 5000 around everything), the ground sphere of the Book-1 scenes -- so nearly every ray meets it.  The traversals test
            // that record ahead of their loop and start at the root's other child (SceneT::pre_slot / bvh_root): one node step and one trip fewer per ray, a
            // tmax from the first instruction on, and node 0 is never visited.  Device view only; RTC_PRE_LEAF=0: off (A/B, tests).
            dt.pre_slot = -1; dt.pre_root = 0;
            bool pre = true;
            if (const char *e = std::getenv("RTC_PRE_LEAF")) pre = std::atoi(e) != 0;
            if (pre && ch.size() > 1) {
                const BvhNodeCH &r0 = ch[0];
                for (int side = 0; side < 2 && dt.pre_slot < 0; side++) {
                    const int child = side ? r0.child1 : r0.child0, other = side ? r0.child0 : r0.child1;
                    const float *oh = side ? r0.h0 : r0.h1;
                    if (child >= 0 || child == RTC_BVH_EMPTY || other == RTC_BVH_EMPTY || ((~child) & 7) != 0) continue;   // a run of exactly one record
                    const int slot = (~child) >> 3;
                    if (slot < 0 || slot >= (int)bl.size()) continue;
                    if (!(oh[0] >= 0.0f) || (bl[(size_t)slot].meta & RTC_LEAF_BOX)) continue;   // (a box record owns six slots: left in the tree)
                    dt.pre_slot = slot; dt.pre_root = other;
                }
            }
        }
        UP(leaf_shade, fs.leaf_shade);
    }
#undef UP
    SceneT<R> &v = dt.view;
    v.prims = (const PrimT<R> *)dt.prims.p; v.n_prims = (int)fs.prims.size();
    v.bprims = (const PrimT<R> *)dt.bprims.p; v.n_bprims = (int)fs.bprims.size();
    v.ops = (const OpT<R> *)dt.ops.p; v.chains = (const ChainRange *)dt.chains.p; v.n_chains = (int)fs.chains.size();
    v.materials = (const MaterialT<R> *)dt.materials.p; v.n_materials = (int)fs.materials.size();
    v.textures = (const TextureT<R> *)dt.textures.p; v.n_textures = (int)fs.textures.size();
    v.lights = (const LightT<R> *)dt.lights.p; v.n_lights = (int)fs.lights.size();
    v.perlin_vec = (const R *)dt.perlin_vec.p; v.perlin_perm = (const int *)dt.perlin_perm.p;
    v.images = (const ImageDesc *)dt.images.p; v.image_data = (const unsigned char *)dt.image_data.p;
    v.nodes = (const BvhNodeCH *)dt.nodes.p; v.leaf_prims = (const int *)dt.leaf_prims.p; v.n_nodes = (int)fs.nodes.size();
    v.bleaves = (const BLeaf *)dt.bleaves.p; v.leaf_shade = (const SmallShade *)dt.leaf_shade.p;
    v.ref_steps = (const RefStep *)dt.ref_steps.p; v.ref_boxes = (const double *)dt.ref_boxes.p;
    v.n_ref_steps = with_ref_steps ? (int)fs.ref_steps.size() : 0;
    v.bvh_root = (fs.nodes.size() == 1 && fs.nodes[0].child1 == RTC_BVH_EMPTY) ? (with_ref_steps ? fs.nodes[0].child0 : dt.root_child0) : 0;
    v.pre_slot = -1;
    if (!with_ref_steps && dt.pre_slot >= 0) { v.pre_slot = dt.pre_slot; v.bvh_root = dt.pre_root; }
    v.bvh_stack = fs.bvh_depth < 1 ? 1 : fs.bvh_depth;   // one entry per level; flatten / load refuse trees deeper than RTC_BVH_MAX_DEPTH: nothing is clamped here
    v.n_media = fs.n_media;
    v.n_bleaves = (int)fs.bleaves.size();
    v.n_general = 0; for (const SmallShade &sh : fs.leaf_shade) v.n_general += (sh.meta & (RTC_SF_GENERAL | RTC_SF_TEXTURED)) != 0;
    v.leaf_kinds = fs.leaf_kinds;
    v.tex_kinds = 0; for (const auto &t : fs.textures) v.tex_kinds |= 1 << t.type;
    v.n_perlin = (int)(fs.perlin_perm.size() / 768);
    v.media_kind = fs.n_media ? (fs.nodes.empty() ? 2 : fs.media_kind) : 0;
    for (int i = 0; i < 3; i++) v.background[i] = (R)fs.background[i];
    dt.ready = true;
    return hipSuccess;
}

template <typename R> CameraT<R> camera_tables(const camera *c) {
    double f[24]; camera_fields(c, f);
    CameraT<R> d;
    R *dst[7] = {d.origin, d.llc, d.horizontal, d.vertical, d.u, d.v, d.w};
    for (int i = 0; i < 7; i++) for (int k = 0; k < 3; k++) dst[i][k] = (R)f[3 * i + k];
    d.lens_radius = (R)f[21]; d.time0 = (R)f[22]; d.time1 = (R)f[23];
    return d;
}

struct StreamState {
    hipEvent_t ev0 = nullptr, ev1 = nullptr; uint64_t samples = 0; int launches = 0; int kernel = 0; bool pending = false; double total_ms = 0;
    // scratch of the small-scene kernel, per stream: renders of one scene enqueued on different streams may overlap on the GPU
    DevBuf tile_scratch; size_t tile_scratch_cap = 0;   // cost + order of the 8x8 tiles
    DevBuf small_partial; size_t small_partial_cap = 0; // per-chunk partial sums (float4 per pixel and chunk)
    std::vector<unsigned char> order_key;               // what the tile order in tile_scratch was probed for (camera, size, depth, integrator)
    // [0] rays, [1] wave bounce-iterations of the last render on THIS stream.  One buffer per stream: renders of one scene on
    // two streams may overlap, and with a buffer per scene they added into -- or wiped -- each other's counts.
    DevBuf stats;
};

}  // namespace

// ---- scratch of the allocation-free entry point (rtc_render_device_noalloc) --------------------------------------------
// Device memory the small-scene render needs beyond its inputs and outputs -- the ray / iteration counters, the cost and
// order of the 8x8 tiles, the per-chunk partial sums -- as ONE file-scope device array instead of hipMalloc'ed buffers:
// 256 B + 1 MiB + 16 MiB.  Every GPU the library runs on holds its own copy, at its own address, until the process ends.
// In host code the array's NAME is only a handle for hipGetSymbolAddress (below); the address it yields, kept per device in
// g_noalloc_ptr, is what kernels, memsets and copies are given.
enum : size_t { RTC_NOALLOC_MAX_TILES = 1u << 17,          // 8x8 tiles: up to 3840 x 2160 (129,600 tiles)
                RTC_NOALLOC_MAX_PARTIAL = 1u << 20,        // float4 partial sums: W * H * chunks, e.g. 1024 x 512 x 2
                RTC_NOALLOC_COUNTER_BYTES = 256,
                RTC_NOALLOC_BYTES = RTC_NOALLOC_COUNTER_BYTES + 2 * RTC_NOALLOC_MAX_TILES * sizeof(int) + RTC_NOALLOC_MAX_PARTIAL * 16 };
static __device__ __align__(256) unsigned char scratch[RTC_NOALLOC_BYTES];
namespace {
enum { RTC_MAX_DEVICES = 64 };
void *g_noalloc_ptr[RTC_MAX_DEVICES];      // device address of `scratch` on device d (nullptr: not asked yet)
std::mutex g_noalloc_mu[RTC_MAX_DEVICES];  // serialises the ENQUEUE of calls on one device; overlapping execution on two streams still races
std::mutex g_noalloc_init_mu;
// once per device, with that device current (rtc_init, or the entry point's first call on the device)
int noalloc_scratch(void **out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= RTC_MAX_DEVICES) return fail_literal(RTC_ERR_DEVICE, "rtc_render_device_noalloc: no current HIP device (or its ordinal is >= 64)");
    std::lock_guard<std::mutex> lk(g_noalloc_init_mu);
    if (!g_noalloc_ptr[dev]) {
        void *ptr = nullptr;
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr) { (void)hipGetLastError(); return fail_literal(RTC_ERR_DEVICE, "rtc_render_device_noalloc: hipGetSymbolAddress(scratch) failed on this device"); }
        g_noalloc_ptr[dev] = ptr;
    }
    *out = g_noalloc_ptr[dev];
    return RTC_OK;
}
}  // namespace

struct rtc_scene {
    FlatScene flat;
    int device = -1;
    DeviceTables<float> f32;
    DeviceTables<double> f64;
    DevBuf ray_counter;                        // wavefront only: striped statistics + queue counters, ordered across streams by wf_done like the pool
    DevBuf wf_pool; size_t wf_pool_bytes = 0;   // wavefront path-slot state + camera-ray hits, grown on demand
    hipEvent_t wf_done = nullptr;              // end of the last wavefront render's final accumulate: the pool is shared by all streams
    hipStream_t wf_stream[4] = {nullptr, nullptr, nullptr, nullptr};   // streams of the parts a wavefront render is split into (part 0 runs on the caller's)
    hipEvent_t wf_fork = nullptr, wf_part_done[4] = {nullptr, nullptr, nullptr, nullptr};
    DevBuf small_prog, small_shade, small_lights;            // the small-scene pack as the kernel reads it: record stream + shading records
    SmallScene small{};   // n_leaves == 0: not eligible
    std::map<void *, StreamState> streams;
    std::mutex mu;
    ~rtc_scene() { if (wf_done) (void)hipEventDestroy(wf_done); if (wf_fork) (void)hipEventDestroy(wf_fork); for (int k = 0; k < 4; k++) { if (wf_part_done[k]) (void)hipEventDestroy(wf_part_done[k]); if (wf_stream[k]) (void)hipStreamDestroy(wf_stream[k]); } for (auto &kv : streams) { if (kv.second.ev0) (void)hipEventDestroy(kv.second.ev0); if (kv.second.ev1) (void)hipEventDestroy(kv.second.ev1); } }
};

static rtc_scene *fail_scene(rtc_scene *doomed, const std::string &msg) { g_error = msg; g_error_literal = nullptr; delete doomed; return nullptr; }

extern "C" {

const char *rtc_last_error(void) { return g_error_literal ? g_error_literal : g_error.c_str(); }

int rtc_init(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(RTC_ERR_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    int dev = 0; HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop; HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(RTC_ERR_DEVICE, std::string("kernels are built for gfx950, device is ") + prop.gcnArchName);
    void *scratch_ptr = nullptr;               // the allocation-free entry point's scratch: resolve its address on this device now
    return noalloc_scratch(&scratch_ptr);
}
int rtc_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
const char *rtc_device_name(void) {
    static thread_local char name[256];
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return "";
    std::snprintf(name, sizeof name, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return name;
}

// The small-scene pack as the record stream k_pathtrace_small reads (rtc_tables.h SmallRec): world-space records first (the world
// group's box and pairs, the spheres, and every y-rotated space that holds one box and nothing else, folded into one self-contained
// record), then the other object spaces, each led by its transform.  Boxes and rects go to centre / half-extent form, the half extent
// rounded outward.
static std::vector<SmallRec> build_small_program(const FlatScene &fs, int counts[6]) {
    auto fbits = [](float f) { uint32_t u; std::memcpy(&u, &f, sizeof u); return u; };
    auto centre = [](float lo, float hi) { return (float)(0.5 * ((double)lo + (double)hi)); };
    auto half = [](float lo, float hi) { const float h = (float)(0.5 * ((double)hi - (double)lo)); return (double)h < 0.5 * ((double)hi - (double)lo) ? std::nextafter(h, HUGE_VALF) : h; };
    std::vector<SmallRec> boxes, xboxes, pairs[3], spheres, general;
    auto box_rec = [&](const SmallBox &B, uint32_t type) {
        SmallRec r{}; r.w[0] = type;
        auto id = [&](uint32_t ids, int e) { return (ids >> (8 * e)) & 0xffu; };
        r.w[1] = id(B.ids_lo, 0) | id(B.ids_hi, 0) << 8 | id(B.ids_lo, 1) << 16 | id(B.ids_hi, 1) << 24;
        r.w[2] = id(B.ids_lo, 2) | id(B.ids_hi, 2) << 8;
        for (int e = 0; e < 3; e++) { r.w[3 + e] = fbits(centre(B.lo[e], B.hi[e])); r.w[6 + e] = fbits(half(B.lo[e], B.hi[e])); }
        return r;
    };
    auto pair_recs = [&](const SmallGroup &g, std::vector<SmallRec> *by_axis, std::vector<SmallRec> *all) {
        const int first[3] = {g.first, g.first + g.n_xy, g.first + g.n_xy + g.n_xz}, count[3] = {g.n_xy, g.n_xz, g.n_yz};
        for (int axis = 0; axis < 3; axis++)                            // RTC_AXIS_XY, _XZ, _YZ
            for (int p = first[axis]; p < first[axis] + count[axis]; p++) {
                const SmallPair &P = fs.small_pairs[(size_t)p];
                int id0, id1; std::memcpy(&id0, &P.f[10], sizeof id0); std::memcpy(&id1, &P.f[11], sizeof id1);
                SmallRec r{};
                r.w[0] = RTC_SR_PAIR | (uint32_t)axis << 8 | (id1 >= 0 ? 1u << 10 : 0u) | (P.f[12] != 0.0f ? 1u << 11 : 0u) | (id1 >= 0 && P.f[13] != 0.0f ? 1u << 12 : 0u)
                       | ((uint32_t)id0 & 0xffu) << 16 | ((uint32_t)(id1 >= 0 ? id1 : 0xff) & 0xffu) << 24;
                r.w[1] = fbits(P.f[0]); r.w[2] = fbits(P.f[1]);
                for (int k = 0; k < 2; k++) {
                    if (k == 1 && id1 < 0) break;
                    const float *b = &P.f[2 + 4 * k];                   // lo_a lo_b hi_a hi_b
                    r.w[3 + 4 * k] = fbits(centre(b[0], b[2])); r.w[4 + 4 * k] = fbits(centre(b[1], b[3]));
                    r.w[5 + 4 * k] = fbits(half(b[0], b[2])); r.w[6 + 4 * k] = fbits(half(b[1], b[3]));
                    r.w[11 + k] = fbits(P.f[12 + k]);
                }
                (by_axis ? by_axis[axis] : *all).push_back(r);
            }
    };
    for (const SmallGroup &g : fs.small_groups) {
        const SmallXform &X = fs.small_xforms[(size_t)g.xform];
        const int n_pairs = g.n_xy + g.n_xz + g.n_yz;
        if (g.xform == 0) {                                             // the world's own space (identity)
            if (g.box >= 0) boxes.push_back(box_rec(fs.small_boxes[(size_t)g.box], RTC_SR_BOX));
            pair_recs(g, pairs, nullptr);
        } else if (g.rot_y && g.box >= 0 && n_pairs == 0) {
            SmallRec r = box_rec(fs.small_boxes[(size_t)g.box], RTC_SR_XBOX_Y);
            r.w[9] = fbits(X.m[0]); r.w[10] = fbits(X.m[2]); r.w[11] = fbits(X.m[6]); r.w[12] = fbits(X.m[8]);
            for (int k = 0; k < 3; k++) r.w[13 + k] = fbits(X.b[k]);
            xboxes.push_back(r);
        } else {
            SmallRec r{}; r.w[0] = RTC_SR_XFORM | (g.rot_y ? 1u << 8 : 0u);
            for (int k = 0; k < 9; k++) r.w[1 + k] = fbits(X.m[k]);
            for (int k = 0; k < 3; k++) r.w[10 + k] = fbits(X.b[k]);
            general.push_back(r);
            if (g.box >= 0) general.push_back(box_rec(fs.small_boxes[(size_t)g.box], RTC_SR_BOX));
            pair_recs(g, nullptr, &general);
        }
    }
    for (size_t i = 0; i < fs.small_spheres.size(); i++) {
        const SmallSphere &sp = fs.small_spheres[i];
        SmallRec r{}; r.w[0] = RTC_SR_SPHERE | ((uint32_t)(fs.small_n_rects + (int)i) & 0xffu) << 8 | (sp.pdf_k != 0.0f ? 1u << 16 : 0u);
        for (int k = 0; k < 3; k++) r.w[1 + k] = fbits(sp.c[k]);
        r.w[4] = fbits(sp.r2); r.w[5] = fbits(sp.pdf_k);
        spheres.push_back(r);
    }
    std::vector<SmallRec> prog;
    const std::vector<SmallRec> *parts[7] = {&boxes, &xboxes, &pairs[0], &pairs[1], &pairs[2], &spheres, &general};
    for (int k = 0; k < 7; k++) { if (k < 6) counts[k] = (int)parts[k]->size(); prog.insert(prog.end(), parts[k]->begin(), parts[k]->end()); }
    return prog;
}

// the light list as the small kernel samples it (rtc_tables.h): same (r1, r2) -> point map as rt_shade.hpp light_random_one
static std::vector<SmallRec> build_small_lights(const FlatScene &fs) {
    auto fbits = [](float f) { uint32_t u; std::memcpy(&u, &f, sizeof u); return u; };
    std::vector<SmallRec> out;
    for (const LightT<double> &l : fs.lights) {
        SmallRec r{}; r.w[0] = (uint32_t)l.kind;
        if (l.kind == RTC_LIGHT_RECT) {
            // in-plane axes a, b and the normal axis n as x / y / z indices: XY -> (0, 1, 2), XZ -> (0, 2, 1), YZ -> (1, 2, 0)
            const int ia = l.axis == RTC_AXIS_YZ ? 1 : 0, ib = l.axis == RTC_AXIS_XY ? 1 : 2, in = l.axis == RTC_AXIS_XY ? 2 : (l.axis == RTC_AXIS_XZ ? 1 : 0);
            const float a0 = (float)l.d[2 * ia], a1 = (float)l.d[2 * ia + 1], b0 = (float)l.d[2 * ib], b1 = (float)l.d[2 * ib + 1];
            float corner[3] = {0, 0, 0}, ea[3] = {0, 0, 0}, eb[3] = {0, 0, 0};
            corner[ia] = a0; corner[ib] = b0; corner[in] = (float)l.d[6]; ea[ia] = a1 - a0; eb[ib] = b1 - b0;
            for (int k = 0; k < 3; k++) { r.w[1 + k] = fbits(corner[k]); r.w[4 + k] = fbits(ea[k]); r.w[7 + k] = fbits(eb[k]); }
        } else if (l.kind == RTC_LIGHT_SPHERE) {
            for (int k = 0; k < 3; k++) r.w[1 + k] = fbits((float)l.d[k]);
            r.w[4] = fbits((float)l.d[4]);
        }
        out.push_back(r);
    }
    return out;
}

// device tables of a flattened scene on the current device (rtc_scene_build and rtc_scene_load)
static rtc_scene *upload_scene(rtc_scene *s, const char *who) {
    if (hipGetDevice(&s->device) != hipSuccess) return fail_scene(s, std::string(who) + ": no current HIP device");
    hipError_t e = upload_tables<float>(s->flat, s->f32, false);
    std::vector<unsigned char> zero(4 * (wf_stats_bytes() + 64), 0);   // per part of a wavefront render: striped statistics, 16 control words (kernels.h WF_*)
    if (e == hipSuccess) e = s->ray_counter.upload(zero.data(), zero.size());
    if (e == hipSuccess) {
        const std::vector<SmallRec> lights = build_small_lights(s->flat);
        e = s->small_lights.upload(lights.data(), lights.size() * sizeof(SmallRec));
        s->small.lights = (const SmallRec *)s->small_lights.p; s->small.n_lights = (int)lights.size();
    }
    if (e == hipSuccess && s->flat.small_ok) {
        const FlatScene &fs = s->flat;
        int counts[6];
        const std::vector<SmallRec> prog = build_small_program(fs, counts);
        e = s->small_prog.upload(prog.data(), prog.size() * sizeof(SmallRec));
        // the kernel's copy of the shading records: g = (n, 0) for a rect, (-c / r, 1 / r) for a sphere -- n_out = p * g3 + g (kernels_small.hip)
        std::vector<SmallShade> shade = fs.small_shade;
        for (SmallShade &sh : shade) {
            if (sh.meta & RTC_SF_SPHERE) for (int k = 0; k < 3; k++) sh.g[k] = -sh.g[k] * sh.g[3];
            else sh.g[3] = 0.0f;
        }
        if (e == hipSuccess) e = s->small_shade.upload(shade.data(), shade.size() * sizeof(SmallShade));
        SmallScene &m = s->small;
        m.prog = (const SmallRec *)s->small_prog.p; m.shade = (const SmallShade *)s->small_shade.p;
        m.n_box = counts[0]; m.n_xbox = counts[1]; m.n_xy = counts[2]; m.n_xz = counts[3]; m.n_yz = counts[4]; m.n_sphere = counts[5]; m.n_recs = (int)prog.size();
        m.n_leaves = (int)fs.small_shade.size(); m.any_fuzz = fs.small_any_fuzz;
        m.fused_lights = fs.small_fused;
    }
    if (e != hipSuccess) return fail_scene(s, std::string(who) + ": upload failed: " + hipGetErrorString(e));
    // the allocation-free entry point's scratch: resolve its address on THIS device now (hipGetSymbolAddress may load the code object,
    // which allocates and is not legal during stream capture) -- a scene upload on a device is that entry point's warm-up there
    void *scratch_ptr = nullptr;
    if (noalloc_scratch(&scratch_ptr) != RTC_OK) { const std::string why = rtc_last_error(); return fail_scene(s, std::string(who) + ": " + why); }
    return s;
}

// Scenes of more leaves than this are traversed through the BVH, smaller ones by the brute-force / small-scene kernels.
// RTC_BVH_THRESHOLD is a development knob (tests build a BVH over one or two primitives with it), like RTC_BUILD_THREADS.
static int bvh_threshold() { const char *e = std::getenv("RTC_BVH_THRESHOLD"); return e ? std::max(0, std::atoi(e)) : 64; }

rtc_scene *rtc_scene_build(const hittable_list *world, const hittable_list *lights, const color *background, double time0, double time1) {
    rtc_scene *s = new rtc_scene();
    // brute force up to 64 leaves: the uniform loop runs out of SGPR-fed scalar loads; above that the linear BVH wins
    if (!flatten_scene(world, lights, background, time0, time1, bvh_threshold(), s->flat)) return fail_scene(s, "flatten: " + s->flat.error);
    return upload_scene(s, "rtc_scene_build");
}
rtc_scene *rtc_scene_flatten_only(const hittable_list *world, const hittable_list *lights, const color *background, double time0, double time1) {
    rtc_scene *s = new rtc_scene();
    if (!flatten_scene(world, lights, background, time0, time1, bvh_threshold(), s->flat)) return fail_scene(s, "flatten: " + s->flat.error);
    return s;   // device == -1: every compute entry point refuses this handle
}
int rtc_scene_save(const rtc_scene *scene, const char *path) {
    if (!scene || !path) return fail(RTC_ERR_ARG, "rtc_scene_save: null argument");
    std::string err;
    if (!save_flat_scene(scene->flat, path, err)) return fail(RTC_ERR_ARG, "rtc_scene_save: " + err);
    return RTC_OK;
}
rtc_scene *rtc_scene_load_host(const char *path) {
    if (!path) return fail_scene(nullptr, "rtc_scene_load: null path");
    rtc_scene *s = new rtc_scene();
    std::string err;
    if (!load_flat_scene(path, s->flat, err)) return fail_scene(s, "rtc_scene_load: " + err);
    return s;   // device == -1, like rtc_scene_flatten_only
}
rtc_scene *rtc_scene_load(const char *path) {
    rtc_scene *s = rtc_scene_load_host(path);
    return s ? upload_scene(s, "rtc_scene_load") : nullptr;
}
static void multi_forget(const rtc_scene *s);
void rtc_scene_free(rtc_scene *scene) { if (scene) multi_forget(scene); delete scene; }
int rtc_scene_num_prims(const rtc_scene *scene) { return scene ? (int)scene->flat.prims.size() : RTC_ERR_ARG; }

int rtc_scene_dump_leaves(const rtc_scene *scene, double *out16, int cap) {
    if (!scene || !out16) return fail(RTC_ERR_ARG, "rtc_scene_dump_leaves: null argument");
    const FlatScene &fs = scene->flat;
    static const int kind_of[4] = {1 /*HITTABLE_SPHERE*/, 2 /*MOVING_SPHERE*/, 3 /*RECT*/, 7 /*CONSTANT_MEDIUM*/};
    int n = 0;
    for (size_t i = 0; i < fs.prims.size(); i++, n++) {
        if (n >= cap) continue;
        const PrimT<double> &p = fs.prims[i];
        double *d = out16 + 16 * (size_t)n; for (int k = 0; k < 16; k++) d[k] = 0;
        d[0] = kind_of[p.type]; d[1] = fs.materials[p.mat].type;
        if (p.type == RTC_PRIM_SPHERE) for (int k = 0; k < 4; k++) d[2 + k] = p.d[k];
        else if (p.type == RTC_PRIM_MSPHERE) for (int k = 0; k < 9; k++) d[2 + k] = p.d[k];
        else if (p.type == RTC_PRIM_RECT) { for (int k = 0; k < 7; k++) d[2 + k] = p.d[k]; d[9] = p.axis; }
        else d[2] = p.d[0];
        d[12] = fs.wrap_depth[i]; d[13] = fs.flipped[i];
    }
    return n;
}

int rtc_scene_bvh_info(const rtc_scene *scene, long long *out4) {
    if (!scene || !out4) return fail(RTC_ERR_ARG, "rtc_scene_bvh_info: null argument");
    const FlatScene &fs = scene->flat;
    out4[0] = (long long)fs.nodes.size(); out4[1] = (long long)fs.bleaves.size(); out4[2] = fs.bvh_depth;
    out4[3] = (long long)(fs.nodes.size() * sizeof(BvhNode) + fs.bleaves.size() * (sizeof(BLeaf) + sizeof(int)));
    return RTC_OK;
}

int rtc_scene_small_info(const rtc_scene *scene, int *out8) {
    if (!scene || !out8) return fail(RTC_ERR_ARG, "rtc_scene_small_info: null argument");
    const FlatScene &fs = scene->flat;
    int fused_rects = 0, fused_spheres = 0;
    for (const SmallPair &P : fs.small_pairs) fused_rects += (P.f[12] != 0.0f) + (P.f[13] != 0.0f);
    for (const SmallSphere &sp : fs.small_spheres) fused_spheres += sp.pdf_k != 0.0f;
    out8[0] = fs.small_ok; out8[1] = (int)fs.small_pairs.size(); out8[2] = (int)fs.small_groups.size(); out8[3] = (int)fs.small_boxes.size();
    out8[4] = (int)fs.small_spheres.size(); out8[5] = fs.small_fused; out8[6] = fused_rects; out8[7] = fused_spheres;
    return RTC_OK;
}

int rtc_scene_extend_info(const rtc_scene *scene, int *out4) {
    if (!scene || !out4) return fail(RTC_ERR_ARG, "rtc_scene_extend_info: null argument");
    if (scene->device < 0) return fail(RTC_ERR_DEVICE, "rtc_scene_extend_info: the scene has no device tables");
    size_t lds = 0;
    const bool on = wf_lds_tables(scene->f32.view, &lds);
    out4[0] = on ? 1 : 0; out4[1] = (int)lds; out4[2] = scene->f32.view.n_nodes; out4[3] = scene->f32.view.n_bleaves;
    return RTC_OK;
}

int rtc_scene_dump_bvh(const rtc_scene *scene, void *nodes16, int node_cap, void *leaves16, int *prim_of_slot, int leaf_cap) {
    if (!scene) return fail(RTC_ERR_ARG, "rtc_scene_dump_bvh: null scene");
    const FlatScene &fs = scene->flat;
    static_assert(sizeof(BvhNode) == 64 && sizeof(BLeaf) == 64, "table layout");
    if (nodes16) std::memcpy(nodes16, fs.nodes.data(), std::min((size_t)(node_cap > 0 ? node_cap : 0), fs.nodes.size()) * sizeof(BvhNode));
    const size_t nl = std::min((size_t)(leaf_cap > 0 ? leaf_cap : 0), fs.bleaves.size());
    if (leaves16) std::memcpy(leaves16, fs.bleaves.data(), nl * sizeof(BLeaf));
    if (prim_of_slot) std::memcpy(prim_of_slot, fs.leaf_prims.data(), nl * sizeof(int));
    return (int)fs.bleaves.size();
}

// Every compute entry point: the scene has device tables, and they live on the calling thread's current device (a kernel
// launched on another GPU with these pointers would fault it).
static int check_device(const rtc_scene *s) {
    if (s->device < 0) return fail(RTC_ERR_DEVICE, "scene was flattened on the host only (rtc_scene_flatten_only / rtc_scene_load_host): no device tables");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return fail(RTC_ERR_DEVICE, "no current HIP device");
    if (cur != s->device) return fail(RTC_ERR_DEVICE, "scene lives on HIP device " + std::to_string(s->device) + ", the current device is " + std::to_string(cur));
    return RTC_OK;
}
// A pointer the caller says is device memory: refuse host memory and memory of another GPU before a kernel meets it.
static int check_device_pointer(const void *ptr, size_t bytes, const char *what) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) { (void)hipGetLastError(); return fail(RTC_ERR_ARG, std::string(what) + " is not memory the HIP runtime knows (a host pointer?)"); }
    int cur = -1; (void)hipGetDevice(&cur);
    if (attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeUnregistered) return fail(RTC_ERR_ARG, std::string(what) + " is host memory; this entry point takes device memory");
    if (attr.type == hipMemoryTypeDevice && attr.device != cur) return fail(RTC_ERR_ARG, std::string(what) + " lives on HIP device " + std::to_string(attr.device) + ", the current device is " + std::to_string(cur));
    hipDeviceptr_t base = nullptr; size_t size = 0;   // the allocation must reach as far as the kernel will write (a sub-allocator's block may hide a short tensor)
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr) == hipSuccess) {
        if ((const char *)ptr + bytes > (const char *)base + size) return fail(RTC_ERR_ARG, std::string(what) + " is shorter than the image needs (" + std::to_string(bytes) + " bytes)");
    } else (void)hipGetLastError();
    return RTC_OK;
}
static int ensure_f64(rtc_scene *s) {
    if (int rc_dev = check_device(s)) return rc_dev;
    if (s->f64.ready) return RTC_OK;
    HIP_TRY(upload_tables<double>(s->flat, s->f64, true));
    return RTC_OK;
}

int rtc_first_hit_f64(const rtc_scene *scene_c, const double *rays7, size_t n, double tmin, double tmax, int32_t *prim_id, double *t, double *rec12) {
    rtc_scene *s = const_cast<rtc_scene *>(scene_c);
    if (!s || (!rays7 && n) || !prim_id || !t) return fail(RTC_ERR_ARG, "rtc_first_hit_f64: null argument");
    if (n == 0) return RTC_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    int rc = ensure_f64(s); if (rc) return rc;
    DevBuf d_rays, d_prim, d_t, d_rec;
    HIP_TRY(d_rays.upload(rays7, n * 7 * sizeof(double)));
    HIP_TRY(hipMalloc(&d_prim.p, n * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&d_t.p, n * sizeof(double)));
    if (rec12) HIP_TRY(hipMalloc(&d_rec.p, n * 12 * sizeof(double)));
    HIP_TRY(launch_first_hit_f64(s->f64.view, (const double *)d_rays.p, n, tmin, tmax, (int32_t *)d_prim.p, (double *)d_t.p, (double *)d_rec.p, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(prim_id, d_prim.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(t, d_t.p, n * sizeof(double), hipMemcpyDeviceToHost));
    if (rec12) HIP_TRY(hipMemcpy(rec12, d_rec.p, n * 12 * sizeof(double), hipMemcpyDeviceToHost));
    return RTC_OK;
}

int rtc_primary_first_hit_f64(const rtc_scene *scene_c, const camera *cam, const double *st2, const double *lens3, size_t n,
                              double tmin, double tmax, int32_t *prim_id, double *t, double *rays7_out) {
    rtc_scene *s = const_cast<rtc_scene *>(scene_c);
    if (!s || !cam || (!st2 && n) || !prim_id || !t) return fail(RTC_ERR_ARG, "rtc_primary_first_hit_f64: null argument");
    if (n == 0) return RTC_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    int rc = ensure_f64(s); if (rc) return rc;
    DevBuf d_st, d_lens, d_prim, d_t, d_rays;
    HIP_TRY(d_st.upload(st2, n * 2 * sizeof(double)));
    if (lens3) HIP_TRY(d_lens.upload(lens3, n * 3 * sizeof(double)));
    HIP_TRY(hipMalloc(&d_prim.p, n * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&d_t.p, n * sizeof(double)));
    if (rays7_out) HIP_TRY(hipMalloc(&d_rays.p, n * 7 * sizeof(double)));
    HIP_TRY(launch_primary_first_hit_f64(s->f64.view, camera_tables<double>(cam), (const double *)d_st.p, (const double *)d_lens.p, n, tmin, tmax,
                                         (int32_t *)d_prim.p, (double *)d_t.p, (double *)d_rays.p, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(prim_id, d_prim.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(t, d_t.p, n * sizeof(double), hipMemcpyDeviceToHost));
    if (rays7_out) HIP_TRY(hipMemcpy(rays7_out, d_rays.p, n * 7 * sizeof(double), hipMemcpyDeviceToHost));
    return RTC_OK;
}

int rtc_first_hit_f32(const rtc_scene *scene_c, const double *rays7, size_t n, double tmin, double tmax, int32_t *prim_id, float *t) {
    rtc_scene *s = const_cast<rtc_scene *>(scene_c);
    if (!s || (!rays7 && n) || !prim_id || !t) return fail(RTC_ERR_ARG, "rtc_first_hit_f32: null argument");
    if (int rc_dev = check_device(s)) return rc_dev;
    if (n == 0) return RTC_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    std::vector<float> rf(n * 7);
    for (size_t i = 0; i < n * 7; i++) rf[i] = (float)rays7[i];
    DevBuf d_rays, d_prim, d_t;
    HIP_TRY(d_rays.upload(rf.data(), rf.size() * sizeof(float)));
    HIP_TRY(hipMalloc(&d_prim.p, n * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&d_t.p, n * sizeof(float)));
    const float tmaxf = tmax > 3.0e38 ? __builtin_huge_valf() : (float)tmax;
    HIP_TRY(launch_first_hit_f32(s->f32.view, (const float *)d_rays.p, n, (float)tmin, tmaxf, (int32_t *)d_prim.p, (float *)d_t.p, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(prim_id, d_prim.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(t, d_t.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return RTC_OK;
}

static int check_settings(const rtc_settings *st) {
    if (!st) return fail(RTC_ERR_ARG, "settings is NULL");
    if (st->width < 2 || st->height < 2) return fail(RTC_ERR_ARG, "width and height must be >= 2 (the reference divides by W-1, H-1)");
    if ((long long)st->width * st->height > (1ll << 30)) return fail(RTC_ERR_ARG, "image too large");
    if (st->sample_end < st->sample_begin || st->sample_begin < 0) return fail(RTC_ERR_ARG, "bad sample range");
    if (st->max_depth < 0 || st->max_depth > 4096) return fail(RTC_ERR_ARG, "bad max_depth");
    if (st->integrator < 0 || st->integrator > 2) return fail(RTC_ERR_ARG, "integrator must be 0, 1 or 2");
    if (st->kernel < 0 || st->kernel > 3) return fail(RTC_ERR_ARG, "kernel must be 0, 1, 2 or 3");
    return RTC_OK;
}

// Wavefront driver (kernels_wf.hip): init, a FIXED schedule of bounces (extend + shade), a compaction every eight of them, then the
// accumulate.  Nothing is read back: the length of the slot list, which of its two buffers is current, whether a compaction is worth
// it and whether anything is still alive are device-side state (WfBuffers::counters), and a kernel of the schedule that finds no
// live slot returns at once -- so the call only ENQUEUES, like the small-scene path, and the caller's stream is free to carry the
// next frame or a reduce behind it.
// The samples are split into up to three parts that run the same schedule on streams of their own, side by side: extend is bound
// by VALU issue and fetch latency, the tail of every launch runs on a thinning set of waves, and a second and third part fill what
// one leaves idle (final scene 1024^2 x 256 spp: 116 -> ... ms, 1M spheres: 48 -> ... ms).  Each part owns a share of
// the path-slot pool and its own counters; their accumulates run in part order, so the image does not depend on timing.
enum { RTC_WF_MAX_PARTS = 4 };
static int render_wavefront(rtc_scene *s, const CameraT<float> &camf, const RenderParams &p_all, float4 *accum, unsigned long long *stats_out, hipStream_t stream) {
    const int spp_all = p_all.s_end - p_all.s_begin;
    if (spp_all <= 0) return RTC_OK;
    const long long per_layer = (long long)p_all.n_tiles * 64;
    // samples per slot aimed at: ONE -- a slot per sample until the cap of 32 M slots is reached (1080p from 16 spp, 1024^2 from 32 spp), because a slot walks
    // through its samples one after the other and the render lasts as many rounds as the unluckiest slot needs.  It was 8 until the end of round 3,
    // which only mattered below the cap: book1 1080p x 8 spp (config 1) 3.33 ms at 8 samples per slot, 2.69 / 2.66 / 2.36 at 4 / 2 / 1; at 64 spp and
    // for the 1M spheres all four are within 1 %.
    int sps = 1;
    if (const char *e = std::getenv("RTC_WF_SPS")) { const int v = std::atoi(e); if (v >= 1) sps = v; }   // A/B knob
    const long long slots_wanted = std::min<long long>((long long)((spp_all + sps - 1) / sps) * per_layer, 32ll << 20);
    // Re-measured at the end of round 3, after the live-slot counter stopped serialising every part on one address (which is what more
    // parts had mostly bought: each has counters of its own).  One GPU run, 1 / 2 / 3 / 4 parts: final 75.5 / 69.3 / 68.8 / 67.9 ms, final + earth
    // 138.5 / 125.5 / 124.6 / 123.0, 1M spheres 32.7 / 30.4 / 30.5 / 30.2; trees traversed from LDS (k_wf_extend_lds: persistent workgroups that
    // fill the GPU by themselves): book1 15.0 / 13.8 / 14.2 / 14.7, Book-2 random 26.1 / 24.3 / 24.5 / 24.3.  So: up to four parts of >= 4 M slots,
    // two for the LDS kernels.
    int n_parts = (int)std::max<long long>(1, std::min<long long>(4, slots_wanted / (4ll << 20)));
    if (wf_lds_tables(s->f32.view, nullptr)) n_parts = std::min(n_parts, 2);
    if (const char *e = std::getenv("RTC_WF_PARTS")) { const int v = std::atoi(e); if (v >= 1 && v <= RTC_WF_MAX_PARTS) n_parts = v; }   // A/B knob
    if (n_parts > spp_all) n_parts = spp_all;
    // Slots per pixel ("layers").  A slot walks through its samples one after the other, so a render lasts as many bounces
    // as the UNLUCKIEST slot needs; with ~8 samples per slot that is <= 8 * max_depth bounces of large launches instead of
    // hundreds of bounces whose tail runs on a few live slots (final scene 1024^2 x 256 spp: 411 ms at 1M slots, 227 ms at
    // 32M).  Past ~32M slots (3 GB of state) the footprint costs more than the shorter tail gains.  RTC_WF_SLOTS overrides
    // the cap for experiments.
    long long max_slots = 32ll << 20;
    if (const char *e = std::getenv("RTC_WF_SLOTS")) { const long long v = std::atoll(e); if (v > 0) max_slots = v; }
    // Camera rays traced ahead (k_wf_primary): the hits of the first phit_k samples of every slot, 8 bytes each, within a budget of 8 GB per
    // render (32 M slots: 32 samples per slot; what lies beyond goes through the slot lists as before).  RTC_WF_PRIMARY=0: none (A/B, tests).
    long long phit_budget = 8ll << 30;
    if (const char *e = std::getenv("RTC_WF_PRIMARY")) { const long long v = std::atoll(e); phit_budget = v <= 0 ? 0 : (v == 1 ? phit_budget : v); }
    int compact_div = 16;
    if (const char *e = std::getenv("RTC_WF_COMPACT_DIV")) compact_div = std::max(2, std::atoi(e));   // A/B knob
    struct Part { RenderParams p; long long layers, n_slots, phit_k; WfBuffers b; long long iterations; };
    Part part[RTC_WF_MAX_PARTS];
    long long total_slots = 0;
    size_t phit_bytes = 0;
    for (int k = 0; k < n_parts; ++k) {
        Part &P = part[k];
        P.p = p_all;
        P.p.s_begin = p_all.s_begin + (int)((long long)spp_all * k / n_parts);
        P.p.s_end = p_all.s_begin + (int)((long long)spp_all * (k + 1) / n_parts);
        const int spp = P.p.s_end - P.p.s_begin;
        long long layers = (spp + sps - 1) / sps;
        const long long part_cap = std::min(max_slots / n_parts, 32ll << 20);   // (<= 32 M slots per part whatever RTC_WF_SLOTS says: the packed statistics of kernels_wf.hip count 1/64 of a part in 20 bits)
        if (layers * per_layer > part_cap) layers = part_cap / per_layer;
        if (layers * per_layer < (1ll << 20) / n_parts) layers = ((1ll << 20) / n_parts + per_layer - 1) / per_layer;   // small images: at least 1M slots in all
        if (layers > spp) layers = spp;
        if (layers < 1) layers = 1;
        P.layers = layers; P.n_slots = per_layer * layers;
        const long long per_slot_samples = (spp + layers - 1) / layers;
        P.iterations = per_slot_samples * (long long)(P.p.max_depth > 0 ? P.p.max_depth : 1);
        P.phit_k = std::max<long long>(0, std::min<long long>(std::min<long long>(per_slot_samples, 65535), phit_budget / n_parts / (P.n_slots * 8)));
        if (P.p.max_depth <= 0) P.phit_k = 0;
        total_slots += P.n_slots;
        phit_bytes += (size_t)(P.phit_k * P.n_slots) * sizeof(float2);
    }
    if (total_slots > (1ll << 30)) return fail(RTC_ERR_ARG, "image too large for the wavefront path");
    const size_t per_slot = wf_bytes_per_slot();
    // The previous render's final accumulate may still be in flight on ANOTHER stream and reads the pool this render is about
    // to overwrite (this call returns with that kernel enqueued, not finished): order this stream behind it.
    if (!s->wf_done) HIP_TRY(hipEventCreateWithFlags(&s->wf_done, hipEventDisableTiming));
    else HIP_TRY(hipStreamWaitEvent(stream, s->wf_done, 0));
    size_t pool_bytes = (size_t)total_slots * per_slot + phit_bytes + 512 * RTC_WF_MAX_PARTS;
    if (pool_bytes > s->wf_pool_bytes) {
        if (s->wf_pool.p) { HIP_TRY(hipEventSynchronize(s->wf_done)); HIP_TRY(hipStreamSynchronize(stream)); (void)hipFree(s->wf_pool.p); s->wf_pool.p = nullptr; s->wf_pool_bytes = 0; }
        if (hipMalloc(&s->wf_pool.p, pool_bytes) != hipSuccess && phit_bytes) {   // no room for the camera-ray hits: render without them
            (void)hipGetLastError();
            for (int k = 0; k < n_parts; ++k) part[k].phit_k = 0;
            pool_bytes -= phit_bytes; phit_bytes = 0; s->wf_pool.p = nullptr;
            if (hipMalloc(&s->wf_pool.p, pool_bytes) != hipSuccess) s->wf_pool.p = nullptr;
        }
        if (!s->wf_pool.p) { (void)hipGetLastError(); return fail(RTC_ERR_NOMEM, "wavefront path-slot state does not fit in device memory"); }
        s->wf_pool_bytes = pool_bytes;
    }
    for (int k = 1; k < n_parts; ++k)
        if (!s->wf_stream[k]) {
            HIP_TRY(hipStreamCreateWithFlags(&s->wf_stream[k], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&s->wf_part_done[k], hipEventDisableTiming));
        }
    if (!s->wf_fork) { HIP_TRY(hipEventCreateWithFlags(&s->wf_fork, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&s->wf_part_done[0], hipEventDisableTiming)); }
    char *base = (char *)s->wf_pool.p;
    for (int k = 0; k < n_parts; ++k) {
        WfBuffers &b = part[k].b; std::memset(&b, 0, sizeof b);
        const size_t n = (size_t)part[k].n_slots;
        b.ro = (float4 *)base; base += n * sizeof(float4);
        b.rd = (float4 *)base; base += n * sizeof(float4);
        b.tm = (float4 *)base; base += n * sizeof(float4);
        b.sum = (float4 *)base; base += n * sizeof(float4);
        b.hit = (float2 *)base; base += n * sizeof(float2);
        b.active[0] = (int *)base; base += n * sizeof(int);
        b.active[1] = (int *)base; base += n * sizeof(int);
        b.key = (unsigned char *)base; base += (n + 255) / 256 * 256;
        b.phit = (float2 *)base; base += (size_t)part[k].phit_k * n * sizeof(float2);
        b.phit_k = (int)part[k].phit_k; b.div_layers = wf_make_div((uint32_t)part[k].layers);
        char *ctl = (char *)s->ray_counter.p + (size_t)k * (wf_stats_bytes() + 64);
        b.stats = (unsigned long long *)ctl; b.stats_out = stats_out;
        b.counters = (unsigned *)(ctl + wf_stats_bytes());
        b.n_slots = (int)n; b.slots_per_layer = (int)per_layer; b.layers = (int)part[k].layers;
        b.quad = part[k].layers % 4 == 0;
        b.compact_div = compact_div;
        b.noise_lds_slots = (std::getenv("RTC_WF_NOISE_LDS") && std::atoi(std::getenv("RTC_WF_NOISE_LDS")) == 0) ? 0 : 2;
        b.div_group = wf_make_div((uint32_t)(b.quad ? 4 * per_layer : per_layer)); b.div_tiles_x = wf_make_div((uint32_t)p_all.tiles_x);
    }
    // fork: the parts' streams start behind everything the caller's stream holds so far
    hipStream_t st[RTC_WF_MAX_PARTS];
    st[0] = stream;
    if (n_parts > 1) HIP_TRY(hipEventRecord(s->wf_fork, stream));
    for (int k = 1; k < n_parts; ++k) { st[k] = s->wf_stream[k]; HIP_TRY(hipStreamWaitEvent(st[k], s->wf_fork, 0)); }
    for (int k = 0; k < n_parts; ++k) HIP_TRY(launch_wf_init(camf, part[k].p, part[k].b, st[k]));
    for (int k = 0; k < n_parts; ++k) HIP_TRY(launch_wf_primary(s->f32.view, camf, part[k].p, part[k].b, st[k]));
    long long longest = 0;
    for (int k = 0; k < n_parts; ++k) longest = std::max(longest, part[k].iterations);
    // Compaction points every 2 rounds, and the device squeezes the list there once a sixteenth of it is holes.  (Rounds 1-3 ran every 8 rounds / a quarter,
    // from the days when extend waves were cheap to leave sparse; with the shading tail in extend a hole costs a lane of a 100-register kernel.  One GPU run,
    // every / fraction: 8 / 4 -> 2 / 16: final 88.8 -> 83.8 ms, final + earth 154.8 -> 149.7, book1 19.5 -> 17.2, 1M spheres 36.7 -> 35.8;
    // 1 / 32: 86.7, 153.7, 17.8, 36.3; never squeezing below the threshold (2 / 10^6): 84.3, 150.3, 17.5, 36.2 -- profiles/r03e/compaction_sweep.txt.)
    long long compact_every = 2;
    if (const char *e = std::getenv("RTC_WF_COMPACT_EVERY")) compact_every = std::min(16ll, std::max(1ll, std::atoll(e)));   // A/B knob (<= 16: the packed statistics of kernels_wf.hip are unpacked at compaction points)
    for (long long it = 0; it < longest; ++it)                       // round robin, so that the parts' kernels reach the GPU interleaved
        for (int k = 0; k < n_parts; ++k) {
            if (it >= part[k].iterations) continue;
            if (it && it % compact_every == 0) HIP_TRY(launch_wf_compact(part[k].b, st[k]));
            HIP_TRY(launch_wf_bounce(s->f32.view, camf, part[k].p, part[k].b, st[k]));
        }
    // join: the accumulates run in part order (each adds to the same pixels), the last one on the caller's stream
    for (int k = n_parts - 1; k >= 1; --k) {
        if (k < n_parts - 1) HIP_TRY(hipStreamWaitEvent(st[k], s->wf_part_done[k + 1], 0));
        HIP_TRY(launch_wf_finish(part[k].p, part[k].b, accum, st[k]));
        HIP_TRY(hipEventRecord(s->wf_part_done[k], st[k]));
    }
    if (n_parts > 1) HIP_TRY(hipStreamWaitEvent(stream, s->wf_part_done[1], 0));
    HIP_TRY(launch_wf_finish(part[0].p, part[0].b, accum, stream));
    HIP_TRY(hipEventRecord(s->wf_done, stream));
    return RTC_OK;
}

int rtc_render_device(const rtc_scene *scene_c, const camera *cam, const rtc_settings *st, float *accum_dev, void *stream_v) {
    rtc_scene *s = const_cast<rtc_scene *>(scene_c);
    if (!s || !cam || !accum_dev) return fail(RTC_ERR_ARG, "rtc_render_device: null argument");
    if (int rc_dev = check_device(s)) return rc_dev;
    int rc = check_settings(st); if (rc) return rc;
    if (int rc_ptr = check_device_pointer(accum_dev, (size_t)st->width * (size_t)st->height * 16, "rtc_render_device: accum_dev")) return rc_ptr;
    hipStream_t stream = (hipStream_t)stream_v;
    std::lock_guard<std::mutex> lk(s->mu);
    StreamState &ss = s->streams[stream_v];
    if (!ss.ev0) { HIP_TRY(hipEventCreate(&ss.ev0)); HIP_TRY(hipEventCreate(&ss.ev1)); }
    RenderParams p; std::memset(&p, 0, sizeof p);
    p.width = st->width; p.height = st->height; p.s_begin = st->sample_begin; p.s_end = st->sample_end;
    p.max_depth = st->max_depth; p.integrator = st->integrator;
    p.seed_lo = (uint32_t)st->seed; p.seed_hi = (uint32_t)(st->seed >> 32);
    // An empty light list leaves nothing to mix: the cosine law alone (weight 0).  The reference has no defined answer here --
    // pdf_generate would call hittable_list_random on an empty list (hittable_list.c:174-180: random_int(0, -1)) -- and keeping
    // the weight would scale every bounce by 1 / (1 - w).
    p.light_weight = s->flat.lights.empty() ? 0.0f : (float)st->mixture_light_weight;
    p.light_weight_each = s->flat.lights.empty() ? 0.0f : p.light_weight / (float)s->flat.lights.size(); p.cosine_weight = 1.0f - p.light_weight;
    p.tiles_x = (st->width + 7) / 8; p.tiles_y = (st->height + 7) / 8; p.n_tiles = p.tiles_x * p.tiles_y;
    const CameraT<float> camf = camera_tables<float>(cam);
    p.use_time = camf.time0 != camf.time1;
    for (int k = 0; k < 3; k++) { p.cam_base[k] = camf.llc[k] - camf.origin[k]; p.cam_hs[k] = camf.horizontal[k] / (float)(p.width - 1); p.cam_vs[k] = camf.vertical[k] / (float)(p.height - 1); }
    if (!ss.stats.p) { const unsigned long long zero[4] = {0, 0, 0, 0}; HIP_TRY(ss.stats.upload(zero, sizeof zero)); }
    unsigned long long *const stats = (unsigned long long *)ss.stats.p;
    HIP_TRY(hipMemsetAsync(stats, 0, 4 * sizeof(unsigned long long), stream));
    HIP_TRY(hipEventRecord(ss.ev0, stream));
    const bool small = st->kernel == 2 || (st->kernel == 0 && s->small.n_leaves > 0);
    if (small && s->small.n_leaves <= 0)
        return fail(RTC_ERR_SCENE, "kernel 2 (small-scene) needs <= 64 rect/sphere leaves with solid textures and no media");
    // the wavefront path state packs (sample, depth) into 24 + 8 bits (kernels.h WfBuffers)
    // ... and its packed statistics count 1/64 of a part's slots in 20 bits: one layer of slots (64 per 8x8 tile) must not exceed 32 M (a 33-megapixel image)
    const bool wf_fits = st->max_depth <= 255 && st->sample_end <= (1 << 24) && (long long)p.n_tiles * 64 <= (32ll << 20);
    const bool wavefront = st->kernel == 3 || (st->kernel == 0 && !small && s->f32.view.n_nodes > 0 && wf_fits);
    if (wavefront && s->f32.view.n_nodes <= 0)
        return fail(RTC_ERR_SCENE, "kernel 3 (wavefront) needs a scene that was flattened into a BVH (more than 64 leaves)");
    if (wavefront && !wf_fits)
        return fail(RTC_ERR_ARG, "kernel 3 (wavefront) needs max_depth <= 255, sample_end <= 2^24 and an image of at most 32 M pixels (in 8x8 tiles)");
    if (small) {
        if ((size_t)p.n_tiles > ss.tile_scratch_cap) {
            if (ss.tile_scratch.p) { HIP_TRY(hipStreamSynchronize(stream)); (void)hipFree(ss.tile_scratch.p); ss.tile_scratch.p = nullptr; ss.tile_scratch_cap = 0; }
            if (hipMalloc(&ss.tile_scratch.p, 2 * (size_t)p.n_tiles * sizeof(int)) == hipSuccess) ss.tile_scratch_cap = (size_t)p.n_tiles;
            else ss.tile_scratch.p = nullptr;                       // no scratch: tiles run in image order
            ss.order_key.clear();
        }
        // The tile order is a property of (camera, image size, depth, integrator): the next pass of a progressive render, or
        // the next frame of a benchmark, reuses it instead of probing again (0.25 ms of the 23 ms frame at 1080p).
        std::vector<unsigned char> key(sizeof camf + 5 * sizeof(int));
        { const int k[5] = {p.width, p.height, p.max_depth, p.integrator, (int)(p.light_weight * 1e6f)};
          std::memcpy(key.data(), &camf, sizeof camf); std::memcpy(key.data() + sizeof camf, k, sizeof k); }
        const bool reuse_order = ss.tile_scratch.p && key == ss.order_key;
        // split a tile's samples over several waves until there are ~60K work items (12 per resident wave slot), keeping at
        // least 32 samples per wave: finer items shorten the tail of the launch, but every wave pays its own ramp-down
        // (measured at 1080p x 256 spp: 1 wave per tile 30.9 ms, 2 waves 29.4, 4 waves 29.9, 8 waves 32.8)
        int log2_chunks = 0;
        const int spp = p.s_end - p.s_begin;
        while (log2_chunks < 3 && ((long long)p.n_tiles << log2_chunks) < 60000 && (spp >> (log2_chunks + 1)) >= 32) ++log2_chunks;
        if (const char *e = std::getenv("RTC_SMALL_LOG2_CHUNKS")) { const int v = std::atoi(e); if (v >= 0 && v <= 3 && (spp >> v) >= 1) log2_chunks = v; }   // tuning knob
        const size_t need = log2_chunks ? ((size_t)p.width * p.height) << log2_chunks : 0;
        if (need > ss.small_partial_cap) {
            if (ss.small_partial.p) { HIP_TRY(hipStreamSynchronize(stream)); (void)hipFree(ss.small_partial.p); ss.small_partial.p = nullptr; ss.small_partial_cap = 0; }
            if (hipMalloc(&ss.small_partial.p, need * sizeof(float4)) == hipSuccess) ss.small_partial_cap = need;
            else { ss.small_partial.p = nullptr; log2_chunks = 0; }   // no scratch: one wave per tile
        }
        int ordered = reuse_order ? 1 : 0;
        HIP_TRY(launch_pathtrace_small(s->small, s->f32.view, camf, p, (float4 *)accum_dev, stats, (int *)ss.tile_scratch.p,
                                       &ordered, log2_chunks ? (float4 *)ss.small_partial.p : nullptr, log2_chunks, stream));
        if (ordered) ss.order_key = key; else ss.order_key.clear();
    }
    else if (wavefront) { rc = render_wavefront(s, camf, p, (float4 *)accum_dev, stats, stream); if (rc) return rc; }
    else HIP_TRY(launch_pathtrace_mega(s->f32.view, camf, p, (float4 *)accum_dev, stats, stream));
    ss.kernel = small ? 2 : (wavefront ? 3 : 1);
    HIP_TRY(hipEventRecord(ss.ev1, stream));
    ss.samples = (uint64_t)st->width * st->height * (uint64_t)(st->sample_end - st->sample_begin);
    ss.launches = 1; ss.pending = true;
    return RTC_OK;
}

// rtc_render_device for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid hipMalloc): the
// same kernels, their scratch in the file-scope device array above.  No hipMalloc / hipFree / event or stream creation, no
// host heap either (no per-stream state is kept: rtc_render_stats does not see these calls).  Small scenes only.
int rtc_render_device_noalloc(const rtc_scene *scene_c, const camera *cam, const rtc_settings *st, float *accum_dev, void *stream_v) {
    const rtc_scene *s = scene_c;
    if (!s || !cam || !accum_dev || !st) return fail_literal(RTC_ERR_ARG, "rtc_render_device_noalloc: null argument");
    if (s->device < 0) return fail_literal(RTC_ERR_DEVICE, "rtc_render_device_noalloc: the scene has no device tables (flattened on the host only)");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != s->device) return fail_literal(RTC_ERR_DEVICE, "rtc_render_device_noalloc: the scene lives on another HIP device than the current one");
    if (st->width < 2 || st->height < 2 || st->sample_end < st->sample_begin || st->sample_begin < 0 || st->max_depth < 0 || st->max_depth > 4096 ||
        st->integrator < 0 || st->integrator > 2) return fail_literal(RTC_ERR_ARG, "rtc_render_device_noalloc: bad settings (size >= 2x2, 0 <= sample_begin <= sample_end, 0 <= max_depth <= 4096, integrator 0..2)");
    if (st->kernel != 0 && st->kernel != 2) return fail_literal(RTC_ERR_SCENE, "rtc_render_device_noalloc: only the small-scene kernel (kernel 0 or 2) has an allocation-free form");
    if (s->small.n_leaves <= 0) return fail_literal(RTC_ERR_SCENE, "rtc_render_device_noalloc: needs a small scene (<= 64 rect/sphere leaves, solid textures, no media)");
    const long long tiles_x = (st->width + 7) / 8, tiles_y = (st->height + 7) / 8;
    if (tiles_x * tiles_y > (long long)RTC_NOALLOC_MAX_TILES)
        return fail_literal(RTC_ERR_ARG, "rtc_render_device_noalloc: image too large for the static scratch (at most 131,072 tiles of 8x8 pixels, e.g. 3840 x 2160)");
    hipPointerAttribute_t attr;                 // a host pointer must not reach a kernel
    if (hipPointerGetAttributes(&attr, accum_dev) != hipSuccess || attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeUnregistered ||
        (attr.type == hipMemoryTypeDevice && attr.device != cur)) { (void)hipGetLastError(); return fail_literal(RTC_ERR_ARG, "rtc_render_device_noalloc: accum_dev is not device memory of the current device"); }
    void *base = nullptr;
    if (int rc = noalloc_scratch(&base)) return rc;
    hipStream_t stream = (hipStream_t)stream_v;
    RenderParams p; std::memset(&p, 0, sizeof p);
    p.width = st->width; p.height = st->height; p.s_begin = st->sample_begin; p.s_end = st->sample_end;
    p.max_depth = st->max_depth; p.integrator = st->integrator;
    p.seed_lo = (uint32_t)st->seed; p.seed_hi = (uint32_t)(st->seed >> 32);
    p.light_weight = s->flat.lights.empty() ? 0.0f : (float)st->mixture_light_weight;
    p.light_weight_each = s->flat.lights.empty() ? 0.0f : p.light_weight / (float)s->flat.lights.size(); p.cosine_weight = 1.0f - p.light_weight;
    p.tiles_x = (int)tiles_x; p.tiles_y = (int)tiles_y; p.n_tiles = p.tiles_x * p.tiles_y;
    const CameraT<float> camf = camera_tables<float>(cam);
    p.use_time = camf.time0 != camf.time1;
    for (int k = 0; k < 3; k++) { p.cam_base[k] = camf.llc[k] - camf.origin[k]; p.cam_hs[k] = camf.horizontal[k] / (float)(p.width - 1); p.cam_vs[k] = camf.vertical[k] / (float)(p.height - 1); }
    // layout of the scratch: counters | tile cost + order | partial planes
    unsigned long long *counters = (unsigned long long *)base;
    int *tile_scratch = (int *)((char *)base + RTC_NOALLOC_COUNTER_BYTES);
    float4 *partial = (float4 *)((char *)base + RTC_NOALLOC_COUNTER_BYTES + 2 * RTC_NOALLOC_MAX_TILES * sizeof(int));
    // the chunking rule of rtc_render_device, then as many chunks as fit the partial planes: with the same chunking the sums are
    // bit-identical to rtc_render_device's, with fewer chunks they differ by float reassociation only
    int log2_chunks = 0;
    const int spp = p.s_end - p.s_begin;
    while (log2_chunks < 3 && ((long long)p.n_tiles << log2_chunks) < 60000 && (spp >> (log2_chunks + 1)) >= 32) ++log2_chunks;
    while (log2_chunks > 0 && (((size_t)p.width * p.height) << log2_chunks) > RTC_NOALLOC_MAX_PARTIAL) --log2_chunks;
    std::lock_guard<std::mutex> lk(g_noalloc_mu[cur]);
    // the array keeps the last call's contents: the counters must start at zero; tile costs, order and partial planes are
    // fully rewritten by the kernels before they are read (probe -> sort -> render -> combine)
    if (hipMemsetAsync(counters, 0, RTC_NOALLOC_COUNTER_BYTES, stream) != hipSuccess) { (void)hipGetLastError(); return fail_literal(RTC_ERR_DEVICE, "rtc_render_device_noalloc: hipMemsetAsync failed"); }
    int ordered = 0;                            // no per-stream cache here: the tile order is probed on every call
    if (launch_pathtrace_small(s->small, s->f32.view, camf, p, (float4 *)accum_dev, counters, tile_scratch, &ordered, log2_chunks ? partial : nullptr, log2_chunks, stream) != hipSuccess)
        return fail_literal(RTC_ERR_DEVICE, "rtc_render_device_noalloc: kernel launch failed");
    return RTC_OK;
}

int rtc_render_stats(const rtc_scene *scene_c, void *stream_v, rtc_stats *out) {
    rtc_scene *s = const_cast<rtc_scene *>(scene_c);
    if (!s || !out) return fail(RTC_ERR_ARG, "rtc_render_stats: null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    auto it = s->streams.find(stream_v);
    if (it == s->streams.end() || !it->second.pending) return fail(RTC_ERR_ARG, "rtc_render_stats: no render was enqueued on this stream");
    StreamState &ss = it->second;
    // the events and counters live on the scene's device, which need not be the caller's current one (rtc_render_multi)
    struct OnDevice { int prev = -1; explicit OnDevice(int d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (d >= 0 && d != prev) (void)hipSetDevice(d); else prev = -1; }
                      ~OnDevice() { if (prev >= 0) (void)hipSetDevice(prev); } } on_device(s->device);
    HIP_TRY(hipEventSynchronize(ss.ev1));
    float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, ss.ev0, ss.ev1));
    unsigned long long cnt[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(cnt, ss.stats.p, sizeof cnt, hipMemcpyDeviceToHost));
    const unsigned long long rays = cnt[0];
    std::memset(out, 0, sizeof *out);
    out->samples = ss.samples; out->rays = rays; out->kernel_ms = ms; out->total_ms = ss.total_ms > 0 ? ss.total_ms : ms;
    out->kernel_launches = ss.launches; out->kernel_used = ss.kernel;
    out->wave_iterations = cnt[1];
    return RTC_OK;
}

int rtc_render(const rtc_scene *scene, const camera *cam, const rtc_settings *st, float *accum_host) {
    if (!scene || !cam || !accum_host) return fail(RTC_ERR_ARG, "rtc_render: null argument");
    int rc = check_settings(st); if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    const size_t bytes = (size_t)st->width * st->height * 4 * sizeof(float);
    DevBuf acc;
    HIP_TRY(acc.upload(accum_host, bytes));
    rc = rtc_render_device(scene, cam, st, (float *)acc.p, nullptr); if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(accum_host, acc.p, bytes, hipMemcpyDeviceToHost));
    rtc_scene *s = const_cast<rtc_scene *>(scene);
    const double total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    { std::lock_guard<std::mutex> lk(s->mu); s->streams[nullptr].total_ms = total_ms; }
    return RTC_OK;
}

// ray_color (ray.c:50-84 and its 02 / 01 forms) for rays the caller supplies: the general bounce loop of kernels_probe.hip.
int rtc_ray_colors(const rtc_scene *scene_c, const double *rays7, size_t n, const rtc_settings *st, double *rgb_sum) {
    rtc_scene *s = const_cast<rtc_scene *>(scene_c);
    if (!s || (!rays7 && n) || (!rgb_sum && n) || !st) return fail(RTC_ERR_ARG, "rtc_ray_colors: null argument");
    if (int rc_dev = check_device(s)) return rc_dev;
    if (st->sample_end < st->sample_begin || st->sample_begin < 0) return fail(RTC_ERR_ARG, "rtc_ray_colors: bad sample range");
    if (st->max_depth < 0 || st->max_depth > 4096) return fail(RTC_ERR_ARG, "rtc_ray_colors: bad max_depth");
    if (st->integrator < 0 || st->integrator > 2) return fail(RTC_ERR_ARG, "rtc_ray_colors: integrator must be 0, 1 or 2");
    if (n > (1u << 24)) return fail(RTC_ERR_ARG, "rtc_ray_colors: at most 2^24 rays per call");
    if (n == 0 || st->sample_end == st->sample_begin) return RTC_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    RenderParams p; std::memset(&p, 0, sizeof p);
    p.s_begin = st->sample_begin; p.s_end = st->sample_end; p.max_depth = st->max_depth; p.integrator = st->integrator;
    p.seed_lo = (uint32_t)st->seed; p.seed_hi = (uint32_t)(st->seed >> 32);
    p.light_weight = s->flat.lights.empty() ? 0.0f : (float)st->mixture_light_weight;
    p.light_weight_each = s->flat.lights.empty() ? 0.0f : p.light_weight / (float)s->flat.lights.size(); p.cosine_weight = 1.0f - p.light_weight;
    std::vector<float> rf(n * 7);
    for (size_t i = 0; i < n * 7; i++) rf[i] = (float)rays7[i];
    DevBuf d_rays, d_sum, d_cnt;
    HIP_TRY(d_rays.upload(rf.data(), rf.size() * sizeof(float)));
    HIP_TRY(d_sum.upload(rgb_sum, n * 3 * sizeof(double)));            // the call ADDS, like rtc_render
    const unsigned long long zero = 0; HIP_TRY(d_cnt.upload(&zero, sizeof zero));
    // enough lanes to fill the chip when few rays carry many samples, never more lanes than samples
    const long long spp = st->sample_end - st->sample_begin;
    long long chunks = (long long)((1u << 20) / n); if (chunks < 1) chunks = 1; if (chunks > spp) chunks = spp; if (chunks > 4096) chunks = 4096;
    HIP_TRY(launch_ray_colors(s->f32.view, p, (const float *)d_rays.p, (unsigned)n, (unsigned)chunks, (double *)d_sum.p, (unsigned long long *)d_cnt.p, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(rgb_sum, d_sum.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return RTC_OK;
}

// Known-answer probe of the device functions the production kernels call (tests only).
int rtc_kat(const rtc_scene *scene_c, int what, const double *in, size_t n, double *out) {
    rtc_scene *s = const_cast<rtc_scene *>(scene_c);
    if (!s || (!in && n) || (!out && n)) return fail(RTC_ERR_ARG, "rtc_kat: null argument");
    if (what < 0 || what >= RTC_KAT_COUNT) return fail(RTC_ERR_ARG, "rtc_kat: unknown function");
    if (int rc_dev = check_device(s)) return rc_dev;
    if (n == 0) return RTC_OK;
    if (n > (1u << 24)) return fail(RTC_ERR_ARG, "rtc_kat: at most 2^24 inputs per call");
    const bool small = what == RTC_KAT_LIGHT_PDF_SMALL || what == RTC_KAT_LIGHT_SAMPLE_SMALL || what == RTC_KAT_DIELECTRIC_SMALL;
    std::lock_guard<std::mutex> lk(s->mu);
    DevBuf d_in, d_out;
    HIP_TRY(d_in.upload(in, n * (size_t)rtc_kat_in_stride(what) * sizeof(double)));
    HIP_TRY(hipMalloc(&d_out.p, n * (size_t)rtc_kat_out_stride(what) * sizeof(double)));
    if (small) HIP_TRY(launch_small_kat(s->f32.view, s->small, what, (const double *)d_in.p, (unsigned)n, (double *)d_out.p, nullptr));
    else HIP_TRY(launch_kat(s->f32.view, what, (const double *)d_in.p, (unsigned)n, (double *)d_out.p, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d_out.p, n * (size_t)rtc_kat_out_stride(what) * sizeof(double), hipMemcpyDeviceToHost));
    return RTC_OK;
}

// ray.h:39 -- color ray_color(ray*, color* background, hittable_list* world, hittable_list* lights, int depth): one sample of
// the 03 estimator for one ray, on the GPU.  The device tables of (world, lights, background) are kept between calls.
color ray_color(ray *r, color *background, hittable_list *world, hittable_list *lights, int recursion_depth) {
    static std::mutex mu;
    static rtc_scene *cached = nullptr;
    static const hittable_list *c_world = nullptr, *c_lights = nullptr;
    static int c_nw = -1, c_nl = -1; static color c_bg = {0, 0, 0};
    static unsigned long long c_gen = 0;   // scene_api.c: bumped by every constructor, list add and free -- a rebuilt scene at the same addresses is seen
    static unsigned long long calls = 0;
    const color black = {0, 0, 0};
    if (!r || !world) return black;
    std::lock_guard<std::mutex> lk(mu);
    const color bg = background ? *background : black;
    const int nw = hittable_list_index(world), nl = lights ? hittable_list_index(lights) : 0;
    if (!cached || c_gen != rt_api_generation() || c_world != world || c_lights != lights || c_nw != nw || c_nl != nl || c_bg.x != bg.x || c_bg.y != bg.y || c_bg.z != bg.z) {
        if (cached) { rtc_scene_free(cached); cached = nullptr; }
        cached = rtc_scene_build(world, lights, &bg, 0.0, 1.0);
        if (!cached) { std::printf("!! ray_color: %s\n", rtc_last_error()); return black; }
        c_world = world; c_lights = lights; c_nw = nw; c_nl = nl; c_bg = bg; c_gen = rt_api_generation();
    }
    rtc_settings st; std::memset(&st, 0, sizeof st);
    st.width = st.height = 2; st.max_depth = recursion_depth; st.integrator = 0; st.mixture_light_weight = 0.1;   // ray.c:72
    st.sample_begin = (int)(calls & 0xffffff); st.sample_end = st.sample_begin + 1; st.seed = calls >> 24; ++calls;
    const double ray7[7] = {r->orig.x, r->orig.y, r->orig.z, r->dir.x, r->dir.y, r->dir.z, r->time};
    double rgb[3] = {0, 0, 0};
    if (rtc_ray_colors(cached, ray7, 1, &st, rgb) != RTC_OK) { std::printf("!! ray_color: %s\n", rtc_last_error()); return black; }
    const color out = {rgb[0], rgb[1], rgb[2]};
    return out;
}

// A copy of a scene's tables on another device (the flattening is not repeated).
rtc_scene *rtc_scene_clone_to_device(const rtc_scene *scene, int device) {
    if (!scene) return fail_scene(nullptr, "rtc_scene_clone_to_device: null scene");
    int n = 0, prev = -1;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return fail_scene(nullptr, "rtc_scene_clone_to_device: no such HIP device");
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return fail_scene(nullptr, "rtc_scene_clone_to_device: hipSetDevice failed");
    rtc_scene *s = new rtc_scene();
    s->flat = scene->flat;
    s = upload_scene(s, "rtc_scene_clone_to_device");
    if (prev >= 0) (void)hipSetDevice(prev);
    return s;
}

// BASELINE config 4 from a C host: disjoint sample shards, one per scene handle (= per device), the accumulators brought to the first
// handle's device by peer copies and added there in handle order, one download.  Every entry point below only enqueues (the wavefront
// path too), so ONE host thread feeds all devices: per handle a stream, an accumulator and -- on the root -- a staging buffer are kept on
// the handle between calls; the N - 1 peer copies run side by side on streams of their own as their renders finish; one kernel adds
// them in handle order; the call synchronises once, at the end.
struct MultiState {                                     // owned by a handle, lives on its device
    hipStream_t stream = nullptr; hipEvent_t rendered = nullptr, staged = nullptr;
    DevBuf acc; DevBuf stage;                           // stage: on the ROOT device, for this handle's accumulator
    int stage_device = -1;
    ~MultiState() { if (stream) (void)hipStreamDestroy(stream); if (rendered) (void)hipEventDestroy(rendered); if (staged) (void)hipEventDestroy(staged); }
};
static std::map<const rtc_scene *, MultiState> g_multi;
static std::mutex g_multi_mu;
static void multi_forget(const rtc_scene *s) {         // rtc_scene_free: the handle's buffers go with it
    std::lock_guard<std::mutex> lk(g_multi_mu);
    auto it = g_multi.find(s);
    if (it == g_multi.end()) return;
    int prev = -1; (void)hipGetDevice(&prev);
    if (s->device >= 0) (void)hipSetDevice(s->device);
    { MultiState &m = it->second; if (m.stage.p && m.stage_device >= 0) { (void)hipSetDevice(m.stage_device); (void)hipFree(m.stage.p); m.stage.p = nullptr; (void)hipSetDevice(s->device); } }
    g_multi.erase(it);
    if (prev >= 0) (void)hipSetDevice(prev);
}
int rtc_render_multi(const rtc_scene *const *scenes, int n_scenes, const camera *cam, const rtc_settings *st, float *accum_host) {
    if (!scenes || n_scenes < 1 || !cam || !accum_host) return fail(RTC_ERR_ARG, "rtc_render_multi: null argument");
    if (n_scenes > 64) return fail(RTC_ERR_ARG, "rtc_render_multi: at most 64 scene handles");
    int rc = check_settings(st); if (rc) return rc;
    for (int i = 0; i < n_scenes; i++) if (!scenes[i] || scenes[i]->device < 0) return fail(RTC_ERR_DEVICE, "rtc_render_multi: scene handle " + std::to_string(i) + " has no device tables");
    for (int i = 0; i < n_scenes; i++) for (int j = 0; j < i; j++) if (scenes[i] == scenes[j]) return fail(RTC_ERR_ARG, "rtc_render_multi: the same scene handle twice (clone it: rtc_scene_clone_to_device)");
    const size_t n_pix = (size_t)st->width * st->height, bytes = n_pix * 4 * sizeof(float);
    const int spp = st->sample_end - st->sample_begin, q = spp / n_scenes, r = spp % n_scenes;
    const int root = scenes[0]->device;
    int prev = -1; (void)hipGetDevice(&prev);
    std::lock_guard<std::mutex> lk(g_multi_mu);
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev};
    auto dev_fail = [&](int i, const char *what, hipError_t e) { return fail(RTC_ERR_DEVICE, std::string("rtc_render_multi, handle ") + std::to_string(i) + ": " + what + ": " + hipGetErrorString(e)); };
    std::vector<MultiState *> ms((size_t)n_scenes);
    hipError_t e;
    // set-up (first call, or another image size): streams, events, accumulators, staging buffers on the root
    for (int i = 0; i < n_scenes; i++) {
        MultiState &m = g_multi[scenes[i]]; ms[(size_t)i] = &m;
        if ((e = hipSetDevice(scenes[i]->device)) != hipSuccess) return dev_fail(i, "hipSetDevice", e);
        if (!m.stream) {
            if ((e = hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking)) != hipSuccess) return dev_fail(i, "hipStreamCreate", e);
            if ((e = hipEventCreateWithFlags(&m.rendered, hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&m.staged, hipEventDisableTiming)) != hipSuccess) return dev_fail(i, "hipEventCreate", e);
        }
        if (m.acc.bytes != bytes) {
            if (m.acc.p) { (void)hipStreamSynchronize(m.stream); (void)hipFree(m.acc.p); m.acc.p = nullptr; m.acc.bytes = 0; }
            if (hipMalloc(&m.acc.p, bytes) != hipSuccess) { m.acc.p = nullptr; return fail(RTC_ERR_NOMEM, "rtc_render_multi: accumulator does not fit in device memory"); }
            m.acc.bytes = bytes;
        }
        if (i > 0 && (m.stage.bytes != bytes || m.stage_device != root)) {
            if (m.stage.p) { (void)hipSetDevice(m.stage_device); (void)hipDeviceSynchronize(); (void)hipFree(m.stage.p); m.stage.p = nullptr; m.stage.bytes = 0; }
            if ((e = hipSetDevice(root)) != hipSuccess) return dev_fail(i, "hipSetDevice(root)", e);
            if (hipMalloc(&m.stage.p, bytes) != hipSuccess) { m.stage.p = nullptr; return fail(RTC_ERR_NOMEM, "rtc_render_multi: staging buffer does not fit in device memory"); }
            m.stage.bytes = bytes; m.stage_device = root;
        }
    }
    // enqueue: the root starts from the caller's sums (the call ADDS, like rtc_render), the others from zero
    for (int i = 0; i < n_scenes; i++) {
        MultiState &m = *ms[(size_t)i];
        if ((e = hipSetDevice(scenes[i]->device)) != hipSuccess) return dev_fail(i, "hipSetDevice", e);
        e = i == 0 ? hipMemcpyAsync(m.acc.p, accum_host, bytes, hipMemcpyHostToDevice, m.stream) : hipMemsetAsync(m.acc.p, 0, bytes, m.stream);
        if (e != hipSuccess) return dev_fail(i, "accumulator set-up", e);
        // shard i: contiguous, disjoint, exhaustive -- earlier handles take the remainder (raytracer_amd/dist.py shard_samples)
        rtc_settings mine = *st;
        mine.sample_begin = st->sample_begin + i * q + std::min(i, r);
        mine.sample_end = mine.sample_begin + q + (i < r ? 1 : 0);
        if (mine.sample_end > mine.sample_begin) {
            const int rc_i = rtc_render_device(scenes[i], cam, &mine, (float *)m.acc.p, m.stream);
            if (rc_i != RTC_OK) { const std::string why = rtc_last_error(); for (int k = 0; k <= i; k++) { (void)hipSetDevice(scenes[k]->device); (void)hipStreamSynchronize(ms[(size_t)k]->stream); } return fail(rc_i, why); }
        }
        if ((e = hipEventRecord(m.rendered, m.stream)) != hipSuccess) return dev_fail(i, "hipEventRecord", e);
    }
    // gather on the root: every copy waits for its own render only (xGMI when the peers see each other, through the host otherwise)
    if ((e = hipSetDevice(root)) != hipSuccess) return dev_fail(0, "hipSetDevice(root)", e);
    std::vector<const float4 *> staged;
    for (int i = 1; i < n_scenes; i++) {
        MultiState &m = *ms[(size_t)i];
        if ((e = hipSetDevice(scenes[i]->device)) != hipSuccess) return dev_fail(i, "hipSetDevice", e);
        if ((e = hipMemcpyPeerAsync(m.stage.p, root, m.acc.p, scenes[i]->device, bytes, m.stream)) != hipSuccess) return dev_fail(i, "hipMemcpyPeerAsync", e);
        if ((e = hipEventRecord(m.staged, m.stream)) != hipSuccess) return dev_fail(i, "hipEventRecord", e);
        if ((e = hipSetDevice(root)) != hipSuccess) return dev_fail(0, "hipSetDevice(root)", e);
        if ((e = hipStreamWaitEvent(ms[0]->stream, m.staged, 0)) != hipSuccess) return dev_fail(i, "hipStreamWaitEvent", e);
        staged.push_back((const float4 *)m.stage.p);
    }
    if (!staged.empty() && (e = launch_accum_add((float4 *)ms[0]->acc.p, staged.data(), (int)staged.size(), n_pix, ms[0]->stream)) != hipSuccess) return dev_fail(0, "sum of the shards", e);
    if ((e = hipMemcpyAsync(accum_host, ms[0]->acc.p, bytes, hipMemcpyDeviceToHost, ms[0]->stream)) != hipSuccess) return dev_fail(0, "download", e);
    if ((e = hipStreamSynchronize(ms[0]->stream)) != hipSuccess) return dev_fail(0, "the render failed on a device", e);   // the one synchronisation of the call
    return RTC_OK;
}

int rtc_tonemap_u8_device(const float *accum_dev, int width, int height, unsigned char *rgb_dev, void *stream) {
    if (!accum_dev || !rgb_dev || width <= 0 || height <= 0) return fail(RTC_ERR_ARG, "rtc_tonemap_u8_device: bad argument");
    if (int rc = check_device_pointer(accum_dev, (size_t)width * (size_t)height * 16, "rtc_tonemap_u8_device: accum_dev")) return rc;
    if (int rc = check_device_pointer(rgb_dev, (size_t)width * (size_t)height * 3, "rtc_tonemap_u8_device: rgb_dev")) return rc;
    HIP_TRY(launch_tonemap_u8((const float4 *)accum_dev, width * height, rgb_dev, (hipStream_t)stream));
    return RTC_OK;
}
int rtc_tonemap_u8(const float *accum_host, int width, int height, unsigned char *rgb_host) {
    if (!accum_host || !rgb_host || width <= 0 || height <= 0) return fail(RTC_ERR_ARG, "rtc_tonemap_u8: bad argument");
    const size_t n = (size_t)width * height;
    DevBuf acc, rgb;
    HIP_TRY(acc.upload(accum_host, n * 4 * sizeof(float)));
    HIP_TRY(hipMalloc(&rgb.p, n * 3));
    int rc = rtc_tonemap_u8_device((const float *)acc.p, width, height, (unsigned char *)rgb.p, nullptr); if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(rgb_host, rgb.p, n * 3, hipMemcpyDeviceToHost));
    return RTC_OK;
}

}  // extern "C"
