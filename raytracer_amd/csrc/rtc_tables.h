// rtc_tables.h -- flat scene tables shared by the host flattener and the HIP kernels (product code).
//
// One scene = a handful of linear arrays in HBM.  Records are small AoS structs because the two access patterns
// that matter both want whole records: (1) the brute-force primitive loop indexes them with a wave-uniform index,
// which hipcc turns into scalar loads (s_load_dwordx8/x16) that cost no VGPRs and no vector-memory issue slots;
// (2) BVH traversal gathers one 64-byte node or one primitive per lane with dwordx4 vector loads.
// Every table exists in an FP32 flavour (production path tracer) and an FP64 flavour (bit-exact first-hit check).
#pragma once
#include <stdint.h>

enum { RTC_PRIM_SPHERE = 0, RTC_PRIM_MSPHERE = 1, RTC_PRIM_RECT = 2, RTC_PRIM_MEDIUM = 3 };
enum { RTC_OP_TRANSLATE = 0, RTC_OP_ROTATE = 1, RTC_OP_FLIP = 2 };
enum { RTC_MAT_LAMBERTIAN = 0, RTC_MAT_METAL = 1, RTC_MAT_DIELECTRIC = 2, RTC_MAT_LIGHT = 3, RTC_MAT_ISOTROPIC = 4 };
enum { RTC_TEX_SOLID = 0, RTC_TEX_CHECKER = 1, RTC_TEX_NOISE = 2, RTC_TEX_IMAGE = 3 };
enum { RTC_LIGHT_NONE = 0, RTC_LIGHT_SPHERE = 1, RTC_LIGHT_RECT = 2 };
enum { RTC_AXIS_XY = 0, RTC_AXIS_XZ = 1, RTC_AXIS_YZ = 2 };   // rect_axis, hittable.h:69
enum { RTC_ROT_X = 0, RTC_ROT_Y = 1, RTC_ROT_Z = 2 };         // rotation_axis, hittable.h:103

// Leaf primitive in the reference's traversal (visit) order; its index is the primitive id.
//   sphere : d = cx cy cz r
//   msphere: d = c0(3) c1(3) time0 time1 r
//   rect   : d = x0 x1 y0 y1 z0 z1 k          (axis = RTC_AXIS_*)
//   medium : d = neg_inv_density              (bnd_first/bnd_count: boundary leaves in the bprims table)
template <typename R> struct alignas(16) PrimT {
    int type, axis, mat, chain;
    int bnd_first, bnd_count, pad0, pad1;
    R d[12];
};
// One wrapper of the reference's instance stack (translate / rotate / flip_face), outermost first.
template <typename R> struct OpT { int type, axis; R a, b, c; };   // translate: offset xyz; rotate: a=sin b=cos
struct ChainRange { int first, count; };                           // chain 0 is the empty chain
template <typename R> struct MaterialT { int type, tex; R albedo[3]; R param; };   // metal: albedo+fuzz; dielectric: param=ir
template <typename R> struct TextureT { int type, a, b, pad; R v[4]; };            // solid: v=rgb; checker: a=odd b=even; noise: a=perlin slot v[0]=scale; image: a=image slot
template <typename R> struct LightT { int kind, axis, pad0, pad1; R d[8]; };       // entries of the `lights` list, flip_face unwrapped;
                                                                                   // rect: x0 x1 y0 y1 z0 z1 k 1/area; sphere: c(3) r r*r
struct ImageDesc { int width, height; uint32_t offset, pad; };                     // RGB8 texels at image_data + offset
// Two-child BVH node with both child boxes inline (one 64-byte fetch decides both children).
// child >= 0: inner node index (count == 0);  child < 0: leaf reference ~((first << 3) | (count - 1)), a run of
// `count` (1..8) consecutive slots of bleaves / leaf_prims starting at `first`.
struct alignas(16) BvhNode { float lo0[3]; int child0; float hi0[3]; int count0; float lo1[3]; int child1; float hi1[3]; int count1; };
enum { RTC_BVH_EMPTY = (int)0x80000000, RTC_BVH_MAX_DEPTH = 64 };
// The same node as the GPU reads it: every child box as CENTRE and HALF-EXTENT (rounded outward at upload, rtc_api.cpp).  The slab
// test is then three FMAs per axis -- m = c * inv - o * inv, near = m - h |inv|, far = m + h |inv| -- with no per-axis min / max:
// on gfx950 v_min_f32 / v_max_f32 (like v_cmp and v_cndmask) issue in 4 SIMD-cycles, an FMA in 2.2 (tools/ubench/valu_issue,
// profiles/r03b_valu_issue.txt), and the lo / hi form spent two thirds of a node step's cycles on them.
struct alignas(16) BvhNodeCH { float c0[3]; int child0; float h0[3]; int count0; float c1[3]; int child1; float h1[3]; int count1; };
inline int rtc_leaf_ref(int first, int count) { return ~((first << 3) | (count - 1)); }
// Traversal leaf of the FP32 production BVH: 64 bytes, the size of a node, so a lane fetches "what it stands on" with the
// same four 16-byte loads whether that is a node or a leaf.  Stored in BVH leaf order (slot s of bleaves is primitive
// leaf_prims[s]).  Rigid wrappers (translate / rotate) are folded in at flatten time: every leaf is in WORLD space.
//   sphere : d = cx cy cz r
//   msphere: d = A(3) B(3) r,  centre(time) = A + B * time   (moving_sphere_center, hittable.c:455-458)
//   rect   : a quad: d = N(3) k | U(3) a_lo | V(3) b_lo | a_hi b_hi -- plane N.x = k, in-plane coordinates a = U.x, b = V.x
//            with N, U, V the rows of the collapsed rotation (unit axes for an unrotated rect, so the test is then exactly
//            rect_hit's: (k - o_n) / d_n and the two interval checks)
//   box    : (rect type + meta bit 10) the six rects of an axis-aligned hittable box as one record: d = lo(3) hi(3).  The
//            record owns SIX consecutive slots, faces x-lo x-hi y-lo y-hi z-lo z-hi; a hit on face f is slot + f, whose
//            leaf_prims entry is that face's rect; slots 1..5 behind the record hold the faces' quads but no BVH
//            reference reaches them
//   medium : boundary = one sphere (meta bit 7): d = cx cy cz r neg_inv_density;
//            any other boundary: nothing here, the record is prims[leaf_prims[s]]
// meta: bits 0-1 RTC_PRIM_*, bits 2-3 rect axis, bits 4-6 material type, bit 7 sphere-bounded medium, bits 8-9 texture
//       type of the material (material + texture type = the ray regrouping key of the wavefront shade stage)
enum { RTC_LEAF_BOX = 1u << 10,
       RTC_LEAF_PAIR = 1u << 11 };   // DEVICE tables only (rtc_api.cpp upload_tables): this sphere record also holds the sphere of the next slot in d[4..7]
struct alignas(64) BLeaf { float d[15]; uint32_t meta; };

// One step of the REFERENCE's own traversal -- the hittable_list scan (hittable_list.c:125-143) with bvh_node_hit's
// recursion (hittable.c:485-490) unrolled depth-first, left subtree before right.  Used by the FP64 check path only: the
// reference's float-reciprocal aabb_hit (aabb.c:33-63, F9) decides which of two leaves with EQUAL t is visited at all, so
// bit-exact primitive ids need the reference's tree and its culling arithmetic, not just its leaf order.
//   RTC_STEP_PRIMS: scan prims [a, a+b) in order (later equal t replaces);
//   RTC_STEP_NODE : aabb_hit of ref_boxes[b] for the ray taken through chain `chain`; on a miss skip `a` steps.
enum { RTC_STEP_PRIMS = 0, RTC_STEP_NODE = 1 };
struct RefStep { int kind, chain, a, b; };

template <typename R> struct SceneT {
    const PrimT<R> *prims;   int n_prims;
    const PrimT<R> *bprims;  int n_bprims;
    const OpT<R> *ops;       const ChainRange *chains; int n_chains;
    const MaterialT<R> *materials; int n_materials;
    const TextureT<R> *textures;   int n_textures;
    const LightT<R> *lights; int n_lights;
    const R *perlin_vec;     const int *perlin_perm;   // per noise slot: 768 reals, 768 ints
    const ImageDesc *images; const unsigned char *image_data;
    const BvhNodeCH *nodes;  const int *leaf_prims; int n_nodes;   // n_nodes == 0: brute-force loop over prims
    const BLeaf *bleaves;                                          // FP32 production BVH only (NULL in the FP64 tables)
    const struct SmallShade *leaf_shade;                           // per leaf slot: everything shading needs of a resolved leaf, one 48-byte read
    int pre_slot;            // >= 0: a sphere-bounded medium that hangs off the tree's root and encloses the rest of it (the global fog of the Book-2 final scene) --
                             // every ray meets it, so a traversal tests that leaf record FIRST and starts at the root's other child (rtc_api.cpp upload_tables)
    int bvh_root;            // where a traversal starts: node 0, or the leaf run itself when the whole scene is one leaf (its node's second child is RTC_BVH_EMPTY and its box is never tested)
    int bvh_stack;           // entries the traversal stack needs: the levels of the binary tree
    const RefStep *ref_steps; const double *ref_boxes; int n_ref_steps;   // FP64 check path; boxes = min xyz, max xyz
    int n_media;             // media among prims (stochastic hits)
    int leaf_kinds;          // BVH leaves: bit 0 rects, bit 1 spheres, bit 2 moving spheres, bit 3 boxes (RTC_KIND_*, rt_core.hpp)
    int tex_kinds;           // texture types in the scene: bit RTC_TEX_*
    int n_perlin;            // noise slots behind perlin_vec / perlin_perm
    int media_kind;          // BVH leaves: 0 no media, 1 sphere-bounded media only (BLeaf carries them), 2 general boundaries
    int n_bleaves;           // records of bleaves (one per leaf slot)
    int n_general;           // leaf slots whose shading record is marked RTC_SF_GENERAL or RTC_SF_TEXTURED (0: the wavefront never launches k_wf_shade)
    R background[3];
};
// ---- small-scene pack (FP32 production path for scenes of <= 64 rect/sphere leaves with solid textures) --------------
// Everything the bounce loop touches, reshaped so that no pointer is chased per ray.  Leaf ORDER inside the pack is
// free (the FP32 path needs the closest hit, not the reference's tie-break), so rects are grouped by object space and,
// inside a space, by axis: the inner loops have compile-time component choices and no per-leaf branching.
//   SmallPair  : 64 B = TWO rects of one (space, axis) run, read with ONE s_load_dwordx16 (the index is wave-uniform);
//                f = k0 k1 | lo_a0 lo_b0 | hi_a0 hi_b0 | lo_a1 lo_b1 | hi_a1 hi_b1   (a,b = in-plane axes: XY->x,y
//                XZ->x,z  YZ->y,z; (lo_a, lo_b) and (hi_a, hi_b) sit in aligned register pairs for packed FP32 math);
//                | id0 id1 (int bits) | f[12], f[13] = 1 / area when that rect is a light-list entry (else 0);  a run of odd
//                length ends with a never-hit rect (lo = +inf)
//   SmallGroup : 32 B per object space: xform id, first pair, pairs per axis
//   SmallSphere: 32 B, world space (rotations/translations only move the centre): cx cy cz r*r | pdf_k = (times the sphere is
//                listed in `lights`) / (2 pi r^2), 0 for a sphere that is no light-list entry
//   SmallShade : 48 B, gathered per lane from LDS by the winning leaf: world-space outward normal (rect) or centre and
//                1/r (sphere), resolved material (solid colour already looked up), flags
//   SmallXform : the translate/rotate wrapper stack of a leaf collapsed to x_obj = M x_world + b (M a rotation)
// Leaf ids (SmallShade index, `src`): rects first (a pair names its two ids in f[10], f[11] as int bits, a box in its id
// bytes), then sphere s -> n_rects + s.
enum { RTC_SMALL_MAX_LEAVES = 64, RTC_SMALL_MAX_XFORMS = 16 };
enum { RTC_SF_FORCED_FRONT = 1u << 8, RTC_SF_FLIP = 1u << 9, RTC_SF_SPHERE = 1u << 10,   // SmallShade.meta bits; bits 0-7 = material type
       // BVH leaf_shade only:
       RTC_SF_GENERAL = 1u << 11,     // not resolvable here (a texture to evaluate, a wrapper chain to replay) -> prims / materials / textures in k_wf_shade
       RTC_SF_MOVING = 1u << 12,      // moving sphere: centre = g[0..2] + pad[0..2] * time (never glass: pad is the glass constants' place)
       RTC_SF_TEXTURED = 1u << 14,    // lambertian with a noise, image or checker texture on a plain sphere: the record shades the hit with albedo 1 (rgb) and the texture index (param, int
                                      // bits) is left for k_wf_shade, which multiplies the slot's throughput by the texture's value at the new ray's origin -- integrators
                                      // 1 and 2, where the scattered ray does not depend on the albedo (under integrator 0 the record counts as RTC_SF_GENERAL)
       RTC_SF_MEDIUM = 1u << 13 };    // constant medium (isotropic phase function, no surface); bits 16-20: regrouping key of k_wf_shade = material type * 4 + texture type
enum { RTC_WF_KEY_DONE = 255, RTC_WF_KEY_ALBEDO = 32 };   // WfBuffers::key of an entry extend has shaded itself; bit of an entry that waits for its albedo only (RTC_SF_TEXTURED)
struct alignas(64) SmallPair { float f[16]; };
struct alignas(32) SmallGroup { int xform, first, n_xy, n_xz, n_yz, box, rot_y, pad2; };   // box >= 0: SmallBox of this space, tested before the pairs;
                                                                                       // rot_y: the space's map leaves y alone up to a shift (m[1] = m[3] = m[5] = m[7] = 0, m[4] = 1)
// Rects of one object space that are faces of ONE axis-aligned box -- all six (hittable_box_new, hittable.c:139-155) or
// some of them (the five walls of the Cornell room): one slab test (~35 VALU instructions) replaces up to three pair
// tests (~78).  ids_lo / ids_hi = leaf ids of the x, y, z faces at lo / hi, one byte each, 0xff = the box has no such
// face (a ray passes through the opening).
struct alignas(32) SmallBox { float lo[3], hi[3]; uint32_t ids_lo, ids_hi; };
struct alignas(32) SmallSphere { float c[3]; float r2; float pdf_k; float pad[3]; };
struct alignas(16) SmallShade { uint32_t meta; float rgb[3]; float g[4]; float param; float pad[3]; };   // param: fuzz / ir; glass: pad[0] = 1 / ir, pad[1] = Schlick r0
struct SmallXform { float m[9]; float b[3]; };
// What the small-scene kernel reads: the pack above as ONE stream of 64-byte records (built at upload time, rtc_api.cpp
// build_small_program), each fetched with one s_load_dwordx16 by a wave-uniform cursor -- no group -> box -> pair indirection
// (every hop was a dependent scalar load the wave waited for), no loops over empty runs.  On gfx950 a scalar instruction takes an
// issue slot like a vector one (tools/ubench/valu_issue: "mix" rows), so the scalar side of the intersection is part of its cost.
//   w[0] = head: bits 0-7 the record type, then per type:
//   RTC_SR_BOX    : slab test of a SmallBox in the current object space.  w[1] = face ids lo_x | hi_x << 8 | lo_y << 16 | hi_y << 24,
//                   w[2] = lo_z | hi_z << 8 (0xff: no such face), f[3..5] = centre, f[6..8] = half extent (rounded outward)
//   RTC_SR_XBOX_Y : the same box behind a y-rotation + translation that holds nothing else (hittable_box under rotate_y + translate):
//                   f[9..12] = m0 m2 m6 m8, f[13..15] = b.  Self-contained: it does not change the current space
//   RTC_SR_PAIR   : two rects of one axis run.  head bits 8-9 axis (RTC_AXIS_*), 10: second rect present, 11 / 12: rect 0 / 1 is a
//                   light-list entry, 16-23 / 24-31 their leaf ids; f[1], f[2] = plane k; f[3..6] / f[7..10] = centre a, centre b, half a,
//                   half b; f[11], f[12] = 1 / area of a light-list entry
//   RTC_SR_SPHERE : head bits 8-15 leaf id, 16: light-list entry; f[1..3] centre, f[4] = r^2, f[5] = count / (2 pi r^2)   (world space)
//   RTC_SR_XFORM  : the records that follow live in this object space.  head bit 8: y-rotation; f[1..9] = M, f[10..12] = b
// Order: the records that run on the world ray as it is come first, SORTED BY TYPE -- boxes, self-contained XBOX_Y, pairs by axis, spheres
// (SmallScene::n_box ... n_sphere) -- so that the kernel runs one counted loop per type and no type dispatch; then the general object
// spaces, each led by its RTC_SR_XFORM (the only part that looks at a record's type).
enum { RTC_SR_BOX = 1, RTC_SR_XBOX_Y = 2, RTC_SR_PAIR = 3, RTC_SR_SPHERE = 4, RTC_SR_XFORM = 5 };
struct alignas(64) SmallRec { uint32_t w[16]; };
// A light-list entry as the small kernel samples it (hittable.c:821-848), one record per entry of `lights`:
//   w[0] = RTC_LIGHT_RECT  : f[1..3] = the corner (a0, b0) of the rect as a point, f[4..6] / f[7..9] = its two edge vectors:
//                            point(r1, r2) = corner + r1 * edge_a + r2 * edge_b -- no axis selects in the kernel
//   w[0] = RTC_LIGHT_SPHERE: f[1..3] = centre, f[4] = r^2
struct SmallScene {
    const SmallRec *prog; const SmallShade *shade;
    const SmallRec *lights; int n_lights;          // sampling records of the light list (all scenes, also those that are not small)
    int n_box, n_xbox, n_xy, n_xz, n_yz, n_sphere; // world-ray records by type, in this order
    int n_recs;                                    // all records (the general object spaces follow the world-ray ones)
    int n_leaves;                                  // shading records (rects first, then spheres)
    int any_fuzz;                                  // some metal has fuzz > 0
    int fused_lights;                              // every light's pdf comes out of the intersection of the next ray (flatten.cpp mark_small_lights)
};

template <typename R> struct CameraT { R origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3], w[3]; R lens_radius, time0, time1; };
