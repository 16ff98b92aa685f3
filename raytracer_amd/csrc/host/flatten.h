// flatten.h -- host flattener: pointer graph of include/raytracer.h -> flat tables of rtc_tables.h (product code).
#pragma once
#include <string>
#include <vector>

#include "../rtc_tables.h"
#include "scene_internal.h"

struct FlatScene {
    std::vector<PrimT<double>> prims, bprims;
    std::vector<OpT<double>> ops;
    std::vector<ChainRange> chains;
    std::vector<MaterialT<double>> materials;
    std::vector<TextureT<double>> textures;
    std::vector<LightT<double>> lights;
    std::vector<double> perlin_vec;
    std::vector<int> perlin_perm;
    std::vector<ImageDesc> images;
    std::vector<unsigned char> image_data;
    std::vector<BvhNode> nodes;
    std::vector<int> leaf_prims;
    std::vector<BLeaf> bleaves;     // FP32 traversal leaves, one per slot of leaf_prims
    std::vector<SmallShade> leaf_shade;   // shading record per slot of leaf_prims (rtc_tables.h SmallShade, RTC_SF_GENERAL)
    int bvh_depth = 0;              // levels of the binary BVH (root = 1)
    std::vector<int> box_first;     // first primitive of every hittable box (its 6 sides follow in creation order)
    std::vector<int> box_slots;     // leaf slots where an axis-aligned box record starts (it owns 6 slots: BLeaf, rtc_tables.h)
    std::vector<RefStep> ref_steps; // the reference's own traversal, depth-first (FP64 check path)
    std::vector<double> ref_boxes;  // 6 doubles per RTC_STEP_NODE
    // small-scene pack (empty when the scene is not eligible: see build_small_pack in flatten.cpp)
    std::vector<SmallPair> small_pairs;
    std::vector<SmallGroup> small_groups;
    std::vector<SmallBox> small_boxes;
    std::vector<SmallSphere> small_spheres;
    std::vector<SmallShade> small_shade;
    std::vector<SmallXform> small_xforms;
    bool small_ok = false, small_any_fuzz = false;
    int small_n_rects = 0;         // leaf ids: rects 0 .. small_n_rects-1, then the spheres
    // Lights whose geometry IS a world-space leaf of the pack (mark_small_lights): the leaf's own intersection test then yields the
    // light's pdf for the ray being traced (SmallPair f[12], f[13] = 1 / area per listing; SmallSphere pdf_k).  Derived, not stored.
    bool small_fused = false;
    std::vector<int> wrap_depth;   // per prim: number of translate/rotate wrappers (for rtc_scene_dump_leaves)
    std::vector<int> flipped;      // per prim: parity of flip_face wrappers
    int n_media = 0;
    int leaf_kinds = 0;             // leaf types in the production BVH: bit 0 rects, bit 1 spheres, bit 2 moving spheres
    int media_kind = 0;             // 0 none, 1 every medium is bounded by one sphere (carried in its BLeaf), 2 general boundaries
    double background[3] = {0, 0, 0};
    std::string error;
};

// lights -> leaves of the small-scene pack (deterministic; also run after a scene file is loaded)
void mark_small_lights(FlatScene &fs);

// Walks world/lights once.  Primitive order = the reference's traversal order (list order, box sides in creation
// order, BVH members in the order recorded by bvh_node_init).  Builds the BVH when the scene has more than
// `bvh_threshold` leaves.  Returns false and sets out.error on unsupported graphs.
bool flatten_scene(const hittable_list *world, const hittable_list *lights, const color *background,
                   double time0, double time1, int bvh_threshold, FlatScene &out);
