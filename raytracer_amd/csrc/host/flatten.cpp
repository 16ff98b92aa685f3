// flatten.cpp -- pointer graph -> flat tables + linear BVH (product code, host C++).
//
// Replaces the reference's pointer-chasing scene (hittable_list.c:125-143 scanning hittable* -> void* -> struct,
// bvh_node_hit hittable.c:485-490 recursing through malloc'd nodes) by arrays the GPU can index.
// The reference's wrappers (translate/rotate/flip_face, hittable.c:527-591) become per-primitive "chains" of
// ops kept in the reference's nesting order so the FP64 verify kernel can replay them op by op.
#include "flatten.h"
#include <atomic>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <unordered_map>

namespace {

struct Box { double lo[3], hi[3]; };
static Box box_empty() { Box b; for (int a = 0; a < 3; a++) { b.lo[a] = HUGE_VAL; b.hi[a] = -HUGE_VAL; } return b; }
static void box_grow(Box &b, const double p[3]) { for (int a = 0; a < 3; a++) { b.lo[a] = std::min(b.lo[a], p[a]); b.hi[a] = std::max(b.hi[a], p[a]); } }
static void box_merge(Box &b, const Box &o) { box_grow(b, o.lo); box_grow(b, o.hi); }

struct Flattener {
    FlatScene &fs;
    double time0, time1;
    std::vector<OpT<double>> cur;                       // wrapper stack, outermost first
    std::map<std::vector<double>, int> chain_ids;       // serialized ops -> chain id
    std::unordered_map<const material *, int> mat_ids;
    std::unordered_map<const texture *, int> tex_ids;
    std::vector<Box> prim_boxes, bprim_boxes;
    explicit Flattener(FlatScene &f, double t0, double t1) : fs(f), time0(t0), time1(t1) {
        fs.chains.push_back(ChainRange{0, 0});
        chain_ids[std::vector<double>()] = 0;
    }

    int chain_id() {
        if (cur.empty()) return 0;                        // no wrapper: the empty chain (nearly every leaf of a large scene)
        std::vector<double> key;
        for (const auto &o : cur) { key.push_back(o.type); key.push_back(o.axis); key.push_back(o.a); key.push_back(o.b); key.push_back(o.c); }
        auto it = chain_ids.find(key);
        if (it != chain_ids.end()) return it->second;
        int id = (int)fs.chains.size();
        fs.chains.push_back(ChainRange{(int)fs.ops.size(), (int)cur.size()});
        for (const auto &o : cur) fs.ops.push_back(o);
        chain_ids[key] = id;
        return id;
    }
    int texture_id(const texture *t) {
        auto it = tex_ids.find(t);
        if (it != tex_ids.end()) return it->second;
        TextureT<double> d; std::memset(&d, 0, sizeof d);
        d.type = (int)t->kind;
        switch (t->kind) {
        case RT_TEX_SOLID: d.v[0] = t->value.x; d.v[1] = t->value.y; d.v[2] = t->value.z; break;
        case RT_TEX_CHECKER: d.a = texture_id(t->odd); d.b = texture_id(t->even); break;
        case RT_TEX_NOISE:
            d.a = (int)(fs.perlin_perm.size() / 768); d.v[0] = t->scale;
            fs.perlin_vec.insert(fs.perlin_vec.end(), t->ranvec, t->ranvec + 768);
            fs.perlin_perm.insert(fs.perlin_perm.end(), t->perm, t->perm + 768);
            break;
        case RT_TEX_IMAGE: {
            d.a = (int)fs.images.size();
            ImageDesc im; im.width = t->width; im.height = t->height; im.offset = (uint32_t)fs.image_data.size(); im.pad = 0;
            fs.images.push_back(im);
            fs.image_data.insert(fs.image_data.end(), t->pixels, t->pixels + (size_t)t->width * t->height * 3);
            while (fs.image_data.size() % 16) fs.image_data.push_back(0);
            break;
        }
        }
        int id = (int)fs.textures.size();
        fs.textures.push_back(d); tex_ids[t] = id;
        return id;
    }
    int material_id(const material *m) {
        auto it = mat_ids.find(m);
        if (it != mat_ids.end()) return it->second;
        MaterialT<double> d; std::memset(&d, 0, sizeof d);
        d.type = (int)m->kind; d.tex = -1;
        if (m->kind == RT_MAT_METAL) { d.albedo[0] = m->albedo.x; d.albedo[1] = m->albedo.y; d.albedo[2] = m->albedo.z; d.param = m->param; }
        else if (m->kind == RT_MAT_DIELECTRIC) d.param = m->param;
        else d.tex = texture_id(m->tex);
        int id = (int)fs.materials.size();
        fs.materials.push_back(d); mat_ids[m] = id;
        return id;
    }

    // world-space bounds of a leaf given its object-space box: replay the wrapper stack inside-out
    Box to_world(Box b) const {
        for (int i = (int)cur.size() - 1; i >= 0; --i) {
            const auto &o = cur[i];
            if (o.type == RTC_OP_TRANSLATE) { double off[3] = {o.a, o.b, o.c}; for (int a = 0; a < 3; a++) { b.lo[a] += off[a]; b.hi[a] += off[a]; } }
            else if (o.type == RTC_OP_ROTATE) {
                Box r = box_empty();
                for (int c = 0; c < 8; c++) {
                    double p[3] = {(c & 1) ? b.hi[0] : b.lo[0], (c & 2) ? b.hi[1] : b.lo[1], (c & 4) ? b.hi[2] : b.lo[2]}, q[3] = {p[0], p[1], p[2]};
                    const double s = o.a, cs = o.b;   // object -> parent uses (+sin, cos), hittable.c:554
                    if (o.axis == RTC_ROT_X) { q[0] = cs * p[0] - s * p[1]; q[1] = s * p[0] + cs * p[1]; }
                    else if (o.axis == RTC_ROT_Y) { q[0] = cs * p[0] + s * p[2]; q[2] = -s * p[0] + cs * p[2]; }
                    else { q[1] = cs * p[1] - s * p[2]; q[2] = s * p[1] + cs * p[2]; }
                    box_grow(r, q);
                }
                b = r;
            }
        }
        return b;
    }

    // kind 2 = a PRIMS step that came out of a BVH splice: its neighbours' skip counts are fixed, never extend it
    enum { STEP_SEALED = 2 };
    void step_prim(std::vector<RefStep> *steps, int id) {
        if (!steps) return;
        if (!steps->empty() && steps->back().kind == RTC_STEP_PRIMS && steps->back().a + steps->back().b == id) { steps->back().b++; return; }
        steps->push_back(RefStep{RTC_STEP_PRIMS, 0, id, 1});
    }

    bool leaf(const hittable *h, bool boundary, std::vector<RefStep> *steps) {
        PrimT<double> p; std::memset(&p, 0, sizeof p);
        Box b = box_empty();
        if (h->kind == RT_H_SPHERE) {
            p.type = RTC_PRIM_SPHERE; p.d[0] = h->c0.x; p.d[1] = h->c0.y; p.d[2] = h->c0.z; p.d[3] = h->r;
            double r = std::fabs(h->r), lo[3] = {h->c0.x - r, h->c0.y - r, h->c0.z - r}, hi[3] = {h->c0.x + r, h->c0.y + r, h->c0.z + r};
            box_grow(b, lo); box_grow(b, hi);
        } else if (h->kind == RT_H_MOVING_SPHERE) {
            p.type = RTC_PRIM_MSPHERE;
            p.d[0] = h->c0.x; p.d[1] = h->c0.y; p.d[2] = h->c0.z; p.d[3] = h->c1.x; p.d[4] = h->c1.y; p.d[5] = h->c1.z;
            p.d[6] = h->t0; p.d[7] = h->t1; p.d[8] = h->r;
            double r = std::fabs(h->r);
            for (double t : {time0, time1}) {
                double k = (t - h->t0) / (h->t1 - h->t0);
                double c[3] = {(h->c1.x - h->c0.x) * k + h->c0.x, (h->c1.y - h->c0.y) * k + h->c0.y, (h->c1.z - h->c0.z) * k + h->c0.z};
                double lo[3] = {c[0] - r, c[1] - r, c[2] - r}, hi[3] = {c[0] + r, c[1] + r, c[2] + r};
                box_grow(b, lo); box_grow(b, hi);
            }
        } else if (h->kind == RT_H_RECT) {
            p.type = RTC_PRIM_RECT; p.axis = h->axis;
            p.d[0] = h->lo.x; p.d[1] = h->hi.x; p.d[2] = h->lo.y; p.d[3] = h->hi.y; p.d[4] = h->lo.z; p.d[5] = h->hi.z; p.d[6] = h->k;
            double lo[3] = {h->lo.x, h->lo.y, h->lo.z}, hi[3] = {h->hi.x, h->hi.y, h->hi.z};
            int thin = h->axis == RTC_AXIS_XY ? 2 : (h->axis == RTC_AXIS_XZ ? 1 : 0);
            lo[thin] = h->k - 0.0001; hi[thin] = h->k + 0.0001;
            box_grow(b, lo); box_grow(b, hi);
        } else return false;
        p.mat = material_id(h->mat);
        p.chain = chain_id();
        (boundary ? fs.bprims : fs.prims).push_back(p);
        (boundary ? bprim_boxes : prim_boxes).push_back(to_world(b));
        if (!boundary) {
            step_prim(steps, (int)fs.prims.size() - 1);
            int depth = 0, flips = 0;
            for (const auto &o : cur) { if (o.type == RTC_OP_FLIP) flips ^= 1; else depth++; }
            fs.wrap_depth.push_back(depth); fs.flipped.push_back(flips);
        }
        return true;
    }

    // bvh_node_constructor (hittable.c:163-215, F8) replayed over the ORIGINAL member order: node box =
    // surrounding_box(left box, right box) (aabb.c:67-84) with the (t0,t1) given to bvh_node_init; span 1 -> the same
    // member is both children (it is visited twice); span 2 -> ordered by box min.x at times (0,0).
    struct Sub { std::vector<RefStep> steps; rt_box box; };
    static rt_box surround(const rt_box &a, const rt_box &b) {
        rt_box r;
        r.lo.x = std::fmin(a.lo.x, b.lo.x); r.lo.y = std::fmin(a.lo.y, b.lo.y); r.lo.z = std::fmin(a.lo.z, b.lo.z);
        r.hi.x = std::fmax(a.hi.x, b.hi.x); r.hi.y = std::fmax(a.hi.y, b.hi.y); r.hi.z = std::fmax(a.hi.z, b.hi.z);
        return r;
    }
    bool ref_tree(const hittable *bvh, const std::vector<std::vector<RefStep>> &msteps, int chain, int start, int end, Sub &out) {
        const hittable_list *src = bvh->members;
        const int span = end - start;
        Sub l, r;
        auto member = [&](int i, Sub &s) {
            if (!rt_bounding_box(src->objs[i], bvh->t0, bvh->t1, &s.box)) { fs.error = "bvh member without a bounding box"; return false; }
            s.steps = msteps[i];
            return true;
        };
        if (span == 1) { if (!member(start, l)) return false; r = l; }
        else if (span == 2) {
            rt_box a, b; int lt = 0;
            if (rt_bounding_box(src->objs[start], 0, 0, &a) && rt_bounding_box(src->objs[start + 1], 0, 0, &b)) lt = a.lo.x < b.lo.x;
            if (!member(lt ? start : start + 1, l) || !member(lt ? start + 1 : start, r)) return false;
        } else {
            const int mid = start + span / 2;
            if (!ref_tree(bvh, msteps, chain, start, mid, l) || !ref_tree(bvh, msteps, chain, mid, end, r)) return false;
        }
        out.box = surround(l.box, r.box);
        out.steps.clear();
        out.steps.reserve(1 + l.steps.size() + r.steps.size());
        out.steps.push_back(RefStep{RTC_STEP_NODE, chain, (int)(1 + l.steps.size() + r.steps.size()), (int)(fs.ref_boxes.size() / 6)});
        const double bx[6] = {out.box.lo.x, out.box.lo.y, out.box.lo.z, out.box.hi.x, out.box.hi.y, out.box.hi.z};
        fs.ref_boxes.insert(fs.ref_boxes.end(), bx, bx + 6);
        out.steps.insert(out.steps.end(), l.steps.begin(), l.steps.end());
        out.steps.insert(out.steps.end(), r.steps.begin(), r.steps.end());
        return true;
    }

    bool walk(const hittable *h, bool boundary, std::vector<RefStep> *steps) {
        switch (h->kind) {
        case RT_H_SPHERE: case RT_H_MOVING_SPHERE: case RT_H_RECT: return leaf(h, boundary, steps);
        case RT_H_BOX: {
            const size_t first = fs.prims.size();
            for (int i = 0; i < h->members->index; i++) if (!walk(h->members->objs[i], boundary, steps)) return false;
            if (!boundary && fs.prims.size() - first == 6) fs.box_first.push_back((int)first);
            return true;
        }
        case RT_H_BVH: {
            if (!steps) { for (int i = 0; i < h->bvh_count; i++) if (!walk(h->members->objs[h->bvh_order[i]], boundary, nullptr)) return false; return true; }
            // leaves are numbered in visit order (bvh_order); each member's own steps are kept aside and spliced into
            // the reference's tree afterwards
            std::vector<std::vector<RefStep>> msteps((size_t)h->members->index);
            for (int i = 0; i < h->bvh_count; i++) {
                const int m = h->bvh_order[i];
                if (!walk(h->members->objs[m], boundary, &msteps[(size_t)m])) return false;
                for (auto &st : msteps[(size_t)m]) if (st.kind == RTC_STEP_PRIMS) st.kind = STEP_SEALED;
            }
            if (h->members->index == 0) return true;
            Sub root;
            if (!ref_tree(h, msteps, chain_id(), 0, h->members->index, root)) return false;
            steps->insert(steps->end(), root.steps.begin(), root.steps.end());
            return true;
        }
        case RT_H_TRANSLATE: case RT_H_ROTATE: case RT_H_FLIP: {
            OpT<double> o; std::memset(&o, 0, sizeof o);
            if (h->kind == RT_H_TRANSLATE) { o.type = RTC_OP_TRANSLATE; o.a = h->offset.x; o.b = h->offset.y; o.c = h->offset.z; }
            else if (h->kind == RT_H_ROTATE) { o.type = RTC_OP_ROTATE; o.axis = h->axis; o.a = h->sin_theta; o.b = h->cos_theta; }
            else o.type = RTC_OP_FLIP;
            cur.push_back(o);
            bool ok = walk(h->child, boundary, steps);
            cur.pop_back();
            return ok;
        }
        case RT_H_MEDIUM: {
            if (boundary) { fs.error = "constant_medium used as the boundary of another constant_medium is not supported"; return false; }
            PrimT<double> p; std::memset(&p, 0, sizeof p);
            p.type = RTC_PRIM_MEDIUM; p.d[0] = h->neg_inv_density; p.mat = material_id(h->mat); p.chain = chain_id();
            p.bnd_first = (int)fs.bprims.size();
            size_t first_box = bprim_boxes.size();
            if (!walk(h->child, true, nullptr)) return false;
            p.bnd_count = (int)fs.bprims.size() - p.bnd_first;
            Box b = box_empty();
            for (size_t i = first_box; i < bprim_boxes.size(); i++) box_merge(b, bprim_boxes[i]);
            fs.prims.push_back(p); prim_boxes.push_back(b);
            step_prim(steps, (int)fs.prims.size() - 1);
            int depth = 0, flips = 0;
            for (const auto &o : cur) { if (o.type == RTC_OP_FLIP) flips ^= 1; else depth++; }
            fs.wrap_depth.push_back(depth); fs.flipped.push_back(flips);
            fs.n_media++;
            return true;
        }
        }
        fs.error = "unknown hittable kind"; return false;
    }
};

// ---- binned-SAH BVH over world-space leaf boxes -------------------------------------------------
// An item is one primitive, or one axis-aligned box whose six rects become ONE traversal record (BLeaf): the record sits
// in the slot of face 0 and the six faces own six consecutive slots in the order x-lo x-hi y-lo y-hi z-lo z-hi, so a
// hit on face f is reported as slot + f and shading finds the face's own rect through leaf_prims as before.
struct BuildItem { int prim; int faces[6]; bool is_box; };
struct Builder {
    const std::vector<Box> &boxes;                      // per item
    std::vector<int> &idx;                              // item order, shared: a builder only touches the range it is given
    std::vector<BvhNode> &nodes;
    std::vector<int> &leaf_prims;
    const std::vector<BuildItem> &items;
    std::vector<int> &box_slots;
    static constexpr int kBins = 16;
    // Records per leaf.  Spheres-only trees: 2 (their leaf test is cheap, and more nodes cost the 1M-sphere scene 3 %); trees with box records, quads,
    // moving spheres or media: 1 (a wasted leaf test there costs what a node step costs: final scene -3 %).  RTC_BVH_LEAF=n forces one value.
    int leaf_max = 2;
    static int leaf_policy(bool spheres_only) {
        if (const char *e = std::getenv("RTC_BVH_LEAF")) { const int k = std::atoi(e); if (k >= 1 && k <= 8) return k; }
        return spheres_only ? 2 : 1;
    }
#define kLeafMax leaf_max
    struct Child { int child, count; Box box; };
    // Parallel build (build_bvh_parallel): the builder of the top of the tree does not descend into ranges of at most
    // `defer_below` items; it records them and leaves (count = kDeferred, child = index into `deferred`) in the parent.
    struct Deferred { int lo, hi, depth; };
    static constexpr int kDeferred = -1;
    int defer_below = 0;
    std::vector<Deferred> deferred;

    static double area(const Box &b) { double d[3] = {b.hi[0] - b.lo[0], b.hi[1] - b.lo[1], b.hi[2] - b.lo[2]}; return 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]); }
    Box range_box(int lo, int hi) const { Box b = box_empty(); for (int i = lo; i < hi; i++) box_merge(b, boxes[idx[i]]); return b; }
    static float down(double v) { float f = (float)v; if ((double)f > v) f = std::nextafterf(f, -HUGE_VALF); return f - std::fabs(f) * 1e-6f - 1e-6f; }
    static float up(double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, HUGE_VALF); return f + std::fabs(f) * 1e-6f + 1e-6f; }

    Child make_leaf(int lo, int hi, const Box &b) {
        Child c; c.child = (int)leaf_prims.size(); c.count = hi - lo; c.box = b;
        std::sort(idx.begin() + lo, idx.begin() + hi, [&](int x, int y) { return items[x].prim < items[y].prim; });
        for (int i = lo; i < hi; i++) {
            const BuildItem &it = items[idx[i]];
            if (!it.is_box) { leaf_prims.push_back(it.prim); continue; }
            box_slots.push_back((int)leaf_prims.size());
            for (int f = 0; f < 6; f++) leaf_prims.push_back(it.faces[f]);
        }
        return c;
    }
    int max_depth = 0;
    Child build(int lo, int hi, int depth = 1) {
        if (depth > max_depth) max_depth = depth;
        Box b = range_box(lo, hi);
        int n = hi - lo;
        if (n <= defer_below) { Child c; c.child = (int)deferred.size(); c.count = kDeferred; c.box = b; deferred.push_back({lo, hi, depth}); return c; }
        if (n <= kLeafMax) return make_leaf(lo, hi, b);
        // centroid bounds
        Box cb = box_empty();
        for (int i = lo; i < hi; i++) { const Box &x = boxes[idx[i]]; double c[3] = {0.5 * (x.lo[0] + x.hi[0]), 0.5 * (x.lo[1] + x.hi[1]), 0.5 * (x.lo[2] + x.hi[2])}; box_grow(cb, c); }
        int best_axis = -1, best_bin = -1; double best_cost = HUGE_VAL;
        // past level 40 only median splits: what is left then adds at most log2(n) levels, and the traversal stack
        // (one entry per level, RTC_BVH_MAX_DEPTH) can never overflow
        for (int a = 0; a < 3 && depth <= 40; a++) {
            double ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0) || !std::isfinite(ext)) continue;
            Box bins[kBins]; int cnt[kBins];
            for (int k = 0; k < kBins; k++) { bins[k] = box_empty(); cnt[k] = 0; }
            for (int i = lo; i < hi; i++) {
                const Box &x = boxes[idx[i]];
                int k = (int)(kBins * ((0.5 * (x.lo[a] + x.hi[a]) - cb.lo[a]) / ext)); k = std::min(std::max(k, 0), kBins - 1);
                cnt[k]++; box_merge(bins[k], x);
            }
            double right_area[kBins]; int right_cnt[kBins];
            Box acc = box_empty(); int c = 0;
            for (int k = kBins - 1; k > 0; k--) { if (cnt[k]) box_merge(acc, bins[k]); c += cnt[k]; right_area[k] = c ? area(acc) : 0; right_cnt[k] = c; }
            acc = box_empty(); c = 0;
            for (int k = 0; k < kBins - 1; k++) {
                if (cnt[k]) box_merge(acc, bins[k]);
                c += cnt[k];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                double cost = c * area(acc) + right_cnt[k + 1] * right_area[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
            }
        }
        int mid;
        if (best_axis < 0) mid = lo + n / 2;   // all centroids coincide: split the range in the middle
        else {
            const int a = best_axis; const double ext = cb.hi[a] - cb.lo[a];
            auto it = std::partition(idx.begin() + lo, idx.begin() + hi, [&](int p) {
                const Box &x = boxes[p];
                int k = (int)(kBins * ((0.5 * (x.lo[a] + x.hi[a]) - cb.lo[a]) / ext)); k = std::min(std::max(k, 0), kBins - 1);
                return k <= best_bin;
            });
            mid = (int)(it - idx.begin());
            if (mid == lo || mid == hi) mid = lo + n / 2;
        }
        int me = (int)nodes.size();
        nodes.push_back(BvhNode());
        Child l = build(lo, mid, depth + 1), r = build(mid, hi, depth + 1);
        BvhNode &nd = nodes[me];
        for (int a = 0; a < 3; a++) { nd.lo0[a] = down(l.box.lo[a]); nd.hi0[a] = up(l.box.hi[a]); nd.lo1[a] = down(r.box.lo[a]); nd.hi1[a] = up(r.box.hi[a]); }
        nd.child0 = l.count > 0 ? rtc_leaf_ref(l.child, l.count) : l.child; nd.count0 = l.count;
        nd.child1 = r.count > 0 ? rtc_leaf_ref(r.child, r.count) : r.child; nd.count1 = r.count;
        Child c; c.child = me; c.count = 0; c.box = b;
        return c;
    }
};

// The same tree as Builder::build(0, n) alone would make -- same partitions, same pre-order node numbering, same leaf slot
// order -- built by several threads: one builder makes the top of the tree and defers every range of at most n / 256
// items; the deferred ranges (disjoint pieces of the shared item order) are built independently into their own vectors;
// a last pass walks the top in pre-order and splices the pieces in, shifting node indices and leaf slots.  1M spheres:
// 1.6 s -> 0.4 s on 8 cores; the scene files of the two builders are byte-identical (tests/test_bvh_host.py).
struct BvhPiece { std::vector<BvhNode> nodes; std::vector<int> leaf_prims, box_slots; Builder::Child root; int max_depth = 0; };
Builder::Child build_bvh_parallel(const std::vector<Box> &item_boxes, const std::vector<BuildItem> &items, std::vector<int> &idx, int threads,
                                  std::vector<BvhNode> &nodes, std::vector<int> &leaf_prims, std::vector<int> &box_slots, int &max_depth, int leaf_max,
                                  int first = 0, int depth0 = 1) {
    // items idx[first ..): what lies before `first` (an item hung off the root by the caller) already has its node and leaf slots in the tables
    const int n = (int)idx.size() - first;
    std::vector<BvhNode> top_nodes; std::vector<int> top_leaves, top_boxes;   // the top builder makes no leaves: every small range is deferred
    Builder top{item_boxes, idx, top_nodes, top_leaves, items, top_boxes};
    top.leaf_max = leaf_max;
    top.defer_below = std::max(n / 256, leaf_max);
    const Builder::Child top_root = top.build(first, first + n, depth0);
    std::vector<BvhPiece> pieces(top.deferred.size());
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (size_t i = next.fetch_add(1); i < pieces.size(); i = next.fetch_add(1)) {
            BvhPiece &pc = pieces[i]; const Builder::Deferred &d = top.deferred[i];
            Builder b{item_boxes, idx, pc.nodes, pc.leaf_prims, items, pc.box_slots};
            b.leaf_max = leaf_max;
            pc.nodes.reserve((size_t)(d.hi - d.lo));
            pc.root = b.build(d.lo, d.hi, d.depth); pc.max_depth = b.max_depth;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(worker);
    worker();
    for (auto &th : pool) th.join();
    max_depth = top.max_depth;
    for (const BvhPiece &pc : pieces) max_depth = std::max(max_depth, pc.max_depth);
    // splice, in pre-order
    if (first == 0) { nodes.clear(); leaf_prims.clear(); box_slots.clear(); }
    nodes.reserve((size_t)n); leaf_prims.reserve((size_t)n);
    auto shift_ref = [](int child, int count, int node_base, int leaf_base) {   // a child reference of a piece, moved to its place in the whole
        if (count == 0) return child + node_base;
        const int v = ~child; return rtc_leaf_ref((v >> 3) + leaf_base, (v & 7) + 1);
    };
    auto emit_piece = [&](int id) -> Builder::Child {
        const BvhPiece &pc = pieces[(size_t)id];
        const int node_base = (int)nodes.size(), leaf_base = (int)leaf_prims.size();
        for (BvhNode nd : pc.nodes) {
            if (nd.child0 != RTC_BVH_EMPTY) nd.child0 = shift_ref(nd.child0, nd.count0, node_base, leaf_base);
            if (nd.child1 != RTC_BVH_EMPTY) nd.child1 = shift_ref(nd.child1, nd.count1, node_base, leaf_base);
            nodes.push_back(nd);
        }
        for (int p : pc.leaf_prims) leaf_prims.push_back(p);
        for (int sl : pc.box_slots) box_slots.push_back(sl + leaf_base);
        Builder::Child c = pc.root;
        c.child = pc.root.count > 0 ? pc.root.child + leaf_base : pc.root.child + node_base;
        return c;
    };
    if (top_root.count == Builder::kDeferred) return emit_piece(top_root.child);   // the whole scene was one deferred range
    // explicit stack instead of recursion: (top node, its index in the whole, which child comes next)
    struct Frame { int top, out, side; };
    std::vector<Frame> st;
    const int out0 = (int)nodes.size();
    nodes.push_back(top_nodes[(size_t)top_root.child]);
    st.push_back({top_root.child, out0, 0});
    while (!st.empty()) {
        Frame &f = st.back();
        if (f.side == 2) { st.pop_back(); continue; }
        const BvhNode &tn = top_nodes[(size_t)f.top];
        const int child = f.side == 0 ? tn.child0 : tn.child1, count = f.side == 0 ? tn.count0 : tn.count1;
        const int out = f.out, side = f.side;
        f.side++;
        if (count == Builder::kDeferred) {
            const Builder::Child c = emit_piece(child);
            BvhNode &on = nodes[(size_t)out];
            (side == 0 ? on.child0 : on.child1) = c.count > 0 ? rtc_leaf_ref(c.child, c.count) : c.child;
            (side == 0 ? on.count0 : on.count1) = c.count;
        } else {   // an inner node of the top (the top builder makes no leaves)
            const int me = (int)nodes.size();
            nodes.push_back(top_nodes[(size_t)child]);
            BvhNode &on = nodes[(size_t)out];
            (side == 0 ? on.child0 : on.child1) = me;
            st.push_back({child, me, 0});
        }
    }
    Builder::Child root = top_root; root.child = out0;
    return root;
}

// Wrapper chain of a leaf collapsed to one affine map x_obj = M x_world + b (M a rotation): translate_hit subtracts the
// offset (hittable.c:529), rotate_hit enters with rotatePoint(p, -sin, cos, axis) (hittable.c:542-545, pair choice F10).
// wrapped = the chain holds a translate or rotate (the re-facing quirk F11 applies); flip = parity of flip_face wrappers.
void collapse_chain(const FlatScene &fs, int chain, double M[9], double b[3], bool &wrapped, bool &flip) {
    const ChainRange cr = fs.chains[chain];
    wrapped = false; flip = false;
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(M, I, sizeof I); b[0] = b[1] = b[2] = 0;
    for (int i = 0; i < cr.count; i++) {
        const OpT<double> &o = fs.ops[cr.first + i];
        if (o.type == RTC_OP_FLIP) { flip = !flip; continue; }
        wrapped = true;
        if (o.type == RTC_OP_TRANSLATE) { b[0] -= o.a; b[1] -= o.b; b[2] -= o.c; continue; }
        const double s = -o.a, c = o.b;
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (o.axis == RTC_ROT_X) { R[0] = c; R[1] = -s; R[3] = s; R[4] = c; }
        else if (o.axis == RTC_ROT_Y) { R[0] = c; R[2] = s; R[6] = -s; R[8] = c; }
        else { R[4] = c; R[5] = -s; R[7] = s; R[8] = c; }
        double M2[9], b2[3];
        for (int r = 0; r < 3; r++) { for (int k = 0; k < 3; k++) M2[3 * r + k] = R[3 * r] * M[k] + R[3 * r + 1] * M[3 + k] + R[3 * r + 2] * M[6 + k]; b2[r] = R[3 * r] * b[0] + R[3 * r + 1] * b[1] + R[3 * r + 2] * b[2]; }
        std::memcpy(M, M2, sizeof M2); std::memcpy(b, b2, sizeof b2);
    }
}
// object-space point -> world space: x_w = M^T (x_obj - b)
void to_world_point(const double M[9], const double b[3], const double p[3], double out[3]) {
    const double c[3] = {p[0] - b[0], p[1] - b[1], p[2] - b[2]};
    for (int k = 0; k < 3; k++) out[k] = M[k] * c[0] + M[3 + k] * c[1] + M[6 + k] * c[2];
}

// Traversal leaves of the FP32 production BVH (rtc_tables.h BLeaf), one per slot of leaf_prims.
void build_bleaves(FlatScene &fs) {
    fs.bleaves.clear();
    fs.media_kind = 0; fs.leaf_kinds = 0;
    std::vector<char> box_start(fs.leaf_prims.size(), 0);
    for (int s : fs.box_slots) box_start[(size_t)s] = 1;
    for (size_t slot = 0; slot < fs.leaf_prims.size(); slot++) {
        const int id = fs.leaf_prims[slot];
        const PrimT<double> &p = fs.prims[id];
        BLeaf L; std::memset(&L, 0, sizeof L);
        double M[9], b[3]; bool wrapped, flip;
        collapse_chain(fs, p.chain, M, b, wrapped, flip);
        if (box_start[slot]) {
            // faces x-lo x-hi y-lo y-hi z-lo z-hi live in slots slot .. slot+5; M is the identity here (box_items), so the
            // world-space planes are the object-space ones shifted by the translation: x_w = x_obj - b
            for (int n = 0; n < 3; n++) {
                L.d[n] = (float)(fs.prims[fs.leaf_prims[slot + 2 * n]].d[6] - b[n]);
                L.d[3 + n] = (float)(fs.prims[fs.leaf_prims[slot + 2 * n + 1]].d[6] - b[n]);
            }
            const MaterialT<double> &pm = fs.materials[p.mat];
            const uint32_t tex_kind = pm.tex >= 0 ? (uint32_t)fs.textures[pm.tex].type : 0u;
            L.meta = (uint32_t)RTC_PRIM_RECT | ((uint32_t)p.axis << 2) | ((uint32_t)pm.type << 4) | (tex_kind << 8) | RTC_LEAF_BOX;
            fs.leaf_kinds |= 8;
            fs.bleaves.push_back(L);
            continue;
        }
        if (p.type == RTC_PRIM_SPHERE) {
            double cw[3]; to_world_point(M, b, p.d, cw);
            for (int k = 0; k < 3; k++) L.d[k] = (float)cw[k];
            L.d[3] = (float)p.d[3];
        } else if (p.type == RTC_PRIM_MSPHERE) {
            double c0[3], c1[3]; to_world_point(M, b, p.d, c0); to_world_point(M, b, p.d + 3, c1);
            const double t0 = p.d[6], t1 = p.d[7];
            for (int k = 0; k < 3; k++) { const double B = (c1[k] - c0[k]) / (t1 - t0); L.d[3 + k] = (float)B; L.d[k] = (float)(c0[k] - B * t0); }
            L.d[6] = (float)p.d[8];
        } else if (p.type == RTC_PRIM_RECT) {
            // x_obj = M x_w + b: the plane x_obj[n] = k is row_n(M) . x_w = k - b[n]; same for the two in-plane intervals
            const int n = p.axis == RTC_AXIS_XY ? 2 : (p.axis == RTC_AXIS_XZ ? 1 : 0);
            const int ia = p.axis == RTC_AXIS_YZ ? 1 : 0, ib = p.axis == RTC_AXIS_XY ? 1 : 2;
            const int a0 = p.axis == RTC_AXIS_YZ ? 2 : 0, b0 = p.axis == RTC_AXIS_XY ? 2 : 4;   // slots of lo_a / lo_b in d[]
            for (int k = 0; k < 3; k++) { L.d[k] = (float)M[3 * n + k]; L.d[4 + k] = (float)M[3 * ia + k]; L.d[8 + k] = (float)M[3 * ib + k]; }
            L.d[3] = (float)(p.d[6] - b[n]);
            L.d[7] = (float)(p.d[a0] - b[ia]); L.d[12] = (float)(p.d[a0 + 1] - b[ia]);
            L.d[11] = (float)(p.d[b0] - b[ib]); L.d[13] = (float)(p.d[b0 + 1] - b[ib]);
        }
        uint32_t sphere_medium = 0;
        if (p.type == RTC_PRIM_MEDIUM && p.bnd_count == 1 && fs.bprims[p.bnd_first].type == RTC_PRIM_SPHERE) {
            // boundary = one sphere (both media of the final scene): its world-space centre and radius travel in the leaf,
            // and the two boundary queries of constant_medium_hit become the two roots of one quadratic
            const PrimT<double> &bp = fs.bprims[p.bnd_first];
            double Mb[9], bb[3]; bool w2, f2;
            collapse_chain(fs, bp.chain, Mb, bb, w2, f2);
            double cw[3]; to_world_point(Mb, bb, bp.d, cw);
            for (int k = 0; k < 3; k++) L.d[k] = (float)cw[k];
            L.d[3] = (float)bp.d[3]; L.d[4] = (float)p.d[0];
            sphere_medium = 1u << 7;
        }
        // (the five face slots behind a box record are never reached by the traversal: they do not count as rect leaves)
        bool behind_box = false;
        for (size_t k = 1; k <= 5 && k <= slot; k++) behind_box |= box_start[slot - k] != 0;
        if (!behind_box) fs.leaf_kinds |= p.type == RTC_PRIM_RECT ? 1 : (p.type == RTC_PRIM_SPHERE ? 2 : (p.type == RTC_PRIM_MSPHERE ? 4 : 0));
        if (p.type == RTC_PRIM_MEDIUM) fs.media_kind = std::max(fs.media_kind, sphere_medium ? 1 : 2);
        const MaterialT<double> &pm = fs.materials[p.mat];
        const uint32_t tex_kind = pm.tex >= 0 ? (uint32_t)fs.textures[pm.tex].type : 0u;
        L.meta = (uint32_t)p.type | ((uint32_t)p.axis << 2) | ((uint32_t)pm.type << 4) | sphere_medium | (tex_kind << 8);
        fs.bleaves.push_back(L);
    }
}

// Shading record of every leaf slot (the wavefront extend stage reads it by the hit's slot and shades the hit in its own tail): the hit
// point needs only the ray, the normal a world-space constant (rect: row of the collapsed rotation, like the quad leaf) or centre and
// 1/r (sphere; a moving sphere's centre is A + B time), the face flags follow hit_record_set_facenormal and the wrappers' re-facing
// (F11, flip_face), the colour is resolved when the texture is a solid one; a constant medium needs its colour only.  Anything else --
// a texture that must be evaluated, a chain whose facing the flags cannot express -- is marked RTC_SF_GENERAL and goes through
// prims / chains / materials / textures in k_wf_shade.
void chain_facing(const FlatScene &fs, int chain, bool &normal_ok, bool &front_ok, bool &flip);   // below, with its explanation
void build_leaf_shade(FlatScene &fs) {
    fs.leaf_shade.clear();
    for (int id : fs.leaf_prims) {
        const PrimT<double> &p = fs.prims[id];
        const MaterialT<double> &m = fs.materials[p.mat];
        SmallShade S; std::memset(&S, 0, sizeof S);
        double M[9], b[3]; bool wrapped, flip;
        collapse_chain(fs, p.chain, M, b, wrapped, flip);
        bool normal_ok, front_ok;
        chain_facing(fs, p.chain, normal_ok, front_ok, flip);
        S.meta = (uint32_t)m.type | (wrapped ? RTC_SF_FORCED_FRONT : 0u) | (flip ? RTC_SF_FLIP : 0u);
        const uint32_t tex_kind = m.tex >= 0 ? (uint32_t)fs.textures[m.tex].type : 0u;
        S.meta |= (((uint32_t)m.type * 4u + tex_kind) & 31u) << 16;                                  // k_wf_shade's regrouping key
        bool general = false;
        if (!normal_ok || ((m.type == RTC_MAT_DIELECTRIC || m.type == RTC_MAT_LIGHT) && !front_ok)) general = true;   // replay the chain (resolve_hit)
        if (m.type == RTC_MAT_METAL) { for (int k = 0; k < 3; k++) S.rgb[k] = (float)m.albedo[k]; S.param = (float)m.param; }
        else if (m.type == RTC_MAT_DIELECTRIC) {   // pad[0] = 1 / ir, pad[1] = Schlick's r0 = ((1 - ir) / (1 + ir))^2 (the same for ir and 1 / ir)
            S.rgb[0] = S.rgb[1] = S.rgb[2] = 1.0f; S.param = (float)m.param; S.pad[0] = (float)(1.0 / m.param);
            const double r = (1.0 - m.param) / (1.0 + m.param); S.pad[1] = (float)(r * r);
        }
        else if (m.type == RTC_MAT_LAMBERTIAN && m.tex >= 0 && fs.textures[m.tex].type != RTC_TEX_SOLID && p.type == RTC_PRIM_SPHERE) {   // noise, image, checker: albedo left to k_wf_shade (RTC_SF_TEXTURED)
            S.rgb[0] = S.rgb[1] = S.rgb[2] = 1.0f; std::memcpy(&S.param, &m.tex, sizeof(int)); S.meta |= RTC_SF_TEXTURED;
        }
        else if (m.tex < 0 || fs.textures[m.tex].type != RTC_TEX_SOLID) general = true;
        else for (int k = 0; k < 3; k++) S.rgb[k] = (float)fs.textures[m.tex].v[k];
        if (p.type == RTC_PRIM_RECT) {
            const int e = p.axis == RTC_AXIS_YZ ? 0 : (p.axis == RTC_AXIS_XZ ? 1 : 2);          // outward normal = +axis unit, object space
            for (int k = 0; k < 3; k++) S.g[k] = (float)M[3 * e + k];                           // world normal = row e of M
        } else if (p.type == RTC_PRIM_SPHERE) {
            double cw[3]; to_world_point(M, b, p.d, cw);
            for (int k = 0; k < 3; k++) S.g[k] = (float)cw[k];
            S.g[3] = (float)(1.0 / p.d[3]);
            S.meta |= RTC_SF_SPHERE;
        } else if (p.type == RTC_PRIM_MSPHERE) {   // centre(time) = A + B * time in world space, like the traversal leaf (build_bleaves)
            if (m.type == RTC_MAT_DIELECTRIC) general = true;   // pad[] holds the glass constants
            else {
                double c0[3], c1[3]; to_world_point(M, b, p.d, c0); to_world_point(M, b, p.d + 3, c1);
                const double t0 = p.d[6], t1 = p.d[7];
                for (int k = 0; k < 3; k++) { const double B = (c1[k] - c0[k]) / (t1 - t0); S.pad[k] = (float)B; S.g[k] = (float)(c0[k] - B * t0); }
                S.g[3] = (float)(1.0 / p.d[8]);
                S.meta |= RTC_SF_SPHERE | RTC_SF_MOVING;
            }
        } else {                                   // constant medium: constant_medium_hit's record (normal (1,0,0), front_face 1: hittable.c:578-582)
            S.meta |= RTC_SF_MEDIUM; S.meta &= ~(uint32_t)(RTC_SF_FORCED_FRONT | RTC_SF_FLIP);
            S.g[0] = 1.0f;
            if (m.type != RTC_MAT_ISOTROPIC && m.type != RTC_MAT_LAMBERTIAN && m.type != RTC_MAT_METAL) general = true;   // glass / lamps read front_face against a real surface
        }
        if (general) { S.meta |= RTC_SF_GENERAL; S.meta &= ~(uint32_t)RTC_SF_TEXTURED; }
        fs.leaf_shade.push_back(S);
    }
}

// Can the resolved shading record (SmallShade: a normal faced against the ray, front_face = "wrapped ? true : natural", then
// the flip parity) stand in for resolve_hit's replay of the wrapper chain (rt_core.hpp)?  The reference re-faces the normal
// in every wrapper (F11): translate_hit against its own ray -- which leaves a faced normal alone and sets front_face = 1 --
// but rotate_hit against the OBJECT-space ray with the normal already turned back to the parent space, so behind a rotate
// front_face depends on the angle, and a chain whose outermost wrapper is a rotate can even leave the normal facing along
// the ray.  flip_face toggles front_face where it stands; a wrapper outside it sets front_face anew.
//   normal_ok: no wrapper, or the outermost one is a translate (it re-faces last, against the true ray)
//   front_ok : no rotate in the chain; flip = parity of the flips OUTSIDE the outermost wrapper (all of them without one)
// Lambertian and metal shading never read front_face, so they only need normal_ok.
void chain_facing(const FlatScene &fs, int chain, bool &normal_ok, bool &front_ok, bool &flip) {
    const ChainRange cr = fs.chains[chain];
    int outer = -1; bool rotate = false; flip = false;
    for (int i = 0; i < cr.count; i++) {
        const OpT<double> &o = fs.ops[cr.first + i];
        if (o.type == RTC_OP_FLIP) { if (outer < 0) flip = !flip; continue; }
        if (outer < 0) outer = i;
        rotate |= o.type == RTC_OP_ROTATE;
    }
    normal_ok = outer < 0 || fs.ops[cr.first + outer].type == RTC_OP_TRANSLATE;
    front_ok = !rotate;
}

// Small-scene pack: eligible when every leaf is a rect or a sphere, there are at most 64 of them, and every
// lambertian / light texture is a solid colour (so the colour can be resolved here, once, instead of per hit).
void build_small_pack(FlatScene &fs) {
    fs.small_ok = false;
    const size_t n = fs.prims.size();
    if (n == 0 || n > RTC_SMALL_MAX_LEAVES || fs.n_media) return;
    std::map<int, int> xform_of_chain;            // chain id -> xform id (0 = identity)
    std::vector<SmallXform> xf(1);
    std::vector<std::array<double, 12>> xfd(1);
    { SmallXform &I = xf[0]; for (int i = 0; i < 9; i++) I.m[i] = (i % 4 == 0); I.b[0] = I.b[1] = I.b[2] = 0; xfd[0] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}; }
    auto chain_xform = [&](int chain, bool &wrapped, bool &flip) -> int {
        double M[9], b[3];
        collapse_chain(fs, chain, M, b, wrapped, flip);
        if (!wrapped) return 0;
        auto it = xform_of_chain.find(chain);
        if (it != xform_of_chain.end()) return it->second;
        SmallXform x; std::array<double, 12> xd;
        for (int i = 0; i < 9; i++) { x.m[i] = (float)M[i]; xd[i] = M[i]; }
        for (int i = 0; i < 3; i++) { x.b[i] = (float)b[i]; xd[9 + i] = b[i]; }
        const int id = (int)xf.size();
        xf.push_back(x); xfd.push_back(xd); xform_of_chain[chain] = id;
        return id;
    };
    // one record per source leaf, keyed for the regrouping below
    struct Item { int key; float rect[5]; SmallSphere sph; SmallShade S; };     // key = xform * 4 + axis for rects, 1 << 20 for spheres
    std::vector<Item> items;
    bool any_fuzz = false;
    for (size_t i = 0; i < n; i++) {
        const PrimT<double> &p = fs.prims[i];
        if (p.type != RTC_PRIM_RECT && p.type != RTC_PRIM_SPHERE) return;
        const MaterialT<double> &m = fs.materials[p.mat];
        Item it; std::memset(&it, 0, sizeof it);
        SmallShade &S = it.S;
        bool wrapped, flip;
        const int x = chain_xform(p.chain, wrapped, flip);
        if ((int)xf.size() > RTC_SMALL_MAX_XFORMS) return;
        {   // the record must answer exactly as the wrapper chain would (chain_facing); otherwise the scene takes the general kernel
            bool normal_ok, front_ok;
            chain_facing(fs, p.chain, normal_ok, front_ok, flip);
            const bool reads_front = m.type == RTC_MAT_DIELECTRIC || m.type == RTC_MAT_LIGHT;
            if (!normal_ok || (reads_front && !front_ok)) return;
        }
        const double *M = xfd[x].data(), *b = M + 9;
        S.meta = (uint32_t)m.type | (wrapped ? RTC_SF_FORCED_FRONT : 0u) | (flip ? RTC_SF_FLIP : 0u);
        if (m.type == RTC_MAT_METAL) { for (int k = 0; k < 3; k++) S.rgb[k] = (float)m.albedo[k]; S.param = (float)m.param; any_fuzz |= m.param > 0; }
        else if (m.type == RTC_MAT_DIELECTRIC) {   // pad[0] = 1 / ir, pad[1] = Schlick's r0 = ((1 - ir) / (1 + ir))^2 (the same for ir and 1 / ir)
            S.rgb[0] = S.rgb[1] = S.rgb[2] = 1.0f; S.param = (float)m.param; S.pad[0] = (float)(1.0 / m.param);
            const double r = (1.0 - m.param) / (1.0 + m.param); S.pad[1] = (float)(r * r);
        }
        else {
            if (m.tex < 0 || fs.textures[m.tex].type != RTC_TEX_SOLID) return;
            for (int k = 0; k < 3; k++) S.rgb[k] = (float)fs.textures[m.tex].v[k];
        }
        if (p.type == RTC_PRIM_RECT) {
            it.key = x * 4 + p.axis;
            const int a0 = p.axis == RTC_AXIS_YZ ? 2 : 0, b0 = p.axis == RTC_AXIS_XY ? 2 : 4;   // slots of lo_a / lo_b in d[]
            it.rect[0] = (float)p.d[6]; it.rect[1] = (float)p.d[a0]; it.rect[2] = (float)p.d[b0]; it.rect[3] = (float)p.d[a0 + 1]; it.rect[4] = (float)p.d[b0 + 1];
            const int e = p.axis == RTC_AXIS_YZ ? 0 : (p.axis == RTC_AXIS_XZ ? 1 : 2);          // outward normal = +axis unit, object space
            for (int k = 0; k < 3; k++) S.g[k] = (float)M[3 * e + k];                           // world normal = M^T e  (row e of M)
        } else {
            it.key = 1 << 20;
            // world-space centre: c_w = M^T (c_obj - b)
            const double c[3] = {p.d[0] - b[0], p.d[1] - b[1], p.d[2] - b[2]};
            double cw[3]; for (int k = 0; k < 3; k++) cw[k] = M[k] * c[0] + M[3 + k] * c[1] + M[6 + k] * c[2];
            for (int k = 0; k < 3; k++) { it.sph.c[k] = (float)cw[k]; S.g[k] = (float)cw[k]; }
            it.sph.r2 = (float)(p.d[3] * p.d[3]);
            S.meta |= RTC_SF_SPHERE;
            S.g[3] = (float)(1.0 / p.d[3]);
        }
        items.push_back(it);
    }
    std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.key < b.key; });
    std::vector<SmallPair> pairs; std::vector<SmallGroup> groups; std::vector<SmallSphere> spheres; std::vector<SmallShade> shade; std::vector<SmallBox> boxes;
    const float never[5] = {0.0f, HUGE_VALF, HUGE_VALF, -HUGE_VALF, -HUGE_VALF};
    // leaf id = index into `shade`: the rects in sorted order, then the spheres
    size_t n_rects = 0;
    while (n_rects < items.size() && items[n_rects].key < (1 << 20)) n_rects++;
    for (const Item &it : items) shade.push_back(it.S);
    auto id_bits = [](int id) { float f; std::memcpy(&f, &id, sizeof f); return f; };
    size_t i = 0;
    while (i < n_rects) {
        SmallGroup g; std::memset(&g, 0, sizeof g);
        g.xform = items[i].key / 4; g.first = (int)pairs.size(); g.box = -1;
        { const SmallXform &X = xf[(size_t)g.xform]; g.rot_y = X.m[1] == 0.0f && X.m[3] == 0.0f && X.m[5] == 0.0f && X.m[7] == 0.0f && X.m[4] == 1.0f; }
        size_t end = i;
        while (end < n_rects && items[end].key / 4 == g.xform) end++;
        // ---- the largest set of rects of this space that are faces of ONE axis-aligned box (SmallBox, rtc_tables.h) ----
        // extent of item r on axis e: its plane (lo == hi == k) on its own axis, its bounds on the other two
        auto axis_of = [&](size_t r) { const int ax = items[r].key & 3; return ax == RTC_AXIS_YZ ? 0 : (ax == RTC_AXIS_XZ ? 1 : 2); };   // normal axis: x, y, z
        auto extent = [&](size_t r, int e, float &lo, float &hi) {
            const float *q = items[r].rect; const int n = axis_of(r);
            if (e == n) { lo = hi = q[0]; return; }
            // in-plane axes (a, b): XY -> x,y  XZ -> x,z  YZ -> y,z
            const int a = n == 0 ? 1 : 0;
            if (e == a) { lo = q[1]; hi = q[3]; } else { lo = q[2]; hi = q[4]; }
        };
        std::vector<int> best_face;   // per chosen candidate: item index of face (axis * 2 + side) or -1
        int best_count = 0; float best_lo[3] = {0, 0, 0}, best_hi[3] = {0, 0, 0};
        for (size_t r0 = i; r0 < end; r0++)
            for (size_t r1 = i; r1 < end; r1++) {
                if (axis_of(r0) == axis_of(r1)) continue;
                float lo[3], hi[3]; const int n0 = axis_of(r0);
                for (int e = 0; e < 3; e++) { if (e == n0) extent(r1, e, lo[e], hi[e]); else extent(r0, e, lo[e], hi[e]); }
                if (!(lo[0] < hi[0] && lo[1] < hi[1] && lo[2] < hi[2]) || !std::isfinite(hi[0] - lo[0]) || !std::isfinite(hi[1] - lo[1]) || !std::isfinite(hi[2] - lo[2])) continue;
                std::vector<int> face(6, -1); int count = 0;
                for (size_t r = i; r < end; r++) {
                    const int n = axis_of(r); bool ok = true; int side = -1;
                    for (int e = 0; e < 3 && ok; e++) {
                        float l, h; extent(r, e, l, h);
                        if (e == n) { side = l == lo[e] ? 0 : (l == hi[e] ? 1 : -1); ok = side >= 0; }
                        else ok = l == lo[e] && h == hi[e];
                    }
                    if (ok && face[2 * n + side] < 0) { face[2 * n + side] = (int)r; count++; }
                }
                if (count > best_count) { best_count = count; best_face = face; for (int e = 0; e < 3; e++) { best_lo[e] = lo[e]; best_hi[e] = hi[e]; } }
            }
        std::vector<char> in_box(end - i, 0);
        if (best_count >= 3) {                                        // one slab test beats two or more pair tests
            SmallBox B; std::memset(&B, 0, sizeof B);
            for (int e = 0; e < 3; e++) {
                B.lo[e] = best_lo[e]; B.hi[e] = best_hi[e];
                const int f_lo = best_face[2 * e], f_hi = best_face[2 * e + 1];
                B.ids_lo |= (uint32_t)(f_lo >= 0 ? f_lo : 0xff) << (8 * e);
                B.ids_hi |= (uint32_t)(f_hi >= 0 ? f_hi : 0xff) << (8 * e);
                if (f_lo >= 0) in_box[(size_t)f_lo - i] = 1;
                if (f_hi >= 0) in_box[(size_t)f_hi - i] = 1;
            }
            g.box = (int)boxes.size(); boxes.push_back(B);
        }
        // ---- the other rects: pairs per axis ----
        for (int axis = 0; axis < 3; axis++) {                      // RTC_AXIS_XY, _XZ, _YZ in this order
            std::vector<size_t> run;
            for (size_t r = i; r < end; r++) if ((items[r].key & 3) == axis && !in_box[r - i]) run.push_back(r);
            int count = 0;
            for (size_t k = 0; k < run.size(); k += 2) {
                const float *r0 = items[run[k]].rect, *r1 = k + 1 < run.size() ? items[run[k + 1]].rect : never;
                SmallPair P; std::memset(&P, 0, sizeof P);
                P.f[0] = r0[0]; P.f[1] = r1[0];
                for (int c = 0; c < 4; c++) { P.f[2 + c] = r0[1 + c]; P.f[6 + c] = r1[1 + c]; }
                P.f[10] = id_bits((int)run[k]); P.f[11] = id_bits(k + 1 < run.size() ? (int)run[k + 1] : -1);
                pairs.push_back(P); count++;
            }
            (axis == 0 ? g.n_xy : axis == 1 ? g.n_xz : g.n_yz) = count;
        }
        groups.push_back(g);
        i = end;
    }
    for (i = n_rects; i < items.size(); i++) spheres.push_back(items[i].sph);
    fs.small_n_rects = (int)n_rects;
    if (shade.size() > RTC_SMALL_MAX_LEAVES) return;
    fs.small_boxes = boxes;
    fs.small_pairs = pairs; fs.small_groups = groups; fs.small_spheres = spheres; fs.small_shade = shade; fs.small_xforms = xf;
    fs.small_any_fuzz = any_fuzz; fs.small_ok = true;
}

// Build items of the production BVH: every primitive on its own, except the six rects of a hittable box
// (hittable.c:146-151) that sits in world space up to a translation -- those become one box item (BuildItem).
void box_items(const FlatScene &fs, const std::vector<Box> &prim_boxes, std::vector<BuildItem> &items, std::vector<Box> &item_boxes) {
    std::vector<int> box_of(fs.prims.size(), -1);
    std::vector<BuildItem> boxes;
    for (int first : fs.box_first) {
        BuildItem it; it.prim = first; it.is_box = true;
        for (int f = 0; f < 6; f++) it.faces[f] = -1;
        double M[9], b[3]; bool wrapped, flip;
        collapse_chain(fs, fs.prims[first].chain, M, b, wrapped, flip);
        bool ok = true;
        for (int i = 0; i < 9; i++) ok &= M[i] == (i % 4 == 0 ? 1.0 : 0.0);
        double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
        for (int k = 0; k < 6 && ok; k++) {
            const PrimT<double> &p = fs.prims[first + k];
            ok &= p.type == RTC_PRIM_RECT && p.chain == fs.prims[first].chain;
            const int n = p.axis == RTC_AXIS_XY ? 2 : (p.axis == RTC_AXIS_XZ ? 1 : 0);
            lo[n] = std::min(lo[n], p.d[6]); hi[n] = std::max(hi[n], p.d[6]);
        }
        for (int k = 0; k < 6 && ok; k++) {
            const PrimT<double> &p = fs.prims[first + k];
            const int n = p.axis == RTC_AXIS_XY ? 2 : (p.axis == RTC_AXIS_XZ ? 1 : 0);
            ok &= lo[n] < hi[n];
            const int f = 2 * n + (p.d[6] == hi[n] ? 1 : 0);
            ok &= it.faces[f] < 0;
            it.faces[f] = first + k;
            // each side spans the box on its two in-plane axes
            for (int a = 0; a < 3; a++) if (a != n) ok &= p.d[2 * a] == lo[a] && p.d[2 * a + 1] == hi[a];
        }
        if (!ok) continue;
        for (int k = 0; k < 6; k++) box_of[(size_t)(first + k)] = (int)boxes.size();
        boxes.push_back(it);
    }
    for (size_t i = 0; i < fs.prims.size(); i++) {
        if (box_of[i] < 0) { BuildItem it; it.prim = (int)i; it.is_box = false; for (int f = 0; f < 6; f++) it.faces[f] = -1; items.push_back(it); item_boxes.push_back(prim_boxes[i]); continue; }
        if ((int)i != boxes[(size_t)box_of[i]].prim) continue;
        Box bx = box_empty();
        for (int k = 0; k < 6; k++) box_merge(bx, prim_boxes[i + (size_t)k]);
        items.push_back(boxes[(size_t)box_of[i]]); item_boxes.push_back(bx);
    }
}

}  // namespace

// Light-list entries that are leaves of the small-scene pack.  hittable_pdf_value (hittable.c:760-788) intersects the ray with the
// light's own geometry; when that geometry is a world-space rect of a pair or a sphere of the pack, the bounce loop's intersection
// of the NEXT ray computes the same t / roots anyway, so the kernel takes the pdf from there instead of a second loop over the
// lights (kernels_small.hip).  Fused only if EVERY light with a pdf has such a leaf (a light that is a face of a SmallBox, sits
// behind a wrapper or is not in the world at all keeps the explicit loop for the whole scene).  A light listed twice counts twice.
void mark_small_lights(FlatScene &fs) {
    auto clear = [&] { fs.small_fused = false; for (SmallPair &P : fs.small_pairs) P.f[12] = P.f[13] = 0.0f; for (SmallSphere &sp : fs.small_spheres) sp.pdf_k = 0.0f; };
    clear();
    if (!fs.small_ok || fs.lights.empty() || std::getenv("RTC_SMALL_NO_FUSE")) return;   // (development knob: the A/B partner of the fused path)
    bool all = true;
    for (const LightT<double> &l : fs.lights) {
        if (l.kind == RTC_LIGHT_NONE) continue;                       // pdf 0 either way
        bool found = false;
        if (l.kind == RTC_LIGHT_SPHERE) {
            for (SmallSphere &sp : fs.small_spheres)
                if (!found && sp.c[0] == (float)l.d[0] && sp.c[1] == (float)l.d[1] && sp.c[2] == (float)l.d[2] && sp.r2 == (float)l.d[4]) {
                    sp.pdf_k += (float)(1.0 / (2.0 * 3.14159265358979323846 * l.d[4])); found = true;
                }
        } else if (l.axis >= 0 && l.axis <= 2) {
            // in-plane bounds in the pair's order: XY -> (x, y), XZ -> (x, z), YZ -> (y, z)
            const float a0 = (float)(l.axis == RTC_AXIS_YZ ? l.d[2] : l.d[0]), a1 = (float)(l.axis == RTC_AXIS_YZ ? l.d[3] : l.d[1]);
            const float b0 = (float)(l.axis == RTC_AXIS_XY ? l.d[2] : l.d[4]), b1 = (float)(l.axis == RTC_AXIS_XY ? l.d[3] : l.d[5]);
            for (const SmallGroup &g : fs.small_groups) {
                if (g.xform != 0 || found) continue;                 // world space only
                const int first[3] = {g.first, g.first + g.n_xy, g.first + g.n_xy + g.n_xz}, count[3] = {g.n_xy, g.n_xz, g.n_yz};
                for (int p = first[l.axis]; p < first[l.axis] + count[l.axis] && !found; p++) {
                    SmallPair &P = fs.small_pairs[(size_t)p];
                    for (int slot = 0; slot < 2 && !found; slot++) {
                        int id; std::memcpy(&id, &P.f[10 + slot], sizeof id);
                        if (id < 0) continue;
                        if (P.f[slot] == (float)l.d[6] && P.f[2 + 4 * slot] == a0 && P.f[3 + 4 * slot] == b0 && P.f[4 + 4 * slot] == a1 && P.f[5 + 4 * slot] == b1) {
                            P.f[12 + slot] += (float)l.d[7]; found = true;
                        }
                    }
                }
            }
        }
        all &= found;
    }
    if (!all) { clear(); return; }
    fs.small_fused = true;
}


bool flatten_scene(const hittable_list *world, const hittable_list *lights, const color *background,
                   double time0, double time1, int bvh_threshold, FlatScene &out) {
    const bool timing = std::getenv("RTC_BUILD_TIMING") != nullptr;   // development aid: phase times on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_prev = now();
    auto lap = [&](const char *what) { if (timing) { auto t = now(); std::fprintf(stderr, "flatten_scene: %-12s %.3f s\n", what, std::chrono::duration<double>(t - t_prev).count()); t_prev = t; } };
    if (!world) { out.error = "world list is NULL"; return false; }
    Flattener f(out, time0, time1);
    for (int i = 0; i < world->index; i++)
        if (!f.walk(world->objs[i], false, &out.ref_steps)) { if (out.error.empty()) out.error = "unsupported hittable in world"; return false; }
    for (auto &st : out.ref_steps) if (st.kind == Flattener::STEP_SEALED) st.kind = RTC_STEP_PRIMS;
    if (background) { out.background[0] = background->x; out.background[1] = background->y; out.background[2] = background->z; }
    // lights list: hittable_pdf_value / hittable_random (hittable.c:792-805,851-864) look through flip_face only;
    // anything else contributes pdf 0 and the fixed direction (1,0,0).
    if (lights)
        for (int i = 0; i < lights->index; i++) {
            const hittable *h = lights->objs[i];
            while (h->kind == RT_H_FLIP) h = h->child;
            LightT<double> l; std::memset(&l, 0, sizeof l);
            if (h->kind == RT_H_SPHERE) { l.kind = RTC_LIGHT_SPHERE; l.d[0] = h->c0.x; l.d[1] = h->c0.y; l.d[2] = h->c0.z; l.d[3] = h->r; l.d[4] = h->r * h->r; }
            else if (h->kind == RT_H_RECT) {
                l.kind = RTC_LIGHT_RECT; l.axis = h->axis;
                l.d[0] = h->lo.x; l.d[1] = h->hi.x; l.d[2] = h->lo.y; l.d[3] = h->hi.y; l.d[4] = h->lo.z; l.d[5] = h->hi.z; l.d[6] = h->k;
                {   // d[7] = 1 / area (hittable.c:783), so that the kernels do not recompute a wave-uniform value per ray
                    const double a = h->axis == RTC_AXIS_YZ ? h->hi.y - h->lo.y : h->hi.x - h->lo.x, b = h->axis == RTC_AXIS_XY ? h->hi.y - h->lo.y : h->hi.z - h->lo.z;
                    l.d[7] = 1.0 / (a * b);
                }
            } else l.kind = RTC_LIGHT_NONE;
            out.lights.push_back(l);
        }
    lap("walk");
    build_small_pack(out);
    mark_small_lights(out);
    if ((int)out.prims.size() > bvh_threshold) {
        std::vector<BuildItem> items; std::vector<Box> item_boxes;
        box_items(out, f.prim_boxes, items, item_boxes);
        out.box_slots.clear();
        std::vector<int> idx(items.size());
        for (size_t i = 0; i < idx.size(); i++) idx[i] = (int)i;
        lap("items");
        // RTC_BUILD_THREADS: 1 = the sequential builder; default = the machine's cores, at most 16, for scenes of >= 50k items
        int threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (const char *e = std::getenv("RTC_BUILD_THREADS")) threads = std::max(1, std::atoi(e));
        Builder::Child root; int max_depth = 0;
        bool spheres_only = true;
        for (const BuildItem &it : items) spheres_only = spheres_only && !it.is_box && out.prims[it.prim].type == RTC_PRIM_SPHERE;
        const int leaf_max = Builder::leaf_policy(spheres_only);
        // One item whose box is (nearly) the whole scene's -- the global fog of the Book-2 final scene, a sphere of r = 5000 around a 2000-unit
        // scene -- must not be sorted into the tree by its centroid: every ancestor it gets inherits its box, and every ray then walks that
        // chain of always-hit nodes down to it (final scene: node 201, a dozen levels deep).  It becomes a child of the ROOT, the tree is built
        // over the rest (tight boxes), and the traversals test it ahead of their loop (rtc_api.cpp upload_tables, SceneT::pre_slot).
        // RTC_BVH_HUGE=0: off (A/B, tests).
        int huge = -1;
        bool extract = idx.size() >= 8;
        if (const char *e = std::getenv("RTC_BVH_HUGE")) extract = extract && std::atoi(e) != 0;
        if (extract) {
            Box all = box_empty(); for (const Box &x : item_boxes) box_merge(all, x);
            int n_huge = 0;
            for (size_t i = 0; i < item_boxes.size(); i++)
                if (Builder::area(item_boxes[i]) >= 0.5 * Builder::area(all)) { ++n_huge; huge = (int)i; }
            if (n_huge != 1) huge = -1;
            if (huge >= 0) {                                      // ... and it must dwarf the rest: the others' common box at most a quarter of its area
                Box rest = box_empty(); for (size_t i = 0; i < item_boxes.size(); i++) if ((int)i != huge) box_merge(rest, item_boxes[i]);
                if (!(Builder::area(rest) <= 0.25 * Builder::area(item_boxes[(size_t)huge]))) huge = -1;
            }
        }
        Builder b{item_boxes, idx, out.nodes, out.leaf_prims, items, out.box_slots};
        b.leaf_max = leaf_max;
        int first = 0, depth0 = 1, root_node = -1;
        Builder::Child huge_leaf;
        if (huge >= 0) {
            std::swap(idx[0], idx[(size_t)huge]);                 // (idx is the identity here)
            root_node = (int)out.nodes.size();
            out.nodes.push_back(BvhNode());
            huge_leaf = b.make_leaf(0, 1, item_boxes[(size_t)idx[0]]);
            first = 1; depth0 = 2;
        }
        if (threads > 1 && (items.size() >= 50000 || std::getenv("RTC_BUILD_THREADS")))
            root = build_bvh_parallel(item_boxes, items, idx, threads, out.nodes, out.leaf_prims, out.box_slots, max_depth, leaf_max, first, depth0);
        else {
            out.nodes.reserve(out.prims.size());
            root = b.build(first, (int)idx.size(), depth0); max_depth = b.max_depth;
        }
        if (huge >= 0) {
            const Builder::Child l = huge_leaf, r = root;
            BvhNode &nd = out.nodes[(size_t)root_node];
            for (int a = 0; a < 3; a++) { nd.lo0[a] = Builder::down(l.box.lo[a]); nd.hi0[a] = Builder::up(l.box.hi[a]); nd.lo1[a] = Builder::down(r.box.lo[a]); nd.hi1[a] = Builder::up(r.box.hi[a]); }
            nd.child0 = rtc_leaf_ref(l.child, l.count); nd.count0 = l.count;
            nd.child1 = r.count > 0 ? rtc_leaf_ref(r.child, r.count) : r.child; nd.count1 = r.count;
            root.child = root_node; root.count = 0; root.box = l.box; box_merge(root.box, r.box);
            max_depth = std::max(max_depth, 2);
        }
        lap("bvh build");
        if (root.count > 0) {   // degenerate: everything fell into one leaf -> wrap it in a node
            BvhNode nd; std::memset(&nd, 0, sizeof nd);
            for (int a = 0; a < 3; a++) { nd.lo0[a] = Builder::down(root.box.lo[a]); nd.hi0[a] = Builder::up(root.box.hi[a]); nd.lo1[a] = 1; nd.hi1[a] = -1; }
            nd.child0 = rtc_leaf_ref(root.child, root.count); nd.count0 = root.count; nd.child1 = RTC_BVH_EMPTY; nd.count1 = 0;
            out.nodes.clear(); out.nodes.push_back(nd);
        }
        out.bvh_depth = max_depth;
        // the traversal stack holds one entry per level: a deeper tree must be refused, not clamped (a dropped push is a missed hit).
        // The builder's own bound is 40 SAH levels + log2(n) median levels, i.e. this needs more than 2^24 leaves in one chain.
        if (max_depth > RTC_BVH_MAX_DEPTH) { out.error = "the BVH is " + std::to_string(max_depth) + " levels deep (limit " + std::to_string((int)RTC_BVH_MAX_DEPTH) + ")"; return false; }
        build_bleaves(out);
        lap("leaves");
        build_leaf_shade(out);
        lap("leaf shade");
    }
    return true;
}
