// scene_io.cpp -- flattened scenes on disk (product code, host only).
//
// SURVEY 8(f)3 "scene (de)serialisation for fixtures": the reference has no scene file -- its scenes are C functions
// (03 main.c:28-78, 02 main.c:76-289) that draw rand() while they build, so the same scene exists only inside the process that built it.
// A FlatScene is a handful of plain tables (flatten.h), so a scene file is those tables, verbatim:
//
//   header  64 B : "RTCSCENE", version, number of sections, payload bytes, CRC-32 of the payload
//   section      : name[24], element size, element count, then the elements (padded to 8 B)
//
// Loading does not trust the file: the CRC catches damage, the section names and element sizes catch a file written by a
// build with other table layouts, and validate_flat_scene() checks every index a kernel follows (primitive -> material
// -> texture -> image, chains -> ops, BVH children and leaf runs, reference traversal steps, the small-scene pack) so that a
// file that passes cannot send a kernel out of its tables or into an endless traversal.
#include "scene_io.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>

namespace {

const char kMagic[8] = {'R', 'T', 'C', 'S', 'C', 'E', 'N', 'E'};
const uint32_t kVersion = 4;   // 4: RTC_SF_TEXTURED records in leaf_shade; 2: glass constants in SmallShade.pad, 1/area in SmallPair, 32-byte SmallSphere (round 2); 3: leaf_shade carries moving spheres, media and the regrouping key

struct Header { char magic[8]; uint32_t version, n_sections; uint64_t payload_bytes; uint32_t payload_crc, pad[9]; };
static_assert(sizeof(Header) == 64, "header layout");
struct SectionHead { char name[24]; uint32_t elem_size, pad; uint64_t count; };
static_assert(sizeof(SectionHead) == 40, "section layout");
struct Scalars { int32_t bvh_depth, small_ok, small_any_fuzz, small_n_rects, n_media, leaf_kinds, media_kind, pad; double background[3]; };

// one list of the tables for the writer and the reader
template <class V> bool visit(FlatScene &fs, Scalars &sc, V &v) {
    return v.vec("prims", fs.prims) && v.vec("bprims", fs.bprims) && v.vec("ops", fs.ops) && v.vec("chains", fs.chains) &&
           v.vec("materials", fs.materials) && v.vec("textures", fs.textures) && v.vec("lights", fs.lights) &&
           v.vec("perlin_vec", fs.perlin_vec) && v.vec("perlin_perm", fs.perlin_perm) && v.vec("images", fs.images) &&
           v.vec("image_data", fs.image_data) && v.vec("nodes", fs.nodes) && v.vec("leaf_prims", fs.leaf_prims) &&
           v.vec("bleaves", fs.bleaves) && v.vec("leaf_shade", fs.leaf_shade) && v.vec("box_first", fs.box_first) &&
           v.vec("box_slots", fs.box_slots) && v.vec("ref_steps", fs.ref_steps) && v.vec("ref_boxes", fs.ref_boxes) &&
           v.vec("small_pairs", fs.small_pairs) && v.vec("small_groups", fs.small_groups) && v.vec("small_boxes", fs.small_boxes) &&
           v.vec("small_spheres", fs.small_spheres) && v.vec("small_shade", fs.small_shade) && v.vec("small_xforms", fs.small_xforms) &&
           v.vec("wrap_depth", fs.wrap_depth) && v.vec("flipped", fs.flipped) && v.one("scalars", sc);
}

struct Writer {
    std::vector<unsigned char> out; uint32_t n = 0;
    void put(const char *name, size_t elem, size_t count, const void *data) {
        SectionHead h; std::memset(&h, 0, sizeof h);
        std::strncpy(h.name, name, sizeof h.name - 1); h.elem_size = (uint32_t)elem; h.count = count;
        const unsigned char *hb = (const unsigned char *)&h;
        out.insert(out.end(), hb, hb + sizeof h);
        const unsigned char *db = (const unsigned char *)data;
        if (count) out.insert(out.end(), db, db + elem * count);
        while (out.size() % 8) out.push_back(0);
        ++n;
    }
    template <class T> bool vec(const char *name, std::vector<T> &v) { put(name, sizeof(T), v.size(), v.data()); return true; }
    template <class T> bool one(const char *name, T &x) { put(name, sizeof(T), 1, &x); return true; }
};

struct Reader {
    const unsigned char *p, *end; std::string &err;
    bool take(const char *name, size_t elem, SectionHead &h) {
        if ((size_t)(end - p) < sizeof h) { err = std::string("scene file ends before section ") + name; return false; }
        std::memcpy(&h, p, sizeof h); p += sizeof h;
        if (std::strncmp(h.name, name, sizeof h.name) != 0) { err = std::string("scene file: expected section ") + name; return false; }
        if (h.elem_size != elem) { err = std::string("scene file: section ") + name + " was written with another table layout"; return false; }
        if (h.count > (uint64_t)(end - p) / (elem ? elem : 1)) { err = std::string("scene file: section ") + name + " is longer than the file"; return false; }
        return true;
    }
    void skip_pad(size_t bytes) { p += bytes; const size_t r = bytes % 8; if (r && (size_t)(end - p) >= 8 - r) p += 8 - r; }
    template <class T> bool vec(const char *name, std::vector<T> &v) {
        SectionHead h; if (!take(name, sizeof(T), h)) return false;
        v.resize((size_t)h.count);
        if (h.count) std::memcpy((void *)v.data(), p, sizeof(T) * (size_t)h.count);
        skip_pad(sizeof(T) * (size_t)h.count);
        return true;
    }
    template <class T> bool one(const char *name, T &x) {
        SectionHead h; if (!take(name, sizeof(T), h)) return false;
        if (h.count != 1) { err = std::string("scene file: section ") + name + " must hold one record"; return false; }
        std::memcpy(&x, p, sizeof(T)); skip_pad(sizeof(T));
        return true;
    }
};

bool in(int i, size_t n) { return i >= 0 && (size_t)i < n; }
bool run_in(long long first, long long count, size_t n) { return first >= 0 && count >= 0 && first + count <= (long long)n; }

bool check_prims(const FlatScene &fs, const std::vector<PrimT<double>> &prims, const char *what, std::string &err) {
    for (size_t i = 0; i < prims.size(); i++) {
        const PrimT<double> &p = prims[i];
        const bool ok = p.type >= RTC_PRIM_SPHERE && p.type <= RTC_PRIM_MEDIUM && in(p.mat, fs.materials.size()) && in(p.chain, fs.chains.size()) &&
                        (p.type != RTC_PRIM_RECT || (p.axis >= 0 && p.axis <= 2)) &&
                        (p.type != RTC_PRIM_MEDIUM || (p.bnd_count >= 1 && run_in(p.bnd_first, p.bnd_count, fs.bprims.size())));
        if (!ok) { err = std::string(what) + " " + std::to_string(i) + ": type, material, chain or boundary range out of range"; return false; }
    }
    return true;
}

}  // namespace

bool validate_flat_scene(const FlatScene &fs, std::string &err) {
#define REQ(cond, msg) do { if (!(cond)) { err = (msg); return false; } } while (0)
    REQ(fs.prims.size() < (1u << 28) && fs.nodes.size() < (1u << 28), "too many primitives");   // leaf references pack first << 3
    REQ(!fs.chains.empty() && fs.chains[0].count == 0, "chain 0 must be the empty chain");
    for (const ChainRange &c : fs.chains) REQ(run_in(c.first, c.count, fs.ops.size()), "a wrapper chain leaves the ops table");
    for (const OpT<double> &o : fs.ops) REQ(o.type >= RTC_OP_TRANSLATE && o.type <= RTC_OP_FLIP && o.axis >= 0 && o.axis <= 2, "wrapper type or axis out of range");
    for (const MaterialT<double> &m : fs.materials) {
        REQ(m.type >= RTC_MAT_LAMBERTIAN && m.type <= RTC_MAT_ISOTROPIC && m.tex >= -1 && m.tex < (int)fs.textures.size(), "material type or texture out of range");
        REQ(m.tex >= 0 || m.type == RTC_MAT_METAL || m.type == RTC_MAT_DIELECTRIC, "a material that reads a texture has none");   // shading evaluates textures[m.tex]
    }
    REQ(fs.perlin_vec.size() == fs.perlin_perm.size() && fs.perlin_vec.size() % 768 == 0, "perlin tables must hold 768 entries per noise texture");
    for (const int v : fs.perlin_perm) REQ(v >= 0 && v < 256, "perlin permutation entry out of range");
    for (const ImageDesc &im : fs.images)
        REQ(im.width > 0 && im.height > 0 && (uint64_t)im.offset + (uint64_t)im.width * (uint64_t)im.height * 3u <= fs.image_data.size(), "an image leaves the texel table");
    for (const TextureT<double> &t : fs.textures) {
        REQ(t.type >= RTC_TEX_SOLID && t.type <= RTC_TEX_IMAGE, "texture type out of range");
        if (t.type == RTC_TEX_CHECKER) REQ(in(t.a, fs.textures.size()) && in(t.b, fs.textures.size()), "checker texture points outside the texture table");
        if (t.type == RTC_TEX_NOISE) REQ(t.a >= 0 && ((size_t)t.a + 1) * 768 <= fs.perlin_vec.size(), "noise texture points outside the perlin tables");
        if (t.type == RTC_TEX_IMAGE) REQ(in(t.a, fs.images.size()), "image texture points outside the image table");
    }
    for (const LightT<double> &l : fs.lights) REQ(l.kind >= RTC_LIGHT_NONE && l.kind <= RTC_LIGHT_RECT && l.axis >= 0 && l.axis <= 2, "light kind or axis out of range");
    if (!check_prims(fs, fs.prims, "primitive", err) || !check_prims(fs, fs.bprims, "boundary primitive", err)) return false;
    for (const PrimT<double> &p : fs.bprims) REQ(p.type != RTC_PRIM_MEDIUM, "a medium cannot bound a medium");
    REQ(fs.wrap_depth.size() == fs.prims.size() && fs.flipped.size() == fs.prims.size(), "per-primitive tables differ in length");
    int media = 0; for (const PrimT<double> &p : fs.prims) media += p.type == RTC_PRIM_MEDIUM;
    REQ(media == fs.n_media, "medium count does not match the primitives");
    for (const int b : fs.box_first) REQ(run_in(b, 6, fs.prims.size()), "a box leaves the primitive table");
    for (int k = 0; k < 3; k++) REQ(std::isfinite(fs.background[k]), "background is not finite");

    // reference traversal steps (FP64 check path, rt_core.hpp closest_hit_ref): every step moves forward
    REQ(fs.ref_boxes.size() % 6 == 0, "reference boxes must hold 6 values each");
    for (size_t i = 0; i < fs.ref_steps.size(); i++) {
        const RefStep &s = fs.ref_steps[i];
        if (s.kind == RTC_STEP_PRIMS) REQ(run_in(s.a, s.b, fs.prims.size()), "a reference step leaves the primitive table");
        else {
            REQ(s.kind == RTC_STEP_NODE, "unknown reference step");
            REQ(s.a >= 1 && i + (size_t)s.a <= fs.ref_steps.size() && in(s.b, fs.ref_boxes.size() / 6) && in(s.chain, fs.chains.size()), "a reference node step is out of range");
        }
    }

    // production BVH: nodes in pre-order (a child lies behind its parent: no cycles), leaf runs inside the leaf tables,
    // depth within what the traversal stack holds
    if (fs.nodes.empty()) REQ(fs.leaf_prims.empty() && fs.bleaves.empty() && fs.leaf_shade.empty() && fs.box_slots.empty(), "leaf tables without a BVH");
    else {
        const size_t nl = fs.leaf_prims.size();
        REQ(fs.bleaves.size() == nl && fs.leaf_shade.size() == nl, "leaf tables differ in length");
        for (const SmallShade &sh : fs.leaf_shade)
            if (sh.meta & RTC_SF_TEXTURED) { int t; std::memcpy(&t, &sh.param, sizeof t); REQ(t >= 0 && (size_t)t < fs.textures.size() && fs.textures[t].type != RTC_TEX_SOLID, "textured shading record without its texture"); }
        for (const int p : fs.leaf_prims) REQ(in(p, fs.prims.size()), "a leaf slot points outside the primitive table");
        std::vector<char> box_start(nl, 0);
        for (const int s : fs.box_slots) { REQ(run_in(s, 6, nl), "a box record leaves the leaf table"); box_start[(size_t)s] = 1; }
        for (size_t s = 0; s < nl; s++) {
            const uint32_t meta = fs.bleaves[s].meta;
            REQ(((meta & RTC_LEAF_BOX) != 0) == (box_start[s] != 0), "box records and box slots disagree");
            const PrimT<double> &p = fs.prims[(size_t)fs.leaf_prims[s]];
            REQ((int)(meta & 3u) == p.type, "a traversal leaf and its primitive differ in type");
        }
        std::vector<int> depth(fs.nodes.size(), 1);
        for (size_t i = fs.nodes.size(); i-- > 0;) {
            const BvhNode &nd = fs.nodes[i];
            const int child[2] = {nd.child0, nd.child1};
            for (int k = 0; k < 2; k++) {
                const int c = child[k];
                if (c == RTC_BVH_EMPTY) continue;
                if (c >= 0) { REQ((size_t)c > i && (size_t)c < fs.nodes.size(), "BVH nodes are not in pre-order"); depth[i] = std::max(depth[i], depth[(size_t)c] + 1); continue; }
                const int first = (~c) >> 3, count = ((~c) & 7) + 1;
                long long slots = 0;   // a box record owns six slots
                for (int j = 0; j < count; j++) { REQ(in(first + (int)slots, nl), "a leaf run leaves the leaf table"); slots += box_start[(size_t)first + (size_t)slots] ? 6 : 1; }
                REQ(run_in(first, slots, nl), "a leaf run leaves the leaf table");
                depth[i] = std::max(depth[i], 2);
            }
        }
        // (a scene that fell into ONE leaf is a root with that leaf and an empty child: recorded depth 1, nothing is ever pushed)
        REQ((depth[0] <= fs.bvh_depth || fs.nodes.size() == 1) && fs.bvh_depth <= RTC_BVH_MAX_DEPTH, "BVH is deeper than its recorded depth");
        REQ(fs.leaf_kinds >= 0 && fs.leaf_kinds <= 15 && fs.media_kind >= 0 && fs.media_kind <= 2, "leaf kinds out of range");
        int kinds = 0, mkind = 0;
        for (size_t s = 0; s < nl; s++) {
            const uint32_t meta = fs.bleaves[s].meta; const int ty = (int)(meta & 3u);
            if (meta & RTC_LEAF_BOX) { kinds |= 8; s += 5; continue; }
            kinds |= ty == RTC_PRIM_RECT ? 1 : (ty == RTC_PRIM_SPHERE ? 2 : (ty == RTC_PRIM_MSPHERE ? 4 : 0));
            if (ty == RTC_PRIM_MEDIUM) mkind = std::max(mkind, (meta & 0x80u) ? 1 : 2);
        }
        REQ((kinds & ~fs.leaf_kinds) == 0 && mkind <= fs.media_kind, "leaf kinds do not cover the leaves");
    }

    // small-scene pack (kernels_small.hip)
    if (!fs.small_ok) REQ(fs.small_pairs.empty() && fs.small_groups.empty() && fs.small_spheres.empty() && fs.small_shade.empty(), "small-scene tables without the small-scene flag");
    else {
        const size_t leaves = fs.small_shade.size();
        REQ(fs.n_media == 0 && leaves >= 1 && leaves <= RTC_SMALL_MAX_LEAVES && leaves == fs.prims.size(), "small-scene pack does not match the primitives");
        REQ(fs.small_n_rects >= 0 && (size_t)fs.small_n_rects + fs.small_spheres.size() == leaves, "small-scene leaf ids do not add up");
        REQ(!fs.small_xforms.empty() && fs.small_xforms.size() <= RTC_SMALL_MAX_XFORMS, "small-scene transform table out of range");
        auto id_ok = [&](uint32_t id) { return id == 0xffu || id < (uint32_t)fs.small_n_rects; };
        for (const SmallGroup &g : fs.small_groups) {
            REQ(in(g.xform, fs.small_xforms.size()) && g.n_xy >= 0 && g.n_xz >= 0 && g.n_yz >= 0 && run_in(g.first, (long long)g.n_xy + g.n_xz + g.n_yz, fs.small_pairs.size()) &&
                g.box >= -1 && g.box < (int)fs.small_boxes.size(), "a small-scene group is out of range");
        }
        for (const SmallPair &p : fs.small_pairs) {
            int id[2]; std::memcpy(id, &p.f[10], sizeof id);
            REQ(in(id[0], (size_t)fs.small_n_rects) && (in(id[1], (size_t)fs.small_n_rects) || !(p.f[6] < 3.0e38f)), "a small-scene rect id is out of range");
        }
        for (const SmallBox &b : fs.small_boxes)
            for (int k = 0; k < 3; k++) REQ(id_ok((b.ids_lo >> (8 * k)) & 0xffu) && id_ok((b.ids_hi >> (8 * k)) & 0xffu), "a small-scene box face id is out of range");
    }
#undef REQ
    return true;
}

bool save_flat_scene(const FlatScene &fs_c, const char *path, std::string &err) {
    FlatScene &fs = const_cast<FlatScene &>(fs_c);   // the visitor is shared with the reader; the writer only reads
    Scalars sc; std::memset(&sc, 0, sizeof sc);
    sc.bvh_depth = fs.bvh_depth; sc.small_ok = fs.small_ok; sc.small_any_fuzz = fs.small_any_fuzz; sc.small_n_rects = fs.small_n_rects;
    sc.n_media = fs.n_media; sc.leaf_kinds = fs.leaf_kinds; sc.media_kind = fs.media_kind;
    for (int k = 0; k < 3; k++) sc.background[k] = fs.background[k];
    Writer w; visit(fs, sc, w);
    Header h; std::memset(&h, 0, sizeof h);
    std::memcpy(h.magic, kMagic, 8); h.version = kVersion; h.n_sections = w.n; h.payload_bytes = w.out.size();
    uLong crc = crc32(0L, Z_NULL, 0);
    for (size_t off = 0; off < w.out.size(); off += (size_t)1 << 30) crc = crc32(crc, w.out.data() + off, (uInt)std::min<size_t>((size_t)1 << 30, w.out.size() - off));
    h.payload_crc = (uint32_t)crc;
    FILE *f = std::fopen(path, "wb");
    if (!f) { err = std::string("cannot open ") + path + " for writing"; return false; }
    const bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 && (w.out.empty() || std::fwrite(w.out.data(), 1, w.out.size(), f) == w.out.size());
    if (std::fclose(f) != 0 || !ok) { err = std::string("short write to ") + path; return false; }
    return true;
}

bool load_flat_scene(const char *path, FlatScene &fs, std::string &err) {
    FILE *f = std::fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return false; }
    Header h;
    if (std::fread(&h, sizeof h, 1, f) != 1 || std::memcmp(h.magic, kMagic, 8) != 0) { std::fclose(f); err = std::string(path) + " is not a scene file"; return false; }
    if (h.version != kVersion) { std::fclose(f); err = std::string(path) + ": scene file version " + std::to_string(h.version) + ", this library reads version " + std::to_string(kVersion); return false; }
    std::fseek(f, 0, SEEK_END); const long long size = std::ftell(f); std::fseek(f, (long)sizeof h, SEEK_SET);
    if (size < 0 || h.payload_bytes != (uint64_t)(size - (long long)sizeof h)) { std::fclose(f); err = std::string(path) + ": payload length does not match the file (truncated?)"; return false; }
    std::vector<unsigned char> buf((size_t)h.payload_bytes);
    const bool got = buf.empty() || std::fread(buf.data(), 1, buf.size(), f) == buf.size();
    std::fclose(f);
    if (!got) { err = std::string("short read from ") + path; return false; }
    uLong crc = crc32(0L, Z_NULL, 0);
    for (size_t off = 0; off < buf.size(); off += (size_t)1 << 30) crc = crc32(crc, buf.data() + off, (uInt)std::min<size_t>((size_t)1 << 30, buf.size() - off));
    if ((uint32_t)crc != h.payload_crc) { err = std::string(path) + ": checksum mismatch (damaged file)"; return false; }
    Scalars sc; std::memset(&sc, 0, sizeof sc);
    Reader r{buf.data(), buf.data() + buf.size(), err};
    if (!visit(fs, sc, r)) return false;
    fs.bvh_depth = sc.bvh_depth; fs.small_ok = sc.small_ok != 0; fs.small_any_fuzz = sc.small_any_fuzz != 0; fs.small_n_rects = sc.small_n_rects;
    fs.n_media = sc.n_media; fs.leaf_kinds = sc.leaf_kinds; fs.media_kind = sc.media_kind;
    for (int k = 0; k < 3; k++) fs.background[k] = sc.background[k];
    std::string why;
    if (!validate_flat_scene(fs, why)) { err = std::string(path) + ": " + why; return false; }
    mark_small_lights(fs);                                            // derived from the verified lights and pack, like the four-child tree   // the traversed four-child tree is derived from the verified binary one
    return true;
}
