"""CPU: the host half of the production BVH (flatten.cpp: binned-SAH builder, 64-byte world-space traversal leaves), checked
without a GPU through rtc_scene_flatten_only + rtc_scene_dump_bvh.

 * structure: every leaf slot is referenced by exactly one leaf run, every inner node by exactly one parent, the depth
   reported by rtc_scene_bvh_info bounds every root-to-leaf path, child boxes contain the geometry below them;
 * geometry: a numpy brute-force closest hit over the dumped FP32 leaves (spheres, moving spheres, rects: what the HIP
   traversal tests) agrees with the oracle's list scan (hittable_list.c:125-143) on the same rays -- same primitive, t
   within 1e-4 relative (FP32 tables against the FP64 oracle).  Media are skipped on both sides.
"""
import numpy as np
import pytest

EMPTY = -(2 ** 31)


def decode(child):
    v = ~int(child)
    return v >> 3, (v & 7) + 1


BOX = 1 << 10                                                    # RTC_LEAF_BOX: one record answers for six slots


def walk(nodes_i, nodes_f, meta):
    """-> (leaf reference count per slot, node reference count, max depth, per-slot list of enclosing boxes)"""
    n_leaves = len(meta)
    leaf_refs = np.zeros(n_leaves, dtype=np.int64)
    node_refs = np.zeros(len(nodes_i), dtype=np.int64)
    boxes = [None] * n_leaves
    max_depth = 0
    stack = [(0, 1)]
    node_refs[0] = 1
    while stack:
        n, depth = stack.pop()
        max_depth = max(max_depth, depth)
        for side in (0, 1):
            child = int(nodes_i[n, 3 + 8 * side])
            lo, hi = nodes_f[n, 8 * side:8 * side + 3], nodes_f[n, 8 * side + 4:8 * side + 7]
            if child == EMPTY:
                continue
            if child >= 0:
                node_refs[child] += 1
                stack.append((child, depth + 1))
            else:
                first, count = decode(child)
                assert count == int(nodes_i[n, 7 + 8 * side])
                max_depth = max(max_depth, depth + 1)
                s = first
                for _ in range(count):                            # `count` records; a box record is six slots wide
                    width = 6 if int(meta[s]) & BOX else 1
                    for k in range(s, s + width):
                        leaf_refs[k] += 1
                        boxes[k] = (lo.copy(), hi.copy())
                    s += width
    return leaf_refs, node_refs, max_depth, boxes


def brute_force(leaves_u, prims, rays, tmin=1e-3):
    """Closest hit over all traversal leaves in float64 arithmetic on the FP32 table values; ties -> highest primitive id."""
    lf = leaves_u.view(np.float32).astype(np.float64)
    meta = leaves_u[:, 15]
    o, d, tm = rays[:, None, 0:3], rays[:, None, 3:6], rays[:, None, 6]
    n, m = len(rays), len(lf)
    t_all = np.full((n, m), np.inf)
    typ = meta & 3
    # spheres / moving spheres
    for kind in (0, 1):
        idx = np.nonzero(typ == kind)[0]
        if not len(idx):
            continue
        L = lf[idx]
        c = L[None, :, 0:3] + (L[None, :, 3:6] * tm[..., None] if kind == 1 else 0.0)
        r = L[None, :, 6] if kind == 1 else L[None, :, 3]
        oc = o - c
        a = (d * d).sum(-1)
        hb = (oc * d).sum(-1)
        cc = (oc * oc).sum(-1) - r * r
        disc = hb * hb - a * cc
        sq = np.sqrt(np.maximum(disc, 0))
        t0, t1 = (-hb - sq) / a, (-hb + sq) / a
        t = np.where(t0 >= tmin, t0, np.where(t1 >= tmin, t1, np.inf))
        t_all[:, idx] = np.where(disc >= 0, t, np.inf)
    idx = np.nonzero((typ == 2) & ((meta & BOX) == 0))[0]
    if len(idx):                                                 # quads: plane N.x = k, a = U.x in [a_lo, a_hi], b = V.x in [b_lo, b_hi]
        L = lf[idx]
        N, U, V = L[None, :, 0:3], L[None, :, 4:7], L[None, :, 8:11]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (L[None, :, 3] - (N * o).sum(-1)) / (N * d).sum(-1)
            p = o + t[..., None] * d
            pa, pb = (U * p).sum(-1), (V * p).sum(-1)
        ok = (t >= tmin) & (pa >= L[None, :, 7]) & (pa <= L[None, :, 12]) & (pb >= L[None, :, 11]) & (pb <= L[None, :, 13])
        t_all[:, idx] = np.where(ok, t, np.inf)
    # box records: the record (slab test, entry point else exit point) answers for its six face slots
    for s0 in np.nonzero((typ == 2) & ((meta & BOX) != 0))[0]:
        lo, hi = lf[s0, 0:3], lf[s0, 3:6]
        with np.errstate(divide="ignore", invalid="ignore"):
            t0, t1 = (lo - o[:, 0]) / d[:, 0], (hi - o[:, 0]) / d[:, 0]
        tn, tf = np.minimum(t0, t1), np.maximum(t0, t1)
        t_in, t_out = tn.max(1), tf.min(1)
        ax_in, ax_out = tn.argmax(1), tf.argmin(1)
        rows = np.arange(n)
        f_in = 2 * ax_in + (d[rows, 0, ax_in] <= 0)
        f_out = 2 * ax_out + (d[rows, 0, ax_out] > 0)
        enter = t_in >= tmin
        t = np.where(enter, t_in, t_out)
        face = np.where(enter, f_in, f_out)
        ok = (t_in <= t_out) & (t >= tmin)
        t_all[:, s0:s0 + 6] = np.inf
        t_all[rows[ok], s0 + face[ok]] = t[ok]
    best = t_all.min(axis=1)
    tie = np.isclose(t_all, best[:, None], rtol=1e-12, atol=0) & np.isfinite(t_all)
    prim = np.where(tie, prims[None, :], -1).max(axis=1)
    return np.where(np.isfinite(best), prim, -1), best


@pytest.mark.parametrize("name,arg", [("book1", 0), ("final", 0), ("book2_random", 0), ("million", 40), ("box_field", 0)])
def test_bvh_structure_and_leaf_geometry(rtc, oracle_lib, name, arg):
    sc, cam, view = rtc.build_named_scene(name, arg=arg, host_only=True)
    try:
        info = sc.bvh_info()
        nodes_u, leaves_u, prims = sc.dump_bvh()
        assert info["nodes"] == len(nodes_u) > 0 and info["leaf_slots"] == len(leaves_u) == sc.num_prims()
        assert sorted(prims.tolist()) == list(range(sc.num_prims()))       # a permutation: slot -> primitive id
        nodes_i, nodes_f = nodes_u.view(np.int32), nodes_u.view(np.float32)
        leaf_refs, node_refs, depth, boxes = walk(nodes_i, nodes_f, leaves_u[:, 15])
        assert (leaf_refs == 1).all() and (node_refs == 1).all()
        assert depth <= info["depth"] <= 64
        lf = leaves_u.view(np.float32)
        typ = leaves_u[:, 15] & 3
        for s in np.nonzero(typ == 0)[0][:500]:                              # spheres: centre +- r inside the parent's child box
            lo, hi = boxes[s]
            c, r = lf[s, 0:3], abs(lf[s, 3])
            assert (c - r >= lo - 1e-3).all() and (c + r <= hi + 1e-3).all()
        # geometry of the leaves against the oracle
        o = oracle_lib
        o.set_stage(3)
        o.build(name, arg=arg)
        g = np.stack(np.meshgrid((np.arange(40) + .5) / 39, (np.arange(24) + .5) / 23), -1).reshape(-1, 2)
        rays = o.camera_rays(g, seed=5)
        rec, prim_ref = o.hit_rays(rays)
        keep = rec[:, 11] != 4                                               # the oracle hit a medium: stochastic, skipped
        prim, t = brute_force(leaves_u, prims, rays)
        assert (prim[keep] == prim_ref[keep]).mean() >= 0.995                # FP32 table values: a grazing ray may flip
        same = keep & (prim == prim_ref) & (prim_ref >= 0)
        assert same.sum() > 200
        assert np.abs(t[same] - rec[same, 1]).max() <= 1e-4 * np.maximum(1.0, rec[same, 1]).max()
    finally:
        sc.close()


def test_small_scenes_have_no_bvh(rtc):
    sc, cam, view = rtc.build_named_scene("cornell", host_only=True)
    try:
        assert sc.bvh_info() == dict(nodes=0, leaf_slots=0, depth=0, bytes=0)
        nodes, leaves, prims = sc.dump_bvh()
        assert len(nodes) == 0 and len(leaves) == 0
    finally:
        sc.close()


def test_parallel_builder_makes_the_sequential_tree(rtc, tmp_path, monkeypatch):
    """flatten.cpp build_bvh_parallel: the top of the tree by one builder, the deferred ranges by several threads, spliced in
    pre-order -- the same partitions, node numbering and leaf slot order as the one-thread builder: the scene files are equal
    byte for byte (the tree, the traversal leaves, the shading records, everything)."""
    for name, arg in (("million", 200), ("final", 0), ("box_field_smoke", 0), ("book1", 0)):
        blobs = []
        for threads in ("1", "5"):
            monkeypatch.setenv("RTC_BUILD_THREADS", threads)             # read by flatten_scene at every build; forces the parallel path
            sc, _, _ = rtc.build_named_scene(name, arg=arg, host_only=True)
            path = str(tmp_path / f"{name}_{threads}.rtcscene")
            sc.save(path); sc.close()
            blobs.append(open(path, "rb").read())
        assert blobs[0] == blobs[1], name
