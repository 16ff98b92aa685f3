"""Host-side model of the BVH traversal of kernels_wf.hip on the dumped production tree (development aid): per-ray trip sequences
(node step or leaf step by kind), then what a 64-lane wave pays for them -- trips = the longest lane, arms = the kinds present per
trip -- and what squeezing the unfinished rays of a 256-lane workgroup together every K trips would save.
    python tools/exp/sim_traversal.py final [n_pixels_side]"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import raytracer_amd as ra
name = sys.argv[1] if len(sys.argv) > 1 else "final"
side = int(sys.argv[2]) if len(sys.argv) > 2 else 64
arg = 200 if name == "million" else 0
sc, cam, view = ra.build_named_scene(name, arg=arg, host_only=True)
nodes_u, leaves_u, prims = sc.dump_bvh()
ni, nf = nodes_u.view(np.int32), nodes_u.view(np.float32).astype(np.float64)
lf = leaves_u.view(np.float32).astype(np.float64); meta = leaves_u[:, 15]
EMPTY = -(2 ** 31); BOX = 1 << 10
print(name, "nodes", len(ni), "leaf slots", len(lf))
# camera
def unit(v): return v / np.linalg.norm(v)
lf_, la, vup = (np.array([getattr(v, k) for k in "xyz"]) for v in (view.lookfrom, view.lookat, view.vup))
theta = np.deg2rad(view.vfov); h = np.tan(theta / 2); vh = 2 * h; vw = view.aspect * vh
w = unit(lf_ - la); u = unit(np.cross(vup, w)); v = np.cross(w, u)
hor = u * vw * view.focus_dist; ver = v * vh * view.focus_dist; llc = lf_ - hor / 2 - ver / 2 - w * view.focus_dist
rng = np.random.default_rng(1)

def slab(o, inv, lo, hi, tmin, tmax):
    t0 = (lo - o) * inv; t1 = (hi - o) * inv
    tn = max(np.minimum(t0, t1).max(), tmin); tf = min(np.maximum(t0, t1).min(), tmax)
    return tn <= tf, tn

def leaf_hit(s, o, d, tm, tmin, tmax):
    typ = int(meta[s]) & 3
    L = lf[s]
    if typ == 3:                                                      # medium: counted as a trip, never a hit here
        return None, 1, "medium"
    if typ == 2 and int(meta[s]) & BOX:
        with np.errstate(divide="ignore", invalid="ignore"):
            t0 = (L[0:3] - o) / d; t1 = (L[3:6] - o) / d
        tn, tf = np.minimum(t0, t1), np.maximum(t0, t1)
        t_in, t_out = tn.max(), tf.min()
        t = t_in if t_in >= tmin else t_out
        return (t if (t_in <= t_out and tmin <= t <= tmax) else None), 6, "box"
    if typ == 2:
        N, U, V = L[0:3], L[4:7], L[8:11]
        den = N @ d
        if den == 0: return None, 1, "rect"
        t = (L[3] - N @ o) / den; p = o + t * d; pa, pb = U @ p, V @ p
        ok = tmin <= t <= tmax and L[7] <= pa <= L[12] and L[11] <= pb <= L[13]
        return (t if ok else None), 1, "rect"
    c = L[0:3] + (L[3:6] * tm if typ == 1 else 0.0); r = L[6] if typ == 1 else L[3]
    oc = o - c; a = d @ d; hb = oc @ d; cc = oc @ oc - r * r; disc = hb * hb - a * cc
    if disc < 0: return None, 1, "sphere" if typ == 0 else "msphere"
    sq = np.sqrt(disc)
    for t in ((-hb - sq) / a, (-hb + sq) / a):
        if tmin <= t <= tmax: return t, 1, "sphere" if typ == 0 else "msphere"
    return None, 1, "sphere" if typ == 0 else "msphere"

def trace(o, d, tm):
    """-> (t, point-ish, list of trip kinds)"""
    with np.errstate(divide="ignore"):
        inv = 1.0 / d
    inv = np.clip(inv, -1e30, 1e30)
    seq = []; stack = []; cur = 0; best = np.inf
    while cur != EMPTY:
        if cur >= 0:
            seq.append("node")
            n = cur
            h = []
            for sd in (0, 1):
                ch = int(ni[n, 3 + 8 * sd])
                if ch == EMPTY: h.append((False, 0, ch)); continue
                ok, tn = slab(o, inv, nf[n, 8 * sd:8 * sd + 3], nf[n, 8 * sd + 4:8 * sd + 7], 1e-3, best)
                h.append((ok, tn, ch))
            (h0, n0, c0), (h1, n1, c1) = h
            if h0 and h1:
                first0 = n0 <= n1
                stack.append(c1 if first0 else c0); cur = c0 if first0 else c1
            elif h0: cur = c0
            elif h1: cur = c1
            else: cur = stack.pop() if stack else EMPTY
        else:
            vv = ~cur; s = vv >> 3; rem = vv & 7
            t, width, kind = leaf_hit(s, o, d, tm, 1e-3, best)
            seq.append(kind)
            if t is not None: best = t
            cur = ~(((s + width) << 3) | (rem - 1)) if rem else (stack.pop() if stack else EMPTY)
    return best, seq

prim_rays = []
for j in range(side):
    for i in range(side):
        s, t = (i + rng.random()) / side, (j + rng.random()) / side
        prim_rays.append((lf_.copy(), llc + hor * s + ver * t - lf_, rng.random()))
seqs_p, seqs_s = [], []
for (o, d, tm) in prim_rays:
    t, seq = trace(o, d, tm)
    seqs_p.append(seq)
    if np.isfinite(t):
        p = o + d * t
        dd = unit(rng.normal(size=3))
        t2, seq2 = trace(p + dd * 1e-3, dd, tm)
        seqs_s.append(seq2)
    else:
        seqs_s.append(None)

def report(tag, seqs, group):
    seqs = [s for s in seqs if s is not None]
    lens = np.array([len(s) for s in seqs])
    kinds = sorted({k for s in seqs for k in s})
    print(f"== {tag}: {len(seqs)} rays, trips per ray mean {lens.mean():.1f} median {np.median(lens):.0f} p90 {np.percentile(lens, 90):.0f} max {lens.max()}")
    frac = {k: sum(s.count(k) for s in seqs) / lens.sum() for k in kinds}
    print("   step mix:", {k: round(v, 3) for k, v in frac.items()})
    # waves of 64 consecutive rays (spatially coherent order: 8x8 tiles)
    cost = {"node": 60, "box": 90, "sphere": 40, "msphere": 45, "rect": 45, "medium": 130}
    waves = [seqs[i:i + 64] for i in range(0, len(seqs) - 63, 64)]
    wt = 0; wi = 0; lane_trips = 0
    for wv in waves:
        M = max(len(s) for s in wv); wt += M; lane_trips += sum(len(s) for s in wv)
        for t in range(M):
            present = {s[t] for s in wv if len(s) > t}
            wi += sum(cost[k] for k in present) + 20
    print(f"   64-lane waves: wave-trips per wave {wt / len(waves):.1f}, lane utilisation {lane_trips / (64.0 * wt):.3f}, modelled VALU per wave {wi / len(waves):.0f}")
    for K in (4, 8, 16):
        for G in (256, 1024):
            blocks = [seqs[i:i + G] for i in range(0, len(seqs) - G + 1, G)]
            tot_wt = 0; tot_wi = 0
            for b in blocks:
                alive = list(b); t0 = 0
                while alive:
                    # pack alive rays into waves of 64, run K trips
                    alive.sort(key=lambda s: 0)  # keep order
                    for wstart in range(0, len(alive), 64):
                        wv = alive[wstart:wstart + 64]
                        for t in range(t0, t0 + K):
                            present = {s[t] for s in wv if len(s) > t}
                            if not present: break
                            tot_wt += 1; tot_wi += sum(cost[k] for k in present) + 20
                    t0 += K
                    alive = [s for s in alive if len(s) > t0]
                    tot_wi += 80 * ((len(alive) + 63) // 64)          # the squeeze itself
            nb = len(blocks) * (G // 64)
            print(f"   squeeze every {K:2d} trips in {G:4d}-lane groups: wave-trips per 64 rays {tot_wt / nb:.1f}, modelled VALU per 64 rays {tot_wi / nb:.0f}")
report("camera rays", seqs_p, 64)
# secondary rays in tile order: reorder rays into 8x8 tiles first
order = [ (j // 8, i // 8, j % 8, i % 8, j * side + i) for j in range(side) for i in range(side)]
order.sort()
report("camera rays (8x8 tiles)", [seqs_p[o[-1]] for o in order], 64)
report("secondary rays (8x8 tiles of their camera rays)", [seqs_s[o[-1]] for o in order], 64)

# ---- wave-uniform traversal of camera rays: how many nodes / leaf records does a wave visit together? -----------------------
def trace_uniform(rays):
    """All rays of a wave walk ONE path: a child is entered if any lane's slab test (against its own closest t) passes; nearer child by vote."""
    n = len(rays)
    o = np.array([r[0] for r in rays]); d = np.array([r[1] for r in rays]); tm = np.array([r[2] for r in rays])
    with np.errstate(divide="ignore"):
        inv = np.clip(1.0 / d, -1e30, 1e30)
    best = np.full(n, np.inf)
    stack = []; cur = 0; n_nodes = 0; n_leaf = 0
    while cur != EMPTY:
        if cur >= 0:
            n_nodes += 1
            res = []
            for sd in (0, 1):
                ch = int(ni[cur, 3 + 8 * sd])
                if ch == EMPTY: res.append((np.zeros(n, bool), np.zeros(n), ch)); continue
                lo, hi = nf[cur, 8 * sd:8 * sd + 3], nf[cur, 8 * sd + 4:8 * sd + 7]
                t0 = (lo - o) * inv; t1 = (hi - o) * inv
                tn = np.maximum(np.minimum(t0, t1).max(1), 1e-3); tf = np.minimum(np.maximum(t0, t1).min(1), best)
                res.append((tn <= tf, tn, ch))
            (h0, n0, c0), (h1, n1, c1) = res
            a0, a1 = h0.any(), h1.any()
            if a0 and a1:
                first0 = (h0 & (~h1 | (n0 <= n1))).sum() >= (h1 & (~h0 | (n1 < n0))).sum()
                stack.append(c1 if first0 else c0); cur = c0 if first0 else c1
            elif a0: cur = c0
            elif a1: cur = c1
            else: cur = stack.pop() if stack else EMPTY
        else:
            vv = ~cur; s = vv >> 3; rem = vv & 7
            while True:
                n_leaf += 1
                width = 1
                for k in range(n):
                    t, width, kind = leaf_hit(s, o[k], d[k], tm[k], 1e-3, best[k])
                    if t is not None: best[k] = t
                if rem == 0: break
                s += width; rem -= 1
            cur = stack.pop() if stack else EMPTY
    return n_nodes, n_leaf
tiles = {}
for idx, (j, i) in enumerate((j, i) for j in range(side) for i in range(side)):
    tiles.setdefault((j // 8, i // 8), []).append(prim_rays[idx])
un, ul = [], []
for key, rays in list(tiles.items())[::3]:
    a, b = trace_uniform(rays); un.append(a); ul.append(b)
print(f"== wave-uniform traversal of camera rays, 8x8-pixel waves at {side}x{side}: node visits per wave mean {np.mean(un):.1f} (max {max(un)}), leaf records {np.mean(ul):.1f} (max {max(ul)})")
