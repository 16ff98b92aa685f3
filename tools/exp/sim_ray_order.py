"""Would re-ordering the rays of a bounce make the waves of k_wf_extend denser?  Host-side model (development aid):
the scattered rays of bounce 1 and 2 are traced through the dumped production tree exactly as bvh_trip_tab does (trip sequences per
ray, as in sim_traversal.py), then 64 consecutive rays of an ordering form a wave: its trips = its longest lane, its cost = the arms
present per trip.  Orderings: the pipeline's tile order, random, global sorts by Morton code of the origin and / or a direction grid,
sorts inside windows of 1 K - 64 K list entries, and the oracle (sorted by trip count).
    python tools/exp/sim_ray_order.py final [pixels per side = 256] [scattered rays per camera ray = 4]
Result for the final scene (134 K bounce-1 rays): tile order 35.0 wave-trips per wave, lane utilisation 0.47; octant inside 1 K-entry
windows 29.8 / 0.55 (modelled VALU -22 %); origin cell (3 bits per axis) then a 4^3 direction grid, sorted globally, 28.1 / 0.58 (-27 %);
oracle 16.4 / 1.0.  The octant regrouping inside a workgroup had been built in round 1 and measured +-1.5 % -- the kernel is not bound by
the instruction count the model prices but by the record fetches, which a re-ordering leaves as they are and whose state
gathers it scatters; a global sort per bounce was therefore not built."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import raytracer_amd as ra
name = sys.argv[1] if len(sys.argv) > 1 else "final"
side = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nsec = int(sys.argv[3]) if len(sys.argv) > 3 else 2
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


import multiprocessing as mp, pickle
def work(args):
    j0, j1, seed = args
    rng = np.random.default_rng(seed)
    out = []
    for j in range(j0, j1):
        for i in range(side):
            s, t = (i + rng.random()) / side, (j + rng.random()) / side
            o, d, tm = lf_.copy(), llc + hor * s + ver * t - lf_, rng.random()
            tt, seq = trace(o, d, tm)
            if not np.isfinite(tt): continue
            p = o + d * tt
            for k in range(nsec):
                dd = unit(rng.normal(size=3))
                o1 = p + dd * 1e-3
                t2, seq2 = trace(o1, dd, tm)
                out.append((1, j, i, k, o1, dd, seq2))
                if np.isfinite(t2) and k == 0:
                    p2 = o1 + dd * t2
                    d3 = unit(rng.normal(size=3))
                    o2 = p2 + d3 * 1e-3
                    t3, seq3 = trace(o2, d3, tm)
                    out.append((2, j, i, k, o2, d3, seq3))
    return out
if __name__ == "__main__":
    nproc = 8
    step = (side + nproc * 4 - 1) // (nproc * 4)
    jobs = [(j, min(j + step, side), 100 + j) for j in range(0, side, step)]
    with mp.Pool(nproc) as pool:
        res = pool.map(work, jobs)
    rays = [r for part in res for r in part]
    print("rays", len(rays))
    cost = {"node": 60, "box": 90, "sphere": 40, "msphere": 45, "rect": 45, "medium": 130}
    def evaluate(tag, seqs):
        waves = [seqs[i:i + 64] for i in range(0, len(seqs) - 63, 64)]
        wt = 0; wi = 0; lane_trips = 0
        for wv in waves:
            M = max(len(s) for s in wv); wt += M; lane_trips += sum(len(s) for s in wv)
            for t in range(M):
                present = {s[t] for s in wv if len(s) > t}
                wi += sum(cost[k] for k in present) + 20
        print(f"{tag:55s} wave-trips/wave {wt / len(waves):6.1f}  lane util {lane_trips / (64.0 * wt):.3f}  modelled VALU/wave {wi / len(waves):7.0f}")
        return wi / len(waves)
    def morton3(q, bits):
        out = np.zeros(len(q), np.int64)
        for b in range(bits):
            for a in range(3):
                out |= ((q[:, a] >> b) & 1) << (3 * b + a)
        return out
    R = [r for r in rays if r[0] == 1]
    o = np.array([r[4] for r in R]); d = np.array([r[5] for r in R]); seqs = [r[6] for r in R]
    pix = np.array([(r[1], r[2], r[3]) for r in R])
    # tile order = 8x8 tiles row-major, 4x4 quadrant, layer, pixel (as slot_pixel)
    order = np.lexsort((pix[:, 1] % 4, pix[:, 0] % 4, pix[:, 2], (pix[:, 1] // 4) % 2, (pix[:, 0] // 4) % 2, pix[:, 1] // 8, pix[:, 0] // 8))
    evaluate("tile order", [seqs[i] for i in order])
    octant = (d[:, 0] < 0) * 1 + (d[:, 1] < 0) * 2 + (d[:, 2] < 0) * 4
    keys = {"octant": octant}
    for nb in (3, 4, 6, 8):
        dq = np.clip(((d + 1) / 2 * nb).astype(np.int64), 0, nb - 1)
        keys[f"dir grid {nb}^3"] = (dq[:, 1] * nb + dq[:, 0]) * nb + dq[:, 2]
    keys["dir.y 8 levels"] = np.clip(((d[:, 1] + 1) * 4).astype(np.int64), 0, 7)
    keys["dir.y 16 levels"] = np.clip(((d[:, 1] + 1) * 8).astype(np.int64), 0, 15)
    for W in (1024, 4096, 16384, 65536):
        for kn, k in keys.items():
            new = []
            for s in range(0, len(order), W):
                win = order[s:s + W]
                new.extend(win[np.argsort(k[win], kind="stable")])
            evaluate(f"window {W:6d} sorted by {kn}", [seqs[i] for i in new])
    lo, hi = o.min(0), o.max(0)
    def mort(bits):
        q = np.clip(((o - lo) / (hi - lo + 1e-9) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
        return morton3(q, bits)
    octant = (d[:, 0] < 0) * 1 + (d[:, 1] < 0) * 2 + (d[:, 2] < 0) * 4
    for nb in (2, 4, 8, 16):
        dq = np.clip(((d + 1) / 2 * nb).astype(np.int64), 0, nb - 1)
        dk = (dq[:, 1] * nb + dq[:, 0]) * nb + dq[:, 2]
        for mb in (3, 5, 8):
            evaluate(f"dir grid {nb}^3 | morton {mb}", [seqs[i] for i in np.lexsort((mort(mb), dk))])
            evaluate(f"morton {mb} | dir grid {nb}^3", [seqs[i] for i in np.lexsort((dk, mort(mb)))])
    lens = np.array([len(s) for s in seqs])
    evaluate("oracle: sorted by trip count", [seqs[i] for i in np.argsort(lens, kind="stable")])
