"""A/B of the nodes-in-LDS extend (kernels_wf.hip k_wf_extend_lds<., ., false>) against the plain extend inside one process:
RTC_WF_LDS_TABLES=1 allows the all-in-LDS mode only (round 2's behaviour), unset allows the node-table-only mode too.
Accumulators must be bit-identical; prints kernel ms (median of reps)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import raytracer_amd as ra
ra.rtc_init()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
jobs = [("final", None, 1024, 1024, 256, 0), ("final_earth", 16 / 9, 1920, 1080, 256, 0), ("book2_random", 16 / 9, 1920, 1080, 64, 0), ("box_field_smoke", None, 768, 768, 64, 0)]
for name, aspect, W, H, spp, arg in jobs:
    try:
        sc, cam, view = ra.build_named_scene(name, arg=arg, aspect=aspect)
    except Exception as e:
        print(name, "skipped:", e); continue
    res = {}
    for mode in ("1", None, "1", None):
        if mode is None: os.environ.pop("RTC_WF_LDS_TABLES", None)
        else: os.environ["RTC_WF_LDS_TABLES"] = mode
        info = sc.extend_info()
        ms = []
        for rep in range(reps):
            acc, st = sc.render(cam, ra.Settings(W, H, spp, view.max_depth, view.integrator, seed=3, kernel=3))
            ms.append(st["kernel_ms"])
        res.setdefault(info["lds_tables"], []).append((sorted(ms)[len(ms) // 2], min(ms), acc.copy(), st["rays"], info))
    os.environ.pop("RTC_WF_LDS_TABLES", None)
    keys = sorted(res)
    for k in keys:
        print(f"{name:14s} {W}x{H}x{spp} lds mode {k}: median {[round(r[0], 2) for r in res[k]]} min {[round(r[1], 2) for r in res[k]]} ms  info {res[k][0][4]}", flush=True)
    if len(keys) == 2:
        a, b = res[keys[0]][0], res[keys[1]][0]
        print(f"{'':14s} bit-identical: {np.array_equal(a[2], b[2])}  rays equal: {a[3] == b[3]}  speed-up {a[0] / b[0]:.3f}x", flush=True)
    sc.close()
