"""A/B of the render's tail inside one process (kernels_wf.hip k_wf_extend_tail): RTC_WF_TAIL_LIVE=0 = one bounce per launch throughout (the
round-3 pipeline before this change); unset = once at most 256 K slots of a part are alive, a lane traces its path to the end inside one launch.
Accumulators must be bit-identical; prints kernel ms (median of reps)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import raytracer_amd as ra
ra.rtc_init()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
only = sys.argv[2:]
jobs = [("final", None, 1024, 1024, 256, 0), ("final_earth", 16 / 9, 1920, 1080, 256, 0), ("book1", 16 / 9, 1920, 1080, 64, 0), ("million", 16 / 9, 1920, 1080, 64, 1000),
        ("book2_random", 16 / 9, 1920, 1080, 64, 0), ("box_field_smoke", None, 768, 768, 64, 0)]
MODES = {"old": {"RTC_WF_LATE_ROUND": "1000000"}, "new": {}, "late20": {"RTC_WF_LATE_ROUND": "20"}, "late40": {"RTC_WF_LATE_ROUND": "40"}}
for name, aspect, W, H, spp, arg in jobs:
    if only and name not in only: continue
    sc, cam, view = ra.build_named_scene(name, arg=arg, aspect=aspect)
    res = {}
    for mode in list(MODES) * 2:
        for k in ("RTC_WF_TAIL_LIVE", "RTC_WF_LATE_ROUND"): os.environ.pop(k, None)
        os.environ.update(MODES[mode])
        ms = []
        for rep in range(reps):
            acc, st = sc.render(cam, ra.Settings(W, H, spp, view.max_depth, view.integrator, seed=3, kernel=3))
            ms.append(st["kernel_ms"])
        res.setdefault(mode, []).append((sorted(ms)[len(ms) // 2], min(ms), acc.copy(), st["rays"]))
    for k in ("RTC_WF_TAIL_LIVE", "RTC_WF_LATE_ROUND"): os.environ.pop(k, None)
    for mode in MODES:
        print(f"{name:16s} {W}x{H}x{spp} {mode:5s}: median {[round(r[0], 2) for r in res[mode]]} min {[round(r[1], 2) for r in res[mode]]} ms", flush=True)
    a = res["old"][0]
    for mode in list(MODES)[1:]:
        b = res[mode][0]
        print(f"{'':16s} {mode}: bit-identical {np.array_equal(a[2], b[2])}  rays {a[3]} vs {b[3]}  speed-up {a[0] / b[0]:.3f}x", flush=True)
    sc.close()
