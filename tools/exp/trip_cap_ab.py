"""tools/exp/trip_cap_ab.py -- RTC_WF_TRIP_CAP (a traversal cut after n trips of a launch and resumed by the next): accumulators against the uncapped
render's (must be bit-identical), ray counts, and kernel time, for a few caps (development aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracer_amd as ra
ra.rtc_init()
caps = [int(c) for c in os.environ.get("CAPS", "0 8 16 24 32").split()]
for scene, arg, W, H, spp in (("final", 0, 256, 256, 64), ("million", 200, 320, 180, 32), ("box_field_smoke", 0, 160, 96, 64), ("final", 0, 1024, 1024, 256), ("million", 1000, 1920, 1080, 64), ("final_earth", 0, 1920, 1080, 256)):
    sc, cam, view = ra.build_named_scene(scene, arg=arg, aspect=(16 / 9 if W * 9 == H * 16 else None))
    ref = None
    for cap in caps:
        os.environ["RTC_WF_TRIP_CAP"] = str(cap)
        best = None
        for rep in range(2):
            acc, st = sc.render(cam, ra.Settings(W, H, spp, view.max_depth, view.integrator, seed=3, kernel=3))
            if best is None or st["kernel_ms"] < best: best = st["kernel_ms"]
        if ref is None: ref = (acc.copy(), st["rays"])
        same = bool(np.array_equal(acc, ref[0]))
        print(f"{scene:16s} {W}x{H}x{spp} cap {cap:3d}: {best:9.3f} ms  rays {st['rays']}  {'bit-identical' if same else 'DIFFERENT max |d| %.3g' % float(np.abs(acc - ref[0]).max())}  rays equal {st['rays'] == ref[1]}", flush=True)
    sc.close()
os.environ.pop("RTC_WF_TRIP_CAP", None)
