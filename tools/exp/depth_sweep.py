"""Where a BVH render's time goes by bounce: kernel ms at max_depth 1, 2, 3, ... (development aid)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import raytracer_amd as ra
ra.rtc_init()
jobs = {"final": (None, 1024, 1024, 256, 0), "book1": (None, 1920, 1080, 64, 0), "million": (None, 1920, 1080, 64, 1000)}
for name in sys.argv[1:] or list(jobs):
    aspect, W, H, spp, arg = jobs[name]
    sc, cam, view = ra.build_named_scene(name, arg=arg, aspect=aspect)
    for depth in (1, 2, 3, 4, view.max_depth):
        best = None
        for rep in range(3):
            acc, st = sc.render(cam, ra.Settings(W, H, spp, depth, view.integrator, seed=rep))
            if best is None or st["kernel_ms"] < best["kernel_ms"]:
                best = st
        print(f"{name:8s} depth {depth:2d}: {best['kernel_ms']:8.2f} ms  rays {best['rays']:.4g}  rays/sample {best['rays']/best['samples']:.3f}", flush=True)
    sc.close()
