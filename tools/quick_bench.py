"""Quick single-GPU timing of named scenes (development aid): kernel ms, Msamples/s, Mrays/s, lane fill."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import raytracer_amd as ra
ra.rtc_init()
jobs = [("cornell", 16 / 9, 1920, 1080, 256, 0), ("cornell", None, 1024, 1024, 256, 0), ("final", None, 1024, 1024, 64, 0), ("book1", None, 1920, 1080, 64, 0), ("million", None, 1920, 1080, 16, 1000)]
only = [a for a in sys.argv[1:] if not a.startswith("k=")]
kern = int(([a[2:] for a in sys.argv[1:] if a.startswith("k=")] or ["0"])[0])
for name, aspect, W, H, spp, arg in jobs:
    if only and name not in only:
        continue
    sc, cam, view = ra.build_named_scene(name, arg=arg, aspect=aspect)
    best = None
    for rep in range(3):
        acc, st = sc.render(cam, ra.Settings(W, H, spp, view.max_depth, view.integrator, seed=rep, kernel=kern))
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = st
    print(f"{name:8s} {W}x{H}x{spp}: kernel {best['kernel_ms']:8.2f} ms  {best['samples']/best['kernel_ms']*1e-3:8.1f} Msamples/s  {best['rays']/best['kernel_ms']*1e-3:8.1f} Mrays/s  rays/sample {best['rays']/best['samples']:.3f}  lane fill {best['lane_fill']:.3f}  kernel {best['kernel_used']}  iters/ray {best['wave_iterations']/max(1,best['rays']):.3f}", flush=True)
    sc.close()
