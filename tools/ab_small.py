"""A/B aid for the small-scene kernel: RTC_LIB=<library build> python tools/ab_small.py <tag> [reps] -> kernel ms of the
headline render (Cornell 1920x1080 x 256 spp, depth 10) and of the reference's default size, the image saved for comparison
with another build's (gpurun_out/ab_<tag>.npy, 240x135 block means)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import raytracer_amd as ra
tag = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ra.rtc_init()
for name, aspect, W, H, spp in (("cornell", 16 / 9, 1920, 1080, 256), ("cornell", None, 1024, 1024, 256)):
    sc, cam, view = ra.build_named_scene(name, aspect=aspect)
    ms = []
    for rep in range(reps):
        acc, st = sc.render(cam, ra.Settings(W, H, spp, 10, 0, seed=3))
        ms.append(st["kernel_ms"])
    img = acc[..., :3] / acc[..., 3:4]
    print(f"{tag:10s} {W}x{H}x{spp}: kernel min {min(ms):7.3f} median {sorted(ms)[len(ms) // 2]:7.3f} ms  rays/sample {st['rays'] / st['samples']:.4f}  "
          f"lane fill {st['lane_fill']:.4f}  mean {img.mean():.6f}", flush=True)
    if W == 1920:
        os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
        np.save(os.path.join(R, "gpurun_out", f"ab_{tag}.npy"), img.reshape(135, 8, 240, 8, 3).mean((1, 3)))
    sc.close()
