"""tools/exp/perlin_cost.py -- what a shading pass of k_wf_shade<true> costs: a marble ground and 100 marble spheres (a BVH scene where EVERY
hit lands on the noise texture, so every hit is parked) at 1080p x 16 spp, and the same scene in grey; run under rocprofv3 --kernel-trace
--stats and read k_wf_shade's total against the number of hits (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracer_amd as ra
ra.rtc_init()
from raytracer_amd import api
L = api.lib(); v3 = api.v3
for marble in (True, False):
    world = L.hittable_list_new(128)
    m = L.material_lambertian_new(L.texture_noise_init_scaled(4.0)) if marble else L.material_lambertian_new_from_color(v3((0.5, 0.5, 0.5)))
    L.hittable_sphere_new(world, v3((0, -1000, 0)), 1000.0, m)
    for i in range(100):
        L.hittable_sphere_new(world, v3((-9 + 2 * (i % 10), 0.8, -9 + 2 * (i // 10))), 0.8, m)
    sc = ra.Scene(world, None, (0.7, 0.8, 1.0))
    cam = ra.Camera((13, 4, 3), (0, 0, 0), (0, 1, 0), 40.0, 16 / 9, 0.0, 10.0)
    for rep in range(3):
        acc, st = sc.render(cam, ra.Settings(1920, 1080, 16, 8, 1, seed=rep))
        print("marble" if marble else "grey  ", st["kernel_used"], round(st["kernel_ms"], 3), "ms", st["rays"], "rays", st["samples"], "samples", flush=True)
    sc.close()
