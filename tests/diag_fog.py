"""Diagnostic: fog-only scene, P(scatter) on the first ray must be 1-exp(-rho*R) = 0.3935 for a camera at the centre."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, raytracer_amd as ra
from raytracer_amd.api import v3
ra.rtc_init()
L = ra.lib()
for extra in (0, 100):
    for radius in (50.0, 5000.0):
        world = L.hittable_list_new(16)
        lights = L.hittable_list_new_no_gc(4)
        b = L.hittable_sphere_new(None, v3((0, 0, 0)), radius, L.material_dielectric_new(1.5))
        L.hittable_constant_medium_init_c(world, b, 0.5 / radius, v3((1, 1, 1)))
        for i in range(extra):
            L.hittable_sphere_new(world, v3((1e5 + 10 * i, 1e5, 1e5)), 1.0, L.material_lambertian_new_from_color(v3((.5, .5, .5))))
        sc = ra.Scene(world, lights, (0, 0, 0), 0.0, 1.0)
        cam = ra.Camera((0, 0, 0), (0, 0, 1), (0, 1, 0), 40, 1.0, 0.0, 10, 0.0, 1.0)
        for depth in (1, 2, 3):
            acc, st = sc.render(cam, ra.Settings(64, 64, 256, depth, 1, seed=1))
            print(f"extra {extra:3d} radius {radius:6.0f} depth {depth}: rays/sample {st['rays']/st['samples']:.4f} (expect {1 + (depth>1)*0.3935:.4f}+)", flush=True)
        sc.close()
