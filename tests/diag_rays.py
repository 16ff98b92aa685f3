"""Diagnostic (not collected by pytest): rays/sample and mean radiance by depth, GPU vs oracle."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, hx, raytracer_amd as ra
ra.rtc_init()
o = hx.oracle()
for name, stage, W, H in (("simple", 3, 96, 96), ("two_perlin", 2, 128, 72), ("simple_light", 2, 128, 72), ("final", 2, 96, 96)):
    o.set_stage(stage); o.build(name)
    sc, cam, view = ra.build_named_scene(name)
    for depth in (1, 2, 3, 4, view.max_depth):
        ref, st = o.oracle_render(W, H, 0, 64, depth=depth, rng="philox", seed=3, threads=0)
        acc, gs = sc.render(cam, ra.Settings(W, H, 1024, depth, view.integrator, seed=5))
        print(f"{name:12s} depth {depth:2d}: rays/sample oracle {st['rays']/st['samples']:.4f} gpu {gs['rays']/gs['samples']:.4f}   mean oracle {ref.mean()/64:.5f} gpu {(acc[...,:3]/acc[...,3:4]).mean():.5f}", flush=True)
    sc.close()
