"""Parameter fuzz of rtc_render on one GPU (development aid): random image sizes, sample ranges, depths, kernels and scenes;
checks that every render returns finite sums, exact sample counts and a plausible ray count.  python tools/fuzz_render.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ra.rtc_init()
scenes = {}
for name in ("cornell", "simple", "cornell02", "book1", "final", "box_field_smoke", "book2_random"):
    scenes[name] = ra.build_named_scene(name, aspect=None)
bad = 0
for it in range(n):
    name = list(scenes)[rng.integers(len(scenes))]
    sc, cam, view = scenes[name]
    big = rng.random() < 0.15
    W, H = (int(rng.integers(300, 2200)), int(rng.integers(200, 1300))) if big else (int(rng.integers(2, 300)), int(rng.integers(2, 200)))
    b = int(rng.integers(0, 50)); spp = int(rng.choice([1, 2, 3, 7, 31, 32, 33, 64, 100, 257]))
    depth = int(rng.choice([0, 1, 2, view.max_depth, 40]))
    small_ok = name in ("cornell", "simple", "cornell02")
    bvh = sc.bvh_info()["nodes"] > 0
    kernel = int(rng.choice([0, 1] + ([2] if small_ok else []) + ([3] if bvh else [])))
    acc, st = sc.render(cam, ra.Settings(W, H, 0, depth, view.integrator, seed=it, sample_begin=b, sample_end=b + spp, kernel=kernel))
    ok = np.isfinite(acc).all() and (acc[..., 3] == spp).all() and st["samples"] == W * H * spp and (st["rays"] >= (st["samples"] if depth > 0 else 0)) and st["rays"] <= st["samples"] * max(depth, 1)
    if not ok:
        bad += 1
        print("BAD", name, W, H, b, spp, depth, kernel, st, flush=True)
print(f"fuzz: {n} renders, {bad} bad", flush=True)
for sc, cam, view in scenes.values():
    sc.close()
sys.exit(1 if bad else 0)
