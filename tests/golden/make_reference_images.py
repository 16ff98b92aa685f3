"""High-spp reference images from the CPU oracle in Philox mode (TEST INFRASTRUCTURE).

The oracle itself is pinned bit-exactly to the compiled reference (tests/test_oracle_golden.py); these images are the
"4096-spp CPU reference, rendered with the same scenes and the reference's pdf/material maths" of BASELINE.json.
Stored as mean linear radiance, float32, row 0 = bottom (the reference's pixel order, main.c:157).
Non-finite samples count as black (scrub_nan), like the GPU path; the reference itself would emit a garbage byte (F16).
Run:  python tests/golden/make_reference_images.py   (about two minutes on 8 cores)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import hx  # noqa: E402

JOBS = [  # name, stage, scene, aspect, W, H, spp, scene argument
    ("cornell_128", 3, "cornell", 0.0, 128, 128, 4096, 0),
    ("cornell_16x9_240", 3, "cornell", 16 / 9, 240, 135, 4096, 0),
    ("simple_96", 3, "simple", 0.0, 96, 96, 4096, 0),
    ("final_96", 2, "final", 0.0, 96, 96, 4096, 0),               # BASELINE config 3 (round 3: 1024 -> 4096 spp)
    ("cornell02_64", 2, "cornell02", 0.0, 64, 64, 2048, 0),
    ("cornell_smoke_64", 2, "cornell_smoke", 0.0, 64, 64, 2048, 0),
    ("book1_160", 1, "book1", 0.0, 160, 90, 2048, 0),
    ("book2_random_128", 2, "book2_random", 0.0, 128, 72, 2048, 0),
    ("two_perlin_128", 2, "two_perlin", 0.0, 128, 72, 2048, 0),
    ("simple_light_128", 2, "simple_light", 0.0, 128, 72, 2048, 0),
    ("two_spheres_128", 2, "two_spheres", 0.0, 128, 72, 2048, 0),
    ("earth_128", 2, "earth", 0.0, 128, 72, 2048, 0),
    ("final_earth_96", 2, "final_earth", 0.0, 96, 96, 4096, 0),   # config 3 with the earth-map globe
    ("box_field_96", 2, "box_field", 0.0, 96, 96, 2048, 0),          # not a reference scene: rotated rects inside a BVH
    ("box_field_smoke_96", 2, "box_field_smoke", 0.0, 96, 96, 2048, 0),   # ... plus a box-bounded and a sphere-bounded medium
    ("million80_160", 1, "million", 0.0, 160, 90, 2048, 80),      # BASELINE config 5's scene at 80 x 80 + 4 spheres (01 integrator: sky)
]


def main():
    o = hx.oracle()
    only = set(sys.argv[1:])
    for name, stage, scene, aspect, W, H, spp, arg in JOBS:
        if only and name not in only:
            continue
        o.set_stage(stage)
        o.build(scene, arg=arg, aspect=aspect)
        t0 = time.time()
        img, st = o.oracle_render(W, H, 0, spp, depth=0, rng="philox", seed=2024, threads=0, iterative=True, scrub_nan=True)
        np.savez_compressed(os.path.join(HERE, f"image_{name}.npz"), mean=(img / spp).astype(np.float32), spp=spp, stage=stage,
                            aspect=aspect, arg=arg, rays_per_sample=st["rays"] / st["samples"])
        print(f"{name}: {W}x{H}x{spp} in {time.time() - t0:.1f}s, rays/sample {st['rays'] / st['samples']:.3f}, mean {img.mean() / spp:.4f}", flush=True)


if __name__ == "__main__":
    main()
