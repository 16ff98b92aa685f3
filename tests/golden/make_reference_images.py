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

JOBS = [  # name, stage, scene, aspect, W, H, spp
    ("cornell_128", 3, "cornell", 0.0, 128, 128, 4096),
    ("cornell_16x9_240", 3, "cornell", 16 / 9, 240, 135, 4096),
    ("simple_96", 3, "simple", 0.0, 96, 96, 4096),
    ("final_96", 2, "final", 0.0, 96, 96, 1024),
    ("cornell02_64", 2, "cornell02", 0.0, 64, 64, 2048),
    ("cornell_smoke_64", 2, "cornell_smoke", 0.0, 64, 64, 2048),
    ("book1_160", 1, "book1", 0.0, 160, 90, 1024),
    ("book2_random_128", 2, "book2_random", 0.0, 128, 72, 512),
    ("two_perlin_128", 2, "two_perlin", 0.0, 128, 72, 512),
    ("simple_light_128", 2, "simple_light", 0.0, 128, 72, 1024),
    ("two_spheres_128", 2, "two_spheres", 0.0, 128, 72, 512),
    ("earth_128", 2, "earth", 0.0, 128, 72, 512),
    ("final_earth_96", 2, "final_earth", 0.0, 96, 96, 1024),
    ("box_field_96", 2, "box_field", 0.0, 96, 96, 2048),          # not a reference scene: rotated rects inside a BVH
    ("box_field_smoke_96", 2, "box_field_smoke", 0.0, 96, 96, 2048),   # ... plus a box-bounded and a sphere-bounded medium
]


def main():
    o = hx.oracle()
    only = set(sys.argv[1:])
    for name, stage, scene, aspect, W, H, spp in JOBS:
        if only and name not in only:
            continue
        o.set_stage(stage)
        o.build(scene, aspect=aspect)
        t0 = time.time()
        img, st = o.oracle_render(W, H, 0, spp, depth=0, rng="philox", seed=2024, threads=0, iterative=True, scrub_nan=True)
        np.savez_compressed(os.path.join(HERE, f"image_{name}.npz"), mean=(img / spp).astype(np.float32), spp=spp, stage=stage,
                            aspect=aspect, rays_per_sample=st["rays"] / st["samples"])
        print(f"{name}: {W}x{H}x{spp} in {time.time() - t0:.1f}s, rays/sample {st['rays'] / st['samples']:.3f}, mean {img.mean() / spp:.4f}", flush=True)


if __name__ == "__main__":
    main()
