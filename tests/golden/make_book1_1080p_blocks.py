"""BASELINE config 1 at its FULL size from the CPU oracle (TEST INFRASTRUCTURE): the Book-1 random-sphere scene of 01_BasicTracer
(01.../main.c:81-106, rand() seed 1 as the reference draws it; camera (13,2,3) -> 0, vfov 30, aperture 0.1, focus 10, main.c:142-154),
1920x1080, camera aspect 16/9, 01's ray_color with the sky background, depth 8, Philox mode, SPP samples per pixel -- stored as the means
of its 8x8-pixel blocks (135 x 240 x 3 float32), 64 x SPP samples per block.
tests/test_gpu_render.py::test_full_size_book1_1080p_against_the_oracle_at_full_size compares the GPU render of the same pixel grid.
Run:  python tests/golden/make_book1_1080p_blocks.py [spp = 256]"""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import hx  # noqa: E402

W, H = 1920, 1080
SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 256
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(HERE, "image_book1_1080p_blocks.npz")
o = hx.oracle()
o.set_stage(1)
o.build("book1", arg=0, aspect=W / H)
t0 = time.time()
img, st = o.oracle_render(W, H, 0, SPP, depth=0, rng="philox", seed=1080, threads=0, iterative=True, scrub_nan=True)
blocks = (img / SPP).reshape(H // 8, 8, W // 8, 8, 3).mean(axis=(1, 3))
np.savez_compressed(OUT, block_mean=blocks.astype(np.float32), spp=SPP, width=W, height=H,
                    rays_per_sample=st["rays"] / st["samples"], mean=float(img.mean() / SPP))
print(f"book1 1080p x {SPP} spp in {time.time() - t0:.1f}s, rays/sample {st['rays'] / st['samples']:.4f}, mean {img.mean() / SPP:.5f}")
