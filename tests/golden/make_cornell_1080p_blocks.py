"""BASELINE config 2 at its FULL size from the CPU oracle (TEST INFRASTRUCTURE): Cornell `setupScene`, 1920x1080, camera aspect 16/9,
depth 10, Philox mode, 1024 spp -- stored as the means of its 8x8-pixel blocks (240 x 135 x 3 float32, 389 KB; the full image
would be 25 MB), 65,536 samples per block.  tests/test_gpu_render.py::test_full_size_cornell_1080p_against_the_oracle_at_full_size
compares the GPU render of the same pixel grid block by block.
Run:  python tests/golden/make_cornell_1080p_blocks.py   (about four minutes on 8 cores)"""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import hx  # noqa: E402

W, H, SPP = 1920, 1080, 1024
o = hx.oracle()
o.set_stage(3)
o.build("cornell", arg=0, aspect=16 / 9)
t0 = time.time()
img, st = o.oracle_render(W, H, 0, SPP, depth=0, rng="philox", seed=1080, threads=0, iterative=True, scrub_nan=True)
blocks = (img / SPP).reshape(H // 8, 8, W // 8, 8, 3).mean(axis=(1, 3))
np.savez_compressed(os.path.join(HERE, "image_cornell_1080p_blocks.npz"), block_mean=blocks.astype(np.float32), spp=SPP, width=W, height=H,
                    rays_per_sample=st["rays"] / st["samples"], mean=float(img.mean() / SPP))
print(f"cornell 1080p x {SPP} spp in {time.time() - t0:.1f}s, rays/sample {st['rays'] / st['samples']:.4f}, mean {img.mean() / SPP:.5f}")
