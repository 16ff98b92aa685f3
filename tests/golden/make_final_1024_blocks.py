"""BASELINE config 3 at its FULL size from the CPU oracle (TEST INFRASTRUCTURE): the 02_AdvancedTracer final scene (02.../main.c:230-289),
1024x1024, depth 8, integrator 1 (02's ray_color), Philox mode, SPP samples per pixel -- stored as the means of its 8x8-pixel blocks
(128 x 128 x 3 float32), 64 x SPP samples per block.  tests/test_gpu_render.py::test_full_size_final_1024_against_the_oracle_at_full_size
compares the GPU render of the same pixel grid block by block.
Run:  python tests/golden/make_final_1024_blocks.py [spp = 256]"""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import hx  # noqa: E402

W, H = 1024, 1024
SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 256
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(HERE, "image_final_1024_blocks.npz")
o = hx.oracle()
o.set_stage(2)
o.build("final", arg=0, aspect=0.0)
t0 = time.time()
img, st = o.oracle_render(W, H, 0, SPP, depth=0, rng="philox", seed=1024, threads=0, iterative=True, scrub_nan=True)
blocks = (img / SPP).reshape(H // 8, 8, W // 8, 8, 3).mean(axis=(1, 3))
np.savez_compressed(OUT, block_mean=blocks.astype(np.float32), spp=SPP, width=W, height=H,
                    rays_per_sample=st["rays"] / st["samples"], mean=float(img.mean() / SPP))
print(f"final 1024^2 x {SPP} spp in {time.time() - t0:.1f}s, rays/sample {st['rays'] / st['samples']:.4f}, mean {img.mean() / SPP:.5f}")
