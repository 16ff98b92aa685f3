"""A PNG written by the REFERENCE's own writer (TEST INFRASTRUCTURE; build container only).

03_FullTracer/main.c:170-171: stbi_flip_vertically_on_write(1); stbi_write_png(path, W, H, 3, pixels, W * 3) -- called here in
oracle/_ref/libref03.so (stb_image_write v1.15 compiled in place through extern_stb_image.c) on a fixed 53 x 37 test image.
tests/golden/png_reference.npz: the pixels handed to the writer and the bytes of the file it wrote.
"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import hx  # noqa: E402

lib = C.CDLL(hx.REF03_SO)
rng = np.random.default_rng(11)
h, w = 37, 53
yy, xx = np.mgrid[0:h, 0:w]
img = np.stack([(xx * 255 // (w - 1)), (yy * 255 // (h - 1)), ((xx + yy) * 7) % 256], -1).astype(np.uint8)
img[10:20, 5:30] = rng.integers(0, 256, (10, 25, 3), dtype=np.uint8)           # a noisy patch: all five row filters get used
img[25:, :] = [12, 200, 99]                                                     # and flat rows
img = np.ascontiguousarray(img)
# round 3: more cases for the byte-exact writer -- a noisy "render" (sqrt-gamma gradient + Monte-Carlo-like noise: long literal runs,
# short matches), a one-pixel-wide image (every row is its own first pixel), a two-row image, a large flat image (258-byte matches and
# the 32 KiB window), and one file written WITHOUT the vertical flip
yy, xx = np.mgrid[0:48, 0:64]
base = np.stack([xx / 63.0, yy / 47.0, (xx + yy) / 110.0], -1)
noisy = np.clip(256 * np.sqrt(np.clip(base + rng.normal(0, 0.08, base.shape), 0, 0.998)), 0, 255).astype(np.uint8)
thin = rng.integers(0, 256, (17, 1, 3), dtype=np.uint8)
two_rows = rng.integers(0, 4, (2, 40, 3), dtype=np.uint8) * 60
flat = np.zeros((150, 100, 3), dtype=np.uint8); flat[..., 1] = 77; flat[40:60, 10:90] = rng.integers(0, 2, (20, 80, 3), dtype=np.uint8) * 255
cases = {"": (img, 1), "noisy": (np.ascontiguousarray(noisy), 1), "thin": (np.ascontiguousarray(thin), 1), "two_rows": (np.ascontiguousarray(two_rows), 1),
         "flat": (np.ascontiguousarray(flat), 1), "noflip": (np.ascontiguousarray(noisy[:20, :31]), 0)}
out = {}
with tempfile.TemporaryDirectory() as tmp:
    for name, (im, flip) in cases.items():
        p = os.path.join(tmp, "ref.png")
        lib.stbi_flip_vertically_on_write(flip)
        assert lib.stbi_write_png(p.encode(), im.shape[1], im.shape[0], 3, im.ctypes.data_as(C.c_void_p), im.shape[1] * 3) == 1
        data = open(p, "rb").read()
        sfx = "_" + name if name else ""
        out["pixels" + sfx] = im; out["png" + sfx] = np.frombuffer(data, dtype=np.uint8); out["flip" + sfx] = flip
        print(f"reference PNG {name or 'base'}: {im.shape[1]}x{im.shape[0]} -> {len(data)} bytes")
np.savez_compressed(os.path.join(HERE, "png_reference.npz"), **out)
