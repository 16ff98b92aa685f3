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
with tempfile.TemporaryDirectory() as tmp:
    p = os.path.join(tmp, "ref.png")
    lib.stbi_flip_vertically_on_write(1)
    assert lib.stbi_write_png(p.encode(), w, h, 3, img.ctypes.data_as(C.c_void_p), w * 3) == 1
    data = open(p, "rb").read()
np.savez_compressed(os.path.join(HERE, "png_reference.npz"), pixels=img, png=np.frombuffer(data, dtype=np.uint8))
print("reference PNG:", len(data), "bytes")
