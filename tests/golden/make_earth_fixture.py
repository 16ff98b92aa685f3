"""Earth-map fixtures from the REFERENCE's own image loader (TEST INFRASTRUCTURE; runs only in the build container).

The reference decodes images/earthmap.jpg with the stb_image that is compiled into oracle/_ref/libref03.so
(extern_stb_image.c, built in place by oracle/Makefile; nothing is copied).  This script calls that stbi_load and writes

  tests/golden/earthmap.jpg          the input asset itself (data, 161 KB): the product's JPEG decoder is tested on it
  tests/golden/earthmap_decode.npz   expected output: every 4th pixel of the reference's decode, row sums, column sums
  tests/golden/earthmap_512.ppm      the decode box-filtered 2x2 -> 512x256 binary PPM: the image asset of the `earth`
                                     and `final_earth` scenes in all three backends (product, oracle, compiled reference:
                                     stb reads binary PPM as well), small enough to travel with the repository

Re-run with:  make -C oracle ref && python tests/golden/make_earth_fixture.py
"""
import ctypes as C
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import hx  # noqa: E402

SRC = "/root/reference/images/earthmap.jpg"

lib = C.CDLL(hx.REF03_SO)
lib.stbi_load.restype = C.POINTER(C.c_ubyte)
lib.stbi_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
w, h, n = C.c_int(), C.c_int(), C.c_int(3)
p = lib.stbi_load(SRC.encode(), C.byref(w), C.byref(h), C.byref(n), 3)      # texture.c:66: components_per_pixel = 3
assert p, "stbi_load failed"
rgb = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
print("reference decode:", rgb.shape, "mean", rgb.mean(axis=(0, 1)))

shutil.copyfile(SRC, os.path.join(HERE, "earthmap.jpg"))
np.savez_compressed(os.path.join(HERE, "earthmap_decode.npz"), width=w.value, height=h.value, sub4=rgb[::4, ::4],
                    row_sums=rgb.astype(np.int64).sum(axis=(1, 2)), col_sums=rgb.astype(np.int64).sum(axis=(0, 2)))
small = rgb.astype(np.uint16).reshape(h.value // 2, 2, w.value // 2, 2, 3).sum(axis=(1, 3))
small = ((small + 2) // 4).astype(np.uint8)
with open(os.path.join(HERE, "earthmap_512.ppm"), "wb") as f:
    f.write(b"P6\n%d %d\n255\n" % (small.shape[1], small.shape[0]))
    f.write(small.tobytes())
print("wrote earthmap.jpg, earthmap_decode.npz, earthmap_512.ppm", small.shape)
