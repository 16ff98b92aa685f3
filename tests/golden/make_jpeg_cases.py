"""JPEG decoder fixtures from the REFERENCE's own loader (TEST INFRASTRUCTURE; runs only in the build container).

Small synthetic JPEG files (written here with PIL: 4:4:4, 4:2:2, 4:2:0, grey, restart intervals, odd and one-pixel sizes)
and images/earthmap.jpg, each decoded by the stb_image compiled into oracle/_ref/libref03.so (stbi_load(path, .., 3), the call
of 03_FullTracer/texture.c:66).  tests/golden/jpeg_cases.npz holds the file bytes and the decoded pixels; the product's
rt_jpeg_decode must return the same bytes.  earthmap_decode.npz gets the FULL reference decode of the 1024x512 asset.
    make -C oracle ref && python tests/golden/make_jpeg_cases.py
"""
import ctypes as C
import io
import os
import sys
import tempfile

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import hx  # noqa: E402

lib = C.CDLL(hx.REF03_SO)
lib.stbi_load.restype = C.POINTER(C.c_ubyte)
lib.stbi_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]


def ref_decode(path):
    w, h, n = C.c_int(), C.c_int(), C.c_int()
    p = lib.stbi_load(path.encode(), C.byref(w), C.byref(h), C.byref(n), 3)
    assert p, path
    return np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()


rng = np.random.default_rng(7)
out = {}
names = []
with tempfile.TemporaryDirectory() as tmp:
    for size in ((131, 93), (64, 48), (17, 9), (1, 1), (2, 33), (33, 2)):
        w, h = size
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([127 + 100 * np.sin(xx / 9.0 + 1), 127 + 100 * np.cos(yy / 7.0), 127 + 90 * np.sin((xx + yy) / 11.0)], -1)
        base = np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8)
        for tag, kw in (("444", dict(subsampling=0)), ("422", dict(subsampling=1)), ("420", dict(subsampling=2)),
                        ("420rst", dict(subsampling=2, restart_marker_blocks=2)), ("grey", dict()), ("444q30", dict(subsampling=0, quality=30))):
            im = Image.fromarray(base[..., 0] if tag == "grey" else base)
            buf = io.BytesIO()
            kw = dict(kw)
            kw.setdefault("quality", 88)
            try:
                im.save(buf, format="JPEG", **kw)
            except TypeError:
                continue
            name = f"{tag}_{w}x{h}"
            path = os.path.join(tmp, name + ".jpg")
            open(path, "wb").write(buf.getvalue())
            out[name + "_jpg"] = np.frombuffer(buf.getvalue(), dtype=np.uint8)
            out[name + "_rgb"] = ref_decode(path)
            names.append(name)
out["names"] = np.array(names)
np.savez_compressed(os.path.join(HERE, "jpeg_cases.npz"), **out)
earth = ref_decode("/root/reference/images/earthmap.jpg")
np.savez_compressed(os.path.join(HERE, "earthmap_decode.npz"), width=earth.shape[1], height=earth.shape[0], rgb=earth,
                    sub4=earth[::4, ::4], row_sums=earth.astype(np.int64).sum(axis=(1, 2)), col_sums=earth.astype(np.int64).sum(axis=(0, 2)))
print(len(names), "cases;", "earth", earth.shape, os.path.getsize(os.path.join(HERE, "earthmap_decode.npz")), os.path.getsize(os.path.join(HERE, "jpeg_cases.npz")))
