"""Image-texture decoder fixtures from the REFERENCE's own loader (TEST INFRASTRUCTURE; runs only in the build container).

Small synthetic PNG, BMP, binary PGM and progressive-JPEG files, each decoded by the stb_image compiled into oracle/_ref/libref03.so
(stbi_load(path, .., 3), the call of 03_FullTracer/texture.c:66).  tests/golden/image_cases.npz holds the file bytes and the decoded
pixels; the product's texture_image_init must load the same texels (tests/test_host_api.py).
  PNG : written here chunk by chunk (zlib from the standard library) so that every colour type, bit depth, row filter and the Adam7
        interlace is covered: grey 1/2/4/8/16, grey+alpha 8/16, RGB 8/16, RGBA 8/16, palette 1/2/4/8 (+ tRNS), split IDAT chunks;
  BMP : Pillow's writers (1, 8, 24 bits, palette) and hand-made headers (4-bit palette, 16-bit 5-5-5 and 5-6-5 bit fields, 32-bit
        with and without masks, top-down rows, V4 header);
  JPEG: Pillow / libjpeg-turbo with progressive=True: 4:4:4, 4:2:2, 4:2:0, grey, odd sizes, restart intervals;
  PGM : binary P5.
    make -C oracle ref && python tests/golden/make_image_cases.py
"""
import ctypes as C
import io
import os
import struct
import sys
import tempfile
import zlib

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
    if not p:
        return None
    return np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()


rng = np.random.default_rng(11)


def chunk(tag, body):
    return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def filter_rows(rows, bpp, filters):
    """rows: list of bytes objects of one (sub-)image; filters: the filter type of each row"""
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for r, f in zip(rows, filters):
        line = bytearray(len(r))
        for i, x in enumerate(r):
            a = r[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, paeth(a, b, c))[f]
            line[i] = (x - pred) & 255
        out.append(f); out += line
        prev = r
    return bytes(out)


def pack_row(samples, depth):
    """samples: 1-D uint16 array of one row's samples -> bytes at `depth` bits, big-endian / most significant first"""
    if depth == 16:
        return samples.astype(">u2").tobytes()
    if depth == 8:
        return samples.astype(np.uint8).tobytes()
    per = 8 // depth
    pad = (-len(samples)) % per
    s = np.concatenate([samples, np.zeros(pad, samples.dtype)]).reshape(-1, per).astype(np.uint32)
    shifts = np.array([8 - depth * (k + 1) for k in range(per)], dtype=np.uint32)
    return (s << shifts).sum(1).astype(np.uint8).tobytes()


def make_png(w, h, color, depth, interlace=0, palette=None, trns=None, split=1, level=6):
    chan = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
    top = len(palette) if color == 3 else (1 << depth)
    img = rng.integers(0, top, size=(h, w, chan)).astype(np.uint16)
    yy, xx = np.mgrid[0:h, 0:w]
    if color != 3:                                               # some structure, so that the filters have something to predict
        img = ((img.astype(np.int64) // 4 + (xx[..., None] * 3 + yy[..., None] * 5) * (top // 64 + 1)) % top).astype(np.uint16)
    bpp = max(1, chan * depth // 8)
    raw = b""
    passes = [(0, 0, 1, 1)] if not interlace else [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    for x0, y0, dx, dy in passes:
        sub = img[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        rows = [pack_row(sub[y].reshape(-1), depth) for y in range(sub.shape[0])]
        raw += filter_rows(rows, bpp, rng.integers(0, 5, size=len(rows)).tolist())
    z = zlib.compress(raw, level)
    body = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, interlace))
    if palette is not None:
        body += chunk(b"PLTE", bytes(np.asarray(palette, np.uint8).reshape(-1)))
    if trns is not None:
        body += chunk(b"tRNS", trns)
    body += chunk(b"tEXt", b"Comment\0made for a test")
    step = (len(z) + split - 1) // split
    for k in range(0, len(z), step):
        body += chunk(b"IDAT", z[k:k + step])
    return body + chunk(b"IEND", b"")


def bmp_file(w, h, bpp, rows, palette=b"", masks=b"", hsz=40, top_down=False, compress=0):
    stride = (len(rows[0]) + 3) & ~3
    data = b"".join(r + bytes(stride - len(r)) for r in (rows if top_down else rows[::-1]))
    if hsz == 40:
        hdr = struct.pack("<IiiHHIIiiII", 40, w, -h if top_down else h, 1, bpp, compress, len(data), 2835, 2835, 0, 0) + masks
    else:                                                         # V4, 108 bytes: masks inside the header
        m = struct.unpack("<4I", masks)
        hdr = struct.pack("<IiiHHIIiiII", 108, w, -h if top_down else h, 1, bpp, compress, len(data), 2835, 2835, 0, 0) + struct.pack("<4I", *m) + b"sRGB"[::-1] + bytes(48)
    off = 14 + len(hdr) + len(palette)
    return b"BM" + struct.pack("<IHHI", off + len(data), 0, 0, off) + hdr + palette + data


files = {}
# ---- PNG
for (w, h) in ((37, 23), (8, 8), (1, 1), (5, 1), (1, 9), (3, 17)):
    for color, depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (4, (8, 16)), (6, (8, 16))):
        for depth in depths:
            for il in (0, 1):
                if (w, h) not in ((37, 23), (3, 17)) and (il or depth in (2, 16)):
                    continue
                files[f"png_c{color}_d{depth}_i{il}_{w}x{h}"] = make_png(w, h, color, depth, il, split=1 + (w % 3))
    for depth in (1, 2, 4, 8):
        npal = min(1 << depth, 200)
        pal = rng.integers(0, 256, size=(npal, 3))
        for il in (0, 1):
            if (w, h) != (37, 23) and il:
                continue
            files[f"png_c3_d{depth}_i{il}_{w}x{h}"] = make_png(w, h, 3, depth, il, palette=pal, trns=bytes(rng.integers(0, 256, size=npal // 2).tolist()) if depth == 8 else None)
files["png_rgb_trns"] = make_png(19, 11, 2, 8, trns=struct.pack(">HHH", 10, 20, 30))
files["png_stored"] = make_png(21, 13, 2, 8, level=0)
files["png_big_fixed"] = make_png(64, 48, 6, 8, level=1)
# ---- BMP
yy, xx = np.mgrid[0:29, 0:43]
rgb = np.stack([(xx * 6) % 256, (yy * 9) % 256, (xx * 3 + yy * 5) % 256], -1).astype(np.uint8)
rgb ^= rng.integers(0, 32, size=rgb.shape).astype(np.uint8)
for mode in ("RGB", "L", "P", "1"):
    b = io.BytesIO(); Image.fromarray(rgb).convert(mode).save(b, format="BMP"); files[f"bmp_pil_{mode}"] = b.getvalue()
h, w = rgb.shape[:2]
files["bmp_24_topdown"] = bmp_file(w, h, 24, [rgb[y, :, ::-1].tobytes() for y in range(h)], top_down=True)
pal16 = rng.integers(0, 256, size=(16, 4)).astype(np.uint8); pal16[:, 3] = 0
idx4 = rng.integers(0, 16, size=(h, w)).astype(np.uint8)
rows4 = []
for y in range(h):
    r = np.concatenate([idx4[y], np.zeros(w % 2, np.uint8)]).reshape(-1, 2)
    rows4.append(((r[:, 0] << 4) | r[:, 1]).astype(np.uint8).tobytes())
files["bmp_4bit"] = bmp_file(w, h, 4, rows4, palette=pal16.tobytes())
v16 = rng.integers(0, 65536, size=(h, w)).astype("<u2")
files["bmp_16_555"] = bmp_file(w, h, 16, [v16[y].tobytes() for y in range(h)])
files["bmp_16_565"] = bmp_file(w, h, 16, [v16[y].tobytes() for y in range(h)], masks=struct.pack("<III", 0xF800, 0x07E0, 0x001F), compress=3)
v32 = rng.integers(0, 2 ** 32, size=(h, w), dtype=np.uint64).astype("<u4")
files["bmp_32_plain"] = bmp_file(w, h, 32, [v32[y].tobytes() for y in range(h)])
files["bmp_32_masks"] = bmp_file(w, h, 32, [v32[y].tobytes() for y in range(h)], masks=struct.pack("<III", 0x0FF00000, 0x0003F000, 0x000000F8), compress=3)
files["bmp_32_v4"] = bmp_file(w, h, 32, [v32[y].tobytes() for y in range(h)], masks=struct.pack("<IIII", 0x00FF0000, 0x0000FF00, 0x000000FF, 0xFF000000), hsz=108, compress=3)
files["bmp_16_v4_odd_masks"] = bmp_file(w, h, 16, [v16[y].tobytes() for y in range(h)], masks=struct.pack("<IIII", 0x7000, 0x0FC0, 0x003E, 0), hsz=108, compress=3)
# ---- progressive JPEG
big = np.stack([(xx * 5 + yy) % 256, (yy * 7) % 256, (xx * 2 + yy * 3) % 256], -1).astype(np.uint8)
for name, arr, kw in (("444", big, dict(subsampling=0, quality=90)), ("422", big, dict(subsampling=1, quality=80)), ("420", big, dict(subsampling=2, quality=75)),
                      ("420_q35", big, dict(subsampling=2, quality=35)), ("grey", big[..., 0], dict(quality=85)), ("odd", big[:17, :9], dict(subsampling=2, quality=92)),
                      ("1x1", big[:1, :1], dict(quality=90)), ("restart", big, dict(subsampling=2, quality=80, restart_marker_blocks=2)),
                      ("wide", np.tile(big, (1, 3, 1))[:19], dict(subsampling=1, quality=70))):
    b = io.BytesIO(); Image.fromarray(arr).save(b, format="JPEG", progressive=True, **kw); files[f"pjpg_{name}"] = b.getvalue()
# ---- PGM
files["pgm_p5"] = b"P5\n# grey\n43 29\n255\n" + rgb[..., 1].tobytes()
files["ppm_p6_max100"] = b"P6 43 29 100\n" + (rgb // 3).tobytes()

out = {}
with tempfile.TemporaryDirectory() as tmp:
    for name, data in files.items():
        path = os.path.join(tmp, name)
        open(path, "wb").write(data)
        px = ref_decode(path)
        assert px is not None, name
        out["file_" + name] = np.frombuffer(data, np.uint8)
        out["rgb_" + name] = px
np.savez_compressed(os.path.join(HERE, "image_cases.npz"), **out)
print(len(files), "files,", sum(len(d) for d in files.values()), "bytes;", os.path.getsize(os.path.join(HERE, "image_cases.npz")), "bytes of fixture")
