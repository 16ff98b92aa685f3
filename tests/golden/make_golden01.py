"""Generate tests/golden/ref01_pins.npz from the compiled 01_BasicTracer (TEST INFRASTRUCTURE).

Runs only in the build container: loads oracle/_ref/libref01.so, which oracle/Makefile compiles from
/root/reference/01_BasicTracer/src/main.c where it lies (nothing is copied).  Every array is an input or an output of the
reference's own setup_scene / camera_get_ray / hittable_list_hit / material_scatter / ray_color / write_color.
  make -C oracle ref && python tests/golden/make_golden01.py [--main]
--main additionally runs the reference's REAL main() (1920x1080 x 8 spp, ~2 min), the harness loop at that size and the
oracle's stage-1 loop at that size, asserts that the three PPM files are byte-identical, and records their SHA-256:
that is what ties the harness loop (used at small sizes by the tests) and the oracle's 01 path to main() itself.
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import hx  # noqa: E402

rng = np.random.default_rng(20261111)
out = {}
r = hx.ref01().build(seed=1)
out["spheres"] = r.spheres()
out["camera"] = r.camera()
g = np.stack(np.meshgrid((np.arange(64) + .5) / 63, (np.arange(36) + .5) / 35), -1).reshape(-1, 2)
out["g_uv"] = g
out["g_rays"] = r.camera_rays(g, seed=7)
out["g_rec"], out["g_mat"], out["g_prim"] = r.hit_rays(out["g_rays"])
assert (out["g_prim"][out["g_rec"][:, 0] == 1] >= 0).all()
# secondary-style rays: origins on the first-hit points, random directions and lengths
hit = out["g_rec"][:, 0] == 1
pts = out["g_rec"][hit, 2:5]
d = rng.normal(size=pts.shape)
d *= rng.uniform(0.2, 30.0, (len(pts), 1)) / np.linalg.norm(d, axis=1, keepdims=True)
out["sec_rays"] = np.concatenate([pts, d], 1)
out["sec_rec"], out["sec_mat"], out["sec_prim"] = r.hit_rays(out["sec_rays"])
out["g_colors"] = r.ray_colors(out["g_rays"], 8, seed=33)                       # 01 main.c:39-65, depth 8 (:154)
out["g_colors_d3"] = r.ray_colors(out["g_rays"], 3, seed=34)
# material_scatter on synthetic records (01 material.h:66-123)
n = 400
dirs = rng.normal(size=(n, 3)) * rng.uniform(0.1, 20, (n, 1))
nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
front = (np.einsum("ij,ij->i", dirs, nrm) < 0)
nrm = np.where(front[:, None], nrm, -nrm) * np.where(rng.uniform(size=(n, 1)) < 0.8, 1, -1)   # mostly faced, some not
out["scatter_in"] = np.concatenate([rng.uniform(-5, 5, (n, 3)), dirs, rng.uniform(-5, 5, (n, 3)), nrm, rng.integers(0, 2, (n, 1))], 1)
for name, mt, matp in (("lambertian", "lambertian", [0.4, 0.2, 0.1, 0]), ("metal", "metal", [0.7, 0.6, 0.5, 0.0]),
                       ("metalfuzz", "metal", [0.8, 0.8, 0.8, 0.5]), ("dielectric", "dielectric", [1.5, 0, 0, 0])):
    out["scatter_" + name] = r.scatter(mt, matp, out["scatter_in"], seed=9)
# the output edge: write_color(FILE*, ...) on byte boundaries, and whole small renders through the loop of main.c:168-191
k = np.arange(0, 257)
edge = np.concatenate([(k / 256.0) ** 2, np.nextafter((k / 256.0) ** 2, 0), np.nextafter((k / 256.0) ** 2, 4), [0.0, 1.0, 4.0, 1e-300]])
wc = np.stack([edge, edge[::-1], np.roll(edge, 7)], 1) * 8
out["wc_in"] = wc
with tempfile.TemporaryDirectory() as tmp:
    p = os.path.join(tmp, "a.ppm")
    r.write_colors(p, wc, 8)
    out["wc_text"] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
    r.build(seed=1)
    r.render_ppm(p, 64, 36, 4, 8)                                                # stream continues from setup_scene, like main()
    out["ppm_64x36x4"] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
    r.build(seed=1)
    r.render_ppm(p, 96, 54, 2, 8, seed=5)
    out["ppm_96x54x2_seed5"] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
    if "--main" in sys.argv:
        pm, ph, po = (os.path.join(tmp, x) for x in ("main.ppm", "harness.ppm", "oracle.ppm"))
        assert r.run_main(pm) == 0
        r.build(seed=1)
        r.render_ppm(ph, 1920, 1080, 8, 8)
        o = hx.oracle()
        o.set_stage(1)
        o.build("book1", aspect=16 / 9, seed=1)
        o.render_ppm01(po, 1920, 1080, 8, 8)
        hm, hh, ho = (hashlib.sha256(open(x, "rb").read()).digest() for x in (pm, ph, po))
        assert hm == hh, "the harness loop does not reproduce the reference's main()"
        assert hm == ho, "the oracle's stage-1 loop does not reproduce the reference's main()"
        out["main_1920x1080x8_sha256"] = np.frombuffer(hm, dtype=np.uint8)
        out["main_1920x1080x8_bytes"] = np.array([os.path.getsize(pm)])
        print("main() == harness loop == oracle stage 1 at 1920x1080x8:", hm.hex())
    else:
        old = os.path.join(HERE, "ref01_pins.npz")
        if os.path.exists(old):
            prev = np.load(old)
            for key in ("main_1920x1080x8_sha256", "main_1920x1080x8_bytes"):
                if key in prev:
                    out[key] = prev[key]
np.savez_compressed(os.path.join(HERE, "ref01_pins.npz"), **out)
print("wrote ref01_pins.npz:", {k: v.shape for k, v in out.items()})
