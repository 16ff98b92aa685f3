"""Generate tests/golden/*.npz from the REFERENCE ITSELF (TEST INFRASTRUCTURE).

Runs only in the build container: it loads oracle/_ref/libref03.so and libref02.so, which oracle/Makefile compiles
from the reference's sources where they lie under /root/reference (nothing is copied).  Every array written here is
an input or an output of the reference's own functions, reached through the probe harness oracle/harness.c.
Re-run with:  make -C oracle ref && python tests/golden/make_golden.py
The reference ships no tests or golden vectors, so these fixtures ARE the parity pin.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import hx  # noqa: E402

rng = np.random.default_rng(20241111)


def grid(nx, ny):
    return np.stack(np.meshgrid((np.arange(nx) + .5) / (nx - 1), (np.arange(ny) + .5) / (ny - 1)), -1).reshape(-1, 2)


def random_rays(n, lo, hi, toward=None, spread=1.0):
    o = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3)) * spread
    if toward is not None:
        d = (np.asarray(toward) - o) / 50.0 + rng.normal(size=(n, 3)) * spread
    return np.concatenate([o, d, rng.uniform(0, 1, (n, 1))], 1)


def first_hit_fixture(h, scene, aspect=0.0):
    out = {}
    h.build(scene, aspect=aspect)
    out["view"] = h.view()
    out["camera"] = h.camera()
    out["leaves"] = h.dump_leaves()
    for tag, g in (("g64x36", grid(64, 36)), ("g32x32", grid(32, 32))):
        rays = h.camera_rays(g, seed=7)
        rec, prim = h.hit_rays(rays)
        assert (prim != -2).all(), "leaf walker lost a hit"
        out[tag + "_uv"], out[tag + "_rays"], out[tag + "_rec"], out[tag + "_prim"] = g, rays, rec, prim
    # secondary-style rays from inside the scene (origins on surfaces: the first-hit points, random directions)
    rec = out["g64x36_rec"]
    hit = rec[:, 0] == 1
    pts = rec[hit, 2:5]
    d = rng.normal(size=pts.shape)
    d *= rng.uniform(0.2, 300.0, (len(pts), 1)) / np.linalg.norm(d, axis=1, keepdims=True)
    rays2 = np.concatenate([pts, d, rng.uniform(0, 1, (len(pts), 1))], 1)
    rec2, prim2 = h.hit_rays(rays2)
    assert (prim2 != -2).all()
    out["sec_rays"], out["sec_rec"], out["sec_prim"] = rays2, rec2, prim2
    return out


FIRST_HIT = (("simple", 3, "ref_simple"), ("cornell", 3, "ref_cornell"), ("cornell_scenes_c", 3, "cornell"),
             ("cornell_16x9", 3, "ref_cornell"), ("book1", 3, "book1"), ("final03", 3, "final"),
             ("final02", 2, "ref_final"), ("final02_scenes_c", 2, "final"), ("cornell02", 2, "ref_cornell02"),
             ("cornell02_scenes_c", 2, "cornell02"), ("cornell_smoke", 2, "cornell_smoke"), ("book2_random", 2, "book2_random"),
             ("two_perlin", 2, "two_perlin"), ("simple_light", 2, "simple_light"), ("two_spheres", 2, "two_spheres"),
             ("earth", 2, "earth"), ("final_earth", 2, "final_earth"),
             ("box_field", 2, "box_field"), ("box_field_smoke", 2, "box_field_smoke"))   # image texture: tests/golden/earthmap_512.ppm (make_earth_fixture.py)


def task_first_hit(name):
    _, stage, scene = next(x for x in FIRST_HIT if x[0] == name)
    h = hx.ref03() if stage == 3 else hx.ref02()
    fx = first_hit_fixture(h, scene, aspect=16 / 9 if name == "cornell_16x9" else 0.0)
    np.savez_compressed(os.path.join(HERE, f"first_hit_{name}.npz"), **fx)
    print("first_hit", name, "leaves", len(fx["leaves"]), "hits", int((fx["g64x36_prim"] >= 0).sum()), flush=True)


def task_unit():
    r3, r2 = hx.ref03(), hx.ref02()
    # ---- unit vectors of pure functions ----------------------------------------------------------------------
    u = {}
    rays = random_rays(600, -100, 700, toward=(278, 278, 278), spread=2.0)
    rays[:50, 3] = 0.0  # zero direction components exercise the NaN/inf branches of rect_hit (F20)
    rays[50:100, 4] = 0.0
    u["rays"] = rays
    for i, (rect, tmin) in enumerate((([0, 0, 0, 555, 0, 555, 555, 2], 0.001), ([213, 343, 0, 0, 227, 332, 554, 1], 0.0001), ([0, 555, 0, 555, 0, 0, 555, 0], 0.001))):
        u[f"rect{i}_p"], u[f"rect{i}_tmin"] = np.array(rect, float), tmin
        u[f"rect{i}_rec"] = r3.rect_hit(rect, rays, tmin)
    for i, sph in enumerate(([190, 90, 190, 90], [278, 278, 278, 400], [0, -1000, 0, 1000])):
        u[f"sph{i}_p"] = np.array(sph, float)
        u[f"sph{i}_rec"] = r3.sphere_hit(sph, rays)
    u["msph_p"] = np.array([400, 400, 200, 430, 400, 200, 0, 1, 50], float)
    u["msph_rec"] = r3.moving_sphere_hit(u["msph_p"], rays)
    for i, box in enumerate(([0, 0, 0, 165, 330, 165, 15, 1, 265, 0, 295], [0, 0, 0, 165, 165, 165, -18, 1, 130, 0, 65], [0, 0, 0, 165, 330, 165, 15, 0, 265, 0, 295],
                             [0, 0, 0, 100, 200, 300, 33, 2, -50, 20, 100])):
        u[f"box{i}_p"] = np.array(box, float)
        u[f"box{i}_rec"] = r3.box_hit(box, rays)
    u["aabb_p"] = np.array([100, 50, 200, 400, 330, 460], float)
    u["aabb_out"] = r3.aabb_hit(u["aabb_p"], rays)
    uvn = rng.normal(size=(300, 3)); uvn /= np.linalg.norm(uvn, axis=1, keepdims=True)
    nn = rng.normal(size=(300, 3)); nn /= np.linalg.norm(nn, axis=1, keepdims=True)
    nn *= -np.sign((uvn * nn).sum(1, keepdims=True))
    u["refract_in"] = np.concatenate([uvn, nn, rng.choice([1.5, 1 / 1.5, 1.33], (300, 1))], 1)
    u["refract_out"] = r3.refract(u["refract_in"])
    u["reflectance_in"] = np.stack([rng.uniform(0, 1, 300), rng.choice([1.5, 1 / 1.5], 300)], 1)
    u["reflectance_out"] = r3.reflectance(u["reflectance_in"])
    u["onb_in"] = np.concatenate([nn, np.eye(3), -np.eye(3), [[0.95, 0.1, 0.1], [0.9, 0.3, 0.3]]])
    u["onb_out"] = r3.onb(u["onb_in"])
    u["cospdf_in"] = np.concatenate([nn, rng.normal(size=(300, 3)) * 5], 1)
    u["cospdf_out"] = r3.cosine_pdf(u["cospdf_in"])
    u["spdf_out"] = r3.scattering_pdf(u["cospdf_in"])
    r3.build("ref_cornell")
    org = rng.uniform(1, 554, (400, 3))
    tgt = np.where(rng.uniform(size=(400, 1)) < 0.5, rng.uniform([213, 554, 227], [343, 554, 332], (400, 3)), [190, 90, 190] + rng.normal(size=(400, 3)) * 60)
    u["lightpdf_in"] = np.concatenate([org, (tgt - org) * rng.uniform(0.01, 2, (400, 1))], 1)
    u["lightpdf_out"] = r3.light_pdf(u["lightpdf_in"])
    u["lightrand_in"] = org
    u["lightrand_out"] = r3.light_random(org, seed=11)
    for k in range(6):
        try:
            u[f"sampler{k}"] = r3.sampler(k, 200, seed=5)
        except RuntimeError:
            pass
    pts = rng.uniform(-300, 600, (300, 3))
    noise, turb, ranvec, perm = r3.perlin(pts, seed=3)
    u["perlin_in"], u["perlin_noise"], u["perlin_turb"], u["perlin_ranvec"], u["perlin_perm"] = pts, noise, turb, ranvec, perm
    tin = np.concatenate([rng.uniform(-0.2, 1.2, (300, 2)), pts], 1)
    u["tex_in"] = tin
    u["tex_solid"] = r3.texture(0, [0.1, 0.2, 0.3], [0, 0, 0], tin)
    u["tex_checker"] = r3.texture(1, [0.5, 0.5, 0.5], [0.2, 0.2, 0.2], tin)
    u["tex_noise"] = r3.texture(2, [5.0, 0, 0], [0, 0, 0], tin, seed=3)
    sums = np.concatenate([rng.uniform(0, 400, (500, 3)), [[0, 0, 0], [256, 255.744, 255.745], [1e9, -1, 64]]])
    u["wc_in"] = sums
    u["wc_out"] = r3.write_color(sums, 256)
    # material scatter, stage 3 and stage 2
    srays = random_rays(200, 0, 555, spread=1.0)
    nrm = rng.normal(size=(200, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm *= -np.sign((srays[:, 3:6] * nrm).sum(1, keepdims=True))
    in14 = np.concatenate([srays, rng.uniform(0, 555, (200, 3)), nrm, rng.integers(0, 2, (200, 1))], 1)
    u["scatter_in"] = in14
    for mname, matp in (("lambertian", [0.65, 0.05, 0.05, 0]), ("metal", [0.8, 0.85, 0.88, 0.0]), ("metalfuzz", [0.8, 0.8, 0.8, 1.0]), ("dielectric", [1.5, 0, 0, 0]),
                        ("light", [15, 15, 15, 0]), ("isotropic", [0.2, 0.4, 0.9, 0])):
        mt = "metal" if mname == "metalfuzz" else mname
        u[f"scatter3_{mname}"] = r3.scatter(mt, matp, in14, seed=9)
        u[f"scatter2_{mname}"] = r2.scatter(mt, matp, in14, seed=9)
    np.savez_compressed(os.path.join(HERE, "unit_vectors.npz"), **u)
    print("unit vectors", len(u))



def task_pins(which):
    """One reference library instance per pin: the reference's registries are fixed 1024-slot arrays that are
    never reset (F17), so every scene is built in a fresh process."""
    import hashlib
    r3 = hx.ref03() if which.startswith(("cornell_", "simple")) and not which.startswith("cornell02") else None
    r2 = hx.ref02() if r3 is None else None
    p = {}
    if which == "cornell_colors":
        r3.build("ref_cornell")
        rays = r3.camera_rays(grid(24, 24), seed=21)
        p["cornell_rays"] = rays
        p["cornell_colors"] = r3.ray_colors(rays, 10, seed=33)
    elif which == "cornell_small":
        r3.build("ref_cornell")
        p["cornell_refloop_64x64x4"] = r3.render_refloop(64, 64, 4, 10)
    elif which == "cornell_big":
        r3.build("ref_cornell")
        big = r3.render_refloop(256, 256, 16, 10)
        p["cornell_refloop_256x256x16_sha256"] = np.frombuffer(hashlib.sha256(big.tobytes()).digest(), dtype=np.uint8)
        p["cornell_refloop_256x256x16_mean"] = big.reshape(-1, 3).mean(0)
    elif which == "simple":
        r3.build("ref_simple")
        p["simple_refloop_48x48x4"] = r3.render_refloop(48, 48, 4, 10)
    elif which == "final_colors":
        r2.build("ref_final")
        rays = r2.camera_rays(grid(16, 16), seed=21)
        p["final_rays"] = rays
        p["final_colors"] = r2.ray_colors(rays, 8, seed=33)
    elif which == "final_small":
        r2.build("ref_final")
        p["final_refloop_32x32x2"] = r2.render_refloop(32, 32, 2, 8)
    elif which == "cornell02":
        r2.build("ref_cornell02")
        p["cornell02_refloop_32x32x4"] = r2.render_refloop(32, 32, 4, 8)
    elif which == "smoke":
        r2.build("cornell_smoke")
        p["smoke_refloop_32x32x4"] = r2.render_refloop(32, 32, 4, 8)
    elif which == "book2":
        r2.build("book2_random")
        p["book2_refloop_32x18x2"] = r2.render_refloop(32, 18, 2, 8)   # height 18 -> step 1: rows 0..15 rendered (F3)
    elif which == "earth":                     # image texture (texture.c:136-152) through the reference's own loader
        r2.build("earth")
        p["earth_refloop_48x32x4"] = r2.render_refloop(48, 32, 4, 8)
        r2.build("earth")
        rays = r2.camera_rays(grid(24, 16), seed=21)
        p["earth_rays"] = rays
        p["earth_colors"] = r2.ray_colors(rays, 8, seed=33)
    np.savez_compressed(os.path.join(HERE, f"_pin_{which}.npz"), **p)
    print("pin", which, flush=True)


PINS = ("cornell_colors", "cornell_small", "cornell_big", "simple", "final_colors", "final_small", "cornell02", "smoke", "book2", "earth")


def main():
    import subprocess
    if len(sys.argv) > 1:
        kind, _, arg = sys.argv[1].partition(":")
        {"first_hit": task_first_hit, "pins": task_pins}.get(kind, lambda a: task_unit())(arg)
        return
    tasks = [f"first_hit:{x[0]}" for x in FIRST_HIT] + ["unit:"] + [f"pins:{w}" for w in PINS]
    for t in tasks:
        subprocess.run([sys.executable, os.path.abspath(__file__), t], check=True)
    merged = {}
    for w in PINS:
        f = os.path.join(HERE, f"_pin_{w}.npz")
        merged.update(np.load(f))
        os.remove(f)
    np.savez_compressed(os.path.join(HERE, "integrator_pins.npz"), **merged)
    print("integrator pins", sorted(merged))


if __name__ == "__main__":
    main()
