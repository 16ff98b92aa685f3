"""GPU: the FP64 closest-hit kernels, through the C-ABI, against (1) vectors produced by the compiled reference
(tests/golden/first_hit_*.npz) and (2) the CPU oracle on seeded random rays.

Bar: BIT-EXACT (np.array_equal on float64 / int32) for primitive id, t, p, normal, front_face and material type --
everything the reference computes with + - * / sqrt.  u,v of spheres go through acos/atan2, where the device libm and
glibc may differ in the last place: tolerance 4 ulp of 1.0 (8.9e-16), stated here.  Constant-medium hits are
stochastic in the reference (hittable.c:574 draws rand()) and are skipped by rtc_first_hit_f64; rays whose reference
hit is a medium are excluded.
"""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

UV_TOL = 4 * 2.220446049250313e-16

# fixture -> (scenes.c scene, aspect, stage the fixture was made with)
FIXTURES = {
    "simple": ("simple", None, 3), "cornell": ("cornell", None, 3), "cornell_16x9": ("cornell", 16 / 9, 3),
    "book1": ("book1", None, 3), "final03": ("final", None, 3), "final02": ("final", None, 2),
    "cornell02": ("cornell02", None, 2), "cornell_smoke": ("cornell_smoke", None, 2), "book2_random": ("book2_random", None, 2),
    "two_perlin": ("two_perlin", None, 2), "simple_light": ("simple_light", None, 2), "two_spheres": ("two_spheres", None, 2),
    "earth": ("earth", None, 2), "final_earth": ("final_earth", None, 2),
    "box_field": ("box_field", None, 2),                         # not a reference scene: rotated rects inside a BVH
    "box_field_smoke": ("box_field_smoke", None, 2),             # ... plus a box-bounded and a sphere-bounded medium
}
EXACT_COLS = [0, 1, 2, 3, 4, 5, 6, 7, 10, 11]   # hit, t, p, normal, front_face, material type


def check_against(rec_ref, prim_ref, prim, t, rec, stage):
    keep = rec_ref[:, 11] != 4                                   # reference hit a medium: stochastic, excluded
    cols = EXACT_COLS                                            # hit functions are the same code in 02 and 03
    assert np.array_equal(prim[keep], prim_ref[keep])
    assert np.array_equal(t[keep][prim_ref[keep] >= 0], rec_ref[keep][prim_ref[keep] >= 0, 1])
    assert np.array_equal(rec[keep][:, cols], rec_ref[keep][:, cols])
    hit = keep & (prim_ref >= 0)
    if hit.any():
        assert np.abs(rec[hit][:, 8:10] - rec_ref[hit][:, 8:10]).max() <= UV_TOL


@pytest.mark.parametrize("name", sorted(FIXTURES))
def test_first_hit_matches_reference_vectors(gpu, name):
    scene_name, aspect, stage = FIXTURES[name]
    fx = golden(f"first_hit_{name}.npz")
    sc, cam, view = gpu.build_named_scene(scene_name, aspect=aspect)
    try:
        assert np.array_equal(cam.fields(), fx["camera"])
        for tag in ("g64x36", "g32x32", "sec"):
            rays = fx[tag + "_rays"]
            prim, t, rec = sc.first_hit(rays, want_rec=True)
            # A ray whose reference hit is NOT a medium never had a medium accepted along the way (an accepted
            # medium hit could only be replaced by something nearer, which the surface-only scan finds as well), so
            # the surface answer must agree exactly; rays that ended in a medium are excluded inside check_against.
            check_against(fx[tag + "_rec"], fx[tag + "_prim"], prim, t, rec, stage)
    finally:
        sc.close()


@pytest.mark.parametrize("name", ["cornell", "cornell_16x9", "simple", "final03", "two_spheres"])
def test_primary_rays_generated_on_device(gpu, name):
    """camera_get_ray on device (camera.c:46-53): rays, prim ids and t bit-exact for aperture-0 cameras."""
    scene_name, aspect, stage = FIXTURES[name]
    fx = golden(f"first_hit_{name}.npz")
    sc, cam, view = gpu.build_named_scene(scene_name, aspect=aspect)
    try:
        for tag in ("g64x36", "g32x32"):
            uv, rays = fx[tag + "_uv"], fx[tag + "_rays"]
            lens = np.zeros((len(uv), 3))
            lens[:, 2] = rays[:, 6]                               # the time the reference drew
            prim, t, rays_dev = sc.primary_first_hit(cam, uv, lens)
            assert np.array_equal(rays_dev, rays)
            keep = fx[tag + "_rec"][:, 11] != 4
            assert np.array_equal(prim[keep], fx[tag + "_prim"][keep])
            hit = keep & (fx[tag + "_prim"] >= 0)
            assert np.array_equal(t[hit], fx[tag + "_rec"][hit, 1])
    finally:
        sc.close()


def _random_rays(rng, n, view, spread, max_zero_components=1):
    """Rays from around the camera and from inside the scene, any direction, with exact-zero components mixed in."""
    lf = np.array([view.lookfrom.x, view.lookfrom.y, view.lookfrom.z])
    la = np.array([view.lookat.x, view.lookat.y, view.lookat.z])
    o = np.where(rng.random((n, 1)) < 0.5, lf + rng.normal(size=(n, 3)) * spread * 0.05, la + (rng.random((n, 3)) - 0.5) * spread)
    d = rng.normal(size=(n, 3))
    # F20: a zero direction component makes rect_hit's t = +-inf; two of them make the in-rect test see NaN
    for k in range(max_zero_components):
        idx = np.nonzero(rng.random(n) < 0.05)[0]
        d[idx, rng.integers(0, 3, size=len(idx))] = 0.0
    tm = rng.random((n, 1))
    return np.concatenate([o, d, tm], axis=1)


@pytest.mark.parametrize("name,spread,zeros", [("cornell", 600.0, 2), ("cornell_smoke", 600.0, 2), ("two_perlin", 30.0, 2),
                                               ("final", 900.0, 1), ("book1", 12.0, 1), ("book2_random", 12.0, 1)])
def test_random_rays_against_oracle(gpu, oracle_lib, name, spread, zeros):
    """Brute-force leaf loop (<= 64 leaves) and linear BVH (> 64) both reproduce the oracle's list scan, ties included.

    The brute-force scenes also get rays with two zero direction components: there rect_hit registers inf/NaN hits
    (F20) and the product, scanning in the same order, must reproduce even those.  BVH scenes get at most one zero
    component (t = +-inf is rejected by the in-rect test): a NaN tmax makes the reference's answer depend on visit
    order, which a BVH does not preserve -- documented in DESIGN.md.
    """
    o = oracle_lib
    o.set_stage(3)                                                # hit functions are identical in 02 and 03
    o.build(name)
    sc, cam, view = gpu.build_named_scene(name)
    try:
        assert np.array_equal(sc.dump_leaves(), o.dump_leaves())
        rays = _random_rays(np.random.default_rng(1234), 20000, view, spread, zeros)
        o.srand(99)
        rec_ref, prim_ref = o.hit_rays(rays)
        prim, t, rec = sc.first_hit(rays, want_rec=True)
        keep = rec_ref[:, 11] != 4
        assert np.array_equal(prim[keep], prim_ref[keep])
        assert np.array_equal(rec[keep][:, EXACT_COLS], rec_ref[keep][:, EXACT_COLS], equal_nan=True)
        hit = keep & (prim_ref >= 0) & np.isfinite(rec_ref[:, 1])
        assert hit.sum() > 1000
        assert np.abs(rec[hit][:, 8:10] - rec_ref[hit][:, 8:10]).max() <= UV_TOL
    finally:
        sc.close()


def test_tmin_tmax_window_and_empty_input(gpu, oracle_lib):
    o = oracle_lib
    o.set_stage(3)
    o.build("cornell")
    sc, cam, view = gpu.build_named_scene("cornell")
    try:
        prim, t = sc.first_hit(np.zeros((0, 7)))
        assert prim.shape == (0,) and t.shape == (0,)
        rays = _random_rays(np.random.default_rng(5), 4000, view, 600.0)
        for tmin, tmax in ((0.001, 50.0), (10.0, 400.0), (0.0001, np.inf)):
            rec_ref, prim_ref = o.hit_rays(rays, tmin, tmax)
            prim, t, rec = sc.first_hit(rays, tmin, tmax, want_rec=True)
            assert np.array_equal(prim, prim_ref)
            assert np.array_equal(rec[:, EXACT_COLS], rec_ref[:, EXACT_COLS], equal_nan=True)
    finally:
        sc.close()


def test_million_sphere_bvh_against_oracle_subset(gpu, oracle_lib):
    """Config 5 stress scene at a reduced grid (arg = 80 spheres per side -> 6404 spheres): BVH result == oracle list scan."""
    o = oracle_lib
    o.set_stage(3)
    o.build("million", arg=80)
    sc, cam, view = gpu.build_named_scene("million", arg=80)
    try:
        assert sc.num_prims() == o.num_leaves() > 6000
        g = np.stack(np.meshgrid((np.arange(96) + .5) / 95, (np.arange(54) + .5) / 53), -1).reshape(-1, 2)
        rays = o.camera_rays(g, seed=3)
        rec_ref, prim_ref = o.hit_rays(rays)
        prim, t, rec = sc.first_hit(rays, want_rec=True)
        assert np.array_equal(prim, prim_ref)
        assert np.array_equal(rec[:, EXACT_COLS], rec_ref[:, EXACT_COLS])
    finally:
        sc.close()


@pytest.mark.parametrize("name,spread", [("cornell", 600.0), ("final", 900.0), ("book1", 12.0), ("book2_random", 12.0), ("box_field", 600.0)])
def test_production_fp32_traversal_agrees_with_fp64(gpu, name, spread):
    """rtc_first_hit_f32 = the traversal the path tracer really runs (scalar brute-force loop / flattened BVH + LDS stack).
    Stated tolerance against the bit-exact FP64 path on 40k random rays: same primitive for >= 99.5 % of the rays;
    where the primitives agree, |t32 - t64| / max(1, t64) <= 2e-4 for 99 % of the rays and <= 1e-2 for all of them
    (rays grazing the r = 1000 ground spheres lose digits in any FP32 formulation); a hit/miss disagreement (rays
    grazing an edge) on <= 0.2 % of the rays."""
    sc, cam, view = gpu.build_named_scene(name)
    try:
        rays = _random_rays(np.random.default_rng(77), 40000, view, spread, 0)
        rays = rays.astype(np.float32).astype(np.float64)          # both paths see the same float-representable rays
        p64, t64 = sc.first_hit(rays)
        p32, t32 = sc.first_hit_f32(rays)
        med = sc.dump_leaves()[:, 0] == 7
        if med.any():                                              # both skip media
            assert not med[p64[p64 >= 0]].any() and not med[p32[p32 >= 0]].any()
        same = p32 == p64
        assert same.mean() >= 0.995
        both = same & (p64 >= 0)
        assert both.sum() > 5000
        err = np.abs(t32[both] - t64[both]) / np.maximum(1.0, t64[both])
        print(f"{name}: same prim {same.mean():.5f}, t error p50 {np.median(err):.2e} p99 {np.quantile(err, 0.99):.2e} max {err.max():.2e}")
        assert np.quantile(err, 0.99) <= 2e-4 and err.max() <= 1e-2
        assert ((p32 >= 0) != (p64 >= 0)).mean() <= 0.002
    finally:
        sc.close()


@pytest.mark.parametrize("n_spheres", [1, 2, 3])
def test_bvh_over_one_leaf(gpu, monkeypatch, n_spheres):
    """A BVH whose root is a single leaf run (one or two primitives: its node's second child is RTC_BVH_EMPTY and must never be
    entered or shadow the first) and the smallest real tree (three).  RTC_BVH_THRESHOLD=0 makes flatten build a BVH for any
    scene.  Rays through the unit cube around the origin -- where an unguarded empty child's box would sit -- must find the
    spheres behind it; FP32 traversal against the FP64 closest-hit kernel."""
    monkeypatch.setenv("RTC_BVH_THRESHOLD", "0")
    L = gpu.lib()
    v3 = gpu.api.v3
    world = L.hittable_list_new(4)
    mat = L.material_lambertian_new_from_color(v3((0.5, 0.5, 0.5)))
    centres = [(5.0, 0.0, 0.0), (0.0, 6.0, 0.5), (-4.0, -4.0, 3.0)][:n_spheres]
    for c in centres:
        L.hittable_sphere_new(world, v3(c), 0.75, mat)
    sc = gpu.Scene(world, None, (0.5, 0.7, 1.0))
    try:
        info = sc.bvh_info()
        assert info["nodes"] == 1 and info["leaf_slots"] == n_spheres
        rng = np.random.default_rng(n_spheres)
        rays = []
        for c in centres:                                           # from beyond the origin, through the unit cube, at each sphere
            c = np.array(c)
            o = -c / np.linalg.norm(c) * 5.0 + rng.uniform(-0.3, 0.3, size=(500, 3))
            target = c + rng.uniform(-0.6, 0.6, size=(500, 3))
            rays.append(np.concatenate([o, target - o, np.zeros((500, 1))], axis=1))
        o = rng.uniform(-0.9, 0.9, size=(1000, 3))                  # from inside the cube, any direction
        rays.append(np.concatenate([o, rng.normal(size=(1000, 3)), np.zeros((1000, 1))], axis=1))
        rays = np.concatenate(rays).astype(np.float32).astype(np.float64)
        p64, t64 = sc.first_hit(rays)
        p32, t32 = sc.first_hit_f32(rays)
        assert (p64 >= 0).sum() > 200 * n_spheres and (p64 < 0).sum() > 300
        assert ((p32 >= 0) != (p64 >= 0)).mean() <= 0.004           # grazing rays only
        both = (p32 >= 0) & (p64 >= 0)
        assert np.array_equal(p32[both], p64[both])
        assert np.abs(t32[both] - t64[both]).max() <= 2e-4 * np.maximum(1.0, t64[both]).max()
        # the render takes the same traversal: a 16 x 16 image from behind the origin sees the first sphere
        cam = gpu.Camera(tuple(-np.array(centres[0])), centres[0], (0, 0, 1), 20.0, 1.0, 0.0, 10.0)
        acc, st = sc.render(cam, gpu.Settings(16, 16, 64, 8, 2, seed=3, kernel=2))
        centre_px = acc[7:9, 7:9, :3].mean(axis=(0, 1)) / 64
        sky = acc[0, 0, :3] / 64
        assert st["kernel_used"] == 2 and np.abs(centre_px - sky).max() > 0.05
    finally:
        sc.close()
