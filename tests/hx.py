"""ctypes binding of the probe harness (oracle/harness.c) -- TEST INFRASTRUCTURE.

The same flat ``hx_*`` ABI is exported by three libraries:
  * ``oracle/_build/liboracle.so``  the CPU restatement (oracle/cpu_ref.c)
  * ``oracle/_ref/libref03.so``     the reference's 03_FullTracer compiled in place
  * ``oracle/_ref/libref02.so``     the reference's 02_AdvancedTracer compiled in place
``oracle/_ref/libref01.so`` (the reference's 01_BasicTracer, one header-only translation unit with an API of its own)
exports the ``hx01_*`` probes of oracle/harness01.c instead: class ``Ref01``.
Only tests/, smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
REF03_SO = os.path.join(ROOT, "oracle", "_ref", "libref03.so")
REF02_SO = os.path.join(ROOT, "oracle", "_ref", "libref02.so")
REF01_SO = os.path.join(ROOT, "oracle", "_ref", "libref01.so")
EARTHMAP = os.path.join(ROOT, "tests", "golden", "earthmap.jpg")   # image asset of the earth scenes: the reference's own images/earthmap.jpg (1024x512)


def earthmap_for(is_oracle):
    """The compiled reference decodes the JPEG itself (stb_image).  The oracle reads binary PPM only (oracle/cpu_ref.c:190), so it
    gets the reference's full decode of that file (tests/golden/earthmap_decode.npz, make_jpeg_cases.py) written out as a PPM."""
    if not is_oracle:
        return EARTHMAP
    import tempfile
    path = os.path.join(tempfile.gettempdir(), f"rtc_earthmap_1024_{os.getuid()}.ppm")
    if not os.path.exists(path):
        rgb = np.load(os.path.join(ROOT, "tests", "golden", "earthmap_decode.npz"))["rgb"]
        tmp = path + f".{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]))
            f.write(np.ascontiguousarray(rgb).tobytes())
        os.replace(tmp, path)
    return path

SCENE = dict(simple=0, cornell=1, final=2, book1=3, cornell02=4, cornell_smoke=5, two_spheres=6,
             two_perlin=7, simple_light=8, book2_random=9, million=10, earth=11, final_earth=12, box_field=13, box_field_smoke=14,
             ref_simple=-1, ref_cornell=-2, ref_final=-8, ref_cornell02=-6)
MAT = dict(lambertian=0, metal=1, dielectric=2, light=3, isotropic=4)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_bp = C.POINTER(C.c_ubyte)


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


class Harness:
    """One loaded backend.  Scene state is global inside the library (one scene at a time)."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self.lib = C.CDLL(path)
        self.lib.hx_stage.restype = C.c_int
        self.is_oracle = hasattr(self.lib, "hx_oracle_render")
        self.lib.hx_set_image_path(earthmap_for(self.is_oracle).encode())

    # -- scene -----------------------------------------------------------------------------
    def stage(self):
        return self.lib.hx_stage()

    def set_stage(self, s):
        return self.lib.hx_set_stage(C.c_int(s))

    def srand(self, seed):
        self.lib.hx_srand(C.c_uint(seed))

    def build(self, scene, arg=0, aspect=0.0, seed=1):
        """srand(seed) then build: rand()-built scenes come out identical in every backend."""
        self.srand(seed)
        sid = SCENE[scene] if isinstance(scene, str) else scene
        rc = self.lib.hx_build_scene(C.c_int(sid), C.c_int(arg), C.c_double(aspect))
        if rc != 0:
            raise RuntimeError(f"scene {scene} not available in {self.path}")
        return self

    def view(self):
        out = np.zeros(20)
        self.lib.hx_get_view(out.ctypes.data_as(_dp))
        return out

    def camera(self):
        out = np.zeros(24)
        self.lib.hx_get_camera(out.ctypes.data_as(_dp))
        return out

    def num_leaves(self):
        return self.lib.hx_num_leaves()

    def dump_leaves(self, cap=1 << 21):
        out = np.zeros((cap, 16))
        n = self.lib.hx_dump_leaves(out.ctypes.data_as(_dp), C.c_int(cap))
        return out[:n].copy()

    # -- rays / hits -------------------------------------------------------------------------
    def camera_rays(self, uv, seed=1):
        uv, p = _d(uv)
        n = uv.size // 2
        out = np.zeros((n, 7))
        self.srand(seed)
        self.lib.hx_camera_rays(C.c_int(n), p, out.ctypes.data_as(_dp))
        return out

    def hit_rays(self, rays, tmin=0.001, tmax=np.inf, want_prim=True):
        rays, p = _d(rays)
        n = rays.size // 7
        rec = np.zeros((n, 12))
        prim = np.zeros(n, dtype=np.int32)
        self.lib.hx_hit_rays(C.c_int(n), p, C.c_double(tmin), C.c_double(tmax), rec.ctypes.data_as(_dp),
                             prim.ctypes.data_as(_ip) if want_prim else None)
        return rec, prim

    def ray_colors(self, rays, depth, seed=1, iterative=False):
        rays, p = _d(rays)
        n = rays.size // 7
        out = np.zeros((n, 3))
        fn = self.lib.hx_ray_colors_iter if iterative else self.lib.hx_ray_colors
        fn(C.c_uint(seed), C.c_int(n), p, C.c_int(depth), out.ctypes.data_as(_dp))
        return out

    def render_refloop(self, w, h, spp, depth, seed=1):
        px = np.zeros((h, w, 3), dtype=np.uint8)
        self.srand(seed)
        self.lib.hx_render_refloop(C.c_int(w), C.c_int(h), C.c_int(spp), C.c_int(depth), px.ctypes.data_as(_bp))
        return px

    def render_refloop_omp(self, w, h, spp, depth, threads=16, seed=1):
        """The reference's loop as shipped: 16 row chunks under `omp parallel for`, one shared rand() (timing row only)."""
        px = np.zeros((h, w, 3), dtype=np.uint8)
        self.srand(seed)
        self.lib.hx_render_refloop_omp(C.c_int(w), C.c_int(h), C.c_int(spp), C.c_int(depth), C.c_int(threads), px.ctypes.data_as(_bp))
        return px

    # -- unit probes ---------------------------------------------------------------------------
    def _hit_probe(self, fn, params, rays, tmin, tmax):
        params, pp = _d(params)
        rays, pr = _d(rays)
        n = rays.size // 7
        rec = np.zeros((n, 12))
        fn(pp, C.c_int(n), pr, C.c_double(tmin), C.c_double(tmax), rec.ctypes.data_as(_dp))
        return rec

    def rect_hit(self, rect8, rays, tmin=0.001, tmax=np.inf):
        return self._hit_probe(self.lib.hx_unit_rect_hit, rect8, rays, tmin, tmax)

    def sphere_hit(self, sph4, rays, tmin=0.001, tmax=np.inf):
        return self._hit_probe(self.lib.hx_unit_sphere_hit, sph4, rays, tmin, tmax)

    def moving_sphere_hit(self, sph9, rays, tmin=0.001, tmax=np.inf):
        return self._hit_probe(self.lib.hx_unit_moving_sphere_hit, sph9, rays, tmin, tmax)

    def box_hit(self, box11, rays, tmin=0.001, tmax=np.inf):
        return self._hit_probe(self.lib.hx_unit_box_hit, box11, rays, tmin, tmax)

    def aabb_hit(self, box6, rays, tmin=0.001, tmax=np.inf):
        box6, pb = _d(box6)
        rays, pr = _d(rays)
        n = rays.size // 7
        out = np.zeros(n, dtype=np.int32)
        self.lib.hx_unit_aabb_hit(pb, C.c_int(n), pr, C.c_double(tmin), C.c_double(tmax), out.ctypes.data_as(_ip))
        return out

    def _vec_probe(self, fn, arr, stride_in, stride_out):
        arr, p = _d(arr)
        n = arr.size // stride_in
        out = np.zeros((n, stride_out)) if stride_out > 1 else np.zeros(n)
        fn(C.c_int(n), p, out.ctypes.data_as(_dp))
        return out

    def refract(self, in7):
        return self._vec_probe(self.lib.hx_unit_refract, in7, 7, 3)

    def reflectance(self, in2):
        return self._vec_probe(self.lib.hx_unit_reflectance, in2, 2, 1)

    def onb(self, n3):
        return self._vec_probe(self.lib.hx_unit_onb, n3, 3, 9)

    def cosine_pdf(self, in6):
        return self._vec_probe(self.lib.hx_unit_cosine_pdf, in6, 6, 1)

    def light_pdf(self, in6):
        return self._vec_probe(self.lib.hx_unit_light_pdf, in6, 6, 1)

    def scattering_pdf(self, in6):
        return self._vec_probe(self.lib.hx_unit_scattering_pdf, in6, 6, 1)

    def light_random(self, orig3, seed=1):
        orig3, p = _d(orig3)
        n = orig3.size // 3
        out = np.zeros((n, 3))
        self.lib.hx_unit_light_random(C.c_uint(seed), C.c_int(n), p, out.ctypes.data_as(_dp))
        return out

    def scatter(self, mtype, matp, in14, seed=1):
        matp, pm = _d(matp)
        in14, pi = _d(in14)
        n = in14.size // 14
        out = np.zeros((n, 12))
        mt = MAT[mtype] if isinstance(mtype, str) else mtype
        self.lib.hx_unit_scatter(C.c_int(mt), pm, C.c_uint(seed), C.c_int(n), pi, out.ctypes.data_as(_dp))
        return out

    def sampler(self, kind, n, seed=1):
        out = np.zeros((n, 3))
        rc = self.lib.hx_unit_sampler(C.c_int(kind), C.c_uint(seed), C.c_int(n), out.ctypes.data_as(_dp))
        if rc != 0:
            raise RuntimeError("sampler kind not available")
        return out

    def perlin(self, p3, seed=1):
        p3, pp = _d(p3)
        n = p3.size // 3
        noise, turb = np.zeros(n), np.zeros(n)
        ranvec = np.zeros((256, 3))
        perm = np.zeros((3, 256), dtype=np.int32)
        self.lib.hx_unit_perlin(C.c_uint(seed), C.c_int(n), pp, noise.ctypes.data_as(_dp), turb.ctypes.data_as(_dp),
                                ranvec.ctypes.data_as(_dp), perm.ctypes.data_as(_ip))
        return noise, turb, ranvec, perm

    def texture(self, kind, c0, c1, in5, seed=1):
        c0, p0 = _d(c0)
        c1, p1 = _d(c1)
        in5, pi = _d(in5)
        n = in5.size // 5
        out = np.zeros((n, 3))
        self.lib.hx_unit_texture(C.c_int(kind), p0, p1, C.c_uint(seed), C.c_int(n), pi, out.ctypes.data_as(_dp))
        return out

    def write_color(self, rgb3, spp):
        rgb3, p = _d(rgb3)
        n = rgb3.size // 3
        out = np.zeros((n, 3), dtype=np.uint8)
        self.lib.hx_unit_write_color(C.c_int(n), p, C.c_int(spp), out.ctypes.data_as(_bp))
        return out

    def render_ppm01(self, path, w, h, spp, depth, seed=None):
        """Oracle only, stage 1: the serial loop of 01 main.c:168-191 into an ASCII PPM; seed None continues the rand() stream."""
        assert self.is_oracle
        return self.lib.hx_render_ppm01(path.encode(), C.c_int(w), C.c_int(h), C.c_int(spp), C.c_int(depth),
                                        C.c_int(0 if seed is None else 1), C.c_uint(seed or 0))

    def write_colors_ppm(self, path, rgb3, spp):
        assert self.is_oracle
        rgb3, p = _d(rgb3)
        return self.lib.hx_write_colors_ppm(path.encode(), C.c_int(rgb3.size // 3), p, C.c_int(spp))

    # -- oracle only ---------------------------------------------------------------------------
    def oracle_render(self, w, h, s_begin, s_end, depth=0, rng="philox", seed=0, threads=0, iterative=True,
                      scrub_nan=True):
        """Sum of radiance over samples [s_begin, s_end): (h, w, 3) float64, row 0 = bottom; plus stats."""
        assert self.is_oracle
        out = np.zeros((h, w, 3))
        stats = np.zeros(3)
        self.lib.hx_oracle_render(C.c_int(w), C.c_int(h), C.c_int(s_begin), C.c_int(s_end), C.c_int(depth),
                                  C.c_int(1 if rng == "philox" else 0), C.c_ulonglong(seed), C.c_int(threads),
                                  C.c_int(int(iterative)), C.c_int(int(scrub_nan)), out.ctypes.data_as(_dp),
                                  stats.ctypes.data_as(_dp))
        return out, dict(samples=stats[0], rays=stats[1], seconds=stats[2])


COUNTER_NAMES = """rays miss emit diffuse specular_metal specular_dielectric isotropic rect_tests rect_hits sphere_tests
sphere_hits msphere_tests aabb_tests medium_tests translate_calls rotate_calls light_pdf_rect light_pdf_sphere
light_sample_rect light_sample_sphere cosine_sample onb_builds camera_rays tex_noise tex_checker tex_image
translate_hits rotate_hits light_pdf_rect_hits light_pdf_sphere_hits samples""".split()


def counters_reset(h):
    h.lib.hx_counters(C.c_int(1), None, C.c_int(0))


def counters_get(h):
    out = np.zeros(len(COUNTER_NAMES), dtype=np.uint64)
    n = h.lib.hx_counters(C.c_int(0), out.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.c_int(len(out)))
    assert n == len(COUNTER_NAMES)
    return dict(zip(COUNTER_NAMES, (int(x) for x in out)))


def oracle():
    return Harness(ORACLE_SO)


def ref03():
    return Harness(REF03_SO)


def ref02():
    return Harness(REF02_SO)


class Ref01:
    """The compiled 01_BasicTracer (oracle/harness01.c).  Rays are 6 doubles (no time), records 9 (no u,v)."""

    def __init__(self, path=REF01_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)

    def build(self, seed=1, aspect=0.0):
        self.n = self.lib.hx01_build(C.c_uint(seed), C.c_double(aspect))
        return self

    def srand(self, seed):
        self.lib.hx01_srand(C.c_uint(seed))

    def camera(self):
        out = np.zeros(22)
        self.lib.hx01_get_camera(out.ctypes.data_as(_dp))
        return out

    def spheres(self):
        out = np.zeros((self.n, 10))
        self.lib.hx01_dump_spheres(out.ctypes.data_as(_dp), C.c_int(self.n))
        return out

    def camera_rays(self, uv, seed=1):
        uv, p = _d(uv)
        n = uv.size // 2
        out = np.zeros((n, 6))
        self.srand(seed)
        self.lib.hx01_camera_rays(C.c_int(n), p, out.ctypes.data_as(_dp))
        return out

    def hit_rays(self, rays6, tmin=0.001, tmax=np.inf):
        rays6, p = _d(rays6)
        n = rays6.size // 6
        rec, mat, prim = np.zeros((n, 9)), np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        self.lib.hx01_hit_rays(C.c_int(n), p, C.c_double(tmin), C.c_double(tmax), rec.ctypes.data_as(_dp), mat.ctypes.data_as(_ip), prim.ctypes.data_as(_ip))
        return rec, mat, prim

    def ray_colors(self, rays6, depth, seed=1):
        rays6, p = _d(rays6)
        n = rays6.size // 6
        out = np.zeros((n, 3))
        self.lib.hx01_ray_colors(C.c_uint(seed), C.c_int(n), p, C.c_int(depth), out.ctypes.data_as(_dp))
        return out

    def scatter(self, mtype, matp, in13, seed=1):
        matp, pm = _d(matp)
        in13, pi = _d(in13)
        n = in13.size // 13
        out = np.zeros((n, 10))
        self.lib.hx01_scatter(C.c_int(MAT[mtype]), pm, C.c_uint(seed), C.c_int(n), pi, out.ctypes.data_as(_dp))
        return out

    def render_ppm(self, path, w, h, spp, depth, seed=None):
        return self.lib.hx01_render_ppm(path.encode(), C.c_int(w), C.c_int(h), C.c_int(spp), C.c_int(depth),
                                        C.c_int(0 if seed is None else 1), C.c_uint(seed or 0))

    def write_colors(self, path, rgb3, spp):
        rgb3, p = _d(rgb3)
        return self.lib.hx01_write_colors(path.encode(), C.c_int(rgb3.size // 3), p, C.c_int(spp))

    def run_main(self, path):
        """The reference's real main(): 1920x1080 x 8 spp x depth 8, about two minutes, prints its progress."""
        return self.lib.hx01_run_main(path.encode())


def ref01():
    return Ref01()
