"""ctypes mirror of include/raytracer.h and include/rtc.h (host-side plumbing, not the product).

Names, argument meaning and error behaviour follow the C headers, which in turn follow the reference's
scene-construction API (03_FullTracer/hittable.h:117-129, material.h:39-52, texture.h:56-68, camera.h:29-35).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class RtcError(RuntimeError):
    pass


def library_path():
    # RTC_LIB: another build of the same library (kernel experiments); never a different implementation
    return os.environ.get("RTC_LIB") or os.path.join(_HERE, "librtc.so")


class Vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]


def v3(a):
    return Vec3(float(a[0]), float(a[1]), float(a[2]))


class SceneView(C.Structure):
    _fields_ = [("lookfrom", Vec3), ("lookat", Vec3), ("vup", Vec3), ("vfov", C.c_double), ("aspect", C.c_double),
                ("aperture", C.c_double), ("focus_dist", C.c_double), ("time0", C.c_double), ("time1", C.c_double),
                ("background", Vec3), ("integrator", C.c_int), ("max_depth", C.c_int)]


class RtcSettings(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("max_depth", C.c_int), ("sample_begin", C.c_int),
                ("sample_end", C.c_int), ("integrator", C.c_int), ("seed", C.c_uint64),
                ("mixture_light_weight", C.c_double), ("kernel", C.c_int), ("reserved", C.c_int)]


class SceneSettings(C.Structure):
    """scene_settings of the library facade (reference src/renderer.h:31-36)."""
    _fields_ = [("res_width", C.c_int), ("res_height", C.c_int), ("samples_per_pixel", C.c_int), ("iterations", C.c_int)]


class RtcStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("kernel_ms", C.c_double), ("total_ms", C.c_double),
                ("kernel_launches", C.c_int), ("kernel_used", C.c_int), ("wave_iterations", C.c_uint64)]


SCENES = dict(simple=0, cornell=1, final=2, book1=3, cornell02=4, cornell_smoke=5, two_spheres=6, two_perlin=7,
              simple_light=8, book2_random=9, million=10, earth=11, final_earth=12, box_field=13, box_field_smoke=14)
EARTHMAP = os.path.join(_HERE, "assets", "earthmap.jpg")   # the reference's images/earthmap.jpg, 1024x512: package data, installed to share/raytracer_amd

_P = C.c_void_p
_dp = C.POINTER(C.c_double)

# every symbol include/raytracer.h and include/rtc.h declare (checked by tests/test_abi.py)
HOST_SYMBOLS = """random_double random_double_scaled random_int deg2rad texture_solid_color_init
texture_solid_color_init_rgb texture_checker_init texture_checker_init_c texture_noise_init_scaled texture_noise_init
texture_image_init texture_image_init_rgb textures_free_all material_lambertian_new material_lambertian_new_from_color
material_metal_new material_dielectric_new material_light_new material_light_new_from_color material_isotropic_new
material_isotropic_new_from_color materials_free_all hittable_list_new hittable_list_new_no_gc hittable_list_add
hittable_list_index hittable_list_objects hittable_list_free hittable_lists_free_all hittable_sphere_new
hittable_rect_new hittable_box_new hittable_moving_sphere_new bvh_node_init hittable_translate_init
hittable_rotate_init hittable_constant_medium_init hittable_constant_medium_init_c hittable_flip_face_init
hittable_free camera_new camera_free camera_fields camera_get_ray write_color ray_color material_free texture_free rt_scene_build rt_scene_set_image_path rt_write_png rt_write_ppm_p3""".split()
RTC_SYMBOLS = """rtc_init rtc_last_error rtc_device_name rtc_device_count rtc_scene_build rtc_scene_flatten_only rtc_scene_save rtc_scene_load rtc_scene_load_host rtc_scene_free rtc_scene_num_prims
rtc_scene_dump_leaves rtc_scene_bvh_info rtc_scene_small_info rtc_scene_extend_info rtc_scene_dump_bvh rtc_first_hit_f64 rtc_first_hit_f32 rtc_primary_first_hit_f64 rtc_render_device rtc_render_device_noalloc rtc_render rtc_render_stats
rtc_tonemap_u8_device rtc_tonemap_u8 rtc_ray_colors rtc_kat rtc_scene_clone_to_device rtc_render_multi make_scene free_scene render_scene render_scene_accumulate""".split()


def lib():
    """Load librtc.so once and declare signatures.  Raises if the HIP extension was not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # A process that also uses torch must end up with ONE HIP runtime: torch bundles its own libamdhip64, librtc.so
    # links the system one, and whichever is loaded second finds "no HIP GPUs".  Loading torch first makes librtc.so
    # resolve against torch's copy (same SONAME).  A C host without torch is not affected.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise RtcError(f"{path} is missing: build it with `make -C raytracer_amd/csrc` (no CPU fallback exists)")
    L = C.CDLL(path)
    for name in ("texture_solid_color_init", "texture_checker_init", "texture_noise_init_scaled", "texture_image_init_rgb",
                 "material_lambertian_new", "material_lambertian_new_from_color", "material_metal_new",
                 "material_dielectric_new", "material_light_new", "material_light_new_from_color",
                 "material_isotropic_new", "hittable_list_new", "hittable_list_new_no_gc", "hittable_sphere_new",
                 "hittable_rect_new", "hittable_box_new", "hittable_moving_sphere_new", "bvh_node_init",
                 "hittable_translate_init", "hittable_rotate_init", "hittable_constant_medium_init_c",
                 "hittable_flip_face_init", "camera_new", "rt_scene_build", "rtc_scene_build", "rtc_scene_flatten_only",
                 "rtc_scene_load", "rtc_scene_load_host"):
        getattr(L, name).restype = _P
    L.texture_solid_color_init.argtypes = [Vec3]
    L.texture_checker_init.argtypes = [_P, _P]
    L.texture_noise_init_scaled.argtypes = [C.c_double]
    L.texture_image_init_rgb.argtypes = [_P, C.c_int, C.c_int]
    L.material_lambertian_new.argtypes = [_P]
    L.material_lambertian_new_from_color.argtypes = [Vec3]
    L.material_metal_new.argtypes = [Vec3, C.c_double]
    L.material_dielectric_new.argtypes = [C.c_double]
    L.material_light_new.argtypes = [_P]
    L.material_light_new_from_color.argtypes = [Vec3]
    L.material_isotropic_new.argtypes = [_P]
    L.hittable_list_new.argtypes = [C.c_int]
    L.hittable_list_new_no_gc.argtypes = [C.c_int]
    L.hittable_list_add.argtypes = [_P, _P]
    L.hittable_list_index.argtypes = [_P]
    L.hittable_sphere_new.argtypes = [_P, Vec3, C.c_double, _P]
    L.hittable_rect_new.argtypes = [_P] + [C.c_double] * 7 + [C.c_int, _P]
    L.hittable_box_new.argtypes = [_P, Vec3, Vec3, _P]
    L.hittable_moving_sphere_new.argtypes = [_P, Vec3, Vec3, C.c_double, C.c_double, C.c_double, _P]
    L.bvh_node_init.argtypes = [_P, _P, C.c_double, C.c_double]
    L.hittable_translate_init.argtypes = [_P, _P, Vec3]
    L.hittable_rotate_init.argtypes = [_P, _P, C.c_double, C.c_int]
    L.hittable_constant_medium_init_c.argtypes = [_P, _P, C.c_double, Vec3]
    L.hittable_flip_face_init.argtypes = [_P, _P]
    L.camera_new.argtypes = [Vec3, Vec3, Vec3] + [C.c_double] * 6
    L.camera_free.argtypes = [_P]
    L.camera_fields.argtypes = [_P, _dp]
    L.rt_scene_build.argtypes = [C.c_int, C.c_int, _P, C.POINTER(SceneView)]
    L.rt_scene_set_image_path.argtypes = [C.c_char_p]
    L.rt_scene_set_image_path(EARTHMAP.encode())
    L.rt_write_png.argtypes = [C.c_char_p, C.c_int, C.c_int, _P, C.c_int]
    L.rt_write_ppm_p3.argtypes = [C.c_char_p, C.c_int, C.c_int, _P, C.c_int]
    L.rtc_last_error.restype = C.c_char_p
    L.rtc_device_name.restype = C.c_char_p
    L.rtc_scene_build.argtypes = [_P, _P, C.POINTER(Vec3), C.c_double, C.c_double]
    L.rtc_scene_flatten_only.argtypes = [_P, _P, C.POINTER(Vec3), C.c_double, C.c_double]
    L.rtc_scene_save.argtypes = [_P, C.c_char_p]
    L.rtc_scene_load.argtypes = [C.c_char_p]
    L.rtc_scene_load_host.argtypes = [C.c_char_p]
    L.rtc_scene_free.argtypes = [_P]
    L.rtc_scene_num_prims.argtypes = [_P]
    L.rtc_scene_dump_leaves.argtypes = [_P, _dp, C.c_int]
    L.rtc_scene_bvh_info.argtypes = [_P, C.POINTER(C.c_longlong)]
    L.rtc_scene_small_info.argtypes = [_P, C.POINTER(C.c_int)]
    L.rtc_scene_extend_info.argtypes = [_P, C.POINTER(C.c_int)]
    L.rtc_scene_dump_bvh.argtypes = [_P, _P, C.c_int, _P, _P, C.c_int]
    L.rtc_first_hit_f64.argtypes = [_P, _dp, C.c_size_t, C.c_double, C.c_double, _P, _dp, _dp]
    L.rtc_first_hit_f32.argtypes = [_P, _dp, C.c_size_t, C.c_double, C.c_double, _P, _P]
    L.rtc_primary_first_hit_f64.argtypes = [_P, _P, _dp, _dp, C.c_size_t, C.c_double, C.c_double, _P, _dp, _dp]
    L.rtc_render_device.argtypes = [_P, _P, C.POINTER(RtcSettings), _P, _P]
    L.rtc_render.argtypes = [_P, _P, C.POINTER(RtcSettings), _P]
    L.rtc_render_device_noalloc.argtypes = [_P, _P, C.POINTER(RtcSettings), _P, _P]
    L.rtc_render_stats.argtypes = [_P, _P, C.POINTER(RtcStats)]
    L.rtc_tonemap_u8_device.argtypes = [_P, C.c_int, C.c_int, _P, _P]
    L.rtc_tonemap_u8.argtypes = [_P, C.c_int, C.c_int, _P]
    if hasattr(L, "rtc_render_multi"):
        L.rtc_scene_clone_to_device.restype = _P
        L.rtc_scene_clone_to_device.argtypes = [_P, C.c_int]
        L.rtc_render_multi.argtypes = [C.POINTER(_P), C.c_int, _P, C.POINTER(RtcSettings), _P]
    L.rtc_ray_colors.argtypes = [_P, _dp, C.c_size_t, C.POINTER(RtcSettings), _dp]
    L.rtc_kat.argtypes = [_P, C.c_int, _dp, C.c_size_t, _dp]
    L.write_color.argtypes = [_P, C.POINTER(Vec3), C.c_int, C.c_int]
    L.write_color.restype = None
    L.ray_color.argtypes = [_P, C.POINTER(Vec3), _P, _P, C.c_int]
    L.ray_color.restype = Vec3
    L.material_free.argtypes = [_P]
    L.texture_free.argtypes = [_P]
    L.make_scene.restype = _P
    L.make_scene.argtypes = [_P, _P, C.c_int]
    L.free_scene.argtypes = [_P]
    L.free_scene.restype = None
    L.render_scene.argtypes = [_P, _P, C.POINTER(SceneSettings), C.c_char_p]
    L.render_scene.restype = None
    L.render_scene_accumulate.argtypes = [_P, _P, C.POINTER(SceneSettings), C.c_int, C.c_int, _P]
    L.srand = C.CDLL(None).srand
    _LIB = L
    return L


def _check(rc, what):
    if rc != 0:
        raise RtcError(f"{what}: {lib().rtc_last_error().decode()} (code {rc})")


def rtc_init():
    _check(lib().rtc_init(), "rtc_init")
    return lib().rtc_device_name().decode()


class Settings:
    """Mirror of rtc_settings; defaults are the reference's (03 main.c:88-92, ray.c:72)."""

    def __init__(self, width=1024, height=1024, spp=256, max_depth=10, integrator=0, seed=0, sample_begin=0,
                 sample_end=None, mixture_light_weight=0.1, kernel=0):
        self.c = RtcSettings(width, height, max_depth, sample_begin, spp if sample_end is None else sample_end,
                             integrator, seed, mixture_light_weight, kernel, 0)


class Camera:
    def __init__(self, lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist, time0=0.0, time1=1.0):
        self.handle = lib().camera_new(v3(lookfrom), v3(lookat), v3(vup), vfov, aspect, aperture, focus_dist, time0, time1)

    @classmethod
    def from_view(cls, view, aspect=None):
        g = lambda v: (v.x, v.y, v.z)
        return cls(g(view.lookfrom), g(view.lookat), g(view.vup), view.vfov, aspect or view.aspect, view.aperture,
                   view.focus_dist, view.time0, view.time1)

    def fields(self):
        out = np.zeros(24)
        lib().camera_fields(self.handle, out.ctypes.data_as(_dp))
        return out


class Scene:
    """A flattened scene resident on the current HIP device (rtc_scene)."""

    def __init__(self, world, lights, background=(0, 0, 0), time0=0.0, time1=1.0, view=None, host_only=False):
        bg = v3(background)
        self.world, self.lights, self.view = world, lights, view
        build = lib().rtc_scene_flatten_only if host_only else lib().rtc_scene_build
        self.handle = build(world, lights, C.byref(bg), time0, time1)
        if not self.handle:
            raise RtcError("rtc_scene_build: " + lib().rtc_last_error().decode())

    @classmethod
    def load(cls, path, view=None, host_only=False):
        """rtc_scene_load / rtc_scene_load_host: a scene saved with save(); the file is verified before anything is uploaded."""
        self = cls.__new__(cls)
        self.world, self.lights, self.view = None, None, view
        self.handle = (lib().rtc_scene_load_host if host_only else lib().rtc_scene_load)(os.fsencode(path))
        if not self.handle:
            raise RtcError(lib().rtc_last_error().decode())
        return self

    def save(self, path):
        _check(lib().rtc_scene_save(self.handle, os.fsencode(path)), "rtc_scene_save")

    def close(self):
        if self.handle:
            lib().rtc_scene_free(self.handle)
            self.handle = None

    def num_prims(self):
        return lib().rtc_scene_num_prims(self.handle)

    def dump_leaves(self):
        n = self.num_prims()
        out = np.zeros((max(n, 1), 16))
        lib().rtc_scene_dump_leaves(self.handle, out.ctypes.data_as(_dp), n)
        return out[:n]

    def bvh_info(self):
        out = (C.c_longlong * 4)()
        _check(lib().rtc_scene_bvh_info(self.handle, out), "rtc_scene_bvh_info")
        return dict(nodes=out[0], leaf_slots=out[1], depth=out[2], bytes=out[3])

    def small_info(self):
        out = (C.c_int * 8)()
        _check(lib().rtc_scene_small_info(self.handle, out), "rtc_scene_small_info")
        return dict(zip(("eligible", "pairs", "spaces", "boxes", "spheres", "fused", "fused_rects", "fused_spheres"), out))

    def extend_info(self):
        out = (C.c_int * 4)()
        _check(lib().rtc_scene_extend_info(self.handle, out), "rtc_scene_extend_info")
        return dict(zip(("lds_tables", "lds_bytes", "nodes", "leaf_records"), out))

    def dump_bvh(self):
        """(nodes uint32[n, 16], leaves uint32[m, 16], prim_of_slot int32[m]); view the float fields with .view(np.float32)."""
        info = self.bvh_info()
        nodes = np.zeros((max(info["nodes"], 1), 16), dtype=np.uint32)
        leaves = np.zeros((max(info["leaf_slots"], 1), 16), dtype=np.uint32)
        prims = np.zeros(max(info["leaf_slots"], 1), dtype=np.int32)
        n = lib().rtc_scene_dump_bvh(self.handle, nodes.ctypes.data, info["nodes"], leaves.ctypes.data, prims.ctypes.data, info["leaf_slots"])
        if n < 0:
            raise RtcError("rtc_scene_dump_bvh: " + lib().rtc_last_error().decode())
        return nodes[:info["nodes"]], leaves[:n], prims[:n]

    def first_hit(self, rays7, tmin=0.001, tmax=np.inf, want_rec=False):
        rays7 = np.ascontiguousarray(rays7, dtype=np.float64)
        n = rays7.size // 7
        prim = np.zeros(n, dtype=np.int32)
        t = np.zeros(n)
        rec = np.zeros((n, 12)) if want_rec else None
        _check(lib().rtc_first_hit_f64(self.handle, rays7.ctypes.data_as(_dp), n, tmin, tmax, prim.ctypes.data, t.ctypes.data_as(_dp),
                                       rec.ctypes.data_as(_dp) if want_rec else None), "rtc_first_hit_f64")
        return (prim, t, rec) if want_rec else (prim, t)

    def first_hit_f32(self, rays7, tmin=0.001, tmax=np.inf):
        """The production FP32 traversal alone: (prim int32, t float32)."""
        rays7 = np.ascontiguousarray(rays7, dtype=np.float64)
        n = rays7.size // 7
        prim = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float32)
        _check(lib().rtc_first_hit_f32(self.handle, rays7.ctypes.data_as(_dp), n, tmin, tmax, prim.ctypes.data, t.ctypes.data), "rtc_first_hit_f32")
        return prim, t

    def primary_first_hit(self, cam, st2, lens3=None, tmin=0.001, tmax=np.inf):
        st2 = np.ascontiguousarray(st2, dtype=np.float64)
        n = st2.size // 2
        prim = np.zeros(n, dtype=np.int32)
        t = np.zeros(n)
        rays = np.zeros((n, 7))
        lp = None
        if lens3 is not None:
            lens3 = np.ascontiguousarray(lens3, dtype=np.float64)
            lp = lens3.ctypes.data_as(_dp)
        _check(lib().rtc_primary_first_hit_f64(self.handle, cam.handle, st2.ctypes.data_as(_dp), lp, n, tmin, tmax, prim.ctypes.data,
                                               t.ctypes.data_as(_dp), rays.ctypes.data_as(_dp)), "rtc_primary_first_hit_f64")
        return prim, t, rays

    def render(self, cam, settings, accum=None):
        """Host-buffer render: returns (accum[h, w, 4] float32 with row 0 = bottom, stats dict)."""
        s = settings.c
        if accum is None:
            accum = np.zeros((s.height, s.width, 4), dtype=np.float32)
        _check(lib().rtc_render(self.handle, cam.handle, C.byref(s), accum.ctypes.data), "rtc_render")
        return accum, self.stats(None)

    def ray_colors(self, rays7, settings, rgb_sum=None):
        """rtc_ray_colors: the estimator of each supplied ray summed over the settings' sample range -> (n, 3) float64."""
        rays7 = np.ascontiguousarray(rays7, dtype=np.float64)
        n = rays7.size // 7
        if rgb_sum is None:
            rgb_sum = np.zeros((n, 3))
        _check(lib().rtc_ray_colors(self.handle, rays7.ctypes.data_as(_dp), n, C.byref(settings.c), rgb_sum.ctypes.data_as(_dp)), "rtc_ray_colors")
        return rgb_sum

    def kat(self, what, inputs):
        """rtc_kat: device functions on supplied inputs (test-only); `what` as in include/rtc.h."""
        stride_in = (5, 6, 6, 6, 2, 7, 6, 6, 9, 9)[what]
        stride_out = (4, 1, 1, 1, 1, 3, 3, 3, 3, 3)[what]
        inputs = np.ascontiguousarray(inputs, dtype=np.float64)
        n = inputs.size // stride_in
        out = np.zeros((n, stride_out))
        _check(lib().rtc_kat(self.handle, what, inputs.ctypes.data_as(_dp), n, out.ctypes.data_as(_dp)), "rtc_kat")
        return out[:, 0] if stride_out == 1 else out

    def render_device(self, cam, settings, accum_ptr, stream=None):
        _check(lib().rtc_render_device(self.handle, cam.handle, C.byref(settings.c), accum_ptr, stream), "rtc_render_device")

    def stats(self, stream=None):
        st = RtcStats()
        _check(lib().rtc_render_stats(self.handle, stream, C.byref(st)), "rtc_render_stats")
        return dict(samples=st.samples, rays=st.rays, kernel_ms=st.kernel_ms, total_ms=st.total_ms,
                    kernel_launches=st.kernel_launches, kernel_used=st.kernel_used, wave_iterations=st.wave_iterations,
                    lane_fill=(st.rays / (64.0 * st.wave_iterations) if st.wave_iterations else 0.0))


def build_named_scene(name, arg=0, aspect=None, seed=1, host_only=False):
    """srand(seed); rt_scene_build(...) -> (Scene, Camera, SceneView).  Scene ids as in raytracer.h."""
    L = lib()
    L.srand(C.c_uint(seed))
    lights = L.hittable_list_new_no_gc(64)
    view = SceneView()
    world = L.rt_scene_build(SCENES[name] if isinstance(name, str) else name, arg, lights, C.byref(view))
    if not world:
        raise RtcError(f"unknown scene {name}")
    cam = Camera.from_view(view, aspect)
    scene = Scene(world, lights, (view.background.x, view.background.y, view.background.z), view.time0, view.time1, view,
                  host_only=host_only)
    return scene, cam, view


def tonemap_u8(accum):
    accum = np.ascontiguousarray(accum, dtype=np.float32)
    h, w = accum.shape[:2]
    out = np.zeros((h, w, 3), dtype=np.uint8)
    _check(lib().rtc_tonemap_u8(accum.ctypes.data, w, h, out.ctypes.data), "rtc_tonemap_u8")
    return out


def png_write(path, rgb_u8, flip=True):
    rgb_u8 = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
    h, w = rgb_u8.shape[:2]
    if lib().rt_write_png(path.encode(), w, h, rgb_u8.ctypes.data, int(flip)) != 1:
        raise RtcError(f"rt_write_png({path}) failed")
