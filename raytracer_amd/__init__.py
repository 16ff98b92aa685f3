"""raytracer_amd -- MI355X-native path tracer behind elkiwy/raytraCer's C scene API.

The product is the C-ABI shared library ``librtc.so`` (include/raytracer.h + include/rtc.h); this package is a
thin ctypes mirror of it for tests, bench.py and torch.distributed plumbing.  It never falls back to a CPU path:
if the library is missing, or no gfx950 GPU is usable, calls raise.
"""
from .api import (Camera, RtcError, Scene, Settings, build_named_scene, lib, library_path, png_write, rtc_init,
                  tonemap_u8, SCENES)

__all__ = ["Camera", "RtcError", "Scene", "Settings", "build_named_scene", "lib", "library_path", "png_write",
           "rtc_init", "tonemap_u8", "SCENES"]
