"""Scene files (rtc_scene_save / rtc_scene_load, csrc/host/scene_io.cpp).  The reference has no scene file
(its scenes are C functions: 03 main.c:28-78, 02 main.c:76-289), so there is nothing of the reference's to pin here; what is pinned
is the round trip and the refusal of anything that is not a sound scene:

 * CPU: every built-in scene survives save -> load -> save byte for byte, with the same leaf table and BVH; damaged,
   truncated, foreign and structurally unsound files (an index out of its table, a BVH cycle) are refused with a message;
 * GPU: a loaded scene renders bit-identically to the scene it was saved from (both small-scene and wavefront kernels) and
   gives the same FP64 first hits.
"""
import os
import struct
import zlib

import numpy as np
import pytest

HOST_SCENES = [("simple", 0), ("cornell", 0), ("final", 0), ("book1", 0), ("cornell02", 0), ("cornell_smoke", 0), ("two_spheres", 0),
               ("two_perlin", 0), ("simple_light", 0), ("book2_random", 0), ("million", 30), ("earth", 0), ("final_earth", 0),
               ("box_field", 0), ("box_field_smoke", 0)]


@pytest.mark.parametrize("name,arg", HOST_SCENES)
def test_roundtrip_on_the_host(rtc, tmp_path, name, arg):
    sc, cam, view = rtc.build_named_scene(name, arg=arg, host_only=True)
    a, b = str(tmp_path / "a.rtcscene"), str(tmp_path / "b.rtcscene")
    try:
        sc.save(a)
        back = rtc.Scene.load(a, host_only=True)
        try:
            back.save(b)
            assert open(a, "rb").read() == open(b, "rb").read()
            assert back.num_prims() == sc.num_prims() and back.bvh_info() == sc.bvh_info()
            assert np.array_equal(back.dump_leaves(), sc.dump_leaves(), equal_nan=True)
            if sc.bvh_info()["nodes"]:
                for x, y in zip(back.dump_bvh(), sc.dump_bvh()):
                    assert np.array_equal(x, y)
        finally:
            back.close()
    finally:
        sc.close()


def _sections(blob):
    """-> {name: (offset of the section's data in the file, element size, count)}"""
    out, p = {}, 64
    while p < len(blob):
        name = blob[p:p + 24].split(b"\0")[0].decode()
        elem, _, count = struct.unpack_from("<IIQ", blob, p + 24)
        out[name] = (p + 40, elem, count)
        p += 40 + elem * count
        p += (-p) % 8
    return out


def _reseal(blob):
    """Recompute the payload CRC of an edited file, so that only the structural checks stand between it and the GPU."""
    blob = bytearray(blob)
    struct.pack_into("<I", blob, 24, zlib.crc32(bytes(blob[64:])) & 0xFFFFFFFF)
    return bytes(blob)


def test_file_layout_is_what_the_header_says(rtc, tmp_path):
    sc, _, _ = rtc.build_named_scene("final", host_only=True)
    path = str(tmp_path / "final.rtcscene")
    try:
        sc.save(path)
        blob = open(path, "rb").read()
        magic, version, n_sections, payload, crc = struct.unpack_from("<8sIIQI", blob, 0)
        assert magic == b"RTCSCENE" and version == 4 and payload == len(blob) - 64 and crc == zlib.crc32(blob[64:]) & 0xFFFFFFFF
        secs = _sections(blob)
        assert len(secs) == n_sections and secs["prims"][2] == sc.num_prims() and secs["nodes"][2] == sc.bvh_info()["nodes"]
        assert secs["nodes"][1] == 64 and secs["bleaves"][1] == 64 and secs["leaf_shade"][1] == 48
    finally:
        sc.close()


def test_unsound_files_are_refused(rtc, tmp_path):
    sc, _, _ = rtc.build_named_scene("final", host_only=True)
    small, _, _ = rtc.build_named_scene("cornell", host_only=True)
    good, path = str(tmp_path / "good.rtcscene"), str(tmp_path / "bad.rtcscene")
    try:
        sc.save(good)
        blob = open(good, "rb").read()
        secs = _sections(blob)

        def refused(data, needle):
            open(path, "wb").write(data)
            with pytest.raises(rtc.RtcError, match=needle):
                rtc.Scene.load(path, host_only=True)

        with pytest.raises(rtc.RtcError, match="cannot open"):
            rtc.Scene.load(str(tmp_path / "missing.rtcscene"), host_only=True)
        refused(b"P6\n2 2\n255\n" + bytes(64), "not a scene file")
        refused(blob[:len(blob) // 2], "truncated")
        refused(blob[:20], "not a scene file")
        flipped = bytearray(blob); flipped[len(blob) // 2] ^= 0x40
        refused(bytes(flipped), "checksum")
        v2 = bytearray(blob); struct.pack_into("<I", v2, 8, 2)
        refused(bytes(v2), "version")

        def poke(section, index, word, value, fmt="<i"):
            off, elem, count = secs[section]
            assert index < count
            b = bytearray(blob); struct.pack_into(fmt, b, off + index * elem + 4 * word, value)
            return _reseal(bytes(b))

        refused(poke("prims", 5, 2, 10 ** 6), "primitive 5")                     # material index
        refused(poke("prims", 7, 3, -1), "primitive 7")                          # chain index
        refused(poke("materials", 0, 1, 10 ** 6), "material")                    # texture index
        refused(poke("leaf_prims", 3, 0, 10 ** 7), "leaf slot")
        refused(poke("nodes", 4, 3, 2), "pre-order")                             # child0 of node 4 -> node 2: a cycle
        refused(poke("nodes", 4, 3, 10 ** 7), "pre-order")
        refused(poke("nodes", 1, 3, ~((10 ** 6 << 3) | 1)), "leaf run")          # a leaf run far behind the table
        refused(poke("chains", 1, 0, 10 ** 6), "chain")
        refused(poke("ref_steps", 0, 2, 0) if struct.unpack_from("<i", blob, secs["ref_steps"][0])[0] == 1 else poke("ref_steps", 0, 2, 10 ** 7), "reference")
        refused(poke("scalars", 0, 0, 3), "deeper")                              # recorded BVH depth below the real one
        swapped = bytearray(blob); off = secs["nodes"][0] - 40; swapped[off:off + 5] = b"nodez"
        refused(_reseal(bytes(swapped)), "expected section nodes")
        resized = bytearray(blob); struct.pack_into("<I", resized, secs["nodes"][0] - 16, 32)
        refused(_reseal(bytes(resized)), "another table layout")

        small.save(good)
        blob = open(good, "rb").read(); secs = _sections(blob)
        refused(poke("small_groups", 0, 1, 10 ** 6), "small-scene group")
        refused(poke("small_pairs", 0, 10, 10 ** 6), "small-scene rect id")
        refused(poke("small_boxes", 0, 6, 0x00FFFF40, "<I"), "box face id")
        assert not os.path.exists(str(tmp_path / "missing.rtcscene"))
    finally:
        sc.close(); small.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,arg,W,H,spp", [("cornell", 0, 96, 64, 32), ("final", 0, 64, 64, 8), ("box_field_smoke", 0, 48, 48, 8), ("million", 60, 64, 48, 4)])
def test_loaded_scene_renders_bit_identically(gpu, tmp_path, name, arg, W, H, spp):
    sc, cam, view = gpu.build_named_scene(name, arg=arg)
    path = str(tmp_path / f"{name}.rtcscene")
    try:
        sc.save(path)
        back = gpu.Scene.load(path)
        try:
            st = gpu.Settings(W, H, spp, view.max_depth, view.integrator, seed=9)
            a, sa = sc.render(cam, st)
            b, sb = back.render(cam, st)
            assert sa["kernel_used"] == sb["kernel_used"] and sa["rays"] == sb["rays"] and np.array_equal(a, b)
            rng = np.random.default_rng(3)
            rays = np.zeros((4096, 7)); rays[:, :3] = np.asarray(cam.fields()[:3]); rays[:, 3:6] = rng.normal(size=(4096, 3)); rays[:, 6] = rng.random(4096)
            pa, ta = sc.first_hit(rays)
            pb, tb = back.first_hit(rays)
            assert np.array_equal(pa, pb) and np.array_equal(ta, tb)
        finally:
            back.close()
    finally:
        sc.close()


@pytest.mark.gpu
def test_command_line_writes_and_reads_scene_files(gpu, tmp_path):
    """rtc_render -W file / -L file (csrc/host/main.c): the render of the file is the render of the scene, byte for byte."""
    import subprocess
    exe = os.path.join(os.path.dirname(gpu.library_path()), "rtc_render")
    scene, a, b = str(tmp_path / "final.rtcscene"), str(tmp_path / "a.png"), str(tmp_path / "b.png")
    common = [exe, "-S", "2", "-w", "64", "-h", "64", "-s", "4", "-r", "7"]
    subprocess.run(common + ["-o", a, "-W", scene], check=True, capture_output=True, timeout=120)
    out = subprocess.run(common + ["-o", b, "-L", scene], check=True, capture_output=True, timeout=120).stdout.decode()
    assert "scene load" in out and open(a, "rb").read() == open(b, "rb").read()
    bad = subprocess.run(common + ["-o", b, "-L", a], capture_output=True, timeout=120)      # a PNG is not a scene file
    assert bad.returncode != 0 and b"not a scene file" in bad.stderr


def test_loader_survives_mutated_files(rtc, tmp_path):
    """Mutation fuzz on the host, in a child process: bit flips, edge-case integers and non-finite floats written over
    random words of a scene file, CRC resealed.  The loader may accept (a changed coordinate is still a scene) or refuse;
    it must do either without crashing."""
    import subprocess
    import sys
    code = """
import struct, sys, zlib
import numpy as np
sys.path.insert(0, %r)
import raytracer_amd as ra
rng = np.random.default_rng(11)
acc = ref = 0
for name in ("final", "cornell", "box_field_smoke", "two_perlin"):
    sc, _, _ = ra.build_named_scene(name, host_only=True)
    sc.save(%r); sc.close()
    blob = open(%r, "rb").read()
    for it in range(120):
        b = bytearray(blob)
        for _ in range(int(rng.integers(1, 4))):
            off = int(rng.integers(64, len(b) - 4)) & ~3
            mode = int(rng.integers(0, 4))
            if mode == 0: b[off + int(rng.integers(0, 4))] ^= 1 << int(rng.integers(0, 8))
            elif mode == 1: struct.pack_into("<i", b, off, int(rng.choice([-1, 0, 1, 7, 255, 65536, 2**31 - 1, -2**31])))
            elif mode == 2: struct.pack_into("<I", b, off, int(rng.integers(0, 2**32)))
            else: struct.pack_into("<f", b, off, float(rng.choice([np.nan, np.inf, -np.inf, 0.0, 1e38])))
        struct.pack_into("<I", b, 24, zlib.crc32(bytes(b[64:])) & 0xFFFFFFFF)
        open(%r, "wb").write(b)
        try:
            ra.Scene.load(%r, host_only=True).close(); acc += 1
        except ra.RtcError:
            ref += 1
print("accepted", acc, "refused", ref)
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "f.rtcscene"), str(tmp_path / "f.rtcscene"),
       str(tmp_path / "g.rtcscene"), str(tmp_path / "g.rtcscene"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=600)
    assert out.returncode == 0 and b"accepted" in out.stdout, out.stderr.decode()[-2000:]
    acc, ref = int(out.stdout.split()[1]), int(out.stdout.split()[3])
    assert acc + ref == 480 and ref > 0


def test_scene_loader_mutation_fuzz_under_sanitizers(tmp_path):
    """tools/fuzz/scene_fuzz.cpp: random payload words of valid scene files overwritten (small integers, sign bits, huge values), the CRC
    brought up to date so that the damage reaches validate_flat_scene, and whatever loads walked by the host code that runs after
    a load (light matching) -- compiled with ASan + UBSan.  A gap in the validator shows up as an out-of-bounds read."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    import raytracer_amd.api as A
    from conftest import ROOT
    host = os.path.join(ROOT, "raytracer_amd", "csrc", "host")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + host, "-I" + os.path.join(ROOT, "raytracer_amd", "csrc")]
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
    objs = []
    for src in ("scene_api.c", "jpeg_decode.c", "image_decode.c"):
        o = str(tmp_path / (src + ".o"))
        r = subprocess.run(["gcc"] + san + ["-c"] + inc + [os.path.join(host, src), "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        objs.append(o)
    exe = str(tmp_path / "scene_fuzz")
    r = subprocess.run(["g++", "-std=c++17"] + san + ["-static-libasan"] + inc + [os.path.join(ROOT, "tools", "fuzz", "scene_fuzz.cpp"), os.path.join(host, "scene_io.cpp"),
                        os.path.join(host, "flatten.cpp")] + objs + ["-lz", "-lpthread", "-lm", "-o", exe], capture_output=True, text=True)
    if r.returncode != 0 and ("asan" in r.stderr.lower() or "ubsan" in r.stderr.lower()):
        pytest.skip("no sanitizer runtime in this image")
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0")
    for name in ("cornell", "cornell_smoke", "book1"):
        sc, cam, view = A.build_named_scene(name, host_only=True)
        path = str(tmp_path / (name + ".rtcscene"))
        sc.save(path); sc.close()
        run = subprocess.run([exe, path, "400", str(tmp_path / "mutated.rtcscene")], capture_output=True, text=True, env=env, timeout=900)
        assert run.returncode == 0 and "loaded" in run.stdout and "runtime error" not in run.stderr, (run.stdout[-300:], run.stderr[-2500:])
