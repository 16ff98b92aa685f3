"""GPU: the multi-rank path end to end on the one GPU a test box has.

Two fresh processes (started with subprocess, so neither inherits an initialised HIP runtime), each rank renders ITS
shard_samples range of every pixel through rtc_render_device on cuda:0 into its own accumulator, the accumulators are summed
onto rank 0 with torch.distributed (gloo over host copies here: RCCL needs one GPU per rank; bench.py uses "nccl" on a node),
rank 0 tonemaps.  The result equals the single-process render of the whole range within float reassociation.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
import raytracer_amd as ra
from raytracer_amd import dist as rd
rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
ra.rtc_init()
res = {{}}
for name, W, H, spp in (("cornell", 200, 120, 67), ("book1", 160, 90, 21)):
    sc, cam, view = ra.build_named_scene(name, aspect=W / H)
    begin, end = rd.shard_samples(spp, rank, world)
    acc = rd.new_accum(H, W, "cuda:0")
    stream = rd.render_into(sc, cam, ra.Settings(W, H, 0, view.max_depth, view.integrator, seed=7, sample_begin=begin, sample_end=end), acc)
    torch.cuda.synchronize()
    st = sc.stats(stream)
    host = acc.cpu()
    rd.reduce_accum(host, dst=0)                       # the one collective of the path
    rays = torch.tensor([float(st["rays"])], dtype=torch.float64)
    dist.reduce(rays, dst=0)
    if rank == 0:
        acc.copy_(host)
        u8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda:0")
        assert ra.lib().rtc_tonemap_u8_device(acc.data_ptr(), W, H, u8.data_ptr(), None) == 0
        torch.cuda.synchronize()
        res[name + "_acc"], res[name + "_u8"], res[name + "_rays"] = host.numpy(), u8.cpu().numpy(), rays.numpy()
    sc.close()
dist.barrier()
if rank == 0:
    np.savez(out, **res)
dist.destroy_process_group()
"""


def test_two_ranks_on_one_gpu_equal_one_process(gpu, tmp_path):
    out = str(tmp_path / "ranks.npz")
    code = WORKER.format(root=ROOT)
    port = str(29500 + os.getpid() % 400)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), "2", port, out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env) for r in range(2)]
    for p in procs:
        try:
            so, se = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, se.decode()[-3000:]
    got = np.load(out)
    for name, W, H, spp in (("cornell", 200, 120, 67), ("book1", 160, 90, 21)):
        sc, cam, view = gpu.build_named_scene(name, aspect=W / H)
        try:
            acc, st = sc.render(cam, gpu.Settings(W, H, 0, view.max_depth, view.integrator, seed=7, sample_begin=0, sample_end=spp))
        finally:
            sc.close()
        a = got[name + "_acc"]
        assert (a[..., 3] == spp).all()                                # sample counts add up
        assert np.allclose(a[..., :3], acc[..., :3], rtol=1e-5, atol=1e-5 * spp)   # float sums reassociated across ranks
        assert int(got[name + "_rays"][0]) == st["rays"]               # the same paths were traced, split in two
        d = np.abs(got[name + "_u8"].astype(int) - gpu.tonemap_u8(acc).astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3                  # bytes: equal except where a sum sits on a rounding boundary


def test_c_abi_multi_device_render_has_no_serial_tail(gpu):
    """rtc_render_multi keeps its accumulators, staging buffers and streams on the handles, enqueues every shard from one thread, runs
    the peer copies side by side and synchronises once: with two handles on ONE GPU (the same total work as one render) a 1080p x
    256 spp call takes at most 1.10 x the wall time of rtc_render on one handle.  (Round 2's form -- hipMalloc / hipMemset / hipFree per
    call, copies one after the other with a device synchronisation each, a scalar host loop over the image -- took 2 x.)"""
    import ctypes as C
    import time
    L = gpu.lib()
    W, H, spp = 1920, 1080, 256
    sc, cam, view = gpu.build_named_scene("cornell", aspect=W / H)
    clone = L.rtc_scene_clone_to_device(sc.handle, 0)
    assert clone, L.rtc_last_error()
    try:
        st = gpu.Settings(W, H, spp, view.max_depth, view.integrator, seed=9)
        handles = (C.c_void_p * 2)(sc.handle, clone)
        one, two = [], []
        for rep in range(6):                                             # the first round allocates; minimum of the other five (a shared box is noisy)
            acc = np.zeros((H, W, 4), dtype=np.float32)
            t0 = time.perf_counter(); sc.render(cam, st, accum=acc); one.append(time.perf_counter() - t0)
            acc2 = np.zeros((H, W, 4), dtype=np.float32)
            t0 = time.perf_counter(); rc = L.rtc_render_multi(handles, 2, cam.handle, C.byref(st.c), acc2.ctypes.data); two.append(time.perf_counter() - t0)
            assert rc == 0, L.rtc_last_error()
        assert (acc2[..., 3] == spp).all() and np.allclose(acc2[..., :3], acc[..., :3], rtol=1e-5, atol=2e-5 * spp)
        t1, t2 = min(one[1:]), min(two[1:])
        print(f"rtc_render {1e3 * t1:.1f} ms, rtc_render_multi with two handles on one GPU {1e3 * t2:.1f} ms ({t2 / t1:.2f} x)")
        assert t2 <= 1.10 * t1
    finally:
        L.rtc_scene_free(clone)
        sc.close()


def test_c_abi_multi_device_render_equals_one_device(gpu):
    """rtc_render_multi (include/rtc.h): one host thread per scene handle, sample shards, peer-copy reduce on the root.  A one-GPU
    box runs it with two and three handles on cuda:0 (rtc_scene_clone_to_device): sums equal the single render within float
    reassociation, sample counts exact, the call adds to what the host buffer held; the same handle twice is refused."""
    import ctypes as C
    L = gpu.lib()
    # (the last case is BASELINE config 4's image size: 3840 x 2160, a few samples of its 16,384)
    for name, W, H, spp in (("cornell", 240, 135, 70), ("book1", 160, 90, 20), ("cornell", 3840, 2160, 7)):
        sc, cam, view = gpu.build_named_scene(name, aspect=W / H)
        clones = []
        try:
            st = gpu.Settings(W, H, 0, view.max_depth, view.integrator, seed=5, sample_begin=3, sample_end=3 + spp)
            want, _ = sc.render(cam, st)
            for n in (2, 3):
                while len(clones) < n - 1:
                    h = L.rtc_scene_clone_to_device(sc.handle, 0)
                    assert h, L.rtc_last_error()
                    clones.append(h)
                handles = (C.c_void_p * n)(sc.handle, *clones[:n - 1])
                acc = np.full((H, W, 4), 1.0, dtype=np.float32)
                rc = L.rtc_render_multi(handles, n, cam.handle, C.byref(st.c), acc.ctypes.data)
                assert rc == 0, L.rtc_last_error()
                assert (acc[..., 3] == spp + 1).all()
                assert np.allclose(acc[..., :3] - 1.0, want[..., :3], rtol=1e-5, atol=2e-5 * spp)
            dup = (C.c_void_p * 2)(sc.handle, sc.handle)
            assert L.rtc_render_multi(dup, 2, cam.handle, C.byref(st.c), acc.ctypes.data) != 0 and b"twice" in L.rtc_last_error()
            assert L.rtc_scene_clone_to_device(sc.handle, 99) is None
        finally:
            for h in clones:
                L.rtc_scene_free(h)
            sc.close()


def test_cli_multi_gpu_flag(gpu, tmp_path):
    """rtc_render -g N (csrc/host/main.c): -g 0 takes every visible device; more devices than the process sees is refused with a
    message, not a fault.  With one GPU the multi-device branch itself is covered by the C-ABI test above."""
    import subprocess
    exe = os.path.join(os.path.dirname(gpu.library_path()), "rtc_render")
    n = gpu.lib().rtc_device_count()
    assert n >= 1
    a, b = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    common = [exe, "-w", "96", "-h", "64", "-s", "8", "-r", "3"]
    one = subprocess.run(common + ["-o", a], check=True, capture_output=True, timeout=120).stdout.decode()
    allg = subprocess.run(common + ["-o", b, "-g", "0"], check=True, capture_output=True, timeout=120).stdout.decode()
    assert "Msamples/s" in one and "Msamples/s" in allg
    if n == 1:
        assert open(a, "rb").read() == open(b, "rb").read()
    too_many = subprocess.run(common + ["-o", b, "-g", str(n + 1)], capture_output=True, timeout=120)
    assert too_many.returncode != 0 and b"device(s)" in too_many.stderr


def test_bench_two_ranks_on_one_gpu(gpu):
    """bench.py's N > 1 path end to end (torch.distributed.run, two ranks, the frame pipeline with its overlapped reduce, the
    max-over-ranks timing, rank 0's JSON line) on the one GPU of a test box: both ranks on cuda:0, gloo instead of RCCL (test
    hooks of bench.py).  Checks the line's bookkeeping, not the speed."""
    import json
    env = dict(os.environ, RTC_BENCH_ONE_DEVICE="1", RTC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = str(29900 + os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--width", "480", "--height", "270", "--spp", "32"]
    p = subprocess.run(cmd, capture_output=True, timeout=400, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["config"]["global_spp"] == 64 and line["config"]["spp_per_gpu"] == 32
    samples = 480 * 270 * 64 * 3
    assert abs(line["value"] * 1e6 * line["ms_per_step"] * 1e-3 * 3 - samples) <= 1e-6 * samples
    assert 2.5 < line["rays_per_sample"] < 4.0 and "cpu_baseline" not in line and "other_configs" not in line


@pytest.mark.gpu
def test_bench_distributed_path_through_rccl_with_one_rank(gpu):
    """The N > 1 code path of bench.py with the backend the driver uses -- RCCL ("nccl"): process group bound to the device, the
    frame pipeline's async reduce and its wait(), barrier, float64 all_reduce of the timings -- on the one GPU of a test box, as a
    one-rank group (RTC_BENCH_FORCE_DIST=1).  The two-rank test above covers the sharding with gloo; this one covers the calls
    into RCCL that gloo replaces there."""
    import json
    env = dict(os.environ, RTC_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = str(29800 + os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--width", "480", "--height", "270", "--spp", "32", "--no-cpu", "--no-other"]
    p = subprocess.run(cmd, capture_output=True, timeout=400, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["steps"] == 3
    samples = 480 * 270 * 32 * 3
    assert abs(line["value"] * 1e6 * line["ms_per_step"] * 1e-3 * 3 - samples) <= 1e-6 * samples
    assert 2.5 < line["rays_per_sample"] < 4.0
