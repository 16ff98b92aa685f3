"""GPU: rtc_render_device_noalloc -- the one entry point that allocates nothing (its scratch is a file-scope device array)
-- against rtc_render_device, through the C-ABI, on two calls in a row, the second with a smaller input.

Bar: bit-exact.  Both entry points launch the same kernels with the same chunking at these sizes; the scratch array keeps the
first call's tile costs, tile order and partial planes, and the second, smaller call must not see any of it.  A third call
whose partial planes would not fit the static array runs with fewer chunks per tile: float sums reassociated, relative 1e-5.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_noalloc_entry_matches_render_device_on_two_calls_in_a_row(gpu):
    import torch
    L = gpu.lib()
    sc, cam, view = gpu.build_named_scene("cornell")
    try:
        jobs = [gpu.Settings(512, 384, 0, 10, 0, seed=11, sample_begin=0, sample_end=64),      # 3,072 tiles, probe + order, 2 chunks per tile
                gpu.Settings(203, 117, 0, 10, 0, seed=12, sample_begin=5, sample_end=45),      # smaller, ragged edges, 1 chunk, no probe
                gpu.Settings(512, 384, 0, 10, 0, seed=13, sample_begin=0, sample_end=64)]      # and the first size again, behind the small one
        want = []
        for st in jobs:
            acc = torch.zeros((st.c.height, st.c.width, 4), dtype=torch.float32, device="cuda:0")
            sc.render_device(cam, st, acc.data_ptr(), None)
            torch.cuda.synchronize()
            want.append(acc.cpu().numpy())
        free0 = torch.cuda.mem_get_info()[0]
        got = []
        for st in jobs:                                                  # back to back on one stream, no synchronisation in between
            acc = torch.full((st.c.height, st.c.width, 4), 0.0, dtype=torch.float32, device="cuda:0")
            rc = L.rtc_render_device_noalloc(sc.handle, cam.handle, C.byref(st.c), acc.data_ptr(), None)
            assert rc == 0, L.rtc_last_error()
            got.append(acc)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0                     # nothing was allocated on the device by the three calls (the
                                                                         # accumulators come out of torch's cached blocks: torch.full above)
        for g, w, st in zip(got, want, jobs):
            g = g.cpu().numpy()
            assert (g[..., 3] == st.c.sample_end - st.c.sample_begin).all()
            assert np.array_equal(g, w)
            assert g[..., :3].sum() > 0
        # it ADDS like rtc_render_device: a second pass with the next sample range on top of the first
        st_a = gpu.Settings(203, 117, 0, 10, 0, seed=12, sample_begin=5, sample_end=25)
        st_b = gpu.Settings(203, 117, 0, 10, 0, seed=12, sample_begin=25, sample_end=45)
        acc = torch.zeros((117, 203, 4), dtype=torch.float32, device="cuda:0")
        ref = torch.zeros_like(acc)
        for st in (st_a, st_b):
            assert L.rtc_render_device_noalloc(sc.handle, cam.handle, C.byref(st.c), acc.data_ptr(), None) == 0
            sc.render_device(cam, st, ref.data_ptr(), None)
        torch.cuda.synchronize()
        assert np.array_equal(acc.cpu().numpy(), ref.cpu().numpy())
        # partial planes that do not fit the static array (1920 x 1080 x 2 chunks = 4 M float4 > 1 M): fewer chunks, same image
        big = gpu.Settings(1920, 1080, 0, 10, 0, seed=14, sample_begin=0, sample_end=64)
        a = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda:0")
        b = torch.zeros_like(a)
        assert L.rtc_render_device_noalloc(sc.handle, cam.handle, C.byref(big.c), a.data_ptr(), None) == 0
        sc.render_device(cam, big, b.data_ptr(), None)
        torch.cuda.synchronize()
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert np.array_equal(a[..., 3], b[..., 3]) and np.allclose(a[..., :3], b[..., :3], rtol=1e-5, atol=1e-4)
        # limits and refusals, each with a message
        too_big = gpu.Settings(8192, 8192, 1, 10, 0)
        assert L.rtc_render_device_noalloc(sc.handle, cam.handle, C.byref(too_big.c), a.ctypes.data, None) != 0
        assert b"too large" in L.rtc_last_error()
        host = np.zeros((117, 203, 4), dtype=np.float32)
        assert L.rtc_render_device_noalloc(sc.handle, cam.handle, C.byref(st_a.c), host.ctypes.data, None) != 0
        assert b"device memory" in L.rtc_last_error()
        assert L.rtc_render_device_noalloc(sc.handle, cam.handle, None, acc.data_ptr(), None) != 0
        # under HIP graph capture: a captured call must not allocate, synchronise or query -- capture fails if it does; the replay
        # renders the same image as the eager call
        st_g = gpu.Settings(203, 117, 0, 10, 0, seed=21, sample_begin=0, sample_end=32)
        eager = torch.zeros((117, 203, 4), dtype=torch.float32, device="cuda:0")
        assert L.rtc_render_device_noalloc(sc.handle, cam.handle, C.byref(st_g.c), eager.data_ptr(), None) == 0
        torch.cuda.synchronize()
        captured = torch.zeros_like(eager)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            graph.capture_begin()
            rc = L.rtc_render_device_noalloc(sc.handle, cam.handle, C.byref(st_g.c), captured.data_ptr(), side.cuda_stream)
            graph.capture_end()
        assert rc == 0, L.rtc_last_error()
        torch.cuda.synchronize()
        assert not captured.any()                                        # captured, not run
        graph.replay(); torch.cuda.synchronize()
        assert np.array_equal(captured.cpu().numpy(), eager.cpu().numpy())
        graph.replay(); torch.cuda.synchronize()                         # the call ADDS: a second replay doubles the sums
        assert np.array_equal(captured.cpu().numpy()[..., 3], 2 * eager.cpu().numpy()[..., 3])
    finally:
        sc.close()
    bvh, cam2, view2 = gpu.build_named_scene("book1")
    try:
        acc = torch.zeros((90, 160, 4), dtype=torch.float32, device="cuda:0")
        st = gpu.Settings(160, 90, 4, view2.max_depth, view2.integrator)
        assert L.rtc_render_device_noalloc(bvh.handle, cam2.handle, C.byref(st.c), acc.data_ptr(), None) != 0
        assert b"small scene" in L.rtc_last_error()
    finally:
        bvh.close()
