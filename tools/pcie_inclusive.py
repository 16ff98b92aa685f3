"""PCIe-inclusive frame time of the host-buffer entry point rtc_render: the caller's accumulator goes up,
the kernel runs, the accumulator comes back.  Never the bench's `value` (that is measured with the buffers resident in HBM)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra
ra.rtc_init()
sc, cam, view = ra.build_named_scene("cornell", aspect=16 / 9)
W, H, spp = 1920, 1080, 256
st = ra.Settings(W, H, spp, 10, view.integrator, seed=0)
acc = np.zeros((H, W, 4), dtype=np.float32)
times = []
for i in range(6):
    acc[:] = 0
    t0 = time.perf_counter(); out, stats = sc.render(cam, st, accum=acc); t1 = time.perf_counter()
    times.append((t1 - t0) * 1e3)
print("rtc_render 1920x1080x256, host buffer: frames %s ms; best %.2f ms = %.1f Msamples/s; kernel %.2f ms" %
      (" ".join("%.1f" % t for t in times), min(times[1:]), W * H * spp / min(times[1:]) * 1e-3, stats["kernel_ms"]), flush=True)
sc.close()
