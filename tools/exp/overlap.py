"""Experiment: do two wavefront renders on two streams overlap on the GPU?  (development aid)
    python tools/exp/overlap.py <scene> <W> <H> <spp> <n_streams>"""
import os, sys, threading, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import raytracer_amd as ra
ra.rtc_init()
name, W, H, spp, ns = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
arg = 1000 if name == "million" else 0
scs = [ra.build_named_scene(name, arg=arg, aspect=W / H) for _ in range(ns)]
streams = [torch.cuda.Stream() for _ in range(ns)]
accs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(ns)]
view = scs[0][2]
def work(k, lo, hi):
    sc, cam, _ = scs[k]
    sc.render_device(cam, ra.Settings(W, H, 0, view.max_depth, view.integrator, seed=1, sample_begin=lo, sample_end=hi), accs[k].data_ptr(), streams[k].cuda_stream)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k, k * spp // ns, (k + 1) * spp // ns)) for k in range(ns)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{name} {W}x{H}x{spp} on {ns} stream(s): {1e3 * (t1 - t0):.2f} ms wall", flush=True)
