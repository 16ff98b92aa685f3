import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import raytracer_amd as ra
ra.rtc_init()
tag = sys.argv[1]
for name, W, H, spp, arg in (("final", 512, 512, 32, 0), ("book1", 640, 360, 32, 0), ("box_field_smoke", 256, 256, 32, 0)):
    sc, cam, view = ra.build_named_scene(name, arg=arg, aspect=W / H)
    acc, st = sc.render(cam, ra.Settings(W, H, spp, view.max_depth, view.integrator, seed=3, kernel=3))
    np.save(f"/root/repo/gpurun_out/abwf_{tag}_{name}.npy", acc)
    print(tag, name, st["kernel_ms"], st["rays"], flush=True)
    sc.close()
