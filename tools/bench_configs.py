"""One-GPU timing of every BASELINE.json configuration that runs on a GPU (measurement tooling; bench.py stays the
headline line).  Prints one JSON object per configuration: kernel time from the library's HIP events, Msamples/s, Mrays/s,
the kernel that ran, and the BVH shape where there is one.

    python tools/bench_configs.py [name ...] [--reps N] [--out FILE]

  cornell_1080p      config 2: 03 Cornell box, 1920x1080, aspect 16/9, 256 spp, depth 10
  cornell_1024       the reference's own defaults (main.c:88-92): 1024x1024, aspect 1, 256 spp
  cornell_4k_share   config 4, one GPU's share: 3840x2160, 2048 of the 16384 spp
  final_1024         config 3: 02 final scene, 1024x1024, 256 spp, depth 8
  final_earth_1080p  config 3 with the earth-map globe, 1920x1080, 256 spp
  book1_1080p        config 1 on the GPU: Book-1 spheres, 1920x1080, 8 spp (the reference's setting) and 64 spp
  book2_1080p        not a BASELINE configuration: the Book-2 random scene (checker ground: every ground hit goes through k_wf_shade), 1920x1080, 64 spp
  million_1080p      config 5: 1,000,004 spheres, 1920x1080, 64 spp, depth 8
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as ra  # noqa: E402

CONFIGS = {  # name: (scene, arg, aspect, W, H, spp)
    "cornell_1080p": ("cornell", 0, 16 / 9, 1920, 1080, 256),
    "cornell_1024": ("cornell", 0, None, 1024, 1024, 256),
    "cornell_4k_share": ("cornell", 0, 16 / 9, 3840, 2160, 2048),
    "final_1024": ("final", 0, None, 1024, 1024, 256),
    "final_earth_1080p": ("final_earth", 0, 16 / 9, 1920, 1080, 256),
    "book1_1080p_8spp": ("book1", 0, None, 1920, 1080, 8),
    "book1_1080p": ("book1", 0, None, 1920, 1080, 64),
    "book2_1080p": ("book2_random", 0, None, 1920, 1080, 64),
    "million_1080p": ("million", 1000, None, 1920, 1080, 64),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dev = ra.rtc_init()
    lines = []
    for name, (scene, arg, aspect, W, H, spp) in CONFIGS.items():
        if a.names and name not in a.names:
            continue
        t0 = time.perf_counter()
        sc, cam, view = ra.build_named_scene(scene, arg=arg, aspect=aspect)
        build_s = time.perf_counter() - t0
        best = None
        for rep in range(a.reps):
            acc, st = sc.render(cam, ra.Settings(W, H, spp, view.max_depth, view.integrator, seed=rep, kernel=a.kernel))
            if best is None or st["kernel_ms"] < best["kernel_ms"]:
                best = st
        info = sc.bvh_info()
        line = dict(config=name, scene=scene, width=W, height=H, spp=spp, max_depth=view.max_depth, integrator=view.integrator,
                    prims=sc.num_prims(), bvh=info, scene_build_s=round(build_s, 3), kernel_used=best["kernel_used"],
                    kernel_ms=round(best["kernel_ms"], 3), msamples_per_s=round(best["samples"] / best["kernel_ms"] * 1e-3, 1),
                    mrays_per_s=round(best["rays"] / best["kernel_ms"] * 1e-3, 1), rays_per_sample=round(best["rays"] / best["samples"], 4),
                    device=dev)
        sc.close()
        lines.append(line)
        print(json.dumps(line), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            for line in lines:
                f.write(json.dumps(line) + "\n")


if __name__ == "__main__":
    main()
