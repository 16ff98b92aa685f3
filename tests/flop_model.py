"""Algorithmic FLOP per ray for the roofline figure of bench.py (measurement tooling kept under tests/ because it runs the oracle: only tests/, smoke() and the cpu_baseline leg of bench.py may).

SURVEY.md 8(d): `roofline.achieved` = algorithmic FLOP per ray x rays per launch / kernel time, where the algorithmic
figure is the MINIMAL algorithm (every leaf of the flattened scene tested once per ray, no BVH / wavefront /
regeneration bookkeeping) counted as add/sub/mul/div/sqrt = 1, FMA = 2, each transcendental = 1, compares/selects
free.  The event counts come from an instrumented oracle run over the same scene and estimator; the per-event costs
below are hand counts of the reference's arithmetic (file:line under /root/reference/03_FullTracer).

Run:  python tests/flop_model.py [scene] [aspect] [W H spp]      (CPU only; prints events/ray and FLOP/ray)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

# FLOP per event.  "test" = the part every candidate pays, "hit" = the extra work of the accepted closest hit.
COST = {
    # hittable.c:494-506: t=(k-o)/d -> 2; two in-plane coordinates o+t*d -> 4
    "rect_tests": 6,
    # hittable.c:509-515 + ray.c:8-11: ray_at 6, face-forward dot 5 (u,v only when a texture needs them)
    "rect_hits": 11,
    # hittable.c:406-429: oc 3, a 5, half_b 5, c 7, disc 3, sqrt 1, root 2
    "sphere_tests": 26,
    # hittable.c:440-445: ray_at 6, (p-c)/r 6, face-forward dot 5
    "sphere_hits": 17,
    "msphere_tests": 26 + 11,          # + moving_sphere_center hittable.c:455-458: 2 + 3*3
    # aabb.c:33-63: per axis rcp 1, 2 sub, 2 mul
    "aabb_tests": 15,
    # hittable.c:563-584: two boundary queries are counted as their own leaf tests; free flight log 1, mul 1, length 6, 4 more
    "medium_tests": 12,
    "translate_calls": 3,              # hittable.c:529 origin - offset
    "translate_hits": 8,               # :533-534 p + offset 3, re-face dot 5
    "rotate_calls": 12,                # :542-545 origin and direction, 2 components x (2 mul + 1 add) each
    "rotate_hits": 17,                 # :552-557 p 6, normal 6, re-face dot 5
    # onb.c:7-12: unit 9, cross 9, unit 9, cross 9
    "onb_builds": 36,
    # vec3.c:97-107 + onb.c:25-30: phi 1, cos 1, sin 1, sqrt(r2) 1, 2 mul, sqrt(1-r2) 2, local 15
    "cosine_sample": 23,
    "light_sample_rect": 9,            # hittable.c:838-848: 2 x (a + (b-a)*xi) 6, point - origin 3
    # hittable.c:821-835: direction 3, len2 5, cone sample (sqrt(1-r^2/d^2) 4, z 3, phi 1, cos/sin 2, sqrt(1-z^2) 3, 2 mul) 15, local 15
    # (its ONB is counted under onb_builds)
    "light_sample_sphere": 38,
    # hittable.c:772-788: the rect test itself is counted under rect_tests
    "light_pdf_rect": 0,
    # on a hit: area 3, distance^2 = t*t*len2 7, cosine = |d.n|/len 3, d2/(cos*area) 2
    "light_pdf_rect_hits": 15,
    "light_pdf_sphere": 0,             # hittable.c:760-768: the sphere test is counted under sphere_tests
    "light_pdf_sphere_hits": 15,       # cos_theta_max (sub 3, len2 5, div, sub, sqrt) 11, 1/(2pi(1-c)) 4
    # per diffuse bounce: texture (solid) 0, pdf.c:100-103 cosine pdf (unit 9, dot 5, /pi 1) 15, mixture pdf.c:119-121 3,
    # list average hittable_list.c:165-172 (2 per light, 2 lights) 4, material.c:205-208 scattering pdf (dot 5, /pi 1; unit shared) 6,
    # ray.c:80-83 throughput (div 1, 3+3 mul) 7, emitted add 3
    "diffuse": 38,
    # material.c:130-138: unit 9, reflect 12, fuzz*xi + add 6, attenuation 3
    "specular_metal": 30,
    # material.c:144-162: unit 9, cos 5, sin 3, ratio*sin 1, reflectance 12, refract vec3.c:44-50 (perp 9, len2 5, 1-x/abs/sqrt/neg 3, par 3, add 3) 23
    "specular_dielectric": 53,
    "isotropic": 12,
    "emit": 6,                         # ray.c:60-63: L += T * emitted
    "miss": 6,                         # ray.c:56: L += T * background
    # main.c:150-151 jitter 4, camera.c:46-53: lens 9, direction 18, origin 3, time 2
    "camera_rays": 36,
    # perlin.c:108-148 turb: 7 octaves x (8 corners x (dot 5, weights 9, acc 1) + hermite 9 + p*2 3) ; texture.c:128-132 sin + 4
    "tex_noise": 7 * (8 * 15 + 12) + 5,
    "tex_checker": 8,                  # texture.c:121-124: 3 sin, 3 mul by 10, 2 mul
    "tex_image": 8,                    # texture.c:136-152
}


def measure(scene="cornell", aspect=16 / 9, W=480, H=270, spp=16, stage=None):
    import hx
    o = hx.oracle()
    if stage is None:
        stage = {"book1": 1}.get(scene, 3 if scene in ("cornell", "simple") else 2)
    o.set_stage(stage)
    o.build(scene, aspect=aspect)
    hx.counters_reset(o)
    _, st = o.oracle_render(W, H, 0, spp, depth=0, rng="philox", seed=1, threads=0)
    c = hx.counters_get(o)
    rays = c["rays"]
    per_ray = {k: c[k] / rays for k in COST}
    flop = {k: per_ray[k] * COST[k] for k in COST}
    return dict(scene=scene, aspect=aspect, W=W, H=H, spp=spp, rays=int(rays), samples=int(c["samples"]),
                rays_per_sample=rays / c["samples"], events_per_ray={k: round(v, 4) for k, v in per_ray.items() if v},
                flop_per_ray_by_event={k: round(v, 2) for k, v in flop.items() if v}, flop_per_ray=sum(flop.values()))


if __name__ == "__main__":
    a = sys.argv[1:]
    scene = a[0] if a else "cornell"
    aspect = float(a[1]) if len(a) > 1 else 16 / 9
    dims = [int(x) for x in a[2:5]] if len(a) >= 5 else [480, 270, 16]
    print(json.dumps(measure(scene, aspect, *dims), indent=1))
