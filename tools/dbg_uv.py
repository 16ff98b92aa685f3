import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, raytracer_amd as ra
ra.rtc_init()
fx = np.load(os.path.join(R, "tests/golden/first_hit_final02.npz"))
sc, cam, view = ra.build_named_scene("final")
np.set_printoptions(linewidth=200, precision=6)
for tag in ("g64x36", "g32x32", "sec"):
    prim, t, rec = sc.first_hit(fx[tag + "_rays"], want_rec=True)
    ref = fx[tag + "_rec"]; keep = (ref[:, 11] != 4) & (fx[tag + "_prim"] >= 0)
    d = np.abs(rec[:, 8:10] - ref[:, 8:10]).max(axis=1)
    bad = np.nonzero(keep & (d > 1e-12))[0]
    print(tag, len(bad))
    for i in bad[:10]:
        print(i, prim[i], fx[tag + "_prim"][i], "leaf", fx["leaves"][prim[i]][:4], "ref uv", ref[i, 8:10], "gpu uv", rec[i, 8:10], "n", ref[i, 5:8])
