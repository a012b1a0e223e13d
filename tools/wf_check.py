"""kernel 6 against kernel 2 on C4-shaped scenes (quick GPU check); usage: python tools/wf_check.py [nu nv w h spp]"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import numpy as np
import rtamd
from rtamd import shapes
a = [int(x) for x in sys.argv[1:]] + [40, 80, 64, 64, 4][len(sys.argv) - 1:]
nu, nv, W, H, spp = a[:5]
P, N, I = shapes.torus(nu, nv)
w = rtamd.World(); w.new(shapes.cornell_with_mesh(w, P, N, I), bvh_seed=1)
cam = rtamd.Camera(((278, 278, -800), (278, 278, 278)), (0, 1, 0), 50, 1.0, 0.0, 10.0)
print(w.info(), flush=True)
t = time.time(); ref, s2 = w.render(cam, width=W, height=H, spp=spp, seed=1, kernel=2); print("k2", time.time() - t, s2["kernel_ms"], flush=True)
t = time.time(); img, s6 = w.render(cam, width=W, height=H, spp=spp, seed=1, kernel=6); print("k6", time.time() - t, s6["kernel_ms"], s6["launches"], flush=True)
bad = (img != ref).any(axis=2)
print("differ:", int(bad.sum()), "of", bad.size, "max", float(img.max()))
if bad.any():
    idx = np.argwhere(bad)[:5]
    for i in idx: print(i, img[tuple(i)], ref[tuple(i)])
    sys.exit(1)
