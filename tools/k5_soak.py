"""One-off soak of kernel 5 against kernel 2: larger frames and many random instance placements, repeated, bit for bit.
usage: python tools/k5_soak.py [trials]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import numpy as np, rtamd
from rtamd import shapes
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cam = rtamd.Camera(((278, 278, -800), (278, 278, 278)), (0, 1, 0), 50, 1.0, 0.0, 10.0)
bad = 0
P, N, I = shapes.torus(160, 320)
w = rtamd.World(); w.new(shapes.cornell_with_mesh(w, P, N, I), bvh_seed=1)
for rep in range(3):
    a, _ = w.render(cam, width=600, height=600, spp=16, seed=rep, kernel=2)
    for integ in (0,):
        b, st = w.render(cam, width=600, height=600, spp=16, seed=rep, kernel=5)
        same = np.array_equal(a, b, equal_nan=True)
        bad += not same
        print("C4 600x600x16 seed %d: kernel 5 == kernel 2: %s (%.0f Msamples/s)" % (rep, same, st["samples"] / st["kernel_ms"] / 1e3), flush=True)
rng = np.random.default_rng(77)
for trial in range(trials):
    meshes = [shapes.torus(int(rng.integers(6, 120)), int(rng.integers(8, 160))) for _ in range(int(rng.integers(1, 5)))]
    ww = rtamd.World()
    P0, N0, I0 = meshes[0]
    items = shapes.cornell_with_mesh(ww, P0, N0, I0, scale=float(rng.uniform(60.0, 160.0)), translate=tuple(rng.uniform(150.0, 400.0, 3)), rotate=tuple(rng.uniform(-180.0, 180.0, 3)))
    mats = [ww.Lambertian(ww.ConstantTexture((0.6, 0.6, 0.6))), ww.Dielectric(1.5, ww.ConstantTexture((1.0, 1.0, 1.0))), ww.Metal(ww.ConstantTexture((0.8, 0.8, 0.9)), 0.05)]
    for k, (Pm, Nm, Im) in enumerate(meshes[1:]):
        mesh = ww.Mesh(Pm, Nm, Im, mats[k % 3], bvh_seed=7 + k)
        items.append(ww.Transform(tuple(rng.uniform(-180.0, 180.0, 3)), tuple(rng.uniform(20.0, 110.0, 3)), tuple(rng.uniform(100.0, 450.0, 3)), mesh))
    ww.new(items, bvh_seed=trial)
    integ = int(trial % 2)
    if integ == 1:
        continue_ok = True
    a, _ = ww.render(cam, width=256, height=256, spp=12, seed=trial, kernel=2)
    b, st = ww.render(cam, width=256, height=256, spp=12, seed=trial, kernel=5)
    same = np.array_equal(a, b, equal_nan=True)
    bad += not same
    print("trial %d: %d instances, %d triangles: kernel 5 == kernel 2: %s" % (trial, len(meshes), ww.info()["n_tris"], same), flush=True)
print("MISMATCHES:", bad)
sys.exit(1 if bad else 0)
