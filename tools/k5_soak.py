"""One-off soak of kernels 5 and 6 against kernel 2: larger frames and many random instance placements (every third trial also with
instances the service cannot defer: a Cube and a sphere under Transforms), repeated, bit for bit.
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
    for k in (5, 6):
        b, st = w.render(cam, width=600, height=600, spp=16, seed=rep, kernel=k)
        same = np.array_equal(a, b, equal_nan=True)
        bad += not same
        print("C4 600x600x16 seed %d: kernel %d == kernel 2: %s (%.0f Msamples/s)" % (rep, k, same, st["samples"] / st["kernel_ms"] / 1e3), flush=True)
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
    if trial % 3 == 2:  # instances that are entered in the lane
        items.append(ww.Transform(tuple(rng.uniform(-90.0, 90.0, 3)), (1.0, 1.5, 1.0), tuple(rng.uniform(50.0, 400.0, 3)), ww.Cube((0.0, 0.0, 0.0), (80.0, 80.0, 80.0), mats[0])))
        items.append(ww.Transform((0.0, 0.0, 0.0), (40.0, 60.0, 40.0), tuple(rng.uniform(100.0, 450.0, 3)), ww.Sphere((0.0, 0.0, 0.0), 1.0, mats[1])))
    ww.new(items, bvh_seed=trial)
    a, _ = ww.render(cam, width=256, height=256, spp=12, seed=trial, kernel=2)
    res = []
    for k in (5, 6):
        b, st = ww.render(cam, width=256, height=256, spp=12, seed=trial, kernel=k)
        same = np.array_equal(a, b, equal_nan=True)
        bad += not same
        res.append(same)
    print("trial %d: %d instances, %d triangles: kernels 5, 6 == kernel 2: %s" % (trial, ww.info()["accel_instances"], ww.info()["n_tris"], res), flush=True)
print("MISMATCHES:", bad)
sys.exit(1 if bad else 0)
