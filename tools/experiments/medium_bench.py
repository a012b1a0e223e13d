"""A scene_500-class sphere field (485 spheres on a ground sphere) wrapped in a thin fog (a ConstantMedium with a sphere boundary around the whole scene): kernel 1 (reference order) against
kernel 2's MEDIA variant (accel for the surfaces, media resolved in the reference's visit order); images must be identical.
usage: python tools/medium_bench.py [spp]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import numpy as np
import rtamd
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
w = rtamd.World()
rng = np.random.default_rng(5)
mats = [w.Lambertian(w.ConstantTexture(tuple(rng.random(3)))) for _ in range(8)] + [w.Metal(w.ConstantTexture((0.8, 0.8, 0.8)), 0.1), w.Dielectric(1.5, w.ConstantTexture((1.0, 1.0, 1.0)))]
items = [w.Sphere((0.0, -1000.0, 0.0), 1000.0, mats[0])]
for a in range(-11, 11):
    for b in range(-11, 11):
        items.append(w.Sphere((a + 0.9 * rng.random(), 0.2, b + 0.9 * rng.random()), 0.2, mats[int(rng.integers(0, len(mats)))]))
fog = w.Isotropic(w.ConstantTexture((0.9, 0.9, 1.0)))
items.append(w.ConstantMedium(0.01, w.Sphere((0.0, 0.0, 0.0), 60.0, mats[1]), fog))
w.new(items, bvh_seed=1)
cam = rtamd.Camera(((13.0, 2.0, 3.0), (0.0, 0.0, 0.0)), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)
out = {}
imgs = {}
for k in (1, 2):
    w.render(cam, width=1200, height=800, spp=2, seed=1, kernel=k)
    imgs[k], st = w.render(cam, width=1200, height=800, spp=spp, seed=1, kernel=k)
    out["kernel %d" % k] = round(st["samples"] / (st["kernel_ms"] * 1e-3) / 1e6, 1)
out["identical"] = bool(np.array_equal(imgs[1], imgs[2]))
print(json.dumps(out))
