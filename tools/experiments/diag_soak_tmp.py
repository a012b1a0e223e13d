import os, sys
sys.argv = ["x", "2", "100000", "inst"]
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import numpy as np, rtamd
from rtamd import shapes
np.set_printoptions(precision=17, linewidth=250)
for sc in (1,):
    rng = np.random.default_rng(4000 + sc)
    w = rtamd.World()
    m = w.Lambertian(w.ConstantTexture((0.5, 0.5, 0.5)))
    glass = w.Dielectric(1.5, w.ConstantTexture((1.0, 1.0, 1.0)))
    lamp = w.DiffuseLight(w.ConstantTexture((3.0, 2.5, 2.0)))
    items = []; desc = []
    for k in range(40):
        lo = rng.integers(0, 6, 3); ext = rng.integers(1, 3, 3)
        items.append(w.Cube(tuple(float(v) for v in lo), tuple(float(v) for v in lo + ext), glass if k % 3 == 0 else lamp if k % 7 == 1 else m))
        desc.append(("cube", lo, lo + ext))
    for k in range(8):
        lo = rng.integers(0, 6, 3); ext = rng.integers(1, 3, 3)
        mesh = shapes.box_mesh(4, tuple(float(v) for v in ext)) if k % 2 == 0 else shapes.sheet(8, (float(ext[0]), float(ext[2])))
        items.append(w.Transform((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), tuple(float(v) for v in lo), w.Mesh(*mesh, glass if k % 4 == 0 else m, bvh_seed=sc + k)))
        desc.append(("boxmesh" if k % 2 == 0 else "sheet", lo, ext))
    for _ in range(16):
        axis = int(rng.integers(0, 3)); a0, b0 = rng.integers(0, 5, 2)
        a1, b1 = a0 + int(rng.integers(1, 4)), b0 + int(rng.integers(1, 4)); k = float(rng.integers(0, 8))
        ctor = (w.YZRectangle, w.XZRectangle, w.XYRectangle)[axis]
        items.append(ctor((float(a0), float(b0)), (float(a1), float(b1)), k, m))
        desc.append(("rect", axis, (a0, b0, a1, b1, k)))
    order = rng.permutation(len(items))
    w.new([items[i] for i in order], bvh_seed=int(sc + 1))
    n_rays = 100000
    o = np.concatenate([rng.uniform(-0.5, 8.5, (n_rays // 2, 3)), rng.uniform(-6.0, 14.0, (n_rays - n_rays // 2, 3))])
    d = rng.normal(size=(n_rays, 3))
    d[: n_rays // 10] = np.round(d[: n_rays // 10] * 2.0) / 2.0 + 0.25
    rays = np.concatenate([o, d], axis=1)
    ref = w.debug_hit(rays, kernel=1)
    got = w.debug_hit(rays, kernel=2)
    bad = np.nonzero(~((got == ref) | (np.isnan(got) & np.isnan(ref))).all(axis=1))[0]
    print("bad", bad)
    for i in bad:
        print("ray", i, rays[i]); print("  k1", ref[i]); print("  k2", got[i])
    for j, dd in enumerate(desc): print(j, dd)
    print("order", order)
