import sys, os
sys.path.insert(0, 'rust-raytracer_amd'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import numpy as np, rtamd, oracle
from conftest import scene_path
np.set_printoptions(precision=17, linewidth=250)
scale = 1e-3
rng = np.random.default_rng(int(scale * 1000) % 9973 + 7)
P, N, I = oracle.load_obj(scene_path("cube.obj"))
for trial in range(3):
    w = rtamd.World()
    m = [w.Lambertian(w.ConstantTexture(tuple(rng.random(3)))) for _ in range(3)]
    items = []
    for _ in range(60):
        c = (rng.random(3) - 0.5) * 20.0 * scale
        items.append(w.Sphere(tuple(c), float(rng.uniform(0.05, 1.5) * scale), m[rng.integers(3)]))
    for _ in range(12):
        a0, b0 = (rng.random(2) - 0.5) * 20.0 * scale
        a1, b1 = a0 + rng.uniform(0.5, 6.0) * scale, b0 + rng.uniform(0.5, 6.0) * scale
        k = float((rng.random() - 0.5) * 20.0 * scale)
        ctor = [w.XYRectangle, w.XZRectangle, w.YZRectangle][rng.integers(3)]
        items.append(ctor((float(a0), float(b0)), (float(a1), float(b1)), k, m[rng.integers(3)]))
    for _ in range(4):
        lo = (rng.random(3) - 0.5) * 16.0 * scale
        items.append(w.Cube(tuple(lo), tuple(lo + rng.uniform(0.3, 2.0, 3) * scale), m[rng.integers(3)]))
    for _ in range(3):
        mesh = w.Mesh(P, N, I, m[rng.integers(3)], bvh_seed=int(rng.integers(1 << 30)))
        items.append(w.Transform(tuple(rng.uniform(-180, 180, 3)), tuple(rng.uniform(0.2, 2.0, 3) * scale),
                                 tuple((rng.random(3) - 0.5) * 16.0 * scale), mesh))
    w.new(items, bvh_seed=int(rng.integers(1 << 30)))
    n = 6000
    o = (rng.random((n, 3)) - 0.5) * 30.0 * scale
    o[: n // 4] = (rng.random((n // 4, 3)) - 0.5) * 4.0 * scale
    d = rng.normal(size=(n, 3))
    d[::7, rng.integers(3)] = 0.0
    d[::11] *= 1e-6
    d[::13] *= 1e6
    rays = np.concatenate([o, d], axis=1)
    a = w.debug_hit(rays, t_min=1e-3, kernel=1)
    b = w.debug_hit(rays, t_min=1e-3, kernel=2)
    bad = np.argwhere((a != b).any(axis=1))[:, 0]
    print("trial", trial, "bad", len(bad))
    for i in bad[:4]:
        print("ray", rays[i]); print(" k1", a[i]); print(" k2", b[i])
