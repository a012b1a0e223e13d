"""One-off stress of the conservative f32 box tests: closest hits of kernel 2 (SAH BVH2, f32 boxes; both its global-memory node form, box32,
and its LDS form, box32w) against kernel 1 (the
reference-order walk with exact f64 boxes) on random scenes at several scales, with axis-parallel, tiny and huge direction
components and origins inside / far outside the scene.  usage: python tools/fuzz_boxes.py [trials] [rays]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, rtamd
from rtamd import shapes
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
total_bad = 0
for trial in range(trials):
    rng = np.random.default_rng(1000 + trial)
    scale = float(10.0 ** rng.uniform(-3, 4))
    w = rtamd.World()
    m = [w.Lambertian(w.ConstantTexture(tuple(rng.random(3)))) for _ in range(3)]
    items = []
    for _ in range(int(rng.integers(20, 400))):
        c = (rng.random(3) - 0.5) * 20.0 * scale
        items.append(w.Sphere(tuple(c), float(rng.uniform(0.02, 1.5) * scale), m[rng.integers(3)]))
    for _ in range(12):
        a0, b0 = (rng.random(2) - 0.5) * 20.0 * scale
        a1, b1 = a0 + rng.uniform(0.5, 6.0) * scale, b0 + rng.uniform(0.5, 6.0) * scale
        ctor = [w.XYRectangle, w.XZRectangle, w.YZRectangle][rng.integers(3)]
        items.append(ctor((float(a0), float(b0)), (float(a1), float(b1)), float((rng.random() - 0.5) * 20.0 * scale), m[rng.integers(3)]))
    small = trial % 2 == 1   # every other trial: meshes small enough for the LDS node table
    for k in range(int(rng.integers(1, 4))):
        P, N, I = shapes.torus(int(rng.integers(4, 9)), int(rng.integers(6, 12))) if small else shapes.torus(int(rng.integers(6, 30)), int(rng.integers(8, 40)))
        mesh = w.Mesh(P, N, I, m[rng.integers(3)], bvh_seed=int(rng.integers(1 << 30)))
        items.append(w.Transform(tuple(rng.uniform(-180, 180, 3)), tuple(rng.uniform(0.2, 3.0, 3) * scale), tuple((rng.random(3) - 0.5) * 16.0 * scale), mesh))
    w.new(items, bvh_seed=int(rng.integers(1 << 30)))
    o = (rng.random((n, 3)) - 0.5) * 30.0 * scale
    o[: n // 4] = (rng.random((n // 4, 3)) - 0.5) * 4.0 * scale
    o[n // 4: n // 3] *= 20.0                                   # far outside (still within 64x the extent)
    d = rng.normal(size=(n, 3))
    d[::7, rng.integers(3)] = 0.0
    d[::11] *= 1e-6
    d[::13] *= 1e6
    d[::17, rng.integers(3)] *= 1e-12
    d[::19, rng.integers(3)] = -0.0
    rays = np.concatenate([o, d], axis=1)
    a = w.debug_hit(rays, t_min=1e-3, kernel=1)
    b = w.debug_hit(rays, t_min=1e-3, kernel=2)
    try:
        c = w.debug_hit(rays, t_min=1e-3, kernel=3)   # pt_kernel's LDS node table (NodeW, box32w: no widening factor since round 3)
    except rtamd.RtError:
        c = b                                          # (the table of this scene does not fit in LDS: pt_kernel would not use it either)
    bad = np.argwhere((a != b).any(axis=1) | (a != c).any(axis=1))[:, 0]
    total_bad += len(bad)
    print("trial %d scale %.3g items %d%s: hit share %.3f, %d of %d rays differ" % (trial, scale, len(items), "" if c is not b else " (no NodeW)", a[:, 0].mean(), len(bad), n), flush=True)
    for i in bad[:3]:
        print("   ray", rays[i], "k1", a[i], "k2", b[i])
print("TOTAL differing rays:", total_bad)
sys.exit(1 if total_bad else 0)
