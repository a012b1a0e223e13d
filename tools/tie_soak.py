"""One-off soak of the exact-tie rule: random scenes of axis-aligned cubes on an integer lattice -- shared faces, stacked and
nested cubes, rectangles lying on cube faces -- and rays that start inside and outside the cubes, so that a large share of the closest
hits is an EXACT tie between two or three objects.  Closest-hit records of the accel walks (kernel 2's global node form and its LDS node
table, with the flag + reference-order re-walk) against the reference-order walk (kernel 1), all 12 fields incl. the winning leaf's program
index, bit for bit.  64 top-level objects per scene (a power of two: BVHNode::new then emits no object twice, so the index is comparable).
usage: python tools/tie_soak.py [scenes] [rays]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import numpy as np, rtamd
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
total_bad = 0
for sc in range(scenes):
    rng = np.random.default_rng(4000 + sc)
    w = rtamd.World()
    m = w.Lambertian(w.ConstantTexture((0.5, 0.5, 0.5)))
    items = []
    for _ in range(48):
        lo = rng.integers(0, 6, 3)
        ext = rng.integers(1, 3, 3)
        items.append(w.Cube(tuple(float(v) for v in lo), tuple(float(v) for v in lo + ext), m))
    for _ in range(16):                       # rectangles on lattice planes: coplanar with cube faces
        axis = int(rng.integers(0, 3))
        a0, b0 = rng.integers(0, 5, 2)
        a1, b1 = a0 + int(rng.integers(1, 4)), b0 + int(rng.integers(1, 4))
        k = float(rng.integers(0, 8))
        ctor = (w.YZRectangle, w.XZRectangle, w.XYRectangle)[axis]
        items.append(ctor((float(a0), float(b0)), (float(a1), float(b1)), k, m))
    order = rng.permutation(len(items))
    as_list = sc % 4 == 3
    if as_list:
        w.set_root(w.HitableList([items[i] for i in order]))
        w.commit()
    else:
        w.new([items[i] for i in order], bvh_seed=int(sc + 1))
    o = np.concatenate([rng.uniform(-0.5, 8.5, (n_rays // 2, 3)), rng.uniform(-6.0, 14.0, (n_rays - n_rays // 2, 3))])
    d = rng.normal(size=(n_rays, 3))
    d[: n_rays // 10] = np.round(d[: n_rays // 10] * 2.0) / 2.0 + 0.25      # some directions with equal / simple components
    rays = np.concatenate([o, d], axis=1)
    ref = w.debug_hit(rays, kernel=1)
    bad = 0
    for k in (2, 3):
        got = w.debug_hit(rays, kernel=k)
        bad += int((~((got == ref) | (np.isnan(got) & np.isnan(ref))).all(axis=1)).sum())
    # how many of these closest hits are ties at all?  (a second hit of another object at the same t: count through t_max = t)
    hit = ref[:, 0] > 0
    total_bad += bad
    print("scene %2d (%s root): hit share %.3f, %d of %d rays differ between the accel walks and the reference-order walk" %
          (sc, "list" if as_list else "BVHNode::new", hit.mean(), bad, n_rays), flush=True)
print("TOTAL differing rays: %d" % total_bad)
sys.exit(1 if total_bad else 0)
