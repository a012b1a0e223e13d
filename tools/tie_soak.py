"""One-off soak of the exact-tie rule: random scenes of axis-aligned cubes on an integer lattice -- shared faces, stacked and
nested cubes, rectangles lying on cube faces -- and rays that start inside and outside the cubes, so that a large share of the closest
hits is an EXACT tie between two or three objects.  Closest-hit records of the accel walks (kernel 2's global node form and its LDS node
table, with the flag + reference-order re-walk) against the reference-order walk (kernel 1), all 12 fields incl. the winning leaf's program
index, bit for bit.  64 top-level objects per scene (a power of two: BVHNode::new then emits no object twice, so the index is comparable).
With `inst` (round 5) every fifth cube sits under a Transform (a translation by lattice steps) and eight of the 64 objects are MESH instances on the same lattice -- closed boxes and flat sheets of >= 128 triangles under
translations, their faces coplanar with cube faces and rectangles -- and the walks of the instance service (rt_debug_hit_device 5 / 6: world-
space walk with the instances deferred + their object-space walks over the Node2 / compact NodeQ records, tie bit, re-walk) are compared as
well; every scene is also RENDERED from inside the lattice (glass cubes, depth 12) with kernels 5 and 6 against kernel 1: pt_kernel_coop's /
pt_kernel_wf's own park / serve / adopt code on paths that tie at every bounce.
usage: python tools/tie_soak.py [scenes] [rays] [inst]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import numpy as np, rtamd
from rtamd import shapes
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
inst = len(sys.argv) > 3 and sys.argv[3] == "inst"
total_bad = 0
total_img = 0
for sc in range(scenes):
    rng = np.random.default_rng(4000 + sc)
    w = rtamd.World()
    m = w.Lambertian(w.ConstantTexture((0.5, 0.5, 0.5)))
    glass = w.Dielectric(1.5, w.ConstantTexture((1.0, 1.0, 1.0)))
    lamp = w.DiffuseLight(w.ConstantTexture((3.0, 2.5, 2.0)))
    items = []
    for k in range(48 - (8 if inst else 0)):
        lo = rng.integers(0, 6, 3)
        ext = rng.integers(1, 3, 3)
        mat = (glass if k % 3 == 0 else lamp if k % 7 == 1 else m) if inst else m
        if inst and k % 5 == 4:   # the same cube under a Transform (a translation by lattice steps): its hits come out of the object-space ray, its box is the transformed one
            off = rng.integers(-3, 4, 3)
            items.append(w.Transform((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), tuple(float(v) for v in off),
                                     w.Cube(tuple(float(v) for v in lo - off), tuple(float(v) for v in lo + ext - off), mat)))
        else:
            items.append(w.Cube(tuple(float(v) for v in lo), tuple(float(v) for v in lo + ext), mat))
    for k in range(8 if inst else 0):          # mesh instances on the lattice: boxes (closed) and sheets (flat), translated by integers
        lo = rng.integers(0, 6, 3)
        ext = rng.integers(1, 3, 3)
        mesh = shapes.box_mesh(4, tuple(float(v) for v in ext)) if k % 2 == 0 else shapes.sheet(8, (float(ext[0]), float(ext[2])))
        items.append(w.Transform((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), tuple(float(v) for v in lo), w.Mesh(*mesh, glass if k % 4 == 0 else m, bvh_seed=sc + k)))
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
    for k in (2, 3) + ((5, 6) if inst else ()):
        got = w.debug_hit(rays, kernel=k)
        nb = int((~((got == ref) | (np.isnan(got) & np.isnan(ref))).all(axis=1)).sum())
        if nb and inst:
            print("   kernel %d: %d records differ" % (k, nb), flush=True)
        bad += nb
    img_bad = 0
    if inst:   # the real kernels, from inside the lattice
        cam = rtamd.Camera(((3.3, 3.6, 2.9), (5.0, 2.0, 6.0)), (0.0, 1.0, 0.0), 100.0, 1.0, 0.3, 3.0)
        a, _ = w.render(cam, width=128, height=128, spp=8, seed=sc, max_depth=12, kernel=1)
        for k in (2, 5, 6):
            b, st = w.render(cam, width=128, height=128, spp=8, seed=sc, max_depth=12, kernel=k)
            assert st["kernel_used"] == k
            img_bad += int((~((a == b) | (np.isnan(a) & np.isnan(b))).all(axis=2)).sum())
        total_img += img_bad
    # how many of these closest hits are ties at all?  (a second hit of another object at the same t: count through t_max = t)
    hit = ref[:, 0] > 0
    total_bad += bad
    print("scene %2d (%s root): hit share %.3f, %d of %d rays differ between the accel walks and the reference-order walk%s" %
          (sc, "list" if as_list else "BVHNode::new", hit.mean(), bad, n_rays,
           "; %d of 16384 pixels differ between kernels 2 / 5 / 6 and kernel 1" % img_bad if inst else ""), flush=True)
print("TOTAL differing rays: %d%s" % (total_bad, "; differing pixels: %d" % total_img if inst else ""))
sys.exit(1 if total_bad or total_img else 0)
