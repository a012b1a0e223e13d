"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bar: BIT-EXACT f64 radiance (integer RNG, same operation order, no FMA contraction)."""
import os

import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


def _pair(name, aspect=None):
    import oracle
    import rtamd
    world, cam = rtamd.load_scene_file(scene_path(name))
    if aspect is not None:
        cam = cam.with_aspect(aspect)
    ref = oracle.load_scene_file(scene_path(name), aspect=aspect)
    return world, cam, ref


def _assert_same(img, ref, what):
    if not np.array_equal(img, ref, equal_nan=True):
        bad = (img != ref).any(axis=2)
        idx = np.argwhere(bad)[:5]
        raise AssertionError("%s: %d / %d pixels differ, first at %s: hip=%s oracle=%s" % (
            what, int(bad.sum()), bad.size, idx.tolist(), img[tuple(idx[0])], ref[tuple(idx[0])]))


def test_device_present():
    import rtamd
    assert rtamd.device_count() >= 1


def test_rng_device_matches_oracle_and_host():
    import oracle
    import rtamd
    for key in [(1, 0, 0), (1, 1439999, 999), (0xDEADBEEFCAFEF00D, 2**40 + 7, 12345)]:
        dev = rtamd.debug_rng(*key, 64, device=True)
        host = rtamd.debug_rng(*key, 64, device=False)
        ref = oracle.rng_u64(*key, 64)
        assert dev == ref == host
        for lo, hi in [(-1.0, 1.0), (0.0, 1.0), (0.0, 7.25)]:   # the float conversions of the spec, computed by the device code
            g, r = rtamd.debug_rng_floats(*key, 64, lo, hi, device=True)
            assert g == oracle.rng_f64(*key, 64) and r == oracle.rng_range(*key, 64, lo, hi)
            assert (g, r) == rtamd.debug_rng_floats(*key, 64, lo, hi, device=False)


def test_device_sqrt_and_divide_are_correctly_rounded():
    """bit-exact CPU<->GPU needs IEEE-correct f64 sqrt and divide on the device."""
    import rtamd
    rng = np.random.default_rng(7)
    a = np.concatenate([rng.random(200000) * 10.0 ** rng.integers(-30, 30, 200000), [0.0, 1.0, 2.0, 4.0, 1e-310, 5e-324, np.inf]])
    b = np.concatenate([(rng.random(200000) - 0.5) * 10.0 ** rng.integers(-30, 30, 200000), [1.0, 3.0, -7.0, 0.1, 1e300, 3.0, 2.0]])
    assert np.array_equal(rtamd.debug_math(0, a), np.sqrt(a))
    with np.errstate(all="ignore"):
        assert np.array_equal(rtamd.debug_math(1, a, b), a / b)


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("name,w,h,spp,aspect", [
    ("scene_10.json", 64, 36, 16, 16.0 / 9.0),      # C1 at reduced size
    ("scene_10.yaml", 40, 24, 4, None),
    ("scene_200_no_bvh.json", 48, 48, 4, None),     # nested lists, no BVH
    ("scene_500.json", 96, 96, 8, None),            # C2 at reduced size
    ("scene_500.json", 37, 21, 3, None),            # ragged: not a multiple of the 8x8 tile
])
def test_render_bit_exact_vs_oracle(name, w, h, spp, aspect, kernel):
    world, cam, ref = _pair(name, aspect)
    img, st = world.render(cam, width=w, height=h, spp=spp, seed=1, kernel=kernel)
    exp, _ = ref.render(w, h, spp, seed=1)
    _assert_same(img, exp, "%s %dx%dx%d kernel %d" % (name, w, h, spp, kernel))
    assert st["samples"] == w * h * spp and st["kernel_used"] == kernel


def test_auto_kernel_is_the_accel_kernel():
    world, cam, _ = _pair("scene_500.json")
    assert world.info()["accel_ok"] == 1
    _, st = world.render(cam, width=16, height=16, spp=1)
    assert st["kernel_used"] == 2 and st["scene_in_lds"] == 1


def test_seed_changes_image_and_is_reproducible():
    world, cam, ref = _pair("scene_10.json", 16.0 / 9.0)
    a, _ = world.render(cam, width=32, height=18, spp=4, seed=1)
    b, _ = world.render(cam, width=32, height=18, spp=4, seed=1)
    c, _ = world.render(cam, width=32, height=18, spp=4, seed=2)
    assert np.array_equal(a, b)
    assert not np.array_equal(a, c)
    exp, _ = ref.render(32, 18, 4, seed=2)
    _assert_same(c, exp, "seed 2")


def test_chunking_and_partition_invariance():
    """the image must not depend on spp chunking or on how tiles are dealt to ranks."""
    world, cam, _ = _pair("scene_500.json")
    w, h, spp = 64, 48, 12
    full, _ = world.render(cam, width=w, height=h, spp=spp, seed=3)
    chunked, _ = world.render(cam, width=w, height=h, spp=spp, seed=3, spp_chunk=5)
    assert np.array_equal(full, chunked)
    for world_size in (2, 3, 8):
        acc = np.zeros_like(full)
        for r in range(world_size):
            part, st = world.render(cam, width=w, height=h, spp=spp, seed=3, rank=r, world=world_size)
            assert np.count_nonzero(acc[part != 0]) == 0  # disjoint ownership
            acc += part
        assert np.array_equal(acc, full), "partition over %d ranks changed the image" % world_size


def test_depth_limit_semantics():
    """Q12: depth is tested after the hit and before emission: max_depth hits contribute."""
    world, cam, ref = _pair("scene_500.json")
    for depth in (0, 1, 2, 5):
        img, _ = world.render(cam, width=32, height=32, spp=4, seed=1, max_depth=depth, kernel=1 + depth % 2)
        exp, _ = ref.render(32, 32, 4, max_depth=depth, seed=1)
        _assert_same(img, exp, "max_depth=%d" % depth)
    z, _ = world.render(cam, width=16, height=16, spp=2, seed=1, max_depth=0)
    assert not z.any()


@pytest.mark.parametrize("kernel", [1, 2])
def test_cornell_box_bit_exact(kernel):
    """C3 geometry: rects, cube, transform(mesh), glass + mirror spheres, rect light (scene.rs:16-112)."""
    import oracle
    import rtamd
    cube = scene_path("cube.obj")
    world, cam = rtamd.select_scene(cube, aspect_ratio=1.0, bvh_seed=1)
    ref = oracle.cornell_box_scene(cube, 1.0, seed=1)
    img, _ = world.render(cam, width=64, height=64, spp=8, seed=1, kernel=kernel)
    exp, _ = ref.render(64, 64, 8, seed=1)
    _assert_same(img, exp, "cornell 64x64x8 kernel %d" % kernel)
    assert img.max() > 0


@pytest.mark.parametrize("kernel", [1, 2, 3])
def test_first_hit_records_match_oracle(kernel):
    """World::hit on explicit rays: t, p, normal, front_face identical to the oracle's HitRecord."""
    import oracle
    import rtamd
    cube = scene_path("cube.obj")
    cases = [(rtamd.select_scene(cube, 1.0, 1)[0], oracle.cornell_box_scene(cube, 1.0, 1), (278.0, 278.0, -800.0), 555.0),
             (rtamd.load_scene_file(scene_path("scene_500.json"))[0], oracle.load_scene_file(scene_path("scene_500.json")), (-6.0, 2.0, -6.0), 8.0)]
    rng = np.random.default_rng(11)
    for world, ref, origin, scale in cases:
        n = 2000
        rays = np.zeros((n, 6))
        rays[:, :3] = origin
        target = rng.random((n, 3)) * scale if scale > 100 else (rng.random((n, 3)) - 0.5) * scale
        rays[:, 3:] = target - rays[:, :3]
        out = world.debug_hit(rays, t_min=1e-3, kernel=kernel)
        nhit = 0
        for i in range(n):
            h = ref.hit(rays[i, :3], rays[i, 3:], t_min=1e-3)
            assert (h is not None) == bool(out[i, 0]), "ray %d hit/miss mismatch" % i
            if h is None:
                continue
            nhit += 1
            assert out[i, 1] == h["t"]
            assert np.array_equal(out[i, 2:5], h["p"])
            assert np.array_equal(out[i, 5:8], h["normal"])
            assert bool(out[i, 8]) == h["front_face"]
        assert nhit > n // 4
        assert np.array_equal(out, world.debug_hit(rays, t_min=1e-3, kernel=1))   # incl. the winning leaf's reference-order index


def test_tonemap_and_png_roundtrip(tmp_path):
    import oracle
    import rtamd
    from PIL import Image
    world, cam, _ = _pair("scene_10.json", 16.0 / 9.0)
    img, _ = world.render(cam, width=64, height=36, spp=4, seed=1)
    u8 = rtamd.tonemap_u8(img)
    assert np.array_equal(u8, oracle.tonemap_u8(img))
    p = str(tmp_path / "out.png")
    rtamd.write_png(p, u8)
    assert np.array_equal(np.asarray(Image.open(p).convert("RGB")), u8)
    assert np.array_equal(cam.capture_image(world, width=64, height=36, sample_per_pixel=4, seed=1), u8)


def test_kernels_agree_on_random_sphere_soups_with_ties():
    """kernel 1 (reference order) vs kernel 2 (accel) on scenes built to provoke the tie rule: duplicated and
    touching spheres, shared centres, a 1-object BVHNode::new (Q14) and rays aimed exactly at sphere centres."""
    import rtamd
    rng = np.random.default_rng(5)
    w = rtamd.World()
    mats = [w.Lambertian(w.ConstantTexture(tuple(rng.random(3)))) for _ in range(4)]
    ids = []
    centres = rng.integers(-4, 5, size=(40, 3)).astype(float)
    for i, c in enumerate(centres):
        ids.append(w.Sphere(tuple(c), 0.5, mats[i % 4]))
        if i % 3 == 0:
            ids.append(w.Sphere(tuple(c), 0.5, mats[(i + 1) % 4]))        # exact duplicate: later one must win
        if i % 5 == 0:
            ids.append(w.Sphere(tuple(c + (1.0, 0, 0)), 0.5, mats[2]))     # touching neighbour: tangent-point ties
    single = w.BVHNode_new([ids[0]], bvh_seed=1)
    w.new(ids + [single], bvh_seed=9)
    n = 4000
    rays = np.zeros((n, 6))
    rays[:, :3] = (0.0, 0.0, -20.0)
    target = centres[rng.integers(0, len(centres), n)] + np.where(rng.random((n, 1)) < 0.5, 0.0, rng.normal(0, 0.4, (n, 3)))
    rays[:, 3:] = target - rays[:, :3]
    a = w.debug_hit(rays, kernel=1)
    b = w.debug_hit(rays, kernel=2)
    assert np.array_equal(a, b)
    assert np.array_equal(a, w.debug_hit(rays, kernel=3))
    assert a[:, 0].sum() > n // 2
    cam = rtamd.Camera(((0, 0, -20), (0, 0, 0)), (0, 1, 0), 40, 1.0, 0.0, 20.0)
    i1, _ = w.render(cam, width=48, height=48, spp=4, kernel=1)
    i2, _ = w.render(cam, width=48, height=48, spp=4, kernel=2)
    assert np.array_equal(i1, i2)


@pytest.mark.parametrize("kernel", [1, 2])
def test_image_texture_and_every_material_bit_exact(kernel):
    """ImageTexture (material.rs:70-84) on a sphere (uv by acos/atan2, sphere.rs:16-20) and on rectangles (uv by
    position), CheckerTexture, fuzzy Metal, Dielectric, a sphere light, a rotated + non-uniformly scaled Transform
    of a mesh and a Cube: every device shading branch in one scene, compared with the oracle."""
    import oracle
    import rtamd
    rng = np.random.default_rng(42)
    img = rng.integers(0, 256, size=(16, 32, 3), dtype=np.uint8)
    P, N, I = oracle.load_obj(scene_path("cube.obj"))

    def build(B, mesh_fn):
        it = B.Lambertian(B.ImageTexture(img))
        ch = B.Lambertian(B.CheckerTexture(B.ConstantTexture((0.2, 0.3, 0.1)), B.ConstantTexture((0.9, 0.9, 0.9))))
        items = [
            B.XZRectangle((-8.0, -8.0), (8.0, 8.0), 0.0, ch),
            B.XYRectangle((-8.0, 0.0), (8.0, 8.0), 6.0, it),
            B.YZRectangle((0.0, -8.0), (8.0, 8.0), -7.0, B.Metal(B.ConstantTexture((0.8, 0.7, 0.6)), 0.3)),
            B.Sphere((-2.0, 1.0, 0.0), 1.0, it),
            B.Sphere((0.5, 1.0, -1.0), 1.0, B.Dielectric(1.5, B.ConstantTexture((0.95, 0.95, 1.0)))),
            B.Sphere((0.5, 1.0, -1.0), -0.8 if False else 0.8, B.Dielectric(1.0 / 1.5, B.ConstantTexture((1.0, 1.0, 1.0)))),
            B.Sphere((3.0, 4.0, 1.0), 0.7, B.DiffuseLight(B.ConstantTexture((9.0, 8.0, 7.0)))),
            B.Cube((-5.0, 0.0, -3.0), (-3.5, 1.5, -1.5), B.Lambertian(B.ConstantTexture((0.3, 0.4, 0.8)))),
            B.Transform((20.0, 35.0, 10.0), (0.7, 1.1, 0.5), (2.5, 1.2, 2.0), mesh_fn(B)),
        ]
        return items

    w = rtamd.World()
    w.new(build(w, lambda B: B.Mesh(P, N, I, B.Lambertian(B.ConstantTexture((0.7, 0.7, 0.2))), bvh_seed=4)), bvh_seed=2)
    o = oracle.Scene()
    o.World(build(o, lambda B: B.Mesh(P, N, I, B.Lambertian(B.ConstantTexture((0.7, 0.7, 0.2))), 4)), 2)
    cam = rtamd.Camera(((0.0, 3.0, -10.0), (0.0, 1.0, 0.0)), (0, 1, 0), 45.0, 1.5, 0.05, 10.0)
    o.Camera((0.0, 3.0, -10.0), (0.0, 1.0, 0.0), (0, 1, 0), 45.0, 1.5, 0.05, 10.0)
    got, _ = w.render(cam, width=96, height=64, spp=8, seed=11, kernel=kernel)
    exp, _ = o.render(96, 64, 8, seed=11)
    _assert_same(got, exp, "material zoo, kernel %d" % kernel)
    assert got.max() > 0.5


@pytest.mark.parametrize("scale", [1e-3, 1.0, 250.0, 1e5])
def test_accel_is_conservative_fuzz(scale):
    """Kernel 2's f32 padded boxes must never cull what kernel 1 (f64 boxes, reference order) finds: random scenes of
    spheres, rectangles, cubes and rotated / non-uniformly scaled mesh instances at coordinate scales from 1e-3 to 1e5,
    rays from inside, outside, grazing and axis-parallel; the two traversals must return identical hit records."""
    import oracle
    import rtamd
    rng = np.random.default_rng(int(scale * 1000) % 9973 + 7)
    P, N, I = oracle.load_obj(scene_path("cube.obj"))
    for trial in range(3):
        w = rtamd.World()
        m = [w.Lambertian(w.ConstantTexture(tuple(rng.random(3)))) for _ in range(3)]
        items = []
        spheres = []
        for _ in range(60):
            c = (rng.random(3) - 0.5) * 20.0 * scale
            spheres.append((tuple(c), float(rng.uniform(0.05, 1.5) * scale)))
            items.append(w.Sphere(spheres[-1][0], spheres[-1][1], m[rng.integers(3)]))
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
        assert w.info()["accel_ok"] == 1
        n = 6000
        o = (rng.random((n, 3)) - 0.5) * 30.0 * scale
        o[: n // 4] = (rng.random((n // 4, 3)) - 0.5) * 4.0 * scale            # origins inside the cloud
        d = rng.normal(size=(n, 3))
        d[::7, rng.integers(3)] = 0.0                                           # axis-parallel components
        d[::11] *= 1e-6                                                          # tiny direction vectors
        d[::13] *= 1e6                                                           # huge ones
        rays = np.concatenate([o, d], axis=1)
        a = w.debug_hit(rays, t_min=1e-3, kernel=1)
        for k in (2, 3):   # 3: the LDS node table of pt_kernel (NodeW: planes selected by address, no widening factor)
            b = w.debug_hit(rays, t_min=1e-3, kernel=k)
            bad = np.argwhere((a != b).any(axis=1))
            assert len(bad) == 0, "trial %d: %d rays differ, first %s:\n k1 %s\n k%d %s" % (trial, len(bad), rays[bad[0, 0]], a[bad[0, 0]], k, b[bad[0, 0]])
        assert a[:, 0].sum() > n // 20
        # grazing rays from FAR origins (up to the 64 x extent the boxes are padded for): aimed at the silhouette of every
        # sphere, just inside and just outside; of = fl32(o) and c = fl32(of * iv) are at their coarsest here
        far, tgt = [], []
        for (c, r) in spheres:
            for eps in (1e-3, 1e-6, 1e-9, -1e-9, -1e-6, -1e-3):
                u = rng.normal(size=3)
                u /= np.linalg.norm(u)
                org = np.asarray(c) + u * rng.uniform(100.0, 600.0) * scale
                p = np.cross(u, rng.normal(size=3))
                p /= np.linalg.norm(p)
                far.append(org)
                tgt.append(np.asarray(c) + p * r * (1.0 - eps))
        far, tgt = np.array(far), np.array(tgt)
        rays = np.concatenate([far, tgt - far], axis=1)
        a = w.debug_hit(rays, t_min=1e-3, kernel=1)
        for k in (2, 3):   # 3: the LDS node table of pt_kernel (NodeW: planes selected by address, no widening factor)
            b = w.debug_hit(rays, t_min=1e-3, kernel=k)
            bad = np.argwhere((a != b).any(axis=1))
            assert len(bad) == 0, "trial %d (grazing): %d rays differ, first %s:\n k1 %s\n k%d %s" % (trial, len(bad), rays[bad[0, 0]], a[bad[0, 0]], k, b[bad[0, 0]])
        assert 0.25 < a[:, 0].mean() < 0.95


@pytest.mark.parametrize("kernel", [1, 2, 5, 6])
def test_large_mesh_instance_outside_lds_bit_exact(kernel):
    """C4's shape at test size: the Cornell box with a 6,400-triangle torus instance (rtamd.shapes).  Its tables exceed
    LDS, so this runs the global-memory variants (kernel 2: depth-sorted Node2 array with the top levels cached in LDS,
    leaf-ordered triangle records, TLAS -> object-space BLAS) against the oracle's reference-order recursion."""
    import oracle
    import rtamd
    from rtamd import shapes
    P, N, I = shapes.torus(40, 80)
    w = rtamd.World()
    w.new(shapes.cornell_with_mesh(w, P, N, I), bvh_seed=1)
    o = oracle.Scene()
    o.World(shapes.cornell_with_mesh(o, P, N, I), 1)
    o.Camera((278, 278, -800), (278, 278, 278), (0, 1, 0), 50, 1.0, 0.0, 10.0)
    cam = rtamd.Camera(((278, 278, -800), (278, 278, 278)), (0, 1, 0), 50, 1.0, 0.0, 10.0)
    info = w.info()
    assert info["n_tris"] == 6400 and info["accel_ok"] == 1 and info["accel_instances"] == 1
    img, st = w.render(cam, width=48, height=48, spp=4, seed=1, kernel=kernel)
    exp, _ = o.render(48, 48, 4, seed=1)
    _assert_same(img, exp, "cornell + torus, kernel %d" % kernel)
    assert st["scene_in_lds"] == 0


_C4 = {}


def _c4_scene():
    """BASELINE config C4 at its own mesh size: Cornell box + 102,400-triangle torus instance (built once per session)."""
    if not _C4:
        import oracle
        import rtamd
        from rtamd import shapes
        P, N, I = shapes.torus(160, 320)
        w = rtamd.World()
        w.new(shapes.cornell_with_mesh(w, P, N, I), bvh_seed=1)
        o = oracle.Scene()
        o.World(shapes.cornell_with_mesh(o, P, N, I), 1)
        o.Camera((278, 278, -800), (278, 278, 278), (0, 1, 0), 50, 1.0, 0.0, 10.0)
        cam = rtamd.Camera(((278, 278, -800), (278, 278, 278)), (0, 1, 0), 50, 1.0, 0.0, 10.0)
        exp, _ = o.render(64, 64, 4, seed=1)
        _C4.update(world=w, oracle=o, cam=cam, exp=exp)
    return _C4


@pytest.mark.parametrize("kernel", [0, 1, 2, 5, 6])
def test_c4_full_size_mesh_bit_exact(kernel):
    """Config C4 (Cornell box + ~100k-triangle mesh) at the configured mesh size, every traversal against the oracle:
    64 x 64 x 4 spp of the 102,400-triangle torus instance (mesh.rs:57-137,144-208; transform.rs:152-165)."""
    c = _c4_scene()
    info = c["world"].info()
    assert info["n_tris"] == 102400 and info["accel_ok"] == 1 and info["accel_instances"] == 1
    img, st = c["world"].render(c["cam"], width=64, height=64, spp=4, seed=1, kernel=kernel)
    _assert_same(img, c["exp"], "C4 (102,400 triangles), kernel %d" % kernel)
    assert st["scene_in_lds"] == 0 and img.max() > 0
    if kernel == 0:
        assert st["kernel_used"] == 5   # instance BVH of 68k nodes: the cooperative kernel is the automatic choice


def test_c4_several_instances_of_one_mesh_bit_exact():
    """five Transform instances sharing one mesh (TLAS with several object-space BVHs): kernel 2 == kernel 1 == oracle."""
    import oracle
    import rtamd
    from rtamd import shapes
    P, N, I = shapes.torus(24, 48)
    cam = _c4_scene()["cam"]

    def build(B, mesh_fn):
        white = B.Lambertian(B.ConstantTexture((0.7, 0.7, 0.7)))
        mesh = mesh_fn(B, white)
        items = shapes.cornell_with_mesh(B, P, N, I, scale=70.0, translate=(150.0, 150.0, 200.0))
        items += [B.Transform((10.0 * i, 25.0 * i, 5.0), (40.0, 55.0, 40.0), (120.0 + 90.0 * i, 330.0, 300.0 + 40.0 * i), mesh) for i in range(4)]
        return items
    w = rtamd.World()
    w.new(build(w, lambda B, m: B.Mesh(P, N, I, m, bvh_seed=2)), bvh_seed=3)
    o = oracle.Scene()
    o.World(build(o, lambda B, m: B.Mesh(P, N, I, m, 2)), 3)
    o.Camera((278, 278, -800), (278, 278, 278), (0, 1, 0), 50, 1.0, 0.0, 10.0)
    assert w.info()["accel_instances"] == 5
    exp, _ = o.render(96, 96, 4, seed=5)
    for k in (1, 2, 5, 6):
        img, _ = w.render(cam, width=96, height=96, spp=4, seed=5, kernel=k)
        _assert_same(img, exp, "five instances, kernel %d" % k)


def test_c4_first_hits_agree_between_traversals():
    """closest-hit records (t, point, normal, winning leaf) of primary and random secondary rays through the 102,400-triangle
    instance: kernels 2 and 4 against kernel 1's reference-order walk."""
    c = _c4_scene()
    rng = np.random.default_rng(77)
    n = 20000
    o = np.empty((n, 3))
    o[: n // 2] = (278.0, 278.0, -800.0)
    o[n // 2:] = rng.uniform(20.0, 535.0, size=(n - n // 2, 3))
    tgt = np.array([278.0, 200.0, 278.0]) + rng.normal(0, 90.0, size=(n, 3))
    rays = np.concatenate([o, tgt - o], axis=1)
    a = c["world"].debug_hit(rays, kernel=1)
    assert 0.3 < a[:, 0].mean()
    for k in (2,):
        b = c["world"].debug_hit(rays, kernel=k)
        bad = np.argwhere((a != b).any(axis=1))
        assert len(bad) == 0, "kernel %d: %d rays differ, first %s" % (k, len(bad), rays[bad[0, 0]] if len(bad) else None)


def _random_instance_scene(rng, B, meshes, mesh_fn):
    """Cornell walls + light with 1..3 randomly placed, rotated and (non-uniformly) scaled mesh instances."""
    from rtamd import shapes
    P0, N0, I0 = meshes[0]
    items = shapes.cornell_with_mesh(B, P0, N0, I0, scale=float(rng.uniform(60.0, 140.0)), translate=tuple(rng.uniform(150.0, 400.0, 3)),
                                     rotate=tuple(rng.uniform(-180.0, 180.0, 3)))
    grey = B.Lambertian(B.ConstantTexture((0.6, 0.6, 0.6)))
    glass = B.Dielectric(1.5, B.ConstantTexture((1.0, 1.0, 1.0)))
    for k, (P, N, I) in enumerate(meshes[1:]):
        mesh = mesh_fn(B, P, N, I, glass if k == 0 else grey, 7 + k)
        items.append(B.Transform(tuple(rng.uniform(-180.0, 180.0, 3)), tuple(rng.uniform(20.0, 90.0, 3)), tuple(rng.uniform(100.0, 450.0, 3)), mesh))
    return items


def test_coop_kernel_equals_kernel2_on_random_instances():
    """kernel 5 (cooperative instance service: 16-bit grid boxes, f32 triangle records, parked paths) against kernel 2 on
    random placements: rotations, non-uniform scales, overlapping instances, glass (paths re-enter the mesh), 1..3 instances."""
    import rtamd
    from rtamd import shapes
    rng = np.random.default_rng(505)
    cam = _c4_scene()["cam"]
    for trial in range(6):
        meshes = [shapes.torus(int(rng.integers(6, 40)), int(rng.integers(8, 60))) for _ in range(int(rng.integers(1, 4)))]
        state = rng.bit_generator.state
        w = rtamd.World()
        w.new(_random_instance_scene(rng, w, meshes, lambda B, P, N, I, m, sd: B.Mesh(P, N, I, m, bvh_seed=sd)), bvh_seed=trial)
        assert w.info()["accel_compact"] == 1
        a, sa = w.render(cam, width=96, height=96, spp=6, seed=trial, kernel=2)
        b, sb = w.render(cam, width=96, height=96, spp=6, seed=trial, kernel=5)
        assert sb["kernel_used"] == 5 and sa["kernel_used"] == 2
        _assert_same(b, a, "random instances, trial %d" % trial)
        c6, s6 = w.render(cam, width=96, height=96, spp=6, seed=trial, kernel=6)
        assert s6["kernel_used"] == 6
        _assert_same(c6, a, "random instances, trial %d, kernel 6" % trial)
        assert a.max() > 0
        if trial == 0:  # and against the oracle once
            import oracle
            rng.bit_generator.state = state
            o = oracle.Scene()
            o.World(_random_instance_scene(rng, o, meshes, lambda B, P, N, I, m, sd: B.Mesh(P, N, I, m, sd)), trial)
            o.Camera((278, 278, -800), (278, 278, 278), (0, 1, 0), 50, 1.0, 0.0, 10.0)
            exp, _ = o.render(32, 32, 2, seed=9)
            img, _ = w.render(cam, width=32, height=32, spp=2, seed=9, kernel=5)
            _assert_same(img, exp, "random instances against the oracle")


def test_coop_kernel_full_size_equals_kernel2_at_more_samples():
    """C4 at 256 x 256 x 8 spp: half a million paths through the 102,400-triangle instance, kernel 5 == kernel 2 bit for bit."""
    c = _c4_scene()
    a, _ = c["world"].render(c["cam"], width=256, height=256, spp=8, seed=3, kernel=2)
    b, st = c["world"].render(c["cam"], width=256, height=256, spp=8, seed=3, kernel=5)
    assert st["kernel_used"] == 5
    _assert_same(b, a, "C4 256x256x8, kernel 5 against kernel 2")
    b6, st6 = c["world"].render(c["cam"], width=256, height=256, spp=8, seed=3, kernel=6)
    assert st6["kernel_used"] == 6
    _assert_same(b6, a, "C4 256x256x8, kernel 6 against kernel 2")


def test_coop_kernel_image_does_not_depend_on_partition_chunking_or_unit_size(tuning):
    """kernel 5 under the schedule knobs: tile partition over ranks, spp chunking (several launches), work-unit size, a
    single tile with many units (every path of the image competes for one ticket chain) -- the same image as kernel 2."""
    c = _c4_scene()
    world, cam = c["world"], c["cam"]
    w, h, spp = 72, 56, 12
    full, st = world.render(cam, width=w, height=h, spp=spp, seed=6, kernel=2)
    coop, st5 = world.render(cam, width=w, height=h, spp=spp, seed=6, kernel=5)
    assert st5["kernel_used"] == 5 and st5["launches"] == 1
    _assert_same(coop, full, "kernel 5 against kernel 2")
    for world_size in (2, 3):
        acc = np.zeros_like(full)
        for r in range(world_size):
            part, _ = world.render(cam, width=w, height=h, spp=spp, seed=6, rank=r, world=world_size, kernel=5)
            assert np.count_nonzero(acc[part != 0]) == 0
            acc += part
        assert np.array_equal(acc, full), "kernel 5: partition over %d ranks changed the image" % world_size
    for sub, chunk in ((3, 0), (1, 5), (8, 7)):
        tuning(sub_spp=sub)
        other, st2 = world.render(cam, width=w, height=h, spp=spp, seed=6, spp_chunk=chunk, kernel=5)
        assert st2["launches"] == (1 if chunk == 0 else -(-spp // chunk))
        assert np.array_equal(other, full), (sub, chunk)
    tuning()
    a, _ = world.render(cam, width=8, height=8, spp=400, seed=2, kernel=2)   # one tile, 50 units
    b, _ = world.render(cam, width=8, height=8, spp=400, seed=2, kernel=5)
    _assert_same(b, a, "one tile, 50 units, kernel 5")


def test_coop_kernel_pool_exhaustion_walks_in_the_lane(tuning):
    """with only 32 parked-path slots per workgroup most deferred walks take the in-lane fallback (coop_walk_inline) and the
    rest trickle through the rings: same image."""
    c = _c4_scene()
    exp = c["exp"]
    tuning(coop_pool=32)
    img, st = c["world"].render(c["cam"], width=64, height=64, spp=4, seed=1, kernel=5)
    assert st["kernel_used"] == 5
    _assert_same(img, exp, "C4, kernel 5 with 32 pool slots")
    tuning(coop_pool=1)
    img, _ = c["world"].render(c["cam"], width=64, height=64, spp=4, seed=1, kernel=5)
    _assert_same(img, exp, "C4, kernel 5 with one pool slot")


def test_coop_kernel_needs_f32_mesh_vertices():
    """vertices that are not f32 values (possible through rt_mesh_data / Mesh(...), never through the OBJ loader) have no compact
    copy: automatic selection stays on kernel 2 and asking for kernel 5 is refused."""
    import rtamd
    from rtamd import shapes
    P, N, I = shapes.torus(160, 320)
    P = P * (1.0 + 2.0 ** -40)   # no longer representable in f32
    w = rtamd.World()
    w.new(shapes.cornell_with_mesh(w, P, N, I), bvh_seed=1)
    assert w.info()["accel_compact"] == 0 and w.info()["accel_ok"] == 1
    cam = _c4_scene()["cam"]
    _, st = w.render(cam, width=32, height=32, spp=2, seed=1)
    assert st["kernel_used"] == 2
    with pytest.raises(rtamd.RtError) as e:
        w.render(cam, width=32, height=32, spp=2, seed=1, kernel=5)
    assert e.value.code == -10   # RT_ERR_UNSUPPORTED


@pytest.mark.parametrize("kernel", [1, 2])
def test_many_spheres_outside_lds_bit_exact(kernel):
    """30,000 spheres: the sphere-only kernels with the scene in L2/HBM instead of LDS (variants <LDS=false, GENERAL=false>)."""
    import oracle
    import rtamd
    rng = np.random.default_rng(2024)
    n = 30000
    centers = (rng.random((n, 3)) - 0.5) * np.array([60.0, 4.0, 60.0])
    radii = rng.uniform(0.05, 0.35, n)
    kinds = rng.integers(0, 4, n)

    def build(B):
        mats = [B.Lambertian(B.ConstantTexture((0.6, 0.5, 0.4))), B.Metal(B.ConstantTexture((0.8, 0.8, 0.9)), 0.1),
                B.Dielectric(1.5, B.ConstantTexture((1.0, 1.0, 1.0))), B.DiffuseLight(B.ConstantTexture((3.0, 2.5, 2.0)))]
        return [B.Sphere(tuple(centers[i]), float(radii[i]), mats[kinds[i]]) for i in range(n)]

    w = rtamd.World()
    w.new(build(w), bvh_seed=5)
    o = oracle.Scene()
    o.World(build(o), 5)
    o.Camera((0.0, 6.0, -40.0), (0.0, 0.0, 0.0), (0, 1, 0), 35.0, 1.0, 0.0, 40.0)
    cam = rtamd.Camera(((0.0, 6.0, -40.0), (0.0, 0.0, 0.0)), (0, 1, 0), 35.0, 1.0, 0.0, 40.0)
    img, st = w.render(cam, width=40, height=40, spp=4, seed=2, kernel=kernel)
    exp, _ = o.render(40, 40, 4, seed=2)
    _assert_same(img, exp, "30k spheres, kernel %d" % kernel)
    assert st["scene_in_lds"] == 0 and img.max() > 0


@pytest.mark.parametrize("w,h", [(1, 1), (1, 9), (9, 1), (2, 2)])
def test_degenerate_image_sizes_match_the_oracle(w, h):
    """W-1 / H-1 divisors (camera.rs:97-98, Q2): a 1-pixel-wide image divides by zero -> inf/NaN rays in the reference; the
    kernels must follow without hanging and give the oracle's (black or NaN) pixels."""
    world, cam, ref = _pair("scene_10.json")
    exp, _ = ref.render(w, h, 3, seed=1)
    for kernel in (1, 2):
        img, _ = world.render(cam, width=w, height=h, spp=3, seed=1, kernel=kernel)
        assert np.array_equal(img, exp, equal_nan=True), (kernel, img, exp)


def test_render_workspace_is_small_and_can_be_released():
    """samples live in per-wave unit rings (8 x 12 KB per resident wave) and are folded into the accumulator inside the
    kernel: the device workspace does not grow with spp, stays well below 1 GiB for a 1200 x 1200 frame, and
    rt_release_workspaces gives it back."""
    import rtamd
    world, cam, _ = _pair("scene_500.json")
    rtamd.release_workspaces()
    _, a = world.render(cam, width=1200, height=1200, spp=2, seed=1)
    _, b = world.render(cam, width=1200, height=1200, spp=24, seed=1)
    assert a["workspace_bytes"] == b["workspace_bytes"] and a["launches"] == b["launches"] == 1
    assert b["workspace_bytes"] < 512 * 1024 * 1024 and b["reduce_ms"] == 0.0
    freed = rtamd.release_workspaces()
    assert freed >= b["workspace_bytes"]
    again, _ = world.render(cam, width=64, height=64, spp=2, seed=1)
    assert again.max() > 0


def test_image_does_not_depend_on_the_work_partition_knobs(tuning):
    """work-unit size and launch size only change the schedule (which wave traces which path, which wave folds which unit,
    in how many launches); units of one tile are folded in sample order through the per-tile tickets."""
    world, cam, _ = _pair("scene_500.json")
    w, h, spp = 96, 72, 24
    full, st = world.render(cam, width=w, height=h, spp=spp, seed=9)
    assert st["launches"] == 1
    for sub, chunk in ((3, 0), (1, 7), (8, 24), (5, 11)):
        tuning(sub_spp=sub)
        other, st2 = world.render(cam, width=w, height=h, spp=spp, seed=9, spp_chunk=chunk)
        assert st2["launches"] == (1 if chunk == 0 else -(-spp // chunk))
        assert np.array_equal(other, full), (sub, chunk)


def test_one_tile_many_units_fold_in_order():
    """an 8 x 8 image is ONE tile: its 125 units (1000 spp) are traced by 125 different waves and must be folded strictly in
    sample order through the tile's ticket; compare with the oracle's sequential sum."""
    world, cam, ref = _pair("scene_10.json", aspect=1.0)
    img, st = world.render(cam, width=8, height=8, spp=1000, seed=4)
    exp, _ = ref.render(8, 8, 1000, seed=4)
    _assert_same(img, exp, "one tile, 125 units")


def test_tapered_end_of_launch_schedule_folds_in_sample_order(tuning):
    """every launch ends in rounds of smaller jobs (make_schedule, kernels.hip: single units of sub_spp, sub_spp / 2, sub_spp / 4 samples, at
    most a quarter of the launch) so that the waves run dry together.  The taper changes which wave traces which samples, never the order
    of a pixel's sum: 250 spp (not a multiple of any unit size) against the oracle, then unit sizes, launch chunks and a 3-rank partition
    against that image, on kernel 2 and on kernel 1."""
    world, cam, ref = _pair("scene_10.json", aspect=40 / 24)
    w, h, spp = 40, 24, 250
    full, st = world.render(cam, width=w, height=h, spp=spp, seed=12)
    assert st["launches"] == 1 and st["kernel_used"] == 2
    exp, _ = ref.render(w, h, spp, seed=12)
    _assert_same(full, exp, "tapered schedule, 250 spp")
    for sub, chunk, kernel in ((8, 97, 2), (5, 0, 2), (3, 113, 2), (2, 0, 1), (8, 0, 1)):
        tuning(sub_spp=sub)
        other, st2 = world.render(cam, width=w, height=h, spp=spp, seed=12, spp_chunk=chunk, kernel=kernel)
        assert st2["launches"] == (1 if chunk == 0 else -(-spp // chunk))
        assert np.array_equal(other, full), (sub, chunk, kernel)
    tuning()
    acc = np.zeros_like(full)
    for r in range(3):
        part, _ = world.render(cam, width=w, height=h, spp=spp, seed=12, rank=r, world=3)
        acc += part
    assert np.array_equal(acc, full)


def test_negative_t_min_goes_through_the_reference_order_kernel():
    """box32's conservativeness proof needs t_min >= 0: with a negative t_min (hits behind the origin count, as in the
    reference's World::hit for such a range) kernel 0 resolves to kernel 1 and kernel 2 is refused."""
    import rtamd
    world, cam, ref = _pair("scene_10.json")
    img, st = world.render(cam, width=40, height=24, spp=3, seed=2, t_min=-0.25)
    assert st["kernel_used"] == 1
    exp, _ = ref.render(40, 24, 3, seed=2, t_min=-0.25)
    _assert_same(img, exp, "t_min = -0.25")
    for k in (2,):
        with pytest.raises(rtamd.RtError) as e:
            world.render(cam, width=8, height=8, spp=1, t_min=-0.25, kernel=k)
        assert e.value.code == -10   # RT_ERR_UNSUPPORTED


def test_coordinates_beyond_2_pow_36_use_the_reference_order_kernel():
    """box32 keeps lo*iv and of*iv finite for coordinates below 2^36; larger scenes get no accel (flatten.cpp) and render
    through kernel 1, still bit-exact."""
    import oracle
    import rtamd
    S = 3.0e11                                                      # > 2^36 / 64 (origin_limit = 64 x extent)
    w, o = rtamd.World(), oracle.Scene()
    ids = []
    for sc in (w, o):
        m = sc.Lambertian(sc.ConstantTexture((0.7, 0.6, 0.5)))
        e = sc.DiffuseLight(sc.ConstantTexture((4.0, 4.0, 4.0)))
        items = [sc.Sphere((0.0, -1000.0 * S, 0.0), 1000.0 * S, m), sc.Sphere((0.0, 1.0 * S, 0.0), 1.0 * S, m),
                 sc.Sphere((2.5 * S, 3.0 * S, 1.0 * S), 1.0 * S, e)]
        ids.append(items)
    w.new(ids[0], bvh_seed=1)
    o.World(ids[1], 1)
    assert w.info()["accel_ok"] == 0
    cam = rtamd.Camera(((0, 2 * S, -8 * S), (0, 1 * S, 0)), (0, 1, 0), 40, 1.5, 0.0, 8 * S)
    o.Camera((0, 2 * S, -8 * S), (0, 1 * S, 0), (0, 1, 0), 40, 1.5, 0.0, 8 * S)
    img, st = w.render(cam, width=48, height=32, spp=4, seed=1)
    assert st["kernel_used"] == 1
    exp, _ = o.render(48, 32, 4, seed=1)
    _assert_same(img, exp, "huge coordinates")
    assert img.max() > 0


def test_render_from_the_stored_camera_frame_equals_render_from_the_constructor_arguments():
    """rt_render_camera_frame (a host that owns a constructed Camera, camera.rs:12-21) == rt_render (Camera::new's arguments)."""
    world, cam, _ = _pair("scene_10.json", aspect=1.5)
    a, _ = world.render(cam, width=60, height=40, spp=5, seed=8)
    b, _ = world.render_camera_frame(cam.frame(), width=60, height=40, spp=5, seed=8)
    assert np.array_equal(a, b) and a.max() > 0


_BUNNY = {}


def _bunny_scene():
    """The reference's one real mesh, data/mesh/bun315.obj (4,968 triangles, no `vn` lines), inside the Cornell box under a
    Transform.  Product: rt_object_mesh_obj with synthesize_normals = 1 (the reference would panic at mesh.rs:62); oracle: its
    own OBJ reader + its own restatement of the area-weighted normals (mesh.rs:149-198 for the loading, 57-137 for the hits)."""
    if not _BUNNY:
        import oracle
        import rtamd
        from rtamd import shapes
        obj = scene_path("bun315.obj")
        kw = dict(scale=1700.0, translate=(310.0, -56.0, 300.0), rotate=(0.0, 200.0, 0.0))
        w = rtamd.World()
        w.new(shapes.cornell_with_mesh(w, None, None, None, mesh_fn=lambda B, m: B.Mesh_load_obj(obj, m, synthesize_normals=True, bvh_seed=1), **kw),
              bvh_seed=1)
        P, N, I = oracle.load_obj(obj)
        assert N is None
        Ns = oracle.synthesize_normals(P, I)
        o = oracle.Scene()
        o.World(shapes.cornell_with_mesh(o, None, None, None, mesh_fn=lambda B, m: B.Mesh(P, Ns, I, m, 1), **kw), 1)
        o.Camera((278, 278, -800), (278, 278, 278), (0, 1, 0), 50, 1.0, 0.0, 10.0)
        cam = rtamd.Camera(((278, 278, -800), (278, 278, 278)), (0, 1, 0), 50, 1.0, 0.0, 10.0)
        exp, _ = o.render(72, 72, 4, seed=3)
        _BUNNY.update(world=w, cam=cam, exp=exp)
    return _BUNNY


@pytest.mark.parametrize("kernel", [0, 1, 2, 5, 6])
def test_reference_bunny_obj_in_the_cornell_box_bit_exact(kernel):
    b = _bunny_scene()
    info = b["world"].info()
    assert info["n_tris"] == 4968 and info["accel_ok"] == 1 and info["accel_compact"] == 1
    img, st = b["world"].render(b["cam"], width=72, height=72, spp=4, seed=3, kernel=kernel)
    _assert_same(img, b["exp"], "Cornell + bun315.obj, kernel %d" % kernel)
    if kernel == 0:
        assert st["kernel_used"] == 5
    # the bunny is really in the picture: the image differs from the same box without it
    assert img.max() > 0


def test_mesh_obj_without_normals_is_refused_through_the_c_abi():
    import rtamd
    w = rtamd.World()
    m = w.Lambertian(w.ConstantTexture((1, 1, 1)))
    with pytest.raises(rtamd.RtError) as e:
        w.Mesh_load_obj(scene_path("bun315.obj"), m)
    assert e.value.code == -7   # RT_ERR_NO_NORMALS (mesh.rs:62 would panic)


def test_wavefront_kernel_image_does_not_depend_on_partition_chunking_unit_size_or_segment_size(tuning):
    """kernel 6 (cycles of path-tracing and walk launches, parked paths in HBM) under the schedule knobs: tile partition over
    ranks, spp chunking, work-unit size, one tile with many units, and SMALL pool segments (4 096 records per workgroup: the gate
    that keeps a segment from overflowing closes and opens many times, units are generated across launch boundaries)."""
    c = _c4_scene()
    world, cam = c["world"], c["cam"]
    w, h, spp = 72, 56, 12
    full, _ = world.render(cam, width=w, height=h, spp=spp, seed=6, kernel=2)
    wf, st6 = world.render(cam, width=w, height=h, spp=spp, seed=6, kernel=6)
    assert st6["kernel_used"] == 6 and st6["launches"] >= 2
    _assert_same(wf, full, "kernel 6 against kernel 2")
    for world_size in (2, 3):
        acc = np.zeros_like(full)
        for r in range(world_size):
            part, _ = world.render(cam, width=w, height=h, spp=spp, seed=6, rank=r, world=world_size, kernel=6)
            assert np.count_nonzero(acc[part != 0]) == 0
            acc += part
        assert np.array_equal(acc, full), "kernel 6: partition over %d ranks changed the image" % world_size
    for sub, chunk in ((3, 0), (1, 5), (8, 7)):
        tuning(sub_spp=sub)
        other, _ = world.render(cam, width=w, height=h, spp=spp, seed=6, spp_chunk=chunk, kernel=6)
        assert np.array_equal(other, full), (sub, chunk)
    tuning()
    a, _ = world.render(cam, width=8, height=8, spp=400, seed=2, kernel=2)   # one tile, 50 units
    b, _ = world.render(cam, width=8, height=8, spp=400, seed=2, kernel=6)
    _assert_same(b, a, "one tile, 50 units, kernel 6")
    tuning(coop_pool=4096)
    a, _ = world.render(cam, width=400, height=400, spp=16, seed=4, kernel=2)    # 2.6 M paths: ~10 k per workgroup
    b, st = world.render(cam, width=400, height=400, spp=16, seed=4, kernel=6)
    _assert_same(b, a, "kernel 6 with 4 096-record segments")
    assert st["launches"] > 8


@pytest.mark.parametrize("kernel", [5, 6])
def test_wavefront_kernel_64_instances_bit_exact(kernel):
    """64 Transform instances of three meshes (one pending bit per instance: the limit of kernels 5 and 6; kernel 5 stopped at 32 until
    round 4 and carries the upper half of its mask in the last unit of a parked path's record)."""
    import rtamd
    from rtamd import shapes
    rng = np.random.default_rng(64)
    meshes = [shapes.torus(8, 12), shapes.torus(10, 16), shapes.torus(12, 20)]
    w = rtamd.World()
    white = w.Lambertian(w.ConstantTexture((0.73, 0.73, 0.73)))
    glass = w.Dielectric(1.5, w.ConstantTexture((1.0, 1.0, 1.0)))
    objs = [w.Mesh(P, N, I, white if i else glass, bvh_seed=i + 1) for i, (P, N, I) in enumerate(meshes)]
    items = shapes.cornell_with_mesh(w, *meshes[0], scale=30.0, translate=(100.0, 480.0, 100.0))
    for k in range(63):
        pos = (60.0 + 62.0 * (k % 8), 60.0 + 55.0 * (k // 8), 120.0 + 45.0 * (k % 5))
        items.append(w.Transform((float(rng.uniform(0, 360)), float(rng.uniform(0, 360)), 0.0), (22.0, 30.0, 22.0), pos, objs[k % 3]))
    w.new(items, bvh_seed=9)
    info = w.info()
    assert info["accel_instances"] == 64 and info["accel_compact"] == 1
    cam = _c4_scene()["cam"]
    a, _ = w.render(cam, width=128, height=128, spp=4, seed=8, kernel=2)
    b, st = w.render(cam, width=128, height=128, spp=4, seed=8, kernel=kernel)
    assert st["kernel_used"] == kernel
    _assert_same(b, a, "64 instances, kernel %d against kernel 2" % kernel)
    auto, st0 = w.render(cam, width=128, height=128, spp=4, seed=8)
    assert st0["kernel_used"] == 5      # kernel 5 applies up to 64 instances: kernel 6 runs by request only
    _assert_same(auto, a, "64 instances, automatic kernel")
    k1, _ = w.render(cam, width=48, height=48, spp=2, seed=8, kernel=1)
    k6, _ = w.render(cam, width=48, height=48, spp=2, seed=8, kernel=kernel)
    _assert_same(k6, k1, "64 instances, kernel %d against the reference-order kernel" % kernel)


def test_flat_and_single_triangle_instances_bit_exact():
    """instances that are FLAT in object space (one triangle; a planar two-triangle quad) beside a large mesh instance: their grids
    for the 16-bit node boxes have a thin axis (widened by flatten.cpp), their BVHs are a single leaf under an inner root with a
    zero-size second box -- kernels 5 and 6 must still apply (accel_compact) and agree with kernels 1 / 2 and the oracle."""
    import oracle
    import rtamd
    from rtamd import shapes
    P, N, I = shapes.torus(24, 48)
    tri = (np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]), np.array([[0.0, 0.0, 1.0]] * 3), np.array([[0, 1, 2]], dtype=np.uint32))
    quad = (np.array([[-1.0, 0.0, -1.0], [1.0, 0.0, -1.0], [1.0, 0.0, 1.0], [-1.0, 0.0, 1.0]]), np.array([[0.0, 1.0, 0.0]] * 4), np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32))

    def build(B, mesh):
        white = B.Lambertian(B.ConstantTexture((0.73, 0.73, 0.73)))
        mirror = B.Metal(B.ConstantTexture((0.9, 0.9, 0.9)), 0.0)
        items = shapes.cornell_with_mesh(B, P, N, I, scale=90.0, translate=(300.0, 200.0, 300.0))
        items.append(B.Transform((20.0, 30.0, 0.0), (160.0, 160.0, 160.0), (60.0, 120.0, 150.0), mesh(B, tri, mirror, 5)))
        items.append(B.Transform((35.0, 0.0, 10.0), (90.0, 90.0, 90.0), (420.0, 330.0, 250.0), mesh(B, quad, white, 6)))
        return items

    w = rtamd.World()
    w.new(build(w, lambda B, m, mat, sd: B.Mesh(m[0], m[1], m[2], mat, bvh_seed=sd)), bvh_seed=2)
    o = oracle.Scene()
    o.World(build(o, lambda B, m, mat, sd: B.Mesh(m[0], m[1], m[2], mat, sd)), 2)
    o.Camera((278, 278, -800), (278, 278, 278), (0, 1, 0), 50, 1.0, 0.0, 10.0)
    cam = _c4_scene()["cam"]
    info = w.info()
    assert info["accel_instances"] == 3 and info["accel_compact"] == 1
    exp, _ = o.render(64, 64, 4, seed=7)
    for k in (1, 2, 5, 6, 0):
        img, st = w.render(cam, width=64, height=64, spp=4, seed=7, kernel=k)
        _assert_same(img, exp, "flat instances, kernel %d" % k)
        if k == 0:
            assert st["kernel_used"] == 5
    a, _ = w.render(cam, width=160, height=160, spp=8, seed=9, kernel=2)
    for k in (5, 6):
        b, _ = w.render(cam, width=160, height=160, spp=8, seed=9, kernel=k)
        _assert_same(b, a, "flat instances at 160 x 160 x 8, kernel %d against kernel 2" % k)


def test_mixed_instances_keep_the_instance_service():
    """a scene with a LARGE triangle-mesh instance AND instances kernels 5 / 6 cannot defer (a Cube -- six rectangles -- and a BVH
    of spheres under Transforms; a mesh whose vertices are not f32 values): the service still applies to the mesh (round 2: one such
    instance switched it off for the whole scene), the others are entered in the lane as kernel 2 enters every instance."""
    import oracle
    import rtamd
    from rtamd import shapes
    P, N, I = shapes.torus(40, 80)
    Pq, Nq, Iq = shapes.torus(5, 7)
    Pq = Pq * (1.0 + 2.0 ** -40)            # not representable in f32
    rng = np.random.default_rng(3)
    centres = rng.uniform(-1.0, 1.0, size=(40, 3))

    def build(B, mesh):
        white = B.Lambertian(B.ConstantTexture((0.73, 0.73, 0.73)))
        glass = B.Dielectric(1.5, B.ConstantTexture((1.0, 1.0, 1.0)))
        items = shapes.cornell_with_mesh(B, P, N, I, scale=100.0, translate=(300.0, 220.0, 300.0))
        items.append(B.Transform((0.0, 25.0, 0.0), (1.0, 2.0, 1.0), (90.0, 0.0, 330.0), B.Cube((0.0, 0.0, 0.0), (110.0, 110.0, 110.0), white)))
        items.append(B.Transform((10.0, 40.0, 0.0), (60.0, 60.0, 60.0), (420.0, 400.0, 200.0), B.BVHNode_new([B.Sphere(tuple(c), 0.18, glass if k % 3 == 0 else white) for k, c in enumerate(centres)], 4)))
        items.append(B.Transform((0.0, 0.0, 30.0), (45.0, 45.0, 45.0), (120.0, 420.0, 250.0), mesh(B, (Pq, Nq, Iq), white, 9)))
        return items

    w = rtamd.World()
    w.new(build(w, lambda B, m, mat, sd: B.Mesh(m[0], m[1], m[2], mat, bvh_seed=sd)), bvh_seed=2)
    o = oracle.Scene()
    o.World(build(o, lambda B, m, mat, sd: B.Mesh(m[0], m[1], m[2], mat, sd)), 2)
    o.Camera((278, 278, -800), (278, 278, 278), (0, 1, 0), 50, 1.0, 0.0, 10.0)
    cam = _c4_scene()["cam"]
    info = w.info()
    assert info["accel_instances"] == 4 and info["accel_compact"] == 1
    exp, _ = o.render(72, 72, 4, seed=12)
    for k in (1, 2, 5, 6, 0):
        img, st = w.render(cam, width=72, height=72, spp=4, seed=12, kernel=k)
        _assert_same(img, exp, "mixed instances, kernel %d" % k)
        if k == 0:
            assert st["kernel_used"] == 5
    a, _ = w.render(cam, width=200, height=200, spp=8, seed=5, kernel=2)
    for k in (5, 6):
        b, _ = w.render(cam, width=200, height=200, spp=8, seed=5, kernel=k)
        _assert_same(b, a, "mixed instances at 200 x 200 x 8, kernel %d against kernel 2" % k)


def test_wavefront_kernel_workspace_stays_within_its_budget(tuning):
    """kernel 6 sizes its unit buffers and record pools from the launch and from rt_tuning.wf_workspace_mb (default 1 900 MB; round 3 took
    8.6 GB flat): C4's frame stays below 2 GB, a smaller budget shrinks it further, and the image does not depend on the budget."""
    import rtamd
    c4 = _c4_scene()
    w, cam = c4["world"], c4["cam"]
    rtamd.release_workspaces()
    img, st = w.render(cam, width=1200, height=1200, spp=8, seed=1, kernel=6)      # C4's own frame (its units per wave exceed any ring)
    assert st["kernel_used"] == 6 and st["workspace_bytes"] < 2.0e9, st["workspace_bytes"]
    tuning(wf_workspace_mb=700)
    small, st2 = w.render(cam, width=1200, height=1200, spp=8, seed=1, kernel=6)
    assert st2["workspace_bytes"] < 1.0e9 and np.array_equal(small, img)
    tuning(wf_workspace_mb=8600)
    rtamd.release_workspaces()
    big, st3 = w.render(cam, width=1200, height=1200, spp=8, seed=1, kernel=6)
    assert st3["workspace_bytes"] > st["workspace_bytes"] and np.array_equal(big, img)
    k5, _ = w.render(cam, width=1200, height=1200, spp=8, seed=1, kernel=5)
    assert np.array_equal(k5, img)
    rtamd.release_workspaces()


def test_resumable_render_equals_the_one_call_render(tuning):
    """rt_render_accumulate_device / rt_accum_finalize_device (checkpoint and restart, SURVEY s5 aux): a frame rendered as consecutive sample
    ranges into the caller's accumulator -- cut unevenly, with the accumulator taken to the host and put back between two calls as a restart
    would, on a 3-rank tile partition too, kernels 1 / 2 / 5, both integrators -- equals rt_render_tiles_device's frame bit for bit."""
    import torch
    import rtamd
    from rtamd import shapes
    world, cam = rtamd.select_scene(scene_path("cube.obj"), 1.0, 1)
    mesh_world = rtamd.World()
    P, N, I = shapes.torus(24, 48)
    mesh_world.new(shapes.cornell_with_mesh(mesh_world, P, N, I), bvh_seed=1)
    mcam = rtamd.Camera(((278, 278, -800), (278, 278, 278)), (0, 1, 0), 50, 1.0, 0.0, 10.0)
    cases = [(world, cam, 1, 0, 1, 0), (world, cam, 2, 0, 1, 0), (world, cam, 2, 1, 3, 1), (world, cam, 2, 0, 3, 2), (mesh_world, mcam, 5, 0, 1, 0)]
    for w, c, kernel, integ, wsize, rank in cases:
        p = rtamd.default_params(width=100, height=76, spp=37, seed=9, kernel=kernel, integrator=integ, rank=rank, world=wsize)
        n = rtamd.tiles_owned(p)
        ref = torch.zeros((n, 64, 3), dtype=torch.float64, device="cuda")
        st = w.render_tiles_device(c, p, ref.data_ptr())
        assert st["kernel_used"] == kernel
        acc = torch.full((n, 64, 3), float("nan"), dtype=torch.float64, device="cuda")      # sample_begin == 0 initialises it
        cuts = [0, 5, 6, 20, 37]
        for a, b in zip(cuts[:-1], cuts[1:]):
            st = w.render_accumulate_device(c, p, a, b, acc.data_ptr())
            assert st["kernel_used"] == kernel
            if b == 6:                                                                      # "checkpoint": the state is the accumulator and b
                saved = acc.cpu().clone()
                del acc
                acc = saved.cuda()
        out = torch.empty((n, 64, 3), dtype=torch.float64, device="cuda")
        rtamd.accum_finalize_device(p, acc.data_ptr(), out.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(out, ref), "kernel %d integrator %d rank %d/%d" % (kernel, integ, rank, wsize)
    # refusals: a range outside the frame, the wavefront kernel, SPPM's integrator
    p = rtamd.default_params(width=16, height=16, spp=4, kernel=2)
    acc = torch.zeros((4, 64, 3), dtype=torch.float64, device="cuda")
    for a, b in [(2, 2), (-1, 2), (0, 5)]:
        with pytest.raises(rtamd.RtError):
            world.render_accumulate_device(cam, p, a, b, acc.data_ptr())
    with pytest.raises(rtamd.RtError):
        world.render_accumulate_device(cam, rtamd.default_params(width=16, height=16, spp=4, kernel=6), 0, 4, acc.data_ptr())


def test_resumable_render_with_host_held_state_equals_rt_render():
    """rt_render_accumulate / rt_accum_finalize (the form for a host without device memory, e.g. the reference's Rust host): the accumulator
    state travels through host memory between the calls; 2 + 9 + 5 of 16 samples give rt_render's frame bit for bit."""
    import rtamd
    world, cam = rtamd.load_scene_file(scene_path("scene_10.json"))
    p = rtamd.default_params(width=120, height=68, spp=16, seed=2)
    ref, _ = world.render(cam, width=120, height=68, spp=16, seed=2)
    state = None
    for a, b in [(0, 2), (2, 11), (11, 16)]:
        state, st = world.render_accumulate(cam, p, a, b, state)
        assert st["samples"] == 120 * 68 * (b - a)
        state = state.copy()                      # (what a restart reads back from disk)
    assert np.array_equal(rtamd.accum_finalize(p, state), ref)
    with pytest.raises(rtamd.RtError):
        rtamd.accum_finalize(rtamd.default_params(width=120, height=68, spp=16, seed=2, rank=1, world=2), state)
