"""ConstantMedium + Isotropic (objects/medium.rs:9-57, material.rs:213-231 -- dead / commented-out code in the
reference, so parity is against the oracle's restatement only, D8 in oracle/rt_oracle.cpp).

CPU: the oracle's medium against closed forms (Beer-Lambert transmittance, mean free path, clipping to [t_min, t_max], RNG draw
only when the boundary is crossed), rtamd-ln-1 against numpy's log, host description / flattening of the product.
GPU: HIP == oracle bit for bit on scenes with media (sphere and box boundaries, inside a BVH, under a Transform, next to glass);
the device's det_ln == the oracle's == numpy within 1 ulp."""
import numpy as np
import pytest

from conftest import scene_path


def _ray_through_sphere_medium(density, n, radius=1.0, t_min=1e-3, t_max=float("inf")):
    """n rays along a diameter of a sphere-bounded medium, each with its own RNG stream; returns hit ts (nan = passed through)."""
    import oracle
    o = oracle.Scene()
    iso = o.Isotropic(o.ConstantTexture((1.0, 1.0, 1.0)))
    s = o.Sphere((0.0, 0.0, 0.0), radius, iso)
    med = o.ConstantMedium(density, s, iso)
    ts = np.full(n, np.nan)
    for i in range(n):
        rec = o.hit((0.0, 0.0, -5.0), (0.0, 0.0, 2.0), t_min, t_max, obj=med, key=(7, i, 0))   # |dir| = 2: t is in units of 2
        if rec is not None:
            ts[i] = rec["t"]
    return ts


def test_det_ln_is_within_one_ulp_of_log_and_exact_on_its_special_values():
    import oracle
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.integers(1, 1 << 53, size=50000).astype(np.float64) / 2.0 ** 53,          # gen::<f64>() values
                         np.ldexp(rng.random(20000) + 0.5, rng.integers(-300, 300, 20000)), [2.0 ** -53, 1.0 - 2.0 ** -53, 0.5, 2.0]])
    d = np.array([oracle.det_ln(x) for x in xs])
    ref = np.log(xs)
    assert np.all(np.abs(d - ref) <= np.spacing(np.abs(ref)))
    assert oracle.det_ln(1.0) == 0.0 and oracle.det_ln(0.0) == -np.inf and np.isnan(oracle.det_ln(-1.0))
    assert oracle.det_ln(np.inf) == np.inf and oracle.det_ln(np.exp(1.0)) == pytest.approx(1.0, abs=2e-16)


def test_oracle_medium_obeys_beer_lambert():
    """P(no scattering along a chord of length L) = exp(-d L); scattering depths are exponentially distributed."""
    n = 20000
    for density, radius in ((0.4, 1.0), (1.5, 0.75)):
        ts = _ray_through_sphere_medium(density, n, radius)
        chord = 2.0 * radius
        passed = np.isnan(ts).mean()
        expect = np.exp(-density * chord)
        assert abs(passed - expect) < 4.0 * np.sqrt(expect * (1 - expect) / n), (density, passed, expect)
        depth = (ts[~np.isnan(ts)] - (5.0 - radius) / 2.0) * 2.0            # distance inside the boundary (|dir| = 2)
        assert depth.min() >= 0.0 and depth.max() <= chord
        # mean of an exponential truncated to [0, chord]
        lam = density
        mean_trunc = 1.0 / lam - chord * np.exp(-lam * chord) / (1.0 - np.exp(-lam * chord))
        assert depth.mean() == pytest.approx(mean_trunc, rel=0.03)


def test_oracle_medium_clipping_and_rng_discipline():
    import oracle
    o = oracle.Scene()
    iso = o.Isotropic(o.ConstantTexture((0.9, 0.8, 0.7)))
    s = o.Sphere((0.0, 0.0, 0.0), 1.0, iso)
    med = o.ConstantMedium(1e6, s, iso)                      # so dense that every crossing scatters at once
    inf = float("inf")
    # from outside: scatters right at the entry point t1 = 4 (unit direction); the arbitrary normal (1,0,0) is perpendicular to
    # this ray, so front_face = (dir . n < 0) is False and HitRecord::new flips it (hit.rs:24-33)
    rec = o.hit((0.0, 0.0, -5.0), (0.0, 0.0, 1.0), 1e-3, inf, obj=med, key=(1, 2, 3))
    assert rec["t"] == pytest.approx(4.0, abs=1e-5) and tuple(rec["normal"]) == (-1.0, 0.0, 0.0) and rec["front_face"] is False
    assert rec["uv"] == (0.0, 0.0) and rec["prim_id"] == med
    rec = o.hit((5.0, 0.0, 0.0), (-1.0, 0.0, 0.0), 1e-3, inf, obj=med, key=(1, 2, 3))
    assert tuple(rec["normal"]) == (1.0, 0.0, 0.0) and rec["front_face"] is True
    # from inside: the entry is clipped to t_min, then to 0 (medium.rs:28,33)
    rec = o.hit((0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 1e-3, inf, obj=med, key=(1, 2, 3))
    assert 1e-3 <= rec["t"] < 1e-3 + 1e-5
    # t_max in front of the boundary: None, and NO random number is drawn (medium.rs:30-32 returns before :37)
    assert o.hit((0.0, 0.0, -5.0), (0.0, 0.0, 1.0), 1e-3, 3.5, obj=med, key=(4, 5, 6)) is None and o.last_draws == 0
    assert o.hit((0.0, 3.0, -5.0), (0.0, 0.0, 1.0), 1e-3, inf, obj=med, key=(4, 5, 6)) is None and o.last_draws == 0   # misses the boundary
    assert o.hit((0.0, 0.0, -5.0), (0.0, 0.0, 1.0), 1e-3, inf, obj=med, key=(4, 5, 6)) is not None and o.last_draws == 1
    thin = o.ConstantMedium(1e-9, s, iso)                     # crossed, one draw, but the free flight is longer than the chord
    assert o.hit((0.0, 0.0, -5.0), (0.0, 0.0, 1.0), 1e-3, inf, obj=thin, key=(4, 5, 6)) is None and o.last_draws == 1
    # bounding box = the boundary's
    assert np.array_equal(o.bounding_box(med), o.bounding_box(s))


def test_isotropic_scatter_is_uniform_on_the_sphere():
    import oracle
    o = oracle.Scene()
    iso = o.Isotropic(o.ConstantTexture((0.2, 0.4, 0.6)))
    dirs = []
    for i in range(4000):
        r = o.scatter(iso, (0, 0, 0), (0, 0, 1), (1.0, 2.0, 3.0), (1.0, 0.0, 0.0), True, key=(9, i, 0))
        assert r["kind"] == 1 and r["scattered"]                              # Interaction::Specular (pass-through), D8
        assert tuple(r["attenuation"]) == (0.2, 0.4, 0.6) and tuple(r["orig"]) == (1.0, 2.0, 3.0) and not r["emitted"].any()
        dirs.append(r["dir"])
    dirs = np.array(dirs)
    assert np.allclose(np.linalg.norm(dirs, axis=1), 1.0, atol=1e-12)          # random_in_unit_sphere: ON the sphere (Q3)
    assert np.all(np.abs(dirs.mean(axis=0)) < 0.05) and np.allclose((dirs ** 2).mean(axis=0), 1.0 / 3.0, atol=0.02)


def _smoke_scene(B, seed_arg):
    """book-2 style: a ground, a light, a glass ball, a dense smoke ball inside a BVH, a thin fog box under a Transform."""
    ground = B.Lambertian(B.CheckerTexture(B.ConstantTexture((0.2, 0.3, 0.1)), B.ConstantTexture((0.9, 0.9, 0.9))))
    white = B.Lambertian(B.ConstantTexture((0.73, 0.73, 0.73)))
    light = B.DiffuseLight(B.ConstantTexture((7.0, 7.0, 7.0)))
    smoke = B.Isotropic(B.ConstantTexture((0.1, 0.1, 0.1)))
    fog = B.Isotropic(B.ConstantTexture((0.9, 0.95, 1.0)))
    glass = B.Dielectric(1.5, B.ConstantTexture((1.0, 1.0, 1.0)))
    ball = B.Sphere((0.0, 1.0, 0.0), 1.0, white)
    box = B.Cube((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0), white)
    items = [
        B.XZRectangle((-20.0, -20.0), (20.0, 20.0), 0.0, ground),
        B.XZRectangle((-3.0, -3.0), (3.0, 3.0), 6.0, light),
        B.Sphere((2.6, 1.0, 0.5), 1.0, glass),
        B.ConstantMedium(1.2, ball, smoke),
        B.ConstantMedium(0.15, B.Transform((0.0, 30.0, 0.0), (1.5, 1.0, 1.5), (-3.0, 1.01, 0.5), box), fog),
        B.Sphere((-0.5, 0.4, -2.2), 0.4, B.Metal(B.ConstantTexture((0.8, 0.8, 0.9)), 0.05)),
    ]
    return items


def test_product_describes_and_flattens_media():
    import rtamd
    w = rtamd.World()
    items = _smoke_scene(w, 1)
    w.new(items, bvh_seed=1)
    info = w.info()
    assert info["accel_ok"] == 1                      # world-level media keep the accel (kernel 2's MEDIA variant, round 3)
    kind, d = w.describe(items[3])
    assert kind == "ConstantMedium" and d["v"][0] == 1.2 and len(d["children"]) == 1 and w.describe(d["children"][0])[0] == "Sphere"
    with pytest.raises(rtamd.RtError) as e:
        w2 = rtamd.World()
        m = w2.Isotropic(w2.ConstantTexture((1, 1, 1)))
        inner = w2.ConstantMedium(1.0, w2.Sphere((0, 0, 0), 1, m), m)
        w2.ConstantMedium(1.0, inner, m)
    assert e.value.code == -10
    with pytest.raises(rtamd.RtError) as e:
        w3 = rtamd.World()
        m = w3.Isotropic(w3.ConstantTexture((1, 1, 1)))
        w3.ConstantMedium(0.0, w3.Sphere((0, 0, 0), 1, m), m)
    assert e.value.code == -1


@pytest.mark.gpu
def test_device_det_ln_equals_the_oracles():
    import oracle
    import rtamd
    rng = np.random.default_rng(11)
    xs = np.concatenate([rng.integers(1, 1 << 53, size=200000).astype(np.float64) / 2.0 ** 53, [2.0 ** -53, 1.0 - 2.0 ** -53, 0.5, 0.0]])
    dev = rtamd.debug_math(2, xs)
    ora = np.array([oracle.det_ln(x) for x in xs])
    assert np.array_equal(dev, ora)
    ref = np.log(xs[:-1])
    assert np.all(np.abs(dev[:-1] - ref) <= np.spacing(np.abs(ref)))


@pytest.mark.gpu
def test_hip_media_bit_exact_vs_oracle():
    import oracle
    import rtamd
    w = rtamd.World()
    w.new(_smoke_scene(w, 1), bvh_seed=5)
    o = oracle.Scene()
    o.World(_smoke_scene(o, 1), 5)
    o.Camera((0.0, 2.5, -9.0), (0.0, 1.0, 0.0), (0, 1, 0), 40.0, 1.5, 0.02, 9.0)
    cam = rtamd.Camera(((0.0, 2.5, -9.0), (0.0, 1.0, 0.0)), (0, 1, 0), 40.0, 1.5, 0.02, 9.0)
    img, st = w.render(cam, width=96, height=64, spp=16, seed=3)
    exp, _ = o.render(96, 64, 16, seed=3)
    assert st["kernel_used"] == 2                     # the accel kernel's MEDIA variant is the automatic choice since round 3
    assert np.array_equal(img, exp), np.abs(img - exp).max()
    img1, st1 = w.render(cam, width=96, height=64, spp=16, seed=3, kernel=1)   # the reference-order kernel stays the cross-check
    assert st1["kernel_used"] == 1 and np.array_equal(img1, exp)
    assert img.max() > 0.5
    # the fog really scatters: the same scene without the two media renders differently
    w2 = rtamd.World()
    items = _smoke_scene(w2, 1)
    w2.new([it for i, it in enumerate(items) if i not in (3, 4)], bvh_seed=5)
    clear, _ = w2.render(cam, width=96, height=64, spp=16, seed=3)
    assert not np.array_equal(clear, img)
    # light sampling, kernels 5 / 6 and the closest-hit diagnostic refuse scenes with media
    for kw in (dict(kernel=5), dict(kernel=6), dict(integrator=1)):
        with pytest.raises(rtamd.RtError) as e:
            w.render(cam, width=8, height=8, spp=1, **kw)
        assert e.value.code in (-10, -1)
    with pytest.raises(rtamd.RtError):
        w.debug_hit(np.zeros((1, 6)) + 1.0, kernel=1)


def _fog_zoo(B, order):
    """media that overlap each other and surfaces, a camera INSIDE a fog, a medium whose boundary is a Transform(Cube), media
    early and late in the list (`order` permutes it: the reference's visit order decides which medium draws first)."""
    white = B.Lambertian(B.ConstantTexture((0.8, 0.8, 0.8)))
    red = B.Lambertian(B.ConstantTexture((0.8, 0.2, 0.2)))
    light = B.DiffuseLight(B.ConstantTexture((5.0, 5.0, 5.0)))
    fog_a = B.Isotropic(B.ConstantTexture((0.9, 0.9, 1.0)))
    fog_b = B.Isotropic(B.ConstantTexture((0.3, 0.6, 0.3)))
    glass = B.Dielectric(1.5, B.ConstantTexture((1.0, 1.0, 1.0)))
    mirror = B.Metal(B.ConstantTexture((0.9, 0.9, 0.9)), 0.0)
    items = [
        B.XZRectangle((-30.0, -30.0), (30.0, 30.0), 0.0, white),
        B.XZRectangle((-4.0, -4.0), (4.0, 4.0), 9.0, light),
        B.ConstantMedium(0.05, B.Sphere((0.0, 3.0, -6.0), 7.0, white), fog_a),           # holds the camera
        B.Sphere((0.0, 1.0, 0.0), 1.0, red),                                               # a surface inside the big fog
        B.ConstantMedium(0.9, B.Sphere((0.8, 1.0, 0.2), 1.2, white), fog_b),              # overlaps the red ball and the big fog
        B.Sphere((-2.5, 1.0, 1.0), 1.0, glass),
        B.ConstantMedium(0.4, B.Transform((0.0, 25.0, 0.0), (1.2, 0.8, 1.2), (2.8, 0.81, -1.0), B.Cube((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0), white)), fog_a),
        B.Sphere((3.0, 1.0, 2.5), 1.0, mirror),
        B.YZRectangle((0.0, -8.0), (6.0, 8.0), -6.0, red),
    ]
    return [items[i] for i in order]


@pytest.mark.gpu
@pytest.mark.parametrize("order,bvh", [((0, 1, 2, 3, 4, 5, 6, 7, 8), 2), ((8, 6, 4, 2, 7, 5, 3, 1, 0), 7), ((4, 3, 2, 6, 0, 1, 8, 7, 5), None)])
def test_hip_media_on_the_accel_path_bit_exact(order, bvh):
    """kernel 2's MEDIA variant (the accel finds the surfaces; the media are resolved in the reference's visit order, one draw
    each at most) against the oracle's recursion, under different visit orders (list permutations, BVH seeds, a plain
    HitableList without BVH) -- which medium draws first, and what clips it, changes with each."""
    import oracle
    import rtamd
    w = rtamd.World()
    o = oracle.Scene()
    if bvh is None:
        w.set_root(w.HitableList(_fog_zoo(w, order)))
        o.set_root(o.HitableList(_fog_zoo(o, order)))
    else:
        w.new(_fog_zoo(w, order), bvh_seed=bvh)
        o.World(_fog_zoo(o, order), bvh)
    args = ((0.0, 2.0, -9.0), (0.0, 1.0, 0.0), (0, 1, 0), 45.0, 1.0, 0.0, 9.0)
    o.Camera(*args)
    cam = rtamd.Camera((args[0], args[1]), *args[2:])
    assert w.info()["accel_ok"] == 1
    exp, _ = o.render(64, 64, 8, seed=11)
    for k in (2, 1):
        img, st = w.render(cam, width=64, height=64, spp=8, seed=11, kernel=k)
        assert st["kernel_used"] == k
        assert np.array_equal(img, exp), (k, int((img != exp).any(axis=2).sum()))
    assert exp.max() > 0


def test_a_medium_under_a_transform_keeps_the_reference_order_kernel_and_a_shared_one_does_not():
    import rtamd
    w = rtamd.World()
    white = w.Lambertian(w.ConstantTexture((0.8, 0.8, 0.8)))
    fog = w.Isotropic(w.ConstantTexture((0.9, 0.9, 1.0)))
    med = w.ConstantMedium(0.3, w.Sphere((0.0, 1.0, 0.0), 1.0, white), fog)
    w.new([w.XZRectangle((-5.0, -5.0), (5.0, 5.0), 0.0, white), w.Transform((0.0, 10.0, 0.0), (1.0, 1.0, 1.0), (1.0, 0.0, 0.0), med)], bvh_seed=1)
    assert w.info()["accel_ok"] == 0
    w2 = rtamd.World()
    white = w2.Lambertian(w2.ConstantTexture((0.8, 0.8, 0.8)))
    fog = w2.Isotropic(w2.ConstantTexture((0.9, 0.9, 1.0)))
    med = w2.ConstantMedium(0.3, w2.Sphere((0.0, 1.0, 0.0), 1.0, white), fog)
    w2.set_root(w2.HitableList([w2.XZRectangle((-5.0, -5.0), (5.0, 5.0), 0.0, white), med, med]))
    assert w2.info()["accel_ok"] == 1          # visited twice (as BVHNode::new's one-object leaves do, Q14): two visits, each may draw


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [0, 1, 2])
def test_final_scene_reduced_bit_exact(kernel):
    """BASELINE config C5 reduced to what the reference has code for (rtamd.shapes.final_scene_reduced: no motion blur, no Perlin
    noise; 400 boxes, two ConstantMedium volumes -- one inside a glass ball, one spanning the scene --, an image texture, a
    1000-sphere cluster under a Transform): the accel kernel's MEDIA variant (the automatic choice) and the reference-order kernel
    against the oracle."""
    import oracle
    import rtamd
    from rtamd import shapes
    w = rtamd.World()
    w.new(shapes.final_scene_reduced(w), bvh_seed=3)
    o = oracle.Scene()
    o.World(shapes.final_scene_reduced(o), 3)
    o.Camera(*shapes.FINAL_SCENE_CAMERA)
    f, t, up, vfov, asp, ap, fd = shapes.FINAL_SCENE_CAMERA
    cam = rtamd.Camera((f, t), up, vfov, asp, ap, fd)
    info = w.info()
    assert info["accel_ok"] == 1 and info["n_rects"] == 1 and info["n_cubes"] == 400 and info["n_spheres"] == 1008
    exp, _ = o.render(80, 80, 6, seed=4)
    img, st = w.render(cam, width=80, height=80, spp=6, seed=4, kernel=kernel)
    assert st["kernel_used"] == (2 if kernel == 0 else kernel)
    assert np.array_equal(img, exp), int((img != exp).any(axis=2).sum())
    assert exp.max() > 0
