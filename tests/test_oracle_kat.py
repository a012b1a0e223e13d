"""Analytic known-answer tests authored in this repo for the parts of the path the reference's
own tests do not pin: intersection, slab test, scatter, camera, tonemap,
quirks Q1-Q16.  Each expected value is derived by hand from the cited reference lines."""
import math

import numpy as np
import pytest

import oracle as O

INF = float("inf")


def sphere_scene(center=(0.0, 0.0, 0.0), r=1.0):
    sc = O.Scene()
    m = sc.Lambertian(sc.ConstantTexture((0.5, 0.5, 0.5)))
    s = sc.Sphere(center, r, m)
    sc.set_root(s)
    return sc, s, m


# ---- Sphere::hit (sphere.rs:24-55) ----
def test_sphere_front_hit_known_t():
    sc, _, _ = sphere_scene()
    h = sc.hit((0, 0, -3), (0, 0, 1), 0.001, INF)
    assert h["t"] == 2.0 and h["front_face"]
    assert np.array_equal(h["p"], (0, 0, -1)) and np.array_equal(h["normal"], (0, 0, -1))


def test_sphere_unnormalised_dir_keeps_t_scale():
    sc, _, _ = sphere_scene()
    h = sc.hit((0, 0, -3), (0, 0, 2), 0.001, INF)   # dir is NOT normalised by the reference (a2)
    assert h["t"] == 1.0


def test_sphere_inside_takes_far_root_and_flips_normal():
    sc, _, _ = sphere_scene()
    h = sc.hit((0, 0, 0), (0, 0, 1), 0.001, INF)    # near root -1 < t_min -> far root
    assert h["t"] == 1.0 and not h["front_face"]
    assert np.array_equal(h["normal"], (0, 0, -1))


def test_sphere_range_is_inclusive_at_both_ends():  # Q5
    sc, _, _ = sphere_scene()
    assert sc.hit((0, 0, -3), (0, 0, 1), 0.001, 2.0)["t"] == 2.0      # root == t_max accepted
    assert sc.hit((0, 0, -3), (0, 0, 1), 2.0, INF)["t"] == 2.0        # root == t_min accepted
    assert sc.hit((0, 0, -3), (0, 0, 1), 0.001, 1.999) is None
    assert sc.hit((0, 0, -3), (0, 0, 1), 2.001, 3.999) is None        # both roots out of range


def test_sphere_tangent_and_miss():
    sc, _, _ = sphere_scene()
    h = sc.hit((1, 0, -3), (0, 0, 1), 0.001, INF)   # discriminant == 0 is a hit (`< 0` rejects)
    assert h is not None and h["t"] == 3.0
    assert sc.hit((1.0000001, 0, -3), (0, 0, 1), 0.001, INF) is None


def test_sphere_uv():  # sphere.rs:16-20
    sc, _, _ = sphere_scene()
    h = sc.hit((0, -3, 0), (0, 1, 0), 0.001, INF)   # outward (0,-1,0): theta = acos(1) = 0
    assert h["uv"][1] == 0.0
    h = sc.hit((3, 0, 0), (-1, 0, 0), 0.001, INF)   # outward (1,0,0): phi = atan2(-0,1)+pi, theta = pi/2
    assert h["uv"] == pytest.approx((0.5, 0.5), abs=1e-15)


# ---- AABB::hit (aabb.rs:15-32) ----
def test_aabb_basic_and_reject_rule():
    assert O.aabb_hit((0, 0, 0), (1, 1, 1), (0.5, 0.5, -1), (0, 0, 1), 0.001, INF)
    assert not O.aabb_hit((0, 0, 0), (1, 1, 1), (1.5, 0.5, -1), (0, 0, 1), 0.001, INF)
    assert not O.aabb_hit((0, 0, 0), (1, 1, 1), (0.5, 0.5, -1), (0, 0, 1), 0.001, 1.0)   # max <= min rejects (interval [1,1])
    assert O.aabb_hit((0, 0, 0), (1, 1, 1), (0.5, 0.5, -1), (0, 0, 1), 0.001, 1.0000001)


def test_aabb_zero_direction_component_gives_inf_slabs():
    # dir.x == 0 -> inv = +inf: inside the slab t0 = -inf, t1 = +inf; outside both are +inf (or -inf) -> reject
    assert O.aabb_hit((0, 0, 0), (1, 1, 1), (0.5, 0.5, -1), (0.0, 0.0, 1.0), 0.001, INF)
    assert not O.aabb_hit((0, 0, 0), (1, 1, 1), (2.0, 0.5, -1), (0.0, 0.0, 1.0), 0.001, INF)
    # origin exactly on the slab plane: (0 - 0) * inf = NaN, which f64::max/min ignore
    assert O.aabb_hit((0, 0, 0), (1, 1, 1), (0.0, 0.5, -1), (0.0, 0.0, 1.0), 0.001, INF)


def test_aabb_negative_direction_swaps():
    assert O.aabb_hit((0, 0, 0), (1, 1, 1), (0.5, 0.5, 2), (0, 0, -1), 0.001, INF)
    assert not O.aabb_hit((0, 0, 0), (1, 1, 1), (0.5, 0.5, 2), (0, 0, 1), 0.001, INF)


# ---- rectangles (rectangle.rs) ----
def test_rect_hit_bounds_inclusive_and_fixed_normal():
    sc = O.Scene()
    m = sc.Lambertian(sc.ConstantTexture((1, 1, 1)))
    r = sc.XZRectangle((0, 0), (2, 4), 1.0, m)
    sc.set_root(r)
    h = sc.hit((1, 3, 1), (0, -1, 0), 0.001, INF)
    assert h["t"] == 2.0 and h["uv"] == (0.5, 0.25)
    assert np.array_equal(h["normal"], (0, 1, 0)) and h["front_face"]      # outward is always +y
    h = sc.hit((1, -3, 1), (0, 1, 0), 0.001, INF)
    assert np.array_equal(h["normal"], (0, -1, 0)) and not h["front_face"]
    assert sc.hit((2, 3, 4), (0, -1, 0), 0.001, INF) is not None             # x == x1, z == z1 inclusive
    assert sc.hit((2.0000001, 3, 4), (0, -1, 0), 0.001, INF) is None
    assert np.allclose(sc.bounding_box(r), (0, 1 - 1e-4, 0, 2, 1 + 1e-4, 4), rtol=0, atol=0)


def test_rect_in_plane_ray_returns_nan_hit():  # 0/0 = NaN passes every reject
    sc = O.Scene()
    m = sc.Lambertian(sc.ConstantTexture((1, 1, 1)))
    sc.set_root(sc.XYRectangle((0, 0), (1, 1), 0.0, m))
    h = sc.hit((0.5, 0.5, 0.0), (1, 0, 0), 0.001, INF)
    assert h is not None and math.isnan(h["t"])


# ---- Cube / list order (cube.rs, hit.rs:56-67) ----
def test_cube_hits_nearest_side():
    sc = O.Scene()
    m = sc.Lambertian(sc.ConstantTexture((1, 1, 1)))
    c = sc.Cube((0, 0, 0), (1, 2, 3), m)
    sc.set_root(c)
    assert sc.hit((0.5, 1, -5), (0, 0, 1), 0.001, INF)["t"] == 5.0
    assert sc.hit((0.5, 1, 10), (0, 0, -1), 0.001, INF)["t"] == 7.0
    assert np.array_equal(sc.bounding_box(c), (0, 0, 0, 1, 2, 3))


def test_later_object_wins_exact_tie():  # Q5
    sc = O.Scene()
    m = sc.Lambertian(sc.ConstantTexture((1, 1, 1)))
    a = sc.Sphere((0, 0, 0), 1.0, m)
    b = sc.Sphere((0, 0, 0), 1.0, m)
    sc.set_root(sc.HitableList([a, b]))
    assert sc.hit((0, 0, -3), (0, 0, 1))["prim_id"] == b
    sc.set_root(sc.BVHNode_construct(a, b))
    assert sc.hit((0, 0, -3), (0, 0, 1))["prim_id"] == b   # right child returned when it hits (bvh.rs:98-101)


# ---- Triangle (mesh.rs:57-137) ----
def tri_scene():
    sc = O.Scene()
    m = sc.Lambertian(sc.ConstantTexture((1, 1, 1)))
    P = [(0, 0, 0), (1, 0, 0), (0, 1, 0)]
    N = [(0, 0, 1)] * 3
    mesh = sc.Mesh(P, N, [(0, 1, 2)], m, seed=1)
    sc.set_root(mesh)
    return sc, mesh


def test_triangle_hit_double_sided_and_edges():
    sc, mesh = tri_scene()
    h = sc.hit((0.25, 0.25, 1), (0, 0, -1), 0.001, INF)
    assert h["t"] == 1.0 and h["front_face"] and np.array_equal(h["normal"], (0, 0, 1)) and h["uv"] == (0.0, 0.0)
    h = sc.hit((0.25, 0.25, -1), (0, 0, 1), 0.001, INF)                   # back side: no culling
    assert h["t"] == 1.0 and not h["front_face"] and np.array_equal(h["normal"], (0, 0, -1))
    assert sc.hit((0.5, 0.5, 1), (0, 0, -1), 0.001, INF) is not None        # b1 + b2 == 1 inclusive
    assert sc.hit((0.51, 0.5, 1), (0, 0, -1), 0.001, INF) is None
    assert sc.hit((0.25, 0.25, 1), (1, 0, 0), 0.001, INF) is None           # parallel: s0.e0 == 0
    # Q9: box padded by +-0.1
    assert np.allclose(sc.bounding_box(mesh), (-0.1, -0.1, -0.1, 1.1, 1.1, 0.1), rtol=0, atol=1e-16)


# ---- Transform (transform.rs) ----
def test_transform_scale_translate_preserves_t_and_quirks():
    sc, mesh = tri_scene()
    t = sc.Transform((0, 0, 0), (2, 2, 2), (10, 0, 0), mesh)
    sc.set_root(t)
    h = sc.hit((10.5, 0.5, 4), (0, 0, -1), 0.001, INF)
    assert h["t"] == 4.0                                  # dir is transformed un-normalised -> same t
    assert np.array_equal(h["p"], (10.5, 0.5, 0.0))
    assert np.array_equal(h["normal"], (0, 0, 1))
    assert h["front_face"]                                # Q8: always true after the object-space flip
    h = sc.hit((10.5, 0.5, -4), (0, 0, 1), 0.001, INF)
    assert np.array_equal(h["normal"], (0, 0, -1)) and h["front_face"]
    bb = sc.bounding_box(t)
    assert np.allclose(bb, (9.8, -0.2, -0.2, 12.2, 2.2, 0.2), rtol=0, atol=1e-12)


def test_transform_rotation_y_90():
    sc, mesh = tri_scene()
    t = sc.Transform((0, 90, 0), (1, 1, 1), (0, 0, 0), mesh)   # Ry(90): x -> -z, z -> x
    sc.set_root(t)
    h = sc.hit((2, 0.25, -0.25), (-1, 0, 0), 0.001, INF)
    assert h is not None and h["t"] == pytest.approx(2.0, abs=1e-12)
    assert np.allclose(h["normal"], (1, 0, 0), atol=1e-12)


# ---- materials (material.rs) ----
def test_reflect_refract_schlick():
    assert np.array_equal(O.vec3_op(13, (1, -1, 0), (0, 1, 0)), (1, 1, 0))           # reflect
    r = O.vec3_op(14, (0, -1, 0), (0, 1, 0), 1.0 / 1.5)                               # refract at normal incidence
    assert np.array_equal(r, (0, -1, 0))
    assert O.schlick(1.0, 1.5) == ((1 - 1.5) / (1 + 1.5)) ** 2                        # Schlick(1, n) = r0
    assert O.schlick(0.0, 1.5) == 1.0


def test_lambertian_scatter_direction_is_normal_plus_unit_vector():
    sc, _, m = sphere_scene()
    key = (5, 17, 3)
    out = sc.scatter(m, (0, 0, -3), (0, 0, 1), (0, 0, -1), (0, 0, -1), True, key=key)
    v = O.sample_helper(1, key)                                                        # random_unit_vector, same stream
    assert out["kind"] == 0 and out["scattered"]
    assert np.array_equal(out["dir"], np.array((0, 0, -1)) + v)
    assert abs(np.linalg.norm(v) - 1) < 1e-15
    assert np.array_equal(out["attenuation"], (0.5, 0.5, 0.5)) and not out["emitted"].any()


def test_metal_mirror_and_absorb():  # Q4, Q15
    sc = O.Scene()
    m = sc.Metal(sc.ConstantTexture((0.8, 0.6, 0.3)), 0.0)
    d = np.array((1.0, -1.0, 0.0))
    out = sc.scatter(m, (0, 1, 0), d, (1, 0, 0), (0, 1, 0), True)
    u = d / np.sqrt(2.0)
    assert out["kind"] == 1 and np.allclose(out["dir"], (u[0], -u[1], 0), atol=1e-16)
    out = sc.scatter(m, (0, 1, 0), d, (1, 0, 0), (0, -1, 0), True)                    # reflected points into the surface
    assert out["kind"] == 2 and not out["scattered"]


def test_dielectric_total_internal_reflection_draws_no_random_number():  # Q16
    sc = O.Scene()
    m = sc.Dielectric(2.0, sc.ConstantTexture((1, 1, 1)))
    d = np.array((1.0, -0.5, 0.0))                                                     # sin > 1/2 from inside (ratio 2)
    out1 = sc.scatter(m, (0, 0, 0), d, (0, 0, 0), (0, 1, 0), False, key=(1, 0, 0))
    out2 = sc.scatter(m, (0, 0, 0), d, (0, 0, 0), (0, 1, 0), False, key=(99, 5, 7))
    assert out1["kind"] == 3 and np.array_equal(out1["dir"], out2["dir"])             # Reflect, RNG-independent


def test_diffuse_light_emits_both_faces_and_scatters():  # Q10
    sc = O.Scene()
    m = sc.DiffuseLight(sc.ConstantTexture((4, 5, 6)))
    for ff in (True, False):
        out = sc.scatter(m, (0, 0, -3), (0, 0, 1), (0, 0, -1), (0, 0, -1), ff)
        assert np.array_equal(out["emitted"], (4, 5, 6)) and out["kind"] == 0
        assert np.array_equal(out["attenuation"], [1.0 / math.pi * 1.0] * 3) or np.allclose(out["attenuation"], 1 / math.pi, rtol=1e-16)


def test_checker_texture_picks_t0_when_sines_negative():
    sc = O.Scene()
    m = sc.Lambertian(sc.CheckerTexture(sc.ConstantTexture((1, 0, 0)), sc.ConstantTexture((0, 1, 0))))
    p_neg = (0.1 * 4.0, 0.1, 0.1)   # sin(4) < 0, others > 0 -> product < 0 -> t0
    p_pos = (0.1, 0.1, 0.1)
    assert np.array_equal(sc.scatter(m, (0, 0, 0), (0, 0, 1), p_neg, (0, 0, -1), True)["attenuation"], (1, 0, 0))
    assert np.array_equal(sc.scatter(m, (0, 0, 0), (0, 0, 1), p_pos, (0, 0, -1), True)["attenuation"], (0, 1, 0))


def test_image_texture_nearest_flipped_and_clamped():  # material.rs:70-84, Q11
    sc = O.Scene()
    img = np.zeros((2, 2, 3), dtype=np.uint8)
    img[0, 0] = (255, 0, 0); img[0, 1] = (0, 255, 0); img[1, 0] = (0, 0, 255); img[1, 1] = (255, 255, 255)
    m = sc.Lambertian(sc.ImageTexture(img))
    att = lambda u, v: sc.scatter(m, (0, 0, 0), (0, 0, 1), (0, 0, 0), (0, 0, -1), True, uv=(u, v))["attenuation"]
    assert np.array_equal(att(0.0, 0.9), (1, 0, 0))        # v flipped: v~1 -> top row
    assert np.array_equal(att(0.9, 0.1), (1, 1, 1))
    assert np.array_equal(att(1.0, 1.0), (0, 1, 0))        # u == 1 clamped to the last column instead of panicking


# ---- sampling helpers (vec3.rs:111-162) ----
def test_random_in_unit_sphere_is_on_the_sphere():  # Q3
    for s in range(50):
        v = O.sample_helper(0, (3, s, 0))
        assert abs(np.dot(v, v) - 1.0) < 1e-14
        d = O.sample_helper(2, (3, s, 1))
        assert d[2] == 0.0 and np.dot(d, d) < 1.0
        h = O.sample_helper(3, (3, s, 2), normal=(0, 0, -1))
        assert np.dot(h, (0, 0, -1)) > 0


def test_rng_stream_basics():
    a = O.rng_f64(1, 0, 0, 1000)
    assert all(0.0 <= x < 1.0 for x in a) and len(set(a)) == 1000
    assert O.rng_u64(1, 0, 0, 4) != O.rng_u64(1, 0, 1, 4) != O.rng_u64(1, 1, 0, 4)
    assert abs(np.mean(O.rng_f64(7, 3, 9, 100000)) - 0.5) < 5e-3


# ---- camera (camera.rs:24-64) ----
def test_camera_basis_and_corner_rays():
    sc, _, _ = sphere_scene()
    sc.Camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, 2.0, 0.0, 1.0)
    b = sc.camera_basis()
    origin, llc, hor, ver, u, v, w, lens = b[0:3], b[3:6], b[6:9], b[9:12], b[12:15], b[15:18], b[18:21], b[21]
    assert np.array_equal(w, (0, 0, 1)) and np.array_equal(u, (1, 0, 0)) and np.array_equal(v, (0, 1, 0))
    h = math.tan(math.radians(90.0) / 2)
    assert np.allclose(hor, (2 * 2 * h, 0, 0)) and np.allclose(ver, (0, 2 * h, 0)) and lens == 0.0
    assert np.allclose(llc, (-2 * h, -h, -1))
    # pixel (0,0): u=(0+xi1)/(W-1), v=(0+xi2)/(H-1), ray = get_ray(u, 1-v)  (Q2)
    W, H = 5, 3
    xi = O.rng_f64(1, 0, 0, 2)
    o, d = sc.camera_ray(W, H, 0, 0, seed=1, sample=0)
    s, t = xi[0] / (W - 1), 1.0 - xi[1] / (H - 1)
    assert np.array_equal(o, (0, 0, 0))
    assert np.allclose(d, llc + s * hor + t * ver, rtol=1e-15)


def test_lens_sample_is_drawn_even_for_zero_aperture():  # Q4: RNG stream order
    sc, _, _ = sphere_scene()
    sc.Camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 40.0, 1.0, 0.0, 1.0)
    a, cnt_a = sc.render(4, 4, 2, seed=1)
    sc.Camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 40.0, 1.0, 1e-300, 1.0)
    b, cnt_b = sc.render(4, 4, 2, seed=1)
    assert cnt_a == cnt_b     # same RNG consumption either way


# ---- tonemap (vec3.rs:223-231, Q13) ----
def test_tonemap_values():
    x = np.array([0.0, 1.0, 4.0, 0.25, -1.0, float("nan"), (254.9999 / 255) ** 2, 1e-12])
    assert O.tonemap_u8(x).tolist() == [0, 255, 255, 127, 0, 0, 254, 0]


# ---- sample_ray (photon_mapper.rs:327-365 + D2) ----
def test_miss_is_black_and_emission_is_added_on_first_hit():
    sc = O.Scene()
    light = sc.DiffuseLight(sc.ConstantTexture((2, 3, 4)))
    sc.set_root(sc.Sphere((0, 0, -5), 1.0, light))
    sc.Camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 10.0, 1.0, 0.0, 1.0)
    img, cnt = sc.render(3, 3, 1, max_depth=1, seed=1)
    assert np.array_equal(img[1, 1], (2, 3, 4))            # centre pixel sees the light: Le, then depth exhausted
    img0, _ = sc.render(3, 3, 1, max_depth=0, seed=1)
    assert not img0.any()                                   # Q12: depth 0 -> no hit contributes
    sc.Camera((0, 0, 0), (0, 0, 1), (0, 1, 0), 10.0, 1.0, 0.0, 1.0)
    assert not sc.render(3, 3, 1, seed=1)[0].any()          # looking away: black background


def test_white_furnace_closed_emitter():
    """inside a closed DiffuseLight sphere of emission E every hit adds beta*E and multiplies beta by 1/pi:
    L = E * sum_{k<depth} pi^-k exactly (Q10 semantics)."""
    sc = O.Scene()
    sc.set_root(sc.Sphere((0, 0, 0), 10.0, sc.DiffuseLight(sc.ConstantTexture((1, 1, 1)))))
    sc.Camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 40.0, 1.0, 0.0, 1.0)
    for depth in (1, 2, 5):
        img, _ = sc.render(2, 2, 1, max_depth=depth, seed=1)
        exp = 0.0
        beta = 1.0
        for _ in range(depth):
            exp = exp + beta * 1.0
            beta = beta * (1.0 * (1 / math.pi))
        assert np.allclose(img, exp, rtol=1e-15, atol=0)


def test_bvh_new_is_deterministic_and_boxes_enclose():
    sc = O.Scene()
    m = sc.Lambertian(sc.ConstantTexture((1, 1, 1)))
    rng = np.random.default_rng(3)
    ids = [sc.Sphere(tuple(rng.random(3) * 10), 0.3, m) for _ in range(37)]
    root = sc.BVHNode_new(ids, seed=5)
    bb = sc.bounding_box(root)
    for i in ids:
        b = sc.bounding_box(i)
        assert (bb[:3] <= b[:3]).all() and (bb[3:] >= b[3:]).all()
    sc2 = O.Scene()
    m2 = sc2.Lambertian(sc2.ConstantTexture((1, 1, 1)))
    rng = np.random.default_rng(3)
    ids2 = [sc2.Sphere(tuple(rng.random(3) * 10), 0.3, m2) for _ in range(37)]
    sc2.set_root(sc2.BVHNode_new(ids2, seed=5))
    sc.set_root(root)
    sc.Camera((5, 5, -20), (5, 5, 5), (0, 1, 0), 40.0, 1.0, 0.0, 10.0)
    sc2.Camera((5, 5, -20), (5, 5, 5), (0, 1, 0), 40.0, 1.0, 0.0, 10.0)
    a, ca = sc.render(16, 16, 2, seed=1)
    b, cb = sc2.render(16, 16, 2, seed=1)
    assert np.array_equal(a, b) and ca == cb
