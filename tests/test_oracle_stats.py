"""Distributions of the oracle's sampling routines against closed forms (a second line of defence for the unpinned
parts of the oracle: the reference is not seedable, so its samplers can only be compared in distribution).
Each moment is checked within 5 standard errors of its exact value over N independent RNG streams."""
import numpy as np

import oracle as O

N = 20000


def _samples(which, normal=(0.0, 0.0, 1.0)):
    return np.array([O.sample_helper(which, (7, i, 0), normal) for i in range(N)])


def _close(sample_mean, exact, std, n=N):
    assert abs(sample_mean - exact) < 5.0 * std / np.sqrt(n), (sample_mean, exact)


def test_random_in_unit_sphere_is_uniform_on_the_sphere():  # vec3.rs:111-129 (Marsaglia), Q3
    v = _samples(0)
    assert np.allclose(np.linalg.norm(v, axis=1), 1.0, atol=1e-14)
    for a in range(3):
        _close(v[:, a].mean(), 0.0, np.sqrt(1 / 3))                 # E[x] = 0, Var[x] = 1/3
        _close((v[:, a] ** 2).mean(), 1 / 3, np.sqrt(4 / 45))       # E[x^2] = 1/3, Var[x^2] = 1/5 - 1/9
    _close((v[:, 0] * v[:, 1]).mean(), 0.0, np.sqrt(1 / 15))        # uncorrelated axes: E[x^2 y^2] = 1/15


def test_random_unit_vector_matches_random_in_unit_sphere_normalised():  # vec3.rs:131-133
    a, b = O.sample_helper(0, (3, 9, 1)), O.sample_helper(1, (3, 9, 1))
    assert np.array_equal(b, a / np.sqrt((a * a).sum())) or np.allclose(b, a, atol=1e-15)


def test_random_in_unit_disk_is_uniform():  # vec3.rs:153-162
    v = _samples(2)
    r2 = v[:, 0] ** 2 + v[:, 1] ** 2
    assert (v[:, 2] == 0).all() and (r2 < 1).all()
    _close(r2.mean(), 0.5, np.sqrt(1 / 12))                         # r^2 ~ U(0, 1)
    _close(v[:, 0].mean(), 0.0, 0.5)                                # Var[x] = 1/4
    _close((v[:, 0] * v[:, 1]).mean(), 0.0, np.sqrt(1 / 24))        # E[x^2 y^2] = 1/24


def test_random_in_hemisphere_is_the_sphere_sample_flipped_into_the_normal_side():  # vec3.rs:144-151
    n = np.array((0.6, 0.0, 0.8))
    v = _samples(3, tuple(n))
    c = v @ n
    assert (c >= 0).all()
    _close(c.mean(), 0.5, np.sqrt(1 / 12))                          # |cos| of a uniform direction ~ U(0, 1)


def test_lambertian_scatter_is_cosine_weighted():  # material.rs:92-98: normal + random_unit_vector
    sc = O.Scene()
    m = sc.Lambertian(sc.ConstantTexture((0.5, 0.5, 0.5)))
    n = np.array((0.0, 1.0, 0.0))
    cos = []
    for i in range(N):
        out = sc.scatter(m, (0, 1, 0), (0.3, -1, 0.1), (0, 0, 0), tuple(n), True, key=(11, i, 0))
        d = out["dir"]
        cos.append(float(d @ n / np.sqrt(d @ d)))
    cos = np.array(cos)
    assert (cos >= -1e-12).all()
    _close(cos.mean(), 2 / 3, np.sqrt(1 / 18))                      # pdf cos/pi: E[cos] = 2/3, E[cos^2] = 1/2
    _close((cos ** 2).mean(), 0.5, np.sqrt(1 / 3 - 1 / 4))


def test_dielectric_reflects_with_schlick_probability():  # material.rs:150-188
    sc = O.Scene()
    ir = 1.5
    m = sc.Dielectric(ir, sc.ConstantTexture((1, 1, 1)))
    n = np.array((0.0, 1.0, 0.0))
    d = np.array((np.sin(1.2), -np.cos(1.2), 0.0))                  # 68.75 degrees off the normal, front face
    refl = 0
    for i in range(N):
        out = sc.scatter(m, (0, 1, 0), tuple(d), (0, 0, 0), tuple(n), True, key=(13, i, 0))
        refl += out["dir"][1] > 0
    p = O.schlick(np.cos(1.2), 1.0 / ir)
    assert 0.05 < p < 0.5
    _close(refl / N, p, np.sqrt(p * (1 - p)))
