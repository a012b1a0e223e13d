"""Statistics of the shipped RNG (rtamd-rng-2, csrc/common/rng.h): gen::<f64>() = next_u32 * 2^-32 of a
xoroshiro64** stream keyed by (seed, pixel, sample).  The path tracer uses ADJACENT keys (neighbouring pixels, consecutive
sample indices), so uniformity and independence are checked across adjacent keys as well as along a stream, on the product's
host implementation (rt_debug_rng_host through the C ABI) and on the oracle's independent restatement; the device
implementation is pinned to both bit for bit elsewhere (tests/test_parity_gpu.py::test_rng_device_matches_oracle_and_host).
Also: a stream whose first draw is exactly 0 -- constructed by inverting the key hash -- through ConstantMedium::hit, where
the draw goes into ln() (medium.rs:37-38)."""
import numpy as np
import pytest

M64 = (1 << 64) - 1


def _first_draws(draw_fn, seed, pixels, samples):
    out = np.empty((len(pixels), len(samples)))
    for i, p in enumerate(pixels):
        for j, s in enumerate(samples):
            out[i, j] = (int(draw_fn(seed, p, s, 1)[0]) >> 32) * 2.0 ** -32     # gen_f64 = the first next_u32 = high half of next_u64
    return out


def _impls():
    import oracle
    import rtamd
    return {"product (host)": lambda seed, p, s, n: rtamd.debug_rng(seed, p, s, n, device=False), "oracle": oracle.rng_u64}


@pytest.mark.parametrize("which", ["product (host)", "oracle"])
def test_first_draws_of_adjacent_keys_are_uniform_and_uncorrelated(which):
    draw = _impls()[which]
    f = _first_draws(draw, 1, range(2048), range(16))          # 32 768 streams: 2 048 neighbouring pixels x 16 consecutive samples
    n = f.size
    hist = np.bincount((f.ravel() * 256).astype(int), minlength=256)
    chi2 = ((hist - n / 256.0) ** 2 / (n / 256.0)).sum()
    assert 160.0 < chi2 < 370.0, chi2                           # 255 degrees of freedom: mean 255, sd 22.6 -> +-5 sd
    assert abs(f.mean() - 0.5) < 4.0 / np.sqrt(12 * n)
    x = f - 0.5
    r_pix = (x[:-1] * x[1:]).sum() / (x * x).sum()              # neighbouring pixels, same sample index
    r_smp = (x[:, :-1] * x[:, 1:]).sum() / (x * x).sum()        # consecutive samples of one pixel
    assert abs(r_pix) < 4.0 / np.sqrt(n) and abs(r_smp) < 4.0 / np.sqrt(n), (r_pix, r_smp)
    bits = (f.ravel() * 2.0 ** 32).astype(np.uint64)
    for b in range(32):                                         # every one of the 32 bits is balanced
        ones = int(((bits >> np.uint64(b)) & np.uint64(1)).sum())
        assert abs(ones - n / 2) < 4.5 * np.sqrt(n) / 2, (b, ones)


@pytest.mark.parametrize("which", ["product (host)", "oracle"])
def test_draws_along_one_stream_are_uniform_and_uncorrelated(which):
    draw = _impls()[which]
    u = np.asarray(draw(7, 123456, 789, 32768), dtype=np.uint64)
    f = np.concatenate([(u >> np.uint64(32)), (u & np.uint64(0xFFFFFFFF))]).reshape(2, -1).T.ravel().astype(np.float64) * 2.0 ** -32  # draw order
    n = f.size
    hist = np.bincount((f * 256).astype(int), minlength=256)
    chi2 = ((hist - n / 256.0) ** 2 / (n / 256.0)).sum()
    assert 160.0 < chi2 < 370.0, chi2
    x = f - 0.5
    for lag in (1, 2, 3):
        r = (x[:-lag] * x[lag:]).sum() / (x * x).sum()
        assert abs(r) < 4.0 / np.sqrt(n), (lag, r)
    assert f.min() >= 0.0 and f.max() < 1.0                     # [0, 1): 1.0 is never produced (the largest value is 1 - 2^-32)


def _unxorshift(z, k):
    x = z
    for _ in range(64 // k + 1):
        x = z ^ (x >> k)
    return x & M64


def _unmix(z):
    """inverse of the SplitMix64 finaliser (rng.h `mix`)"""
    z = _unxorshift(z, 31)
    z = (z * pow(0x94D049BB133111EB, -1, 1 << 64)) & M64
    z = _unxorshift(z, 27)
    z = (z * pow(0xBF58476D1CE4E5B9, -1, 1 << 64)) & M64
    return _unxorshift(z, 30)


def _key_with_first_draw_zero():
    """(seed, pixel, sample) = (seed, 0, 0) whose stream starts in the state (s0, s1) = (0, 1): next_u32 = rotl(s0 * K, 5) * 5 = 0."""
    s = 1 << 32
    h = (_unmix(s) - 0xD1B54A32D192ED03 * 1) & M64
    return (_unmix(h) - 0x9E3779B97F4A7C15 * 1) & M64, 0, 0


def test_a_zero_draw_exists_and_every_implementation_returns_it():
    import oracle
    import rtamd
    key = _key_with_first_draw_zero()
    a = rtamd.debug_rng(*key, 3, device=False)
    b = oracle.rng_u64(*key, 3)
    assert list(a) == list(b) and (int(a[0]) >> 32) == 0        # gen_f64() == 0.0 exactly
    assert oracle.rng_f64(*key, 1)[0] == 0.0
    assert (int(a[0]) & 0xFFFFFFFF) != 0                         # ... and the stream goes on normally


def test_constant_medium_with_a_zero_draw_does_not_scatter():
    """gen::<f64>() == 0 -> ln(0) = -inf -> hit_distance = -1/density * -inf = +inf > any chord: the ray passes, whatever the
    density (medium.rs:37-40); one draw is consumed.  With 32-bit draws this happens once per 2^32 medium crossings."""
    import oracle
    o = oracle.Scene()
    iso = o.Isotropic(o.ConstantTexture((0.9, 0.8, 0.7)))
    s = o.Sphere((0.0, 0.0, 0.0), 1.0, iso)
    med = o.ConstantMedium(1e6, s, iso)                           # so dense that every other draw scatters at the entry point
    key = _key_with_first_draw_zero()
    assert o.hit((0.0, 0.0, -5.0), (0.0, 0.0, 1.0), 1e-3, float("inf"), obj=med, key=key) is None and o.last_draws == 1
    other = (key[0] ^ 1, 0, 0)
    assert o.hit((0.0, 0.0, -5.0), (0.0, 0.0, 1.0), 1e-3, float("inf"), obj=med, key=other) is not None
