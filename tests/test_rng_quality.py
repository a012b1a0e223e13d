"""Statistics of the shipped RNG (rtamd-rng-3, csrc/common/rng.h): gen::<f64>() = (next_u64 >> 11) * 2^-53 (rand 0.8.4's
conversion) of a xoroshiro64 stream keyed by (seed, pixel, sample); one step yields 64 bits (** scrambler << 32 | + scrambler).  The path tracer uses ADJACENT keys (neighbouring pixels, consecutive
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
            out[i, j] = (int(draw_fn(seed, p, s, 1)[0]) >> 11) * 2.0 ** -53     # gen_f64 of the first next_u64
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
    bits = (f.ravel() * 2.0 ** 53).astype(np.uint64)
    for b in range(53):                                         # every one of the 53 bits is balanced
        ones = int(((bits >> np.uint64(b)) & np.uint64(1)).sum())
        assert abs(ones - n / 2) < 4.5 * np.sqrt(n) / 2, (b, ones)


@pytest.mark.parametrize("which", ["product (host)", "oracle"])
def test_draws_along_one_stream_are_uniform_and_uncorrelated(which):
    draw = _impls()[which]
    u = np.asarray(draw(7, 123456, 789, 65536), dtype=np.uint64)
    n = u.size

    def check(f, bins=256):
        hist = np.bincount((f * bins).astype(int), minlength=bins)
        chi2 = ((hist - n / bins) ** 2 / (n / bins)).sum()
        assert 160.0 < chi2 < 370.0, chi2
        x = f - 0.5
        for lag in (1, 2, 3):
            r = (x[:-lag] * x[lag:]).sum() / (x * x).sum()
            assert abs(r) < 4.0 / np.sqrt(n), (lag, r)
        assert f.min() >= 0.0 and f.max() < 1.0                 # [0, 1): 1.0 is never produced

    g = (u >> np.uint64(11)).astype(np.float64) * 2.0 ** -53    # gen::<f64>()
    check(g)
    check((u >> np.uint64(32)).astype(np.float64) * 2.0 ** -32)                     # the ** half alone (next_u32)
    low21 = ((u >> np.uint64(11)) & np.uint64(0x1FFFFF)).astype(np.float64) * 2.0 ** -21   # the 21 bits the + scrambler contributes to a draw
    check(low21)
    check(((u >> np.uint64(11)) & np.uint64(0xFF)).astype(np.float64) / 256.0)      # ... and the lowest 8 of them
    x, y = low21[:-1] - 0.5, g[1:] - 0.5                         # low bits of one draw against the next draw
    assert abs((x * y).sum() / np.sqrt((x * x).sum() * (y * y).sum())) < 4.0 / np.sqrt(n)
    x, y = low21 - 0.5, (u >> np.uint64(32)).astype(np.float64) * 2.0 ** -32 - 0.5   # ... and against the upper half of the SAME step
    assert abs((x * y).sum() / np.sqrt((x * x).sum() * (y * y).sum())) < 4.0 / np.sqrt(n)


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
    """(seed, pixel, sample) = (seed, 0, 0) whose stream starts in the state (s0, s1) = (0, 1): the step's upper half is
    rotl(s0 * K, 5) * 5 = 0 and its lower half s0 + s1 = 1, so next_u64 >> 11 == 0 and gen::<f64>() == 0.0 exactly."""
    s = 1 << 32
    h = (_unmix(s) - 0xD1B54A32D192ED03 * 1) & M64
    return (_unmix(h) - 0x9E3779B97F4A7C15 * 1) & M64, 0, 0


def test_a_zero_draw_exists_and_every_implementation_returns_it():
    import oracle
    import rtamd
    key = _key_with_first_draw_zero()
    a = rtamd.debug_rng(*key, 3, device=False)
    b = oracle.rng_u64(*key, 3)
    assert list(a) == list(b) and int(a[0]) == 1                 # 53 upper bits clear: gen_f64() == 0.0 exactly
    assert oracle.rng_f64(*key, 1)[0] == 0.0 and rtamd.debug_rng_floats(*key, 1, device=False)[0][0] == 0.0
    assert int(a[1]) >> 11 != 0                                  # ... and the stream goes on normally


def test_constant_medium_with_a_zero_draw_does_not_scatter():
    """gen::<f64>() == 0 -> ln(0) = -inf -> hit_distance = -1/density * -inf = +inf > any chord: the ray passes, whatever the
    density (medium.rs:37-40); one draw is consumed.  With 53-bit draws (the reference's) this happens once per 2^53 medium crossings."""
    import oracle
    o = oracle.Scene()
    iso = o.Isotropic(o.ConstantTexture((0.9, 0.8, 0.7)))
    s = o.Sphere((0.0, 0.0, 0.0), 1.0, iso)
    med = o.ConstantMedium(1e6, s, iso)                           # so dense that every other draw scatters at the entry point
    key = _key_with_first_draw_zero()
    assert o.hit((0.0, 0.0, -5.0), (0.0, 0.0, 1.0), 1e-3, float("inf"), obj=med, key=key) is None and o.last_draws == 1
    other = (key[0] ^ 1, 0, 0)
    assert o.hit((0.0, 0.0, -5.0), (0.0, 0.0, 1.0), 1e-3, float("inf"), obj=med, key=other) is not None
