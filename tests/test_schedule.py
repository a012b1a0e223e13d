"""Host logic of the per-launch job sequence (make_schedule in csrc/device/kernels.hip, through rt_debug_schedule): no GPU needed.

The reference accumulates a pixel's samples in index order (camera.rs:96-101); the device kernels cut every tile's samples into units,
deal them to waves in rounds and fold them through a per-tile ticket in unit order.  Whatever the cut, it must cover [s_begin, s_end)
exactly once, in order, with unit indices and rounds that the kernel's arithmetic (next_unit) walks without gaps."""
import random

import pytest


def _units(levels):
    """(unit index, s0, s1, round) of every unit of one tile, as next_unit() derives them from the level table"""
    out = []
    closing = levels[-1]
    for i, (r0, u0, s0, size, ju) in enumerate(levels[:-1]):
        nxt = levels[i + 1]
        n_units = nxt[1] - u0
        for k in range(n_units):
            a = s0 + k * size
            out.append((u0 + k, a, min(a + size, nxt[2]), r0 + k // ju))
        assert nxt[0] - r0 == -(-n_units // ju), "rounds of level %d" % i
    assert len(out) == closing[1]
    return out


def _check(tiles, waves, s_begin, s_end, sub, ju):
    import rtamd
    rounds, levels = rtamd.debug_schedule(tiles, waves, s_begin, s_end, sub, ju)
    assert 2 <= len(levels) <= 5 and levels[-1] == (rounds, levels[-1][1], s_end, 0, 0)
    assert levels[0][:3] == (0, 0, s_begin) and levels[0][3] == sub and levels[0][4] == ju
    units = _units(levels)
    # the units tile [s_begin, s_end) in order, none empty, none longer than its level allows
    assert units[0][1] == s_begin and units[-1][2] == s_end
    for (i, a, b, r), (j, c, d, q) in zip(units, units[1:]):
        assert j == i + 1 and c == b and q >= r
    assert all(0 < b - a <= sub for (_, a, b, _) in units)
    assert max(r for (_, _, _, r) in units) == rounds - 1
    # the taper: single units, sizes never grow, at most a quarter of the launch's samples (plus nothing when the launch is tiny)
    sizes = [lv[3] for lv in levels[:-1]]
    assert sizes == sorted(sizes, reverse=True)
    assert all(lv[4] == 1 for lv in levels[1:-1])
    taper = s_end - levels[1][2] if len(levels) > 2 else 0
    assert taper <= (s_end - s_begin) // 4
    return rounds, levels, taper


def test_headline_and_rank_share_schedules():
    """whole headline frame on one GPU: 2 rounds per taper level; a rank's eighth: 9; the taper grows as tiles per wave shrink"""
    _, lv1, t1 = _check(22500, 4096, 0, 1000, 8, 2)
    assert [l[3] for l in lv1[:-1]] == [8, 8, 4, 2] and t1 == 2 * 14
    _, lv8, t8 = _check(2813, 4096, 0, 1000, 8, 2)
    assert t8 == 9 * 14
    _, lv32, t32 = _check(704, 4096, 0, 1000, 8, 2)
    assert t32 == (1000 // 4 // 14) * 14  # capped at a quarter of the launch
    assert t1 < t8 < t32


def test_tiny_launches_have_no_taper():
    for spp in (1, 2, 7, 8, 24, 55):
        rounds, levels, taper = _check(100, 4096, 0, spp, min(8, spp), 1)
        assert taper == 0 or spp >= 4 * (levels[1][3] + (levels[2][3] if len(levels) > 3 else 0))
    rounds, levels, taper = _check(100, 4096, 0, 8, 8, 1)
    assert rounds == 1 and taper == 0 and len(levels) == 2


def test_random_schedules_tile_the_sample_range():
    rng = random.Random(5)
    for _ in range(400):
        s_begin = rng.choice([0, 0, 3, 97, 1000])
        n = rng.choice([1, 2, 5, 8, 9, 16, 31, 56, 57, 100, 250, 999, 1000, 4000])
        sub = rng.randint(1, 8)
        ju = rng.choice([1, 2, 2, 3])
        tiles = rng.choice([1, 2, 63, 704, 1407, 2813, 22500, 40000])
        waves = rng.choice([64, 1024, 4096])
        _check(tiles, waves, s_begin, s_begin + n, sub, ju)


def test_bad_arguments_are_refused():
    import rtamd
    for args in ((0, 4096, 0, 8, 8, 2), (10, 0, 0, 8, 8, 2), (10, 4096, 5, 5, 8, 2), (10, 4096, 0, 8, 9, 2), (10, 4096, 0, 8, 8, 0)):
        with pytest.raises(rtamd.RtError):
            rtamd.debug_schedule(*args)
