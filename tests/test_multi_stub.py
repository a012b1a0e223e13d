"""rt_render_multi's HOST logic -- argument checks, the deal of tiles to ranks, the slot of every rank's rows in the gathered buffer,
which rows travel and between which communicator ranks, the stitch -- on the sanitizer build's FAKE devices (tests/asan/device_stub.cpp:
host memory, rows filled with a pattern of (global tile, pixel, channel), exchange = memcpy).  Runs only inside tests/asan/run_host_asan.sh
(RTAMD_STUB_DEVICES set, librtamd_host_asan.so loaded under ASan + UBSan); the real thing -- HIP kernels and RCCL -- is covered by the
-m gpu tests of tests/test_multi_gpu.py."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.skipif(not os.environ.get("RTAMD_STUB_DEVICES"), reason="needs the sanitizer build's fake devices (run_host_asan.sh)")


def _expected(width, height, seed):
    y, x = np.mgrid[0:height, 0:width]
    t = (y // 8) * ((width + 7) // 8) + x // 8
    pix = (y % 8) * 8 + x % 8
    return np.stack([t * 1000.0 + pix * 3 + c + (seed % 7) * 0.125 for c in range(3)], axis=-1)


def _world():
    import rtamd
    from conftest import scene_path
    return rtamd.load_scene_file(scene_path("scene_10.json"))


@pytest.mark.parametrize("devices", [[0], [0, 1], [0, 1, 2, 3], [2, 0, 2, 1, 0], [1, 1, 1], [3, 3, 0, 0, 3, 0, 1]])
@pytest.mark.parametrize("size", [(64, 40), (37, 21), (8, 8), (5, 3)])
def test_every_tile_reaches_its_place_in_the_frame(devices, size):
    import rtamd
    world, cam = _world()
    w, h = size
    for force in (0, 1):
        rtamd.set_tuning(multi_force_rccl=force)
        try:
            img, st = world.render_multi(cam, devices=devices, width=w, height=h, spp=1, seed=3)
        finally:
            rtamd.set_tuning()
        assert np.array_equal(img, _expected(w, h, 3))
        assert len(st) == len(devices)
        total = (w * h)
        assert sum(s["samples"] for s in st) == total             # every pixel belongs to exactly one rank
        travelled = sum(1 for i, d in enumerate(devices) if (force or d != devices[0]) and (((w + 7) // 8) * ((h + 7) // 8) > i))
        assert st[0]["rows_through_rccl"] == travelled            # rows on the root's device stay where they were rendered


def test_gpus_n_means_devices_0_to_n_minus_1_and_zero_means_all():
    world, cam = _world()
    a, st = world.render_multi(cam, gpus=3, width=40, height=24, spp=1, seed=5)
    assert np.array_equal(a, _expected(40, 24, 5)) and len(st) == 3 and st[0]["rows_through_rccl"] == 2
    b, st = world.render_multi(cam, gpus=0, width=40, height=24, spp=1, seed=5)
    assert np.array_equal(b, a) and len(st) == int(os.environ["RTAMD_STUB_DEVICES"])


def test_more_ranks_than_tiles_leaves_empty_ranks():
    world, cam = _world()
    img, st = world.render_multi(cam, devices=[0, 1, 2, 3, 0, 1], width=16, height=8, spp=1, seed=1)   # 2 tiles, 6 ranks
    assert np.array_equal(img, _expected(16, 8, 1))
    assert [s["samples"] for s in st] == [64, 64, 0, 0, 0, 0] and st[0]["rows_through_rccl"] == 1


def test_argument_errors_come_back_as_status_codes():
    import ctypes as C
    import rtamd
    world, cam = _world()
    with pytest.raises(rtamd.RtError) as e:
        world.render_multi(cam, devices=[0, 7], width=16, height=16, spp=1)
    assert e.value.code == -9 and "rank 1" in str(e.value)                       # RT_ERR_NO_DEVICE names the rank
    with pytest.raises(rtamd.RtError):
        world.render_multi(cam, devices=[-1], width=16, height=16, spp=1)
    p = rtamd.default_params(width=16, height=16, spp=1, rank=1, world=2)        # the call partitions the frame itself
    out = np.zeros((16, 16, 3))
    rc = world.L.rt_render_multi(world.h, C.byref(cam.c), C.byref(p), 2, None, out.ctypes.data_as(C.POINTER(C.c_double)), None)
    assert rc == -1 and b"rank / world must be 0 / 1" in world.L.rt_last_error()
    p = rtamd.default_params(width=16, height=16, spp=1)
    assert world.L.rt_render_multi(world.h, C.byref(cam.c), C.byref(p), 2, None, None, None) == -1
    assert world.L.rt_render_multi(world.h, C.byref(cam.c), C.byref(p), -2, None, out.ctypes.data_as(C.POINTER(C.c_double)), None) == -1


def test_the_callers_current_device_is_restored():
    import rtamd
    world, cam = _world()
    world.render_multi(cam, devices=[3, 1], width=16, height=16, spp=1)
    img, _ = world.render_multi(cam, devices=[0], width=16, height=16, spp=1, seed=2)   # would fail the stub's "rows on the current device" check otherwise
    assert np.array_equal(img, _expected(16, 16, 2))


def test_sppm_multi_takes_the_same_path():
    world, cam = _world()
    img, st = world.render_sppm_multi(cam, devices=[1, 0, 1], width=24, height=16, spp=1, seed=4, iterations=1, photons_per_iter=10)
    assert np.array_equal(img, _expected(24, 16, 4)) and len(st) == 3


def test_resumable_render_with_host_held_state_on_the_stub():
    """rt_accum_state_doubles / rt_render_accumulate / rt_accum_finalize: state size, the copies to and from the (fake) device, the sample
    range checks and the final stitch, under ASan (a wrong size is a heap overflow there); 16 samples as 1 + 9 + 6."""
    import rtamd
    world, cam = _world()
    p = rtamd.default_params(width=52, height=28, spp=16, seed=6)
    state = None
    for a, b in [(0, 1), (1, 10), (10, 16)]:
        state, st = world.render_accumulate(cam, p, a, b, state)
        assert st["samples"] == 52 * 28 * (b - a) and state.size == 7 * 4 * 64 * 3
    assert np.array_equal(rtamd.accum_finalize(p, state), _expected(52, 28, 6))
    for a, b in [(3, 3), (-1, 4), (0, 17)]:
        with pytest.raises(rtamd.RtError):
            world.render_accumulate(cam, p, a, b, state)
    with pytest.raises(rtamd.RtError):
        rtamd.accum_finalize(rtamd.default_params(width=52, height=28, spp=16, rank=1, world=2), state)



def test_rows_travel_as_their_ranks_finish_not_after_the_join(monkeypatch):
    """RTAMD_STUB_STAGGER_MS makes rank r's (stub) render take 30 r ms: a rank's rows are handed to the exchange by its own thread the moment
    it is done (rt_stats.posted_ms = ms after the call began), so the early ranks' rows are on their way long before the last rank has
    finished -- round 4 posted all rows in one group after the join.  The exchange figure counts from the join only."""
    monkeypatch.setenv("RTAMD_STUB_STAGGER_MS", "30")
    world, cam = _world()
    img, st = world.render_multi(cam, devices=[0, 1, 2, 3], width=64, height=40, spp=1, seed=3)
    assert np.array_equal(img, _expected(64, 40, 3))
    posted = [s["posted_ms"] for s in st]
    assert posted[0] == 0.0                                         # rendered in place on the root's device
    assert 20.0 < posted[1] < posted[2] < posted[3]                 # each about when its own render ended ...
    assert posted[1] < posted[3] - 40.0 and posted[2] < posted[3] - 15.0   # ... not all together behind the slowest rank
    assert st[0]["seconds"] * 1e3 >= posted[3] and st[0]["exchange_ms"] < 25.0 and st[0]["rows_through_rccl"] == 3
    assert st[0]["stitch_copy_ms"] >= 0.0 and st[0]["comm_init_ms"] >= 0.0


def test_a_failed_exchange_is_an_error_and_its_communicators_are_not_reused(monkeypatch):
    """a send that fails inside a rank's thread comes back as that rank's status; the communicator set is destroyed, not cached (the next
    call creates a new one: comm_init_ms > 0 only then)"""
    import rtamd
    world, cam = _world()
    _, st = world.render_multi(cam, devices=[0, 2], width=32, height=16, spp=1)          # creates (or reuses) the set for [0, 2]
    _, st = world.render_multi(cam, devices=[0, 2], width=32, height=16, spp=1)
    assert st[0]["comm_init_ms"] == 0.0                                                     # cached
    monkeypatch.setenv("RTAMD_STUB_FAIL_POST", "1")
    with pytest.raises(rtamd.RtError) as e:
        world.render_multi(cam, devices=[0, 2], width=32, height=16, spp=1)
    assert "rank 1" in str(e.value) and "ncclSend failed" in str(e.value)
    monkeypatch.delenv("RTAMD_STUB_FAIL_POST")
    img, st = world.render_multi(cam, devices=[0, 2], width=32, height=16, spp=1, seed=2)
    assert np.array_equal(img, _expected(32, 16, 2)) and st[0]["comm_init_ms"] > 0.0       # a NEW set was made


def test_idle_frame_buffers_are_capped(monkeypatch):
    """rt_render_multi keeps its frame-sized device buffers between calls; a host that renders many resolutions must not pile up one set per
    shape: at most 1 GiB of idle buffers (least recently returned freed first), and rt_release_workspaces frees what is left."""
    import rtamd
    world, cam = _world()
    rtamd.lib().rt_release_workspaces()
    for k in range(9):                                               # 9 shapes x (gathered + frame + a rank's row) of ~ 3 x 48 MB each = 1.3 GB
        w = 1400 + 8 * k
        img, _ = world.render_multi(cam, devices=[0, 1], width=w, height=1400, spp=1, seed=1)
        assert img.shape == (1400, w, 3)
    freed = rtamd.lib().rt_release_workspaces()
    assert (1 << 29) < freed <= (1 << 30), freed                    # without the cap: everything, 1.3 GB
