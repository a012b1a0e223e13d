"""rt_render_multi on the one-GPU box (-m gpu): the frame across "N GPUs" behind the C ABI, with the real HIP kernels and the real
RCCL.  One device is all there is here, so the N-rank path is exercised two ways:
  * a device list that repeats ordinal 0 -- N host threads, N tile partitions, N slots of the gathered buffer, one stitch: the
    fan-out, partition and stitch logic of the library with the kernels rendering (rows on the root's device never travel);
  * rt_tuning.multi_force_rccl -- the same rows sent through the communicator of ncclCommInitAll (one rank on this box) with grouped
    ncclSend / ncclRecv, a rank sending to itself: every RCCL call of the N-GPU exchange executes.
Every frame must equal rt_render's bit for bit (and the oracle's).  N distinct devices are the driver's 8-GPU node; the host logic
for them runs under ASan on fake devices in tests/test_multi_stub.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, scene_path

pytestmark = pytest.mark.gpu


def _scene10():
    import rtamd
    world, cam = rtamd.load_scene_file(scene_path("scene_10.json"))
    return world, cam.with_aspect(16.0 / 9.0)


def test_librtamd_is_linked_with_rccl():
    import rtamd
    v = rtamd.lib().rt_rccl_version()
    assert v >= 20000, v                                           # ncclGetVersion of the linked RCCL (2.26.x -> 226xx)
    out = subprocess.run(["readelf", "-d", os.path.join(ROOT, "rust-raytracer_amd", "librtamd.so")], capture_output=True, text=True).stdout
    assert "librccl.so" in out


@pytest.mark.parametrize("devices,force", [([0], 1), ([0, 0, 0], 0), ([0, 0, 0], 1), ([0] * 8, 1)])
def test_multi_frame_equals_the_single_device_frame_and_the_oracle(devices, force, tuning):
    import oracle
    world, cam = _scene10()
    ref, _ = world.render(cam, width=100, height=57, spp=6, seed=1)            # ragged: 13 x 8 tiles, partial right / bottom tiles
    tuning(multi_force_rccl=force)
    img, st = world.render_multi(cam, devices=devices, width=100, height=57, spp=6, seed=1)
    assert np.array_equal(img, ref)
    exp, _ = oracle.load_scene_file(scene_path("scene_10.json"), aspect=16.0 / 9.0).render(100, 57, 6, seed=1)
    assert np.array_equal(img, exp)
    assert len(st) == len(devices) and sum(s["samples"] for s in st) == 100 * 57 * 6
    assert all(s["kernel_ms"] > 0 for s in st)
    assert st[0]["rows_through_rccl"] == (len(devices) if force else 0)


def test_multi_cornell_with_the_mixture_integrator_and_both_camera_forms(tuning):
    import rtamd
    world, cam = rtamd.select_scene(scene_path("cube.obj"), 1.0, 1)
    ref, _ = world.render(cam, width=48, height=48, spp=8, seed=1, integrator=1)
    tuning(multi_force_rccl=1)
    img, st = world.render_multi(cam, devices=[0, 0], width=48, height=48, spp=8, seed=1, integrator=1)
    assert np.array_equal(img, ref) and st[0]["rows_through_rccl"] == 2
    # the host that owns a constructed Camera hands over its stored frame (what the Rust capture_image does)
    frame = rtamd.rt_camera_frame()
    assert world.L.rt_camera_frame_from(C.byref(cam.c), C.byref(frame)) == 0
    p = rtamd.default_params(width=48, height=48, spp=8, seed=1, integrator=1)
    out = np.zeros((48, 48, 3))
    stats = (rtamd.rt_stats * 3)()
    ids = (C.c_int * 3)(0, 0, 0)
    assert world.L.rt_render_multi_camera_frame(world.h, C.byref(frame), C.byref(p), 3, ids, out.ctypes.data_as(C.POINTER(C.c_double)), stats) == 0
    assert np.array_equal(out, ref)


def test_multi_sppm_equals_the_single_device_sppm(tuning):
    import rtamd
    world, cam = rtamd.select_scene(scene_path("cube.obj"), 1.0, 1)
    ref, _, _, _ = world.render_sppm(cam, width=24, height=24, spp=3, seed=1, iterations=3, photons_per_iter=6000)
    assert np.isfinite(ref).mean() > 0.9
    tuning(multi_force_rccl=1)
    img, st = world.render_sppm_multi(cam, devices=[0, 0, 0], width=24, height=24, spp=3, seed=1, iterations=3, photons_per_iter=6000)
    assert np.array_equal(img, ref, equal_nan=True) and len(st) == 3     # (under-filled photon maps leave NaN estimates, in the reference too)


def test_multi_errors_are_status_codes():
    import rtamd
    world, cam = _scene10()
    with pytest.raises(rtamd.RtError) as e:
        world.render_multi(cam, devices=[0, rtamd.device_count()], width=16, height=16, spp=1)
    assert e.value.code == -9 and "rank 1" in str(e.value)
    with pytest.raises(rtamd.RtError) as e:                                     # a rank's own failure comes back with its rank
        world.render_multi(cam, devices=[0, 0], width=16, height=16, spp=1, kernel=5)
    assert e.value.code == -10 and "rank" in str(e.value)
    img, _ = world.render_multi(cam, gpus=0, width=16, height=16, spp=1)        # 0 = every visible device
    assert np.array_equal(img, world.render(cam, width=16, height=16, spp=1)[0])


def test_release_workspaces_also_drops_the_idle_communicators(tuning):
    import rtamd
    world, cam = _scene10()
    tuning(multi_force_rccl=1)
    a, _ = world.render_multi(cam, devices=[0, 0], width=32, height=24, spp=2)
    assert rtamd.release_workspaces() > 0
    b, _ = world.render_multi(cam, devices=[0, 0], width=32, height=24, spp=2)   # a new communicator is made
    assert np.array_equal(a, b)


def test_cpp_host_renders_across_devices_in_one_capture_image(tmp_path):
    """rtamd_render --devices 0,0,0 = main.rs:52-54 with the N-GPU frame inside the one capture_image call."""
    from PIL import Image
    exe = os.path.join(ROOT, "rust-raytracer_amd", "rtamd_render")
    one, many = str(tmp_path / "one.png"), str(tmp_path / "many.png")
    args = ["--cube", scene_path("cube.obj"), "-w", "48", "-h", "48", "--spp", "8", "--seed", "1"]
    r1 = subprocess.run([exe, *args, "-o", one], capture_output=True, text=True, timeout=300)
    r2 = subprocess.run([exe, *args, "--devices", "0,0,0", "-o", many], capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and r2.returncode == 0, r1.stderr + r2.stderr
    assert "rank 2:" in r2.stdout and "rows through RCCL" in r2.stdout
    assert np.array_equal(np.asarray(Image.open(one)), np.asarray(Image.open(many)))
    r3 = subprocess.run([exe, *args, "--gpus", "0", "-o", many], capture_output=True, text=True, timeout=300)     # all visible devices
    assert r3.returncode == 0 and np.array_equal(np.asarray(Image.open(one)), np.asarray(Image.open(many)))
    r4 = subprocess.run([exe, *args, "--devices", "0,9", "-o", many], capture_output=True, text=True, timeout=300)
    assert r4.returncode == 1 and "error -9" in r4.stderr


def test_multi_stats_tell_upload_init_render_exchange_and_copy_apart(tuning):
    """rt_stats (ABI version 2) of one rt_render_multi call: the scene upload of the first render on a device, the creation of the
    communicators on the first call with a device list (and after rt_release_workspaces), when each rank's rows were handed to RCCL --
    by the rank's own thread, right after its render --, the exchange proper (last rank's finish -> rows on the root) and the stitch +
    host copy.  What a first run on N > 1 devices needs to tell init cost from render from exchange."""
    import rtamd
    rtamd.lib().rt_release_workspaces()                              # drops cached communicators: the next forced exchange creates them
    world, cam = rtamd.load_scene_file(scene_path("scene_10.json"))  # a fresh scene: its first render uploads it
    tuning(multi_force_rccl=1)
    img, st = world.render_multi(cam, devices=[0, 0, 0], width=96, height=64, spp=4, seed=2)
    assert sum(1 for s in st if s["upload_ms"] > 0.0) == 1
    assert st[0]["comm_init_ms"] > 0.0                               # ncclCommInitAll happened in this call, before the renders
    assert all(s["posted_ms"] > 0.0 for s in st)                     # every rank's rows went through the exchange, each posted by its rank
    assert all(s["posted_ms"] <= st[0]["seconds"] * 1e3 for s in st)
    assert st[0]["exchange_ms"] >= 0.0 and abs(st[0]["exchange_seconds"] * 1e3 - st[0]["exchange_ms"]) < 1e-2
    assert st[0]["stitch_copy_ms"] > 0.0 and st[0]["rows_through_rccl"] == 3
    img2, st2 = world.render_multi(cam, devices=[0, 0, 0], width=96, height=64, spp=4, seed=2)
    assert np.array_equal(img, img2)
    assert st2[0]["comm_init_ms"] == 0.0 and all(s["upload_ms"] == 0.0 for s in st2)   # cached communicators, scene already on the device
