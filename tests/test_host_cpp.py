"""The C++ host side above the C ABI (rust-raytracer_amd/host_cpp/rtamd.hpp + main.cpp = the
reference's main.rs on librtamd): builds, replays the reference's Vec3 tests against the mirrored
Vec3, walks the Cornell object graph through the builders, and (GPU) renders the same image as the
oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, scene_path

EXE = os.path.join(ROOT, "rust-raytracer_amd", "rtamd_render")


def run(*args):
    return subprocess.run([EXE, *args], capture_output=True, text=True, timeout=300)


def test_vec3_selftest_of_the_cpp_mirror():
    r = run("--vec3-selftest")
    assert r.returncode == 0 and "vec3 selftest: ok" in r.stdout, r.stdout + r.stderr


def test_cornell_graph_walk_and_scene_files():
    r = run("--describe", "--cube", scene_path("cube.obj"))
    assert r.returncode == 0 and "55 nodes (26 boxes, 2 spheres, 6 rects, 12 tris, 1 xforms)" in r.stdout, r.stdout + r.stderr
    r = run("--describe", "--scene", scene_path("scene_500.json"))
    assert r.returncode == 0 and "999 boxes, 1005 spheres" in r.stdout
    r = run("--describe", "--scene", scene_path("test.json"))
    assert r.returncode == 1 and "error -6" in r.stderr          # RT_ERR_SCHEMA, no abort
    r = run("--describe", "--cube", "/nonexistent/cube.obj")
    assert r.returncode == 1 and "error -5" in r.stderr          # "Failed to load OBJ file." -> RT_ERR_IO


@pytest.mark.gpu
def test_cpp_main_renders_the_oracles_image(tmp_path):
    import oracle
    from PIL import Image
    out = str(tmp_path / "test.png")
    r = run("--cube", scene_path("cube.obj"), "-w", "48", "-h", "48", "--spp", "8", "--seed", "1", "-o", out)
    assert r.returncode == 0 and "Msamples/s" in r.stdout, r.stdout + r.stderr
    gold = np.load(os.path.join(ROOT, "tests", "golden", "cornell_48x48_8spp_seed1.npy"))
    assert np.array_equal(np.asarray(Image.open(out)), oracle.tonemap_u8(gold))


@pytest.mark.gpu
def test_cpp_main_is_the_reference_binary_with_sppm(tmp_path):
    """main.rs:49-72 end to end through the C++ host: cornell_box_scene -> SPPMIntegrator::new -> capture_image -> save."""
    import oracle
    from PIL import Image
    out = str(tmp_path / "test.png")
    r = run("--cube", scene_path("cube.obj"), "-w", "24", "-h", "24", "--spp", "3", "--seed", "1", "--sppm", "3", "6000", "-o", out)
    assert r.returncode == 0 and "SPPM:" in r.stdout, r.stdout + r.stderr
    exp, _, _ = oracle.cornell_box_scene(scene_path("cube.obj"), 1.0, seed=1).render_sppm(24, 24, 3, iterations=3, photons_per_iter=6000,
                                                                                          k_global=100, k_caustic=50, seed=1)
    assert np.array_equal(np.asarray(Image.open(out)), oracle.tonemap_u8(exp))


@pytest.mark.gpu
def test_cpp_main_resumes_a_frame_from_its_checkpoint_file(tmp_path):
    """rtamd_render --checkpoint FILE --run-samples K (World::capture_image_resumable -> rt_render_accumulate / rt_accum_finalize): the frame
    rendered in instalments by SEPARATE processes -- 3 + 3 + 2 of 8 samples, the state in a file between them -- is the frame of one
    plain run, byte for byte; a state file of another frame is refused."""
    plain, resumed, ck = str(tmp_path / "plain.png"), str(tmp_path / "resumed.png"), str(tmp_path / "state.bin")
    common = ["--scene", scene_path("scene_10.json"), "-w", "96", "-h", "54", "--spp", "8", "--seed", "3"]
    r = run(*common, "-o", plain)
    assert r.returncode == 0, r.stderr
    for i, want in enumerate(["3 of 8", "6 of 8", "8 of 8"]):
        r = run(*common, "-o", resumed, "--checkpoint", ck, "--run-samples", "3")
        assert r.returncode == 0 and want in r.stdout, (r.stdout, r.stderr)
        assert os.path.exists(resumed) == (i == 2)
    assert open(plain, "rb").read() == open(resumed, "rb").read()
    r = run(*common[:-2], "--seed", "4", "-o", resumed, "--checkpoint", ck)
    assert r.returncode == 1 and "not the state of this frame" in r.stderr
    # ... and so is the state of the same frame parameters over ANOTHER scene (the header carries a fingerprint of the committed scene, one of
    # the camera and the library's image-spec version: ADVICE r04)
    other = ["--scene", scene_path("scene_200_no_bvh.json")] + common[2:]
    r = run(*other, "-o", resumed, "--checkpoint", ck)
    assert r.returncode == 1 and "not the state of this frame" in r.stderr and "rtamd-image-" in r.stderr
