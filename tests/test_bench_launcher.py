"""bench.py's own N-rank launcher (`python bench.py --gpus N` without a rank environment) and its roofline objects.

CPU half: the launcher must fail loudly -- never render on fewer GPUs than asked -- and the rank command must be the
driver's.  GPU half (one-GPU box): the world-size-2 branch of bench.py rehearsed end to end through the same launcher
(both ranks on device 0, gather over gloo through host memory; RCCL refuses two ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout)


def _no_gpu_here():
    import rtamd
    return rtamd.device_count() == 0


def test_gpus_2_without_devices_fails_loudly():
    if not _no_gpu_here():
        pytest.skip("box has a HIP device")
    r = _run(["--gpus", "2", "--steps", "1"])
    assert r.returncode != 0
    assert "HIP device" in r.stderr and "--gpus 2" in r.stderr
    assert r.stdout.strip() == ""      # no JSON line pretending to be a result


def test_world_size_contradicting_gpus_is_an_error():
    r = _run(["--gpus", "4", "--steps", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "contradicts WORLD_SIZE=2" in r.stderr


def test_rank_command_is_the_drivers():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.rank_command(8, ["--gpus", "8", "--steps", "3"], 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == [BENCH, "--gpus", "8", "--steps", "3"][-5:]


def test_launcher_source_never_execs_or_touches_the_gpu_before_spawning():
    src = open(BENCH).read()
    head = src.split("import torch\n")[0]          # everything that runs before the first torch import in main()
    assert "launch_ranks(" in head and "os.exec" not in src and "execv" not in src
    body = src.split("def launch_ranks")[1].split("\ndef ")[0]
    assert "torch" not in body.replace("torch.distributed.run", "") and "rtamd." not in body


def test_roofline_objects_are_physical(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    model = json.load(open(bench.MODEL))
    # (the arithmetic is what is tested here: a copy of the model stamped with this tree's source hash; that the COMMITTED model
    # belongs to the committed sources is test_committed_pmc_model_belongs_to_the_committed_kernel_sources)
    model["kernel_source_sha16"] = bench.kernel_source_sha16()
    mp = tmp_path / "model.json"
    mp.write_text(json.dumps(model))
    monkeypatch.setattr(bench, "MODEL", str(mp))
    # a launch at exactly the modelled rate: samples/s = SIMDs * clock / (insts/sample * cycles/inst) * valu_busy
    sps = bench.N_SIMDS * bench.MAX_CLOCK_GHZ * 1e9 / (model["valu_insts_per_sample"] * model["valu_issue_cycles_per_inst"]) * 0.9
    acc = {"kernel_ms": 100.0, "launches": 2, "samples": int(sps * 0.1)}
    roof, contract, hbm = bench.roofline_objects(acc, 0.1)
    assert roof["bound"] == "valu_issue" and 0.0 < roof["frac"] <= 1.05 and abs(roof["frac"] - 0.9) < 1e-6
    assert 0.0 < roof["useful_frac"] < roof["frac"] and 0.0 < roof["lane_utilisation"] <= 1.0
    assert roof["traffic"] == pytest.approx(model["hbm_bytes_per_sample"] * acc["samples"] / 2)
    assert contract["bound"] == "hbm" and contract["alg_bytes_per_sample"] > 1000
    assert hbm is not None and 0.0 < hbm["frac"] < 1.0
    # a live rate the model cannot explain is reported as such, not as a fraction above 1
    acc["kernel_ms"] = 10.0
    roof, _, _ = bench.roofline_objects(acc, 0.01)
    assert roof["frac"] is None and "stale" in roof["note"]


def test_committed_pmc_model_belongs_to_the_committed_kernel_sources():
    """the bench line's roofline carries per-sample counts over from profiles/pt_kernel_model.json: the model must have been
    measured on the device sources of this tree (bench.py reports frac = None otherwise)"""
    sys.path.insert(0, ROOT)
    import bench
    model = json.load(open(bench.MODEL))
    assert model.get("kernel_source_sha16") == bench.kernel_source_sha16(), "re-run tools/r03_headline_pmc.sh and copy its pt_kernel_model.json to profiles/"


@pytest.mark.gpu
def test_two_rank_branch_of_bench_rehearsed_on_one_gpu():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--width", "96", "--height", "72", "--spp", "8", "--cpu-spp", "0"],
             {"RTAMD_BENCH_REHEARSE": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and "rehearsal" in out
    assert out["value"] > 0 and "cpu_baseline" not in out
    # the per-rank record that makes an N-GPU line diagnosable: tiles, samples, kernel time, gather + stitch time, loop wall time
    assert [r["rank"] for r in out["ranks"]] == [0, 1] and sum(r["tiles"] for r in out["ranks"]) == 12 * 9
    assert sum(r["samples"] for r in out["ranks"]) == 96 * 72 * 8 and all(r["kernel_ms"] > 0 and r["loop_wall_s"] > 0 for r in out["ranks"])
    assert all("exchange_ms" in r for r in out["ranks"]) and out["ranks"][0]["exchange_ms"] > 0
    # without the rehearsal switch the same command must refuse to run two ranks on one device
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--width", "96", "--height", "72", "--spp", "8", "--cpu-spp", "0"])
    import rtamd
    if rtamd.device_count() < 2:
        assert r.returncode != 0 and "only 1 HIP device" in r.stderr


def _plain_frame(width, height, spp):
    import rtamd
    world, cam = rtamd.load_scene_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene_500.json"))
    img, _ = world.render(cam, width=width, height=height, spp=spp, seed=1)
    return img


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["flag", "torchrun_env"])
def test_forced_process_group_runs_the_rccl_exchange_at_world_1(how, tmp_path):
    """The N-rank path's RCCL calls executed on the one GPU there is: init_process_group("nccl"), dist.gather of the f64 device
    rows into the receive buffer, barrier, all_reduce(MAX), then the device stitch -- in a child process that starts before this
    one touches the GPU for it.  The stitched frame must equal the plain render bit for bit (camera.rs:77,108,115-123)."""
    import numpy as np
    out_npy = str(tmp_path / "frame.npy")
    args = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--width", "96", "--height", "72", "--spp", "8", "--cpu-spp", "2" if how == "flag" else "0",
            "--frame-out", out_npy]
    env = {}
    if how == "flag":
        args.append("--force-pg")
    else:  # the environment torch.distributed.run gives a rank
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = {"RTAMD_BENCH_FORCE_PG": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
    env_all = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "RTAMD_BENCH_FORCE_PG"):
        env_all.pop(k, None)
    env_all.update(env)
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env_all, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    pg = out["process_group"]
    assert pg["backend"].startswith("nccl") and pg["world"] == 1 and pg["forced_at_world_1"] is True
    assert out["n_gpus"] == 1 and out["value"] > 0
    # the bench contract: one JSON line with these keys
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "roofline_contract"):
        assert k in out, k
    assert out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True and out["vs_baseline"] is None and out["dtype"] == "f64"
    assert "workload" in out["config"] and "model" not in out["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "valu_busy_measured", "cost_table"):
        assert k in out["roofline"], k
    assert "measured_classes" in out["roofline"]["cost_table"] and "other_class" in out["roofline"]["cost_table"]
    assert len(out["ranks"]) == 1 and out["ranks"][0]["tiles"] == 12 * 9 and out["ranks"][0]["samples"] == 96 * 72 * 8 * 2
    assert out["ranks"][0]["kernel_ms"] > 0 and out["ranks"][0]["exchange_ms"] > 0 and "all_gather" in pg["calls"]
    if how == "flag":
        cb = out["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb and "cpu_model" in cb and cb["reference_default"]["cores"] <= 8
    assert len([ln for ln in r.stdout.splitlines() if ln.strip()]) == 1      # ONE line on stdout (RCCL's banner goes to stderr)
    got = np.load(out_npy)
    assert got.shape == (72, 96, 3) and np.array_equal(got, _plain_frame(96, 72, 8))


@pytest.mark.gpu
def test_single_process_mode_renders_through_rt_render_multi(tmp_path):
    """`bench.py --single-process --devices 0,0`: the frame through ONE rt_render_multi call per step (two ranks on the one device there
    is), same JSON contract, frame bit-identical to the plain render."""
    import numpy as np
    out_npy = str(tmp_path / "frame.npy")
    r = _run(["--single-process", "--devices", "0,0", "--steps", "2", "--warmup", "1", "--width", "96", "--height", "72", "--spp", "8", "--cpu-spp", "0",
              "--frame-out", out_npy])
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["launch"] == "single-process" and out["n_gpus"] == 1 and out["value"] > 0 and out["rccl_version"] >= 20000
    assert [x["device"] for x in out["ranks"]] == [0, 0] and sum(x["samples"] for x in out["ranks"]) == 96 * 72 * 8 * 2
    assert "rehearsal" in out and out["roofline"]["frac"] is None     # ranks that share a device: the line says it is no utilisation figure
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert np.array_equal(np.load(out_npy), _plain_frame(96, 72, 8))
