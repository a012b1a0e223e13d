#!/usr/bin/env python3
"""bench.py -- headline benchmark of the radiance path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

Workload (config.workload): data/scene_500.json, 1200x1200, 1000 spp, depth 50, seed 1 -- the
configuration BASELINE.json's metric is quoted on.  One "step" = one full frame = 1.44e9
pixel-samples through the HIP path: rt_render_tiles_device on every rank's tiles, then the
framebuffer gather to rank 0 (RCCL) and the stitch.  The scene, camera and all buffers are
resident in HBM before the timed region.  value = W*H*spp*K / max-over-ranks wall time.

N > 1: one rank per GPU; the frame is dealt to ranks by 8x8 tile (tile t -> rank t % N), total work
fixed => "scaling": "strong".  Started either by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE
in the environment) or directly: `python bench.py --gpus N` starts the N ranks itself, as child
processes, BEFORE anything in this process has touched the GPU, and exits with their status.  Fewer
than N visible devices, or a WORLD_SIZE that contradicts --gpus, is an error -- never a silent
single-GPU run.

Extra objects on the JSON line (definitions and formulas: DESIGN.md s5):
  roofline          -- the bound that binds pt_kernel: VALU issue.  achieved = VALU wave-instructions/s,
                       peak = SIMDs x max clock / issue cycles per instruction of the kernel's own mix, class costs MEASURED on
                       this GPU (tools/microbench/valu_cost.hip: f32 fma 2.5, f64 4, compare / select / integer / min / max 3.5,
                       transcendental 7 / 14 cycles per wave64 instruction); lane_utilisation beside it.
                       Per-sample instruction counts come from the PMC passes committed under profiles/ (model
                       file named in `source`); the kernel time is measured live (HIP events on the launch stream).
  roofline_contract -- SURVEY s8d: algorithmic bytes per sample in the REFERENCE's traversal order / kernel time
                       against the 8 TB/s HBM peak.  The scene is LDS-resident, so this is not a physical fraction.
  roofline_hbm      -- physical HBM traffic (PMC, carried over) / kernel time against the HBM peak.
  cpu_baseline      -- the oracle (a C++ port of the reference's CPU path; the Rust binary cannot be
                       built here) on this box's host cores, bounded sample, rank 0 at N=1 only.
  ranks             -- one entry per rank (all_gather at the end of the timed region): tiles owned, samples, summed pt_kernel time
                       (HIP events inside librtamd), time of gather + stitch (HIP events on the step's stream; the RCCL gather is
                       ordered against it by torch), wall time of the rank's timed loop.  Lets an N-GPU line tell imbalance from
                       exchange from launch cost.  Present at every world size (one entry at world 1).

`--single-process` renders the same frames through rt_render_multi instead (ONE process, one host thread per GPU inside librtamd,
RCCL gather linked into the library -- what a Rust / C++ host calls); same JSON line, "launch": "single-process".  The default,
and what the driver starts, is one process per GPU over torch.distributed.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))

SCENE = os.path.join(ROOT, "tests", "golden", "scenes", "scene_500.json")
ALG_BYTES = os.path.join(ROOT, "tests", "golden", "alg_bytes_scene_500.json")
MODEL = os.path.join(ROOT, "profiles", "pt_kernel_model.json")
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_SIMDS = 256 * 4      # 256 CUs x 4 SIMDs
MAX_CLOCK_GHZ = 2.4    # MI355X max engine clock (MI355X_MICROARCH.md): the roofline PEAK is priced at it, not at the clock a profiled pass happened to hold
CPU_BASELINE_THREADS = 16


def cpu_model():
    """CPU model string of the box (BASELINE.md s3 asks for it beside the baseline)."""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(width, height, spp_cpu, seed):
    """Oracle timed on the host cores (test infrastructure used as the reported CPU baseline).  Two points, as BASELINE.md s3
    asks: the reference's own default `n_workers = 8` (main.rs:42) and every core this box gives one GPU (16)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    # the GPU box gives one GPU's share of the host (16 threads, see the task's process guard); never more
    # threads than that even though the host shows 256
    cores = min(len(os.sched_getaffinity(0)), CPU_BASELINE_THREADS)
    sc = oracle.load_scene_file(SCENE)

    def timed(spp, workers):
        t0 = time.perf_counter()
        sc.render(width, height, spp, seed=seed, n_jobs=64, n_workers=workers)  # 64 row bands = camera.rs:79
        dt = time.perf_counter() - t0
        return width * height * spp / dt / 1e6, dt

    v_all, dt_all = timed(spp_cpu, cores)
    ref_workers = min(8, cores)
    spp_ref = max(1, spp_cpu // 2)
    v_ref, dt_ref = timed(spp_ref, ref_workers)
    return {
        "value": v_all, "unit": "Msamples/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
        "sample": "scene_500 %dx%d at %d spp (%.1f s), 64 row-band jobs on %d threads; C++ restatement of the "
                  "reference, not the Rust binary" % (width, height, spp_cpu, dt_all, cores),
        "reference_default": {"value": v_ref, "unit": "Msamples/s", "cores": ref_workers,
                              "sample": "same frame at %d spp (%.1f s) with the reference's default n_workers = 8 (main.rs:42)" % (spp_ref, dt_ref)},
    }


def visible_devices():
    """HIP devices visible to librtamd, counted in a CHILD process: the launcher itself must never initialise the GPU."""
    code = "import sys; sys.path.insert(0, %r); import rtamd; print(rtamd.device_count())" % os.path.join(ROOT, "rust-raytracer_amd")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("bench.py: cannot load librtamd.so to count devices:\n" + r.stderr[-2000:])
    return int(r.stdout.strip().splitlines()[-1])


def rank_command(n_ranks, argv, port):
    """The command the driver uses for N > 1 (one rank per GPU over torch.distributed.run)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n_ranks, "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no rank environment: start the N ranks as children and return their status.
    Nothing here initialises the GPU runtime (no framework import, no HIP call), and no process image is replaced."""
    n_dev = visible_devices()
    rehearse = os.environ.get("RTAMD_BENCH_REHEARSE") == "1"
    if n_dev < 1:
        raise SystemExit("bench.py needs a HIP device: the product has no CPU fallback (0 devices visible, --gpus %d)" % args.gpus)
    if n_dev < args.gpus and not rehearse:
        raise SystemExit("bench.py: --gpus %d but only %d HIP device(s) visible; refusing to render on fewer GPUs than asked "
                         "(RTAMD_BENCH_REHEARSE=1 rehearses the N-rank path on one device over gloo)" % (args.gpus, n_dev))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    rc = 1
    for attempt in range(3):
        # a free port found by bind / close can be taken by another process before the rendezvous binds it (parallel jobs on one
        # box): the ranks' stderr is passed through and, when the launch died on "address already in use", another port is tried
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        p = subprocess.Popen(rank_command(args.gpus, argv, port), env=env, stderr=subprocess.PIPE, text=True)
        tail = []
        for line in p.stderr:
            sys.stderr.write(line)
            tail.append(line)
            del tail[:-400]
        rc = p.wait()
        if rc == 0 or not any("ddress already in use" in ln or "EADDRINUSE" in ln for ln in tail):
            break
    return rc


def kernel_source_sha16():
    """hash of the device sources of this tree (tools/make_pt_model.py stores the same hash in a PMC model file)"""
    import hashlib
    root = os.path.join(ROOT, "rust-raytracer_amd", "csrc")
    h = hashlib.sha256()
    for rel in ("device/kernels.hip", "device/wavefront.inc", "device/sppm.inc", "device/device.h", "common/flat.h", "common/rng.h", "common/detlog.h", "common/schedule.h", "host/schedule.cpp"):
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def roofline_objects(stats_acc, dt_kernel_s, clock_note=None, model_path=None, alg_path=None):
    """roofline / roofline_contract / roofline_hbm for pt_kernel (see the module docstring and DESIGN.md s5).  model_path / alg_path:
    another configuration's PMC model and algorithmic-bytes fixture (tools/config_bench.py); default: the bench workload's."""
    model_path = model_path or MODEL
    alg_path = alg_path or ALG_BYTES
    launches = max(1, stats_acc["launches"])
    samples_per_launch = stats_acc["samples"] / launches
    ms_per_launch = stats_acc["kernel_ms"] / launches
    sps = stats_acc["samples"] / (stats_acc["kernel_ms"] * 1e-3) if stats_acc["kernel_ms"] > 0 else 0.0  # samples/s inside pt_kernel
    with open(alg_path) as f:
        b_alg = json.load(f)["bytes_per_sample"]
    contract = {"bound": "hbm", "achieved": sps * b_alg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sps * b_alg / 1e9 / HBM_PEAK_GBS,
                "alg_bytes_per_sample": b_alg,
                "note": "SURVEY s8d contract figure: bytes the REFERENCE's traversal order would touch; served from LDS here and mostly "
                        "never touched (near-first SAH traversal), so frac > 1 is expected and is not a physical utilisation"}
    model = None
    if os.path.exists(model_path):
        with open(model_path) as f:
            model = json.load(f)
    roof = {"bound": "valu_issue", "achieved": None, "peak": None, "unit": "G wave-instructions/s", "frac": None, "traffic": None,
            "kernel": "pt_kernel", "samples_per_launch": samples_per_launch, "ms_per_launch": ms_per_launch}
    hbm = None
    if model:
        ipc = model["valu_insts_per_sample"]            # VALU wave-instructions per sample (SQ_INSTS_VALU / samples)
        cyc = model["valu_issue_cycles_per_inst"]       # minimum SIMD cycles per wave64 instruction of the kernel's own mix (measured class costs: tools/make_pt_model.py)
        clk = MAX_CLOCK_GHZ                             # (the PMC pass itself held model["clock_ghz"], 2.37: profiled passes clock lower)
        achieved = sps * ipc / 1e9
        peak = N_SIMDS * clk / cyc
        frac = achieved / peak if peak > 0 else None
        roof.update({"achieved": achieved, "peak": peak, "frac": frac, "lane_utilisation": model["lane_utilisation"],
                     # the counter that does not depend on any cycle table: 4 * SQ_ACTIVE_INST_VALU / SIMD cycles of the PMC pass
                     "valu_busy_measured": model.get("valu_busy_measured"),
                     "cost_table": {"cycles_per_wave64_instruction": model.get("valu_class_cycles"),
                                    "anchor": model.get("valu_cost_anchor", "rounds 3-4: ratios anchored at f64 fma = 4 cycles"),
                                    "measured_classes": "GRBM cycles per wave-instruction per SIMD at 4 waves per SIMD, 2.38 GHz held: f32 add/mul/fma 2.56, f64 add/mul/fma/ldexp/div_fixup 4.25, "
                                                        "conversions 4.25, v_mul_lo_u32 4.26, integer add/xor/and/or/shift/alignbit/lshl_add/bfe 3.65, compare + select pair 3.67 each, "
                                                        "lone v_cmp / v_cndmask_e64 / f32 min/max / min3/max3 / v_readlane / v_writelane 4.25, v_mov 2.55, 64-bit add 2 x 4.29, "
                                                        "f32 rcp/rsq/sqrt 8.28, f64 rcp/rsq/sqrt 16.3; one wave per SIMD: 5.1 for all but the transcendentals "
                                                        "(tools/microbench/valu_cost.hip, profiles/r05/valu_cost_microbench.txt)",
                                    "other_class": "the model's 'other' (SQ_INSTS_VALU minus the typed counters: compares, selects, min/max, logic, moves, lane moves) is "
                                                   "priced at 3.65 cycles, its cheapest members besides v_mov (2.55); compares / selects / min-max cost 4.25: the class's true "
                                                   "cost is higher, the peak lower and frac higher than stated"},
                     "useful_frac": frac * model["lane_utilisation"] if frac else None,
                     "valu_insts_per_sample": ipc, "valu_issue_cycles_per_inst": cyc, "clock_ghz": clk, "clock_ghz_in_pmc_pass": model["clock_ghz"],
                     "valu_mix_per_sample": model.get("valu_mix_per_sample"),
                     "source": "per-sample counts carried over from %s; kernel time measured in this run" % model.get("source", "profiles/")})
        if model.get("kernel_source_sha16") not in (None, kernel_source_sha16()):
            roof.update({"frac": None, "useful_frac": None,
                         "note": "the PMC model was measured on other device sources (%s) than this tree's (%s): per-sample counts not carried over" %
                                 (model.get("kernel_source_sha16"), kernel_source_sha16())})
        elif frac is not None and not (0.0 < frac <= 1.05):
            roof.update({"frac": None, "useful_frac": None,
                         "note": "live kernel rate and the PMC model disagree (frac %.3f): the model file is stale for this build" % frac})
        bps = model.get("hbm_bytes_per_sample")
        if bps is not None:
            roof["traffic"] = bps * samples_per_launch  # bytes per launch, PMC (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), carried over
            hbm = {"bound": "hbm", "achieved": sps * bps / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sps * bps / 1e9 / HBM_PEAK_GBS,
                   "hbm_bytes_per_sample": bps, "source": "carried over from %s" % model.get("source", "profiles/")}
    return roof, contract, hbm


def single_process(args):
    """`--single-process`: the same frames through rt_render_multi -- what a Rust / C++ host's one capture_image call does.  The timed
    region per step is the whole call: fan-out over the devices, render, RCCL gather of the rows to devices[0], stitch, copy to the host
    (34.6 MB: the PCIe-inclusive figure; the per-GPU-process mode above keeps the frame on the device)."""
    import rtamd
    n_dev = rtamd.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs a HIP device: the product has no CPU fallback")
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if len(devices) != args.gpus and args.devices:
        args.gpus = len(devices)
    if max(devices) >= n_dev:
        raise SystemExit("bench.py: device ordinal %d asked for, %d HIP device(s) visible" % (max(devices), n_dev))
    world, cam = rtamd.load_scene_file(SCENE)
    kw = dict(devices=devices, width=args.width, height=args.height, spp=args.spp, max_depth=50, t_min=1e-3, seed=args.seed, kernel=args.kernel)
    first_st = None   # the first call: scene uploads (one per device) and the creation of the communicators
    for _ in range(args.warmup):
        _, stw = world.render_multi(cam, **kw)
        first_st = first_st or stw
    acc = [{"kernel_ms": 0.0, "samples": 0, "launches": 0, "posted_ms": 0.0} for _ in devices]
    exch = 0.0
    stitch = 0.0
    rows = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img, st = world.render_multi(cam, **kw)
        first_st = first_st or st
        for a, s in zip(acc, st):
            a["kernel_ms"] += s["kernel_ms"]; a["samples"] += s["samples"]; a["launches"] += s["launches"]; a["posted_ms"] += s["posted_ms"]
        exch += st[0]["exchange_seconds"]
        stitch += st[0]["stitch_copy_ms"]
        rows = st[0]["rows_through_rccl"]
    dt = time.perf_counter() - t0
    total = args.width * args.height * args.spp * args.steps
    slowest = max(acc, key=lambda a: a["kernel_ms"])
    roof, contract, hbm = roofline_objects(slowest, dt)
    out = {"metric": "Msamples/sec (px*spp), scene_500 %dx%d %dspp" % (args.width, args.height, args.spp), "value": total / dt / 1e6, "unit": "Msamples/s",
           "n_gpus": len(set(devices)), "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64",
           "data": "the reference's own scene file data/scene_500.json (committed copy, minified); no dataset or checkpoint involved",
           "config": {"workload": "tests/golden/scenes/scene_500.json: 1005 spheres, 999-node file BVH, %dx%d, %d spp, depth 50, seed %d" % (args.width, args.height, args.spp, args.seed),
                      "parallelism": "rt_render_multi: image tiles 8x8 dealt round-robin to %d rank(s) on HIP devices %s, one host thread per rank, RCCL gather inside librtamd" % (len(devices), devices)},
           "launch": "single-process", "wall_s": dt, "rccl_version": rtamd.lib().rt_rccl_version(), "rows_through_rccl_per_step": rows,
           "includes": "host copy of the stitched frame (PCIe) in every step",
           "devices": devices, "comm_init_ms_first_call": first_st[0]["comm_init_ms"],
           "exchange_ms_per_step": exch * 1e3 / max(1, args.steps), "stitch_and_host_copy_ms_per_step": stitch / max(1, args.steps),
           "ranks": [{"rank": i, "device": d, "samples": a["samples"], "kernel_ms": a["kernel_ms"], "kernel_ms_per_step": a["kernel_ms"] / max(1, args.steps),
                      "rows_posted_ms_after_call_began_per_step": a["posted_ms"] / max(1, args.steps), "scene_upload_ms": first_st[i]["upload_ms"]}
                     for i, (d, a) in enumerate(zip(devices, acc))],
           "exchange_stitch_copy_ms_per_step": exch / max(1, args.steps) * 1e3,
           "roofline": roof, "roofline_contract": contract}
    if hbm is not None:
        out["roofline_hbm"] = hbm
    if len(set(devices)) < len(devices):  # ranks that share a device queue behind each other: a rank's kernel time then contains its neighbours'
        out["rehearsal"] = "%d ranks on %d device(s): kernel times of ranks that share a device overlap; the roofline objects are not utilisations, NOT a scaling measurement" % (len(devices), len(set(devices)))
        out["roofline"]["frac"] = None
    if args.frame_out:
        import numpy as np
        np.save(args.frame_out, img)
    print(json.dumps(out))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--kernel", type=int, default=0, help="0 = library default; 1 / 2 / 5 force a traversal (A/B runs only)")
    ap.add_argument("--cpu-spp", type=int, default=32, help="spp of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--force-pg", action="store_true",
                    help="world 1 only: create the RCCL process group anyway and drive the N-rank exchange through it (init, gather of "
                         "the f64 device rows, barrier, all_reduce(MAX)); also RTAMD_BENCH_FORCE_PG=1")
    ap.add_argument("--frame-out", default=None, help="rank 0 writes the stitched f64 frame [H, W, 3] of the last step to this .npy file")
    ap.add_argument("--single-process", action="store_true",
                    help="render through rt_render_multi: one process, one host thread per GPU inside librtamd, RCCL gather linked into the library")
    ap.add_argument("--devices", default=None, help="--single-process only: comma-separated HIP ordinals, one per rank (may repeat; default 0..gpus-1)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    if args.single_process:
        sys.exit(single_process(args))
    ws_env = os.environ.get("WORLD_SIZE")
    if ws_env is None:
        if args.gpus > 1:
            sys.exit(launch_ranks(args, sys.argv[1:]))
        world_size = 1
    else:
        world_size = int(ws_env)
        if world_size != args.gpus:
            raise SystemExit("bench.py: --gpus %d contradicts WORLD_SIZE=%d" % (args.gpus, world_size))

    import torch
    import rtamd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearse = os.environ.get("RTAMD_BENCH_REHEARSE") == "1"  # all ranks on device 0, gather over gloo through host memory
    n_dev = rtamd.device_count()
    if not torch.cuda.is_available() or n_dev < 1:
        raise SystemExit("bench.py needs a HIP device: the product has no CPU fallback")
    if n_dev < world_size and not rehearse:
        raise SystemExit("bench.py: %d ranks but only %d HIP device(s) visible" % (world_size, n_dev))
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)  # before the process group: RCCL binds the communicator to the current device
    dev = torch.device("cuda", dev_index)
    dist = None
    pg_init_s = 0.0
    force_pg = world_size == 1 and (args.force_pg or os.environ.get("RTAMD_BENCH_FORCE_PG") == "1")
    backend = None
    json_fd = None
    if world_size > 1 or force_pg:
        # RCCL prints its version banner on stdout when the communicator is created (NCCL_DEBUG=VERSION/WARN on these boxes):
        # everything the libraries write to fd 1 goes to stderr, and the ONE JSON line goes to the real stdout at the end
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_pg and "MASTER_PORT" not in os.environ:  # plain `python bench.py --force-pg`: a one-rank rendezvous of its own
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            s.close()
        backend = "gloo" if rehearse else "nccl"
        t_pg = time.perf_counter()
        dist.init_process_group(backend=backend, rank=rank, world_size=world_size)
        pg_init_s = time.perf_counter() - t_pg

    world, cam = rtamd.load_scene_file(SCENE)
    params = rtamd.default_params(width=args.width, height=args.height, spp=args.spp, max_depth=50, t_min=1e-3, seed=args.seed,
                                  rank=rank, world=world_size, kernel=args.kernel, device=dev_index)
    from rtamd.distributed import TileLayout, TileGather
    layout = TileLayout(args.width, args.height, world_size)
    p0 = rtamd.default_params(width=args.width, height=args.height, spp=args.spp, rank=0, world=world_size)
    stride = layout.stride  # rank 0 owns the most tiles; every rank pads to it for the gather
    assert stride == rtamd.tiles_owned(p0)
    d_tiles = torch.zeros(stride * 64 * 3, dtype=torch.float64, device=dev)
    frame = torch.zeros(args.height * args.width * 3, dtype=torch.float64, device=dev) if rank == 0 else None
    gather = TileGather(layout, rank, dist, d_tiles, dst=0, host_staged=rehearse, force=force_pg)  # receive buffer allocated once, outside the loop
    stream = torch.cuda.current_stream().cuda_stream

    stats_acc = {"kernel_ms": 0.0, "launches": 0, "samples": 0}
    first = {}   # the stats of this rank's first render: the one that uploads the scene to its device (rt_stats.upload_ms)
    last = {}
    ex_events = []  # (before the gather, after the stitch) per timed step, on the step's stream

    def step(timed):
        st = world.render_tiles_device(cam, params, d_tiles.data_ptr(), stream)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        # the stitch of camera.rs:115-123 across GPUs: ONE framebuffer gather to rank 0 over RCCL/xGMI
        src = gather(d_tiles)
        if rank == 0:
            rtamd.assemble_frame_device(p0, src.data_ptr(), stride, frame.data_ptr(), stream)
        if timed:
            e1.record()
            ex_events.append((e0, e1))
        if timed:
            stats_acc["kernel_ms"] += st["kernel_ms"]
            stats_acc["launches"] += st["launches"]
            stats_acc["samples"] += st["samples"]
        last.update(st)
        if not first:
            first.update(st)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    sync()
    dt = time.perf_counter() - t0
    # per-rank record: [tiles owned, samples, pt_kernel ms (HIP events in librtamd), gather + stitch ms (HIP events), wall s of the timed loop]
    # ... device ordinal, scene upload ms (first render of the process), process-group init s
    mine = [float(layout.owned(rank)), float(stats_acc["samples"]), stats_acc["kernel_ms"], sum(a.elapsed_time(b) for a, b in ex_events), dt,
            float(dev_index), float(first.get("upload_ms", 0.0)), pg_init_s]
    per_rank = [mine]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        rec = torch.tensor(mine, dtype=torch.float64, device="cpu" if rehearse else dev)
        recs = [torch.zeros_like(rec) for _ in range(world_size)]
        dist.all_gather(recs, rec)
        per_rank = [[float(x) for x in r.cpu()] for r in recs]

    if rank == 0:
        total = args.width * args.height * args.spp * args.steps
        value = total / dt / 1e6
        roof, contract, hbm = roofline_objects(stats_acc, dt)
        kernel_names = {1: "reference-order stackless", 2: "SAH-BVH2 accel, f32 conservative boxes"}
        out = {
            "metric": "Msamples/sec (px*spp), scene_500 %dx%d %dspp" % (args.width, args.height, args.spp),
            "value": value, "unit": "Msamples/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "the reference's own scene file data/scene_500.json (committed copy, minified); no dataset or checkpoint involved",
            "config": {"workload": "tests/golden/scenes/scene_500.json (data/scene_500.json of the reference, minified): 1005 spheres, "
                                   "999-node file BVH, %dx%d, %d spp, depth 50, seed %d" % (args.width, args.height, args.spp, args.seed),
                       "parallelism": "image tiles 8x8 dealt round-robin to %d GPU(s), RCCL framebuffer gather" % world_size,
                       "kernel": "pt_kernel<%s>(f64 primitives, scene %s)" % (kernel_names.get(last.get("kernel_used"), "?"),
                                                                              "in LDS" if last.get("scene_in_lds") else "in L2/HBM"),
                       "block_threads": last.get("block_threads"), "grid_blocks": last.get("grid_blocks"),
                       "spp_chunk": last.get("spp_chunk")},
            "wall_s": dt,
            "roofline": roof, "roofline_contract": contract,
        }
        if hbm is not None:
            out["roofline_hbm"] = hbm
        out["launch"] = "one process per GPU (torch.distributed)" if world_size > 1 or force_pg else "one process"
        # per rank: what a first SCALE line needs to tell init cost from render from exchange
        out["ranks"] = [{"rank": i, "device": int(r[5]), "tiles": int(r[0]), "samples": int(r[1]), "kernel_ms": r[2], "exchange_ms": r[3], "loop_wall_s": r[4],
                         "kernel_ms_per_step": r[2] / max(1, args.steps), "exchange_ms_per_step": r[3] / max(1, args.steps),
                         "scene_upload_ms": r[6], "process_group_init_s": r[7]} for i, r in enumerate(per_rank)]
        out["devices"] = [int(r[5]) for r in per_rank]
        out["rccl_version"] = rtamd.lib().rt_rccl_version()   # the RCCL librtamd is linked with (rt_render_multi); torch's own is in process_group
        if backend is not None:
            nccl_v = None
            try:
                nccl_v = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:
                pass
            out["process_group"] = {"backend": backend + (" (RCCL)" if backend == "nccl" else ""), "world": world_size, "forced_at_world_1": bool(force_pg),
                                    "torch_rccl_version": nccl_v, "init_s_max": max(r[7] for r in per_rank),
                                    "calls": ["init_process_group", "gather", "barrier", "all_reduce(MAX)", "all_gather"]}
        if args.frame_out:
            import numpy as np
            np.save(args.frame_out, frame.cpu().numpy().reshape(args.height, args.width, 3))
        if rehearse:
            out["rehearsal"] = "all %d ranks share HIP device 0, gather over gloo through host memory: NOT a scaling measurement" % world_size
        if world_size == 1 and args.cpu_spp > 0:
            out["cpu_baseline"] = cpu_baseline(args.width, args.height, args.cpu_spp, args.seed)
        if json_fd is not None:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
