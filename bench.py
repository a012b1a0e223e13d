#!/usr/bin/env python3
"""bench.py -- headline benchmark of the radiance path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

Workload (config.workload): data/scene_500.json, 1200x1200, 1000 spp, depth 50, seed 1 -- the
configuration BASELINE.json's metric is quoted on.  One "step" = one full frame = 1.44e9
pixel-samples through the HIP path: rt_render_tiles_device on every rank's tiles, then the
framebuffer gather to rank 0 (RCCL) and the stitch.  The scene, camera and all buffers are
resident in HBM before the timed region.  value = W*H*spp*K / max-over-ranks wall time.

N > 1: launched by torch.distributed.run, one rank per GPU; the frame is dealt to ranks by
8x8 tile (tile t -> rank t % N), total work fixed => "scaling": "strong".

Extra objects on the JSON line:
  roofline     -- SURVEY s8d contract: algorithmic bytes per sample (reference-order AABB/sphere
                  test counts x reference payload sizes, tests/golden/alg_bytes_scene_500.json)
                  x samples per launch / mean path-trace kernel duration (HIP events recorded by
                  the library on its launch stream), against the 8 TB/s HBM peak.
  cpu_baseline -- the oracle (a C++ port of the reference's CPU path; the Rust binary cannot be
                  built here) on this box's host cores, bounded sample, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))

SCENE = os.path.join(ROOT, "tests", "golden", "scenes", "scene_500.json")
ALG_BYTES = os.path.join(ROOT, "tests", "golden", "alg_bytes_scene_500.json")
TRAFFIC = os.path.join(ROOT, "profiles", "hbm_traffic.json")
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CPU_BASELINE_THREADS = 16


def cpu_baseline(width, height, spp_cpu, seed):
    """Oracle timed on the host cores (test infrastructure used as the reported CPU baseline)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    # the GPU box gives one GPU's share of the host (16 threads, see the task's process guard); never more
    # threads than that even though the host shows 256
    cores = min(len(os.sched_getaffinity(0)), CPU_BASELINE_THREADS)
    sc = oracle.load_scene_file(SCENE)
    t0 = time.perf_counter()
    _, cnt = sc.render(width, height, spp_cpu, seed=seed, n_jobs=64, n_workers=cores)
    dt = time.perf_counter() - t0
    return {
        "value": width * height * spp_cpu / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": "scene_500 %dx%d at %d spp (%.1f s), 64 row-band jobs on %d threads; C++ restatement of the "
                  "reference, not the Rust binary" % (width, height, spp_cpu, dt, cores),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--kernel", type=int, default=0, help="0 = library default; 1/2/3 force a path-trace kernel (A/B runs only)")
    ap.add_argument("--cpu-spp", type=int, default=32, help="spp of the bounded CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    import torch
    import rtamd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available() or rtamd.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the product has no CPU fallback")
    dev_index = int(os.environ.get("RTAMD_BENCH_DEVICE", local_rank))  # override only to rehearse N>1 on a 1-GPU box
    torch.cuda.set_device(dev_index)  # before the process group: RCCL binds the communicator to the current device
    dev = torch.device("cuda", dev_index)
    dist = None
    if world_size > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world_size)

    world, cam = rtamd.load_scene_file(SCENE)
    params = rtamd.default_params(width=args.width, height=args.height, spp=args.spp, max_depth=50, t_min=1e-3, seed=args.seed,
                                  rank=rank, world=world_size, kernel=args.kernel)
    from rtamd.distributed import TileLayout, gather_tiles
    layout = TileLayout(args.width, args.height, world_size)
    p0 = rtamd.default_params(width=args.width, height=args.height, spp=args.spp, rank=0, world=world_size)
    stride = layout.stride  # rank 0 owns the most tiles; every rank pads to it for the gather
    assert stride == rtamd.tiles_owned(p0)
    d_tiles = torch.zeros(stride * 64 * 3, dtype=torch.float64, device=dev)
    frame = torch.zeros(args.height * args.width * 3, dtype=torch.float64, device=dev) if rank == 0 else None
    stream = torch.cuda.current_stream().cuda_stream

    stats_acc = {"kernel_ms": 0.0, "launches": 0, "samples": 0}
    last = {}

    def step(timed):
        st = world.render_tiles_device(cam, params, d_tiles.data_ptr(), stream)
        # the stitch of camera.rs:115-123 across GPUs: ONE framebuffer gather to rank 0 over RCCL/xGMI
        src = gather_tiles(d_tiles, layout, rank, dist, dst=0)
        if rank == 0:
            rtamd.assemble_frame_device(p0, src.data_ptr(), stride, frame.data_ptr(), stream)
        if timed:
            stats_acc["kernel_ms"] += st["kernel_ms"]
            stats_acc["launches"] += st["launches"]
            stats_acc["samples"] += st["samples"]
        last.update(st)

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
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total = args.width * args.height * args.spp * args.steps
        value = total / dt / 1e6
        with open(ALG_BYTES) as f:
            alg = json.load(f)
        b_alg = alg["bytes_per_sample"]
        # dominant kernel = pt_kernel; per launch: samples/launch * B_alg / mean launch duration (this rank)
        launches = max(1, stats_acc["launches"])
        samples_per_launch = stats_acc["samples"] / launches
        ms_per_launch = stats_acc["kernel_ms"] / launches
        achieved = (samples_per_launch * b_alg) / (ms_per_launch * 1e-3) / 1e9 if ms_per_launch > 0 else 0.0
        traffic = None
        if os.path.exists(TRAFFIC):
            with open(TRAFFIC) as f:
                per_sample = json.load(f).get("hbm_bytes_per_sample")
                if per_sample is not None:  # PMC-measured (profiles/), scaled to this run's launch size
                    traffic = per_sample * samples_per_launch
        out = {
            "metric": "Msamples/sec (px*spp), scene_500 %dx%d %dspp" % (args.width, args.height, args.spp),
            "value": value, "unit": "Msamples/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "tests/golden/scenes/scene_500.json (data/scene_500.json of the reference, minified): 1005 spheres, "
                                   "999-node file BVH, %dx%d, %d spp, depth 50, seed %d" % (args.width, args.height, args.spp, args.seed),
                       "parallelism": "image tiles 8x8 dealt round-robin to %d GPU(s), RCCL framebuffer gather" % world_size,
                       "kernel": "pt_kernel<%s>(f64 primitives, scene %s)" % ({1: "reference-order stackless", 2: "SAH-BVH2 accel, f32 conservative boxes", 3: "SAH-BVH2 accel, early-restart schedule"}.get(last.get("kernel_used"), "?"),
                                                                              "in LDS" if last.get("scene_in_lds") else "in L2/HBM"),
                       "block_threads": last.get("block_threads"), "grid_blocks": last.get("grid_blocks"),
                       "spp_chunk": last.get("spp_chunk")},
            "wall_s": dt,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "pt_kernel", "alg_bytes_per_sample": b_alg, "samples_per_launch": samples_per_launch,
                         "ms_per_launch": ms_per_launch,
                         "note": "algorithmic bytes in the REFERENCE's traversal order (SURVEY s8d); the scene's traversal tables are "
                                 "LDS-resident, so physical HBM traffic is far below this figure"},
        }
        if world_size == 1 and args.cpu_spp > 0:
            out["cpu_baseline"] = cpu_baseline(args.width, args.height, args.cpu_spp, args.seed)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
