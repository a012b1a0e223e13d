"""Throughput of the BASELINE.json configurations C1..C4 on one GPU with the library's automatic kernel, each with the three objects
of the bench line: `roofline` (VALU issue, per-sample counts carried over from the configuration's own PMC model under
profiles/r05/model_<config>.json), `roofline_contract` (algorithmic bytes in the REFERENCE's traversal order, tests/golden/
alg_bytes_<config>.json, against the HBM peak) and `roofline_hbm` (physical traffic).  Kernel time is measured in this run.
usage: python tools/config_bench.py [scale]   (scale < 1 shrinks spp for a quick look) -> gpurun_out/config_bench.json"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import configs
import bench

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
rows = []
for key in ("scene_10", "scene_500_c2", "scene_500", "cornell", "cornell_mix", "c4", "c5r", "c5"):
    label, W, H, spp_cfg, _ = configs.CONFIGS[key]
    spp = max(1, int(spp_cfg * scale))
    world, cam = configs.product(key)
    integ = configs.INTEGRATOR.get(key, 0)
    world.render(cam, width=W, height=H, spp=min(spp, 4), seed=1, integrator=integ, shutter=configs.SHUTTER.get(key, (0.0, 0.0)))  # warm-up (workspace, code load)
    _, st = world.render(cam, width=W, height=H, spp=spp, seed=1, integrator=integ, shutter=configs.SHUTTER.get(key, (0.0, 0.0)))
    acc = {"kernel_ms": st["kernel_ms"], "launches": st["launches"], "samples": st["samples"]}
    model = os.path.join(ROOT, "profiles", "pt_kernel_model.json") if key == "scene_500" else os.path.join(ROOT, "profiles", "r05", "model_%s.json" % key)
    roof, contract, hbm = bench.roofline_objects(acc, st["kernel_ms"] * 1e-3, model_path=model,
                                                 alg_path=os.path.join(ROOT, "tests", "golden", "alg_bytes_%s.json" % key))
    row = dict(config=label, key=key, width=W, height=H, spp=spp, integrator=integ, kernel=st["kernel_used"], lds=st["scene_in_lds"],
               msamples_per_s_kernel=st["samples"] / (st["kernel_ms"] * 1e-3) / 1e6, msamples_per_s_wall=st["samples"] / st["seconds"] / 1e6,
               wall_s=st["seconds"], roofline=roof, roofline_contract=contract, roofline_hbm=hbm)
    rows.append(row)
    print(json.dumps({k: (v if not isinstance(v, dict) else {kk: v[kk] for kk in ("bound", "achieved", "peak", "frac") if kk in v}) for k, v in row.items()}), flush=True)
    del world
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "config_bench.json"), "w"), indent=1)
