#!/usr/bin/env python3
"""Measure the algorithmic bytes per sample of SURVEY.md s8d for one BASELINE configuration with the instrumented oracle (the
REFERENCE's traversal order) and commit them as a fixture:
    B_alg = 56*N_aabb + 36*N_sphere + 44*N_rect + 160*N_tri + 256*N_xform + 24/spp   [bytes/sample]
Usage: python tests/golden/make_alg_bytes.py [config ...]      (configs: tools/configs.py; default: all)
The counts are statistics of random paths: they are taken at a reduced spp (the per-sample figure does not depend on it beyond
sampling noise) and, for C4, on a reduced image (300 x 300: the oracle walks the reference's own 102,400-triangle BVH)."""
import json, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import configs
import oracle

for key in (sys.argv[1:] or list(configs.CONFIGS)):
    label, W, H, spp_cfg, spp = configs.CONFIGS[key]
    w, h = (300, 300) if key == "c4" else (400, 400) if key in ("c5r", "c5") else (W, H)
    sc = configs.oracle_scene(key)
    t0 = time.time()
    _, cnt = sc.render(w, h, spp, max_depth=50, seed=1, integrator=configs.INTEGRATOR.get(key, 0))
    dt = time.time() - t0
    n = cnt["n_samples"]
    per = {k: v / n for k, v in cnt.items()}
    b = oracle.algorithmic_bytes(cnt, spp_cfg)  # 24/spp uses the CONFIG's spp: the framebuffer term of the contract figure
    out = {"config": label, "key": key, "width": w, "height": h, "config_width": W, "config_height": H, "config_spp": spp_cfg, "spp_measured": spp,
           "max_depth": 50, "seed": 1, "counters": cnt, "per_sample": per, "bytes_per_sample": b,
           "weights": {"aabb": 56, "sphere": 36, "rect": 44, "tri": 160, "xform": 256, "framebuffer": "24/spp"}, "oracle_seconds": dt}
    with open(os.path.join(HERE, "alg_bytes_%s.json" % key), "w") as f:
        json.dump(out, f, indent=1)
    print(key, json.dumps(per), b, "%.1f s" % dt, flush=True)
