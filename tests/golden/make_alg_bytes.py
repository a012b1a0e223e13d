#!/usr/bin/env python3
"""Measure the algorithmic bytes per sample of SURVEY.md s8d for the bench workload with the
instrumented oracle (reference traversal order) and commit them as a fixture:
    B_alg = 56*N_aabb + 36*N_sphere + 44*N_rect + 160*N_tri + 256*N_xform + 24/spp   [bytes/sample]
Usage: python tests/golden/make_alg_bytes.py [spp_measured]   (scene_500, 1200x1200, depth 50, seed 1)"""
import json, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W = H = 1200
sc = oracle.load_scene_file(os.path.join(HERE, "scenes", "scene_500.json"))
t0 = time.time()
_, cnt = sc.render(W, H, spp, max_depth=50, seed=1)
dt = time.time() - t0
n = cnt["n_samples"]
per = {k: v / n for k, v in cnt.items()}
# 24/spp uses the BENCH spp (1000): the framebuffer term of the contract figure
b = oracle.algorithmic_bytes(cnt, 1000)
out = {"scene": "scene_500.json", "width": W, "height": H, "spp_measured": spp, "max_depth": 50, "seed": 1,
       "counters": cnt, "per_sample": per, "bytes_per_sample": b,
       "weights": {"aabb": 56, "sphere": 36, "rect": 44, "tri": 160, "xform": 256, "framebuffer": "24/spp"},
       "oracle_seconds": dt}
with open(os.path.join(HERE, "alg_bytes_scene_500.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out["per_sample"]), b, dt)
