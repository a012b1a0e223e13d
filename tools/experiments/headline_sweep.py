"""scene_500 (headline frame, reduced spp) against the BVH builder's knobs; usage: python tools/headline_sweep.py [spp]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import rtamd
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 96
S = os.path.join(ROOT, "tests", "golden", "scenes", "scene_500.json")
for tun in [dict(), dict(sah_box_cost=0.5), dict(sah_box_cost=0.75), dict(sah_box_cost=1.5), dict(sah_box_cost=2.0), dict(sah_box_cost=3.0),
            dict(max_leaf=2), dict(max_leaf=3), dict(max_leaf=2, sah_box_cost=2.0), dict(max_leaf=3, sah_box_cost=1.5)]:
    rtamd.set_tuning(**tun)
    w, cam = rtamd.load_scene_file(S)
    w.render(cam, width=1200, height=1200, spp=4, seed=1)
    best = 0.0
    for _ in range(2):
        _, st = w.render(cam, width=1200, height=1200, spp=spp, seed=1)
        best = max(best, st["samples"] / (st["kernel_ms"] * 1e-3) / 1e6)
    i = w.info()
    print(json.dumps(dict(tuning=tun, msamples_per_s=round(best, 1), nodes=i["accel_nodes"], stack=i["accel_stack"], lds=st["scene_in_lds"])), flush=True)
