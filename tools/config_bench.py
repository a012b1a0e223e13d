"""Throughput of the BASELINE.json configs on one GPU (kernel rate and wall), for DESIGN.md.
usage: python tools/config_bench.py [scale]   (scale < 1 shrinks spp for a quick look)"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import rtamd
from rtamd import shapes
S = os.path.join(ROOT, "tests", "golden", "scenes")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
rows = []

def run(name, world, cam, w, h, spp, kernels=(2, 1)):
    spp = max(1, int(spp * scale))
    for k in kernels:
        world.render(cam, width=w, height=h, spp=min(spp, 4), seed=1, kernel=k)  # warm-up (workspace, code load)
        _, st = world.render(cam, width=w, height=h, spp=spp, seed=1, kernel=k)
        rows.append(dict(config=name, width=w, height=h, spp=spp, kernel=k, lds=st["scene_in_lds"], msamples_per_s_wall=st["samples"] / st["seconds"] / 1e6,
                         msamples_per_s_kernel=st["samples"] / (st["kernel_ms"] * 1e-3) / 1e6, wall_s=st["seconds"]))
        print(json.dumps(rows[-1]), flush=True)

w10, c10 = rtamd.load_scene_file(os.path.join(S, "scene_10.json"))
run("C1 scene_10 400x225x100", w10, c10.with_aspect(16 / 9), 400, 225, 100)
w500, c500 = rtamd.load_scene_file(os.path.join(S, "scene_500.json"))
run("C2 scene_500 1200x800x500", w500, c500.with_aspect(1.5), 1200, 800, 500)
run("C2h scene_500 1200x1200x1000 (headline)", w500, c500, 1200, 1200, 1000)
wc, cc = rtamd.select_scene(os.path.join(S, "cube.obj"), 1.0, 1)
run("C3 cornell 800x800x2000 (brute force)", wc, cc, 800, 800, 2000)
P, N, I = shapes.torus(160, 320)
wm = rtamd.World(); wm.new(shapes.cornell_with_mesh(wm, P, N, I), bvh_seed=1)
cm = rtamd.Camera(((278, 278, -800), (278, 278, 278)), (0, 1, 0), 50, 1.0, 0.0, 10.0)
run("C4 cornell + 102,400-tri torus 1200x1200x1000 (one GPU)", wm, cm, 1200, 1200, 1000, kernels=(5, 2))
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "config_bench.json"), "w"), indent=1)
