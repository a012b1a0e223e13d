"""Kernel time of rank 0's share, N renders back to back (is the first one after an idle period slower?).  argv: world spp [repeats]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import rtamd
w, c = rtamd.load_scene_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene_500.json"))
world = int(sys.argv[1]); spp = int(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
w.render(c, width=1200, height=1200, spp=8, seed=1, rank=0, world=world)
time.sleep(0.5)
out = []
for i in range(n):
    t0 = time.perf_counter()
    _, st = w.render(c, width=1200, height=1200, spp=spp, seed=1, rank=0, world=world)
    out.append((round(st["kernel_ms"], 2), round((time.perf_counter() - t0) * 1e3, 2)))
print("world", world, "spp", spp, "(kernel_ms, wall_ms):", out)
print("min", world, min(o[0] for o in out))
