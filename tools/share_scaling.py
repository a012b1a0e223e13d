"""kernel time of ONE rank's share of the scene_500 frame as a function of the share (world) and of spp: is the loss of efficiency
at small shares a fixed cost per launch or a per-sample-layer effect?  usage: python tools/share_scaling.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import rtamd
w, c = rtamd.load_scene_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene_500.json"))
w.render(c, width=1200, height=1200, spp=8)
for world in (1, 4, 8, 16, 32):
    for spp in (250, 1000):
        w.render(c, width=1200, height=1200, spp=spp, seed=1, rank=0, world=world)
        _, st = w.render(c, width=1200, height=1200, spp=spp, seed=1, rank=0, world=world)
        ideal = 1200 * 1200 * spp / world / 2.84e9 * 1e3
        print("world %2d spp %4d: kernel %8.2f ms  (%.0f Msamples/s; ideal at 2840: %.2f ms; excess %.2f ms)" % (world, spp, st["kernel_ms"], st["samples"] / st["kernel_ms"] / 1e3, ideal, st["kernel_ms"] - ideal), flush=True)
