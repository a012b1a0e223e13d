import os, sys, json
sys.path.insert(0, "rust-raytracer_amd")
import rtamd
w, cam = rtamd.select_scene("tests/golden/scenes/cube.obj", 1.0, 1)
for integ in (0, 1):
    w.render(cam, width=800, height=800, spp=4, seed=1, integrator=integ)
    _, st = w.render(cam, width=800, height=800, spp=500, seed=1, integrator=integ)
    print("integrator", integ, round(st["samples"] / (st["kernel_ms"] * 1e-3) / 1e6, 1), "Msamples/s", st["launches"])
