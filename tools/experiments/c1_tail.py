import os, sys
sys.path.insert(0, "/root/repo/rust-raytracer_amd")
import rtamd
w, c = rtamd.load_scene_file("/root/repo/tests/golden/scenes/scene_10.json")
w.render(c, width=400, height=225, spp=8, seed=1)
sys.stderr.write("---- measured render\n")
for sub in (0, 8):
    rtamd.set_tuning(sub_spp=sub)
    _, st = w.render(c, width=400, height=225, spp=100, seed=1)
    print("sub", sub, st["kernel_ms"])
