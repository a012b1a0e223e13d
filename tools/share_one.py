import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import rtamd
w, c = rtamd.load_scene_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene_500.json"))
world = int(sys.argv[1]); spp = int(sys.argv[2])
if os.environ.get("SUB_SPP"): rtamd.set_tuning(sub_spp=int(os.environ["SUB_SPP"]))  # A/B of the unit size
w.render(c, width=1200, height=1200, spp=8, seed=1, rank=0, world=world)
sys.stderr.write("---- measured render\n")
_, st = w.render(c, width=1200, height=1200, spp=spp, seed=1, rank=0, world=world)
print("world", world, "spp", spp, "kernel_ms", round(st["kernel_ms"], 2), "sub_spp", os.environ.get("SUB_SPP", "auto"))
