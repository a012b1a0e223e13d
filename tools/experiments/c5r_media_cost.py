"""What the media machinery costs on the reduced book-2 final scene: the same geometry with and without its two ConstantMedium objects
(kernel 2: MEDIA variant against the plain GENERAL variant), kernel Msamples/s.  NOT the same work: a ray that leaves the scene travels
5 000 units through the thin fog and scatters with probability 0.39, so paths in fog have twice the segments (7.3 M against 3.7 M
segment-iterations at 32 spp, -DRTAMD_PHASE_STATS); per segment-iteration the MEDIA variant takes 1.2x the wave clocks (141 k against 116 k).  usage: python tools/experiments/c5r_media_cost.py [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import rtamd
from rtamd import shapes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
f, t, up, vfov, asp, ap, fd = shapes.FINAL_SCENE_CAMERA
cam = rtamd.Camera((f, t), up, vfov, asp, ap, fd)
for label, drop in (("with media", False), ("without media", True)):
    w = rtamd.World()
    items = shapes.final_scene_reduced(w)
    if drop:
        items = [i for i in items if w.describe(i)[0] != "ConstantMedium"]
    w.new(items, bvh_seed=3)
    w.render(cam, width=1600, height=1600, spp=2, seed=1)
    _, st = w.render(cam, width=1600, height=1600, spp=spp, seed=1)
    print("%-14s kernel %d  %.1f Msamples/s" % (label, st["kernel_used"], st["samples"] / st["kernel_ms"] / 1e3), flush=True)
