"""Strong-scaling rehearsal on ONE GPU: rank 0's share of the headline frame for world = 1, 2, 4, 8.
If a rank's time is ~ t1/world, the N-GPU run is bound only by the (tiny) gather."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import rtamd
w, c = rtamd.load_scene_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene_500.json"))
w.render(c, width=1200, height=1200, spp=8)
t1 = None
for world in (1, 2, 4, 8):
    worst = 0
    for r in ([0] if world == 1 else [0, world - 1]):
        w.render(c, width=1200, height=1200, spp=1000, seed=1, rank=r, world=world)  # sizes the cached workspace
        _, st = w.render(c, width=1200, height=1200, spp=1000, seed=1, rank=r, world=world)
        worst = max(worst, st["seconds"])
        print("world %d rank %d: %.3f s wall (host buffers), kernel %.2f ms = %.0f Msamples/s on this rank, launches %d chunk %d" % (
            world, r, st["seconds"], st["kernel_ms"], st["samples"] / st["kernel_ms"] / 1e3, st["launches"], st["spp_chunk"]))
    t1 = t1 or worst
    print("   -> projected speed-up at %d GPUs (render only): %.2fx" % (world, t1 / worst))
