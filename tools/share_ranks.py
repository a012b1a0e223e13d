"""Kernel time of EVERY rank's share of the headline frame at world W (best of 2 each): is the tile -> rank deal balanced?  argv: worlds..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import rtamd
w, c = rtamd.load_scene_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene_500.json"))
w.render(c, width=1200, height=1200, spp=8, seed=1)
t1 = min(w.render(c, width=1200, height=1200, spp=1000, seed=1)[1]["kernel_ms"] for _ in range(2))
print("world 1: %.2f ms" % t1)
for W in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
    ts = [min(w.render(c, width=1200, height=1200, spp=1000, seed=1, rank=r, world=W)[1]["kernel_ms"] for _ in range(2)) for r in range(W)]
    print("world %d: ranks %s  sum %.2f (x%.3f of world 1)  max %.2f -> speed-up %.2f of %d" % (W, " ".join("%.2f" % t for t in ts), sum(ts), sum(ts) / t1, max(ts), t1 / max(ts), W))
