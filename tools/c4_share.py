"""C4 (Cornell + 102,400-triangle torus, kernel 5): kernel time of rank 0's share at world 1, 2, 4, 8 (best of 3); argv: [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import rtamd
from rtamd import shapes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P, N, I = shapes.torus(160, 320)
w = rtamd.World(); w.new(shapes.cornell_with_mesh(w, P, N, I), bvh_seed=1)
cam = rtamd.Camera(((278, 278, -800), (278, 278, 278)), (0, 1, 0), 50, 1.0, 0.0, 10.0)
w.render(cam, width=1200, height=1200, spp=2, seed=1)
t1 = None
for world in (1, 2, 4, 8, 16):
    best = min(w.render(cam, width=1200, height=1200, spp=spp, seed=1, rank=0, world=world)[1]["kernel_ms"] for _ in range(3))
    t1 = t1 or best
    print("world %2d: %.2f ms (ideal %.2f, x%.3f)" % (world, best, t1 / world, best / (t1 / world)))
