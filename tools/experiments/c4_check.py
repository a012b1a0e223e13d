"""C4 sanity: Cornell + 102,400-triangle torus; parity vs oracle at small size, then timing of kernels 1 and 2."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import rtamd, oracle
from rtamd import shapes

nu, nv = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (160, 320)
P, N, I = shapes.torus(nu, nv)
t0 = time.time()
w = rtamd.World(); w.new(shapes.cornell_with_mesh(w, P, N, I), bvh_seed=1)
print("commit %.2fs" % (time.time() - t0), w.info())
cam = rtamd.Camera(((278, 278, -800), (278, 278, 278)), (0, 1, 0), 50, 1.0, 0.0, 10.0)
t0 = time.time()
o = oracle.Scene(); o.World(shapes.cornell_with_mesh(o, P, N, I), 1); o.Camera((278, 278, -800), (278, 278, 278), (0, 1, 0), 50, 1.0, 0.0, 10.0)
print("oracle build %.2fs" % (time.time() - t0))
for k in (1, 2):
    img, st = w.render(cam, width=64, height=64, spp=4, seed=1, kernel=k)
    if k == 1:
        t0 = time.time(); ref, cnt = o.render(64, 64, 4, seed=1); print("oracle render %.2fs" % (time.time() - t0), {a: b / cnt["n_samples"] for a, b in cnt.items()})
    print("kernel", k, "bit-exact vs oracle:", np.array_equal(img, ref), "lds", st["scene_in_lds"])
for k in (2, 1):
    img, st = w.render(cam, width=600, height=600, spp=32, seed=1, kernel=k)
    print("kernel %d: 600x600x32: %.1f Msamples/s (kernel %.1f ms, wall %.3f s)" % (k, st["samples"] / (st["kernel_ms"] * 1e-3) / 1e6, st["kernel_ms"], st["seconds"]))
rtamd.write_png(os.path.join(ROOT, "gpurun_out", "c4.png"), rtamd.tonemap_u8(img))
