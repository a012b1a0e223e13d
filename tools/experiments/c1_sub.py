"""C1 (scene_10 400x225x100): kernel time against the unit size (rt_tuning.sub_spp); best of 5"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import rtamd
w, c = rtamd.load_scene_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene_10.json"))
for (W, H, spp) in ((400, 225, 100), (400, 225, 1000), (1200, 675, 100)):
    for sub in (0, 8, 6, 4, 3, 2):
        rtamd.set_tuning(sub_spp=sub)
        w.render(c, width=W, height=H, spp=spp, seed=1)
        best = min(w.render(c, width=W, height=H, spp=spp, seed=1)[1]["kernel_ms"] for _ in range(5))
        print("%dx%dx%d sub_spp %d: %.3f ms = %.0f Msamples/s" % (W, H, spp, sub, best, W * H * spp / best / 1e3))
