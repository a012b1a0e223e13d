"""diagnostic: rt_render_sppm vs rt_render_sppm_multi vs the oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import rtamd, oracle
cube = os.path.join(ROOT, "tests/golden/scenes/cube.obj")
world, cam = rtamd.select_scene(cube, 1.0, 1)
kw = dict(width=24, height=24, spp=3, seed=1, iterations=3, photons_per_iter=6000)
exp, exp_stats, _ = oracle.cornell_box_scene(cube, 1.0, seed=1).render_sppm(24, 24, 3, iterations=3, photons_per_iter=6000, k_global=100, k_caustic=50, seed=1)
def show(tag, img):
    bad = (img != exp).any(axis=2)
    print(tag, "== oracle:", not bad.any(), int(bad.sum()))
    for y, x in np.argwhere(bad)[:3]:
        print("    px", (int(x), int(y)), "got", img[y, x], "exp", exp[y, x])
for k in (0, 1, 2):
    img, stats, tot, st = world.render_sppm(cam, kernel=k, **kw)
    show("single kernel=%d (used %d, lds %d)" % (k, st["kernel_used"], st["scene_in_lds"]), img)
img, st = world.render_sppm_multi(cam, devices=[0], **kw); show("multi [0]", img)
img, st = world.render_sppm_multi(cam, devices=[0, 0], **kw); show("multi [0,0]", img)
img, stats, tot, st = world.render_sppm(cam, **kw); show("single again", img)
rtamd.set_tuning(); img, stats, tot, st = world.render_sppm(cam, **kw); show("single after set_tuning()", img)
