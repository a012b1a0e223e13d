"""diagnostic: which book-2 feature differs between product and oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("rust-raytracer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import rtamd, oracle
import test_book2 as tb

def run(tag, pick, shutter, kernel):
    w = rtamd.World(); items = tb._scene(w); w.new([items[i] for i in pick], bvh_seed=4)
    o = oracle.Scene(); items = tb._scene(o); o.World([items[i] for i in pick], 4)
    cam = ((0.0, 3.0, -9.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 40.0, 4.0 / 3.0, 0.1, 9.0)
    o.Camera(*cam); o.set_shutter(*shutter)
    f, t, up, vfov, asp, ap, fd = cam
    c = rtamd.Camera((f, t), up, vfov, asp, ap, fd)
    img, st = w.render(c, width=96, height=72, spp=4, seed=2, kernel=kernel, shutter=shutter)
    exp, _ = o.render(96, 72, 4, seed=2)
    bad = (img != exp).any(axis=2)
    print("%-34s shutter %s kernel %d (used %d lds %d): %4d pixels differ, max |d| %.3g" % (tag, shutter, kernel, st["kernel_used"], st["scene_in_lds"], int(bad.sum()), float(np.abs(img - exp).max())), flush=True)

names = ["floor(marble)", "sphere(marble)", "moving1", "moving2(glass)", "xform(moving,metal)", "cube", "light", "sky"]
for k in (1, 2):
    run("sky + light + white cube", [5, 6, 7], (0.0, 1.0), k)
    run("+ marble floor", [0, 5, 6, 7], (0.0, 1.0), k)
    run("+ marble sphere", [1, 5, 6, 7], (0.0, 1.0), k)
    run("+ moving1", [2, 5, 6, 7], (0.0, 1.0), k)
    run("+ moving1, shutter closed", [2, 5, 6, 7], (0.0, 0.0), k)
    run("+ moving2 (glass, 0.25..0.75)", [3, 5, 6, 7], (0.0, 1.0), k)
    run("+ transformed moving metal", [4, 5, 6, 7], (0.0, 1.0), k)
    run("all", list(range(8)), (0.0, 1.0), k)
