"""C5r (reduced final scene) with the accel builder's leaf size / SAH box cost swept through rt_tuning (read at commit)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import configs, rtamd
key = sys.argv[1] if len(sys.argv) > 1 else "c5r"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
label, W, H, _, _ = configs.CONFIGS[key]
for max_leaf in (0, 2, 3, 4):
    for cbox in (0.0, 0.5, 2.0, 4.0):
        rtamd.set_tuning(max_leaf=max_leaf, sah_box_cost=cbox)
        world, cam = configs.product(key)
        info = world.info()
        integ = configs.INTEGRATOR.get(key, 0)
        world.render(cam, width=W, height=H, spp=2, seed=1, integrator=integ)
        _, st = world.render(cam, width=W, height=H, spp=spp, seed=1, integrator=integ)
        print("max_leaf %d c_box %.1f: nodes %5d stack %2d  kernel %d lds %d  %.1f Msamples/s" % (max_leaf, cbox, info["accel_nodes"], info["accel_stack"], st["kernel_used"], st["scene_in_lds"],
              st["samples"] / st["kernel_ms"] / 1e3), flush=True)
        del world
rtamd.set_tuning()
