"""kernel 6 (wavefront instance service) on C4 under different workspace budgets (rt_tuning.wf_workspace_mb)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import configs, rtamd
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
world, cam = configs.product("c4")
world.render(cam, width=1200, height=1200, spp=2, seed=1, kernel=5)
_, st = world.render(cam, width=1200, height=1200, spp=spp, seed=1, kernel=5)
print("kernel 5: %.1f Msamples/s, workspace %.2f GB" % (st["samples"] / st["kernel_ms"] / 1e3, st["workspace_bytes"] / 1e9), flush=True)
for mb in (1900, 2600, 3500, 5000, 8600):
    rtamd.set_tuning(wf_workspace_mb=mb)
    rtamd.release_workspaces()
    world.render(cam, width=1200, height=1200, spp=2, seed=1, kernel=6)
    _, st = world.render(cam, width=1200, height=1200, spp=spp, seed=1, kernel=6)
    print("kernel 6, budget %4d MB: %.1f Msamples/s, workspace %.2f GB, %d launches" % (mb, st["samples"] / st["kernel_ms"] / 1e3, st["workspace_bytes"] / 1e9, st["launches"]), flush=True)
rtamd.set_tuning()
