"""Kernel rates of the BASELINE configurations at a fraction of their spp, one line each (A/B runs: RTAMD_LIB=rust-raytracer_amd/variants/librtamd_X.so).
usage: python tools/rates.py [scale] [keys...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import configs
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
keys = sys.argv[2:] or ["scene_500", "cornell", "cornell_mix", "c4", "c5r", "c5"]
out = []
for key in keys:
    label, W, H, spp_cfg, _ = configs.CONFIGS[key]
    spp = max(1, int(spp_cfg * scale))
    if key in ("c5r", "c5"):
        spp = max(1, spp // 8)
    world, cam = configs.product(key)
    kw = dict(width=W, height=H, seed=1, integrator=configs.INTEGRATOR.get(key, 0), shutter=configs.SHUTTER.get(key, (0.0, 0.0)))
    world.render(cam, spp=min(spp, 4), **kw)
    best = 0.0
    for _ in range(2):
        _, st = world.render(cam, spp=spp, **kw)
        best = max(best, st["samples"] / (st["kernel_ms"] * 1e-3) / 1e6)
    out.append("%s[k%d] %.1f" % (key, st["kernel_used"], best))
    del world
print(os.environ.get("RTAMD_LIB", "product").split("/")[-1], " ".join(out), flush=True)
