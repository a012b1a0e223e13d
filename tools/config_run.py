"""One BASELINE configuration on one GPU at a given spp (warm-up render first); for PMC passes and quick looks.
usage: python tools/config_run.py <config> [spp] [kernel]      configs: tools/configs.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import configs
key = sys.argv[1]
label, W, H, spp_cfg, _ = configs.CONFIGS[key]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else spp_cfg
kernel = int(sys.argv[3]) if len(sys.argv) > 3 else 0
world, cam = configs.product(key)
integ = configs.INTEGRATOR.get(key, 0)
world.render(cam, width=W, height=H, spp=min(spp, 2), seed=1, kernel=kernel, integrator=integ, shutter=configs.SHUTTER.get(key, (0.0, 0.0)))
_, st = world.render(cam, width=W, height=H, spp=spp, seed=1, kernel=kernel, integrator=integ, shutter=configs.SHUTTER.get(key, (0.0, 0.0)))
print(json.dumps(dict(config=label, key=key, width=W, height=H, spp=spp, kernel=st["kernel_used"], lds=st["scene_in_lds"], samples=st["samples"],
                      kernel_ms=st["kernel_ms"], launches=st["launches"], msamples_per_s_kernel=st["samples"] / (st["kernel_ms"] * 1e-3) / 1e6,
                      msamples_per_s_wall=st["samples"] / st["seconds"] / 1e6)))
