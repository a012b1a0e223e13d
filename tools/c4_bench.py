"""C4 (Cornell + 102,400-triangle torus) throughput on one GPU; usage: python tools/c4_bench.py [spp]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import rtamd
from rtamd import shapes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nu, nv = [int(x) for x in os.environ.get("C4_TORUS", "160x320").split("x")]  # C4_TORUS=40x80: smaller meshes
P, N, I = shapes.torus(nu, nv)
tun = {k: (float(v) if "." in v else int(v)) for k, v in (kv.split("=") for kv in os.environ.get("C4_TUNING", "").split(",") if kv)}  # C4_TUNING=max_leaf=2,sah_box_cost=0.5
if tun:
    rtamd.set_tuning(**tun)
w = rtamd.World(); w.new(shapes.cornell_with_mesh(w, P, N, I), bvh_seed=1)
cam = rtamd.Camera(((278, 278, -800), (278, 278, 278)), (0, 1, 0), 50, 1.0, 0.0, 10.0)
w.render(cam, width=1200, height=1200, spp=2, seed=1)
_, st = w.render(cam, width=1200, height=1200, spp=spp, seed=1, kernel=int(os.environ.get("C4_KERNEL", "0")))
print(json.dumps(dict(msamples_per_s=st["samples"] / (st["kernel_ms"] * 1e-3) / 1e6, kernel_ms=st["kernel_ms"], kernel=st["kernel_used"], launches=st["launches"],
                      workspace_mb=round(st["workspace_bytes"] / 1e6), lds=st["scene_in_lds"], info=w.info())))
