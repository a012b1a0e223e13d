"""The reference binary's own default workload (main.rs:34-55): Cornell box, 800x800, SPPM pre-pass of 50 iterations x
500,000 photons, then capture_image at 256 spp with the SPPM sample_ray; writes output/test.png like main.rs does."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import numpy as np
import rtamd
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
photons = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
w, cam = rtamd.select_scene(os.path.join(ROOT, "tests", "golden", "scenes", "cube.obj"), 1.0, 1)
w.render(cam, width=64, height=64, spp=1)  # warm-up
t0 = time.time()
img, st, tot, info = w.render_sppm(cam, width=800, height=800, spp=256, iterations=iters, photons_per_iter=photons, seed=1)
total = time.time() - t0
out = dict(iterations=iters, photons_per_iter=photons, total_s=total, sppm_s=info["prepass_seconds"], rt_s=total - info["prepass_seconds"],
           photons_global=tot[0], photons_caustic=tot[1], nan_pixels=int(np.isnan(img).any(axis=2).sum()), mean_radiance=float(np.nanmean(img)),
           kernel_ms=info["kernel_ms"])
print("Total: %.2fs\n\tSPPM: %.2fs\n\tRT: %.2fs" % (total, out["sppm_s"], out["rt_s"]))   # main.rs:57-71
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
rtamd.write_png(os.path.join(ROOT, "gpurun_out", "sppm_test.png"), rtamd.tonemap_u8(img))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "sppm_reference_workload.json"), "w"), indent=1)
