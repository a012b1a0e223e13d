"""One-off soak of the job schedule (tapered end of launch, early fold, 8 unit buffers): the image must not depend on how the samples
are cut into units, launches and ranks.  Random image sizes / sample counts / unit sizes / chunks / partitions on three scenes and
kernels 1, 2, 5, each against the plain render of the same frame, bit for bit; plus the headline frame at 64 spp as the sum of 8 ranks.
usage: python tools/schedule_soak.py [trials]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import numpy as np, rtamd
from rtamd import shapes
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(11)
G = os.path.join(ROOT, "tests", "golden", "scenes")
scenes = []
w10, c10 = rtamd.load_scene_file(os.path.join(G, "scene_10.json")); scenes.append(("scene_10", w10, c10, (1, 2)))
w500, c500 = rtamd.load_scene_file(os.path.join(G, "scene_500.json")); scenes.append(("scene_500", w500, c500, (1, 2)))
P, N, I = shapes.torus(40, 80)
wm = rtamd.World(); wm.new(shapes.cornell_with_mesh(wm, P, N, I), bvh_seed=1)
cm = rtamd.Camera(((278, 278, -800), (278, 278, 278)), (0, 1, 0), 50, 1.0, 0.0, 10.0)
scenes.append(("cornell + 6 400-triangle torus", wm, cm, (2, 5)))
bad = 0
for t in range(trials):
    name, w, c, kernels = scenes[t % len(scenes)]
    W = rng.choice([8, 9, 40, 64, 100, 200]); H = rng.choice([8, 15, 48, 96, 150])
    spp = rng.choice([1, 7, 8, 33, 64, 130, 250, 500]) if W * H <= 4096 else rng.choice([1, 7, 24, 57, 120])
    seed = rng.randint(0, 99)
    cam = c.with_aspect(W / H) if name != "cornell + 6 400-triangle torus" else c
    rtamd.set_tuning()
    base, _ = w.render(cam, width=W, height=H, spp=spp, seed=seed, kernel=kernels[0])
    for k in kernels:
        sub = rng.choice([0, 1, 2, 3, 5, 8]); chunk = rng.choice([0, 0, max(1, spp // 3), 17]); world = rng.choice([1, 1, 2, 3, 8])
        rtamd.set_tuning(sub_spp=sub)
        acc = np.zeros_like(base)
        for r in range(world):
            part, _ = w.render(cam, width=W, height=H, spp=spp, seed=seed, kernel=k, spp_chunk=chunk, rank=r, world=world)
            acc += part
        ok = np.array_equal(acc, base)
        bad += not ok
        print("trial %2d %-32s %3dx%-3d spp %3d kernel %d sub_spp %d chunk %2d world %d: %s" % (t, name, W, H, spp, k, sub, chunk, world, "same" if ok else "DIFFERENT"))
rtamd.set_tuning()
full, st = w500.render(c500, width=1200, height=1200, spp=64, seed=1)
acc = np.zeros_like(full)
for r in range(8):
    part, _ = w500.render(c500, width=1200, height=1200, spp=64, seed=1, rank=r, world=8)
    acc += part
ok = np.array_equal(acc, full); bad += not ok
print("headline frame at 64 spp = sum of 8 ranks' tiles: %s" % ("same" if ok else "DIFFERENT"))
print("schedule soak: %d trials, %d differences" % (trials, bad))
sys.exit(1 if bad else 0)
