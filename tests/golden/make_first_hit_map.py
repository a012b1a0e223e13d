#!/usr/bin/env python3
"""Golden first-hit map: for scene_500 at 96x64, the sample-0 camera ray of every pixel and the
t / normal of its closest hit from the oracle.  Pins World::hit (traversal + sphere tests) independently of shading."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle
W, H = 96, 64
sc = oracle.load_scene_file(os.path.join(HERE, "scenes", "scene_500.json"), aspect=W / H)
rays = np.zeros((H, W, 6)); out = np.full((H, W, 4), np.nan)
for y in range(H):
    for x in range(W):
        o, d = sc.camera_ray(W, H, x, y, seed=1, sample=0)
        rays[y, x, :3], rays[y, x, 3:] = o, d
        h = sc.hit(o, d, 1e-3)
        if h is not None:
            out[y, x, 0] = h["t"]; out[y, x, 1:] = h["normal"]
np.savez_compressed(os.path.join(HERE, "scene_500_first_hit_96x64.npz"), rays=rays, hit=out)
print("hits:", int(np.isfinite(out[..., 0]).sum()), "of", W * H)
