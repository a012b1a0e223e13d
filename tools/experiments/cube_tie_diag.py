import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("rust-raytracer_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import test_cube_gpu as tc
world, _, ref = tc._pair(1, False)
rays = tc._rays()
o1 = world.debug_hit(rays, t_min=1e-3, kernel=1); o2 = world.debug_hit(rays, t_min=1e-3, kernel=2)
n = 0
for i, r in enumerate(rays):
    if np.isfinite(o1[i, 1]) and np.isfinite(o2[i, 1]) and not np.array_equal(o1[i], o2[i]):
        n += 1
        if n <= 8:
            print(i, "ray", r, "\n   k1", o1[i, [0, 1, 11]], "k2", o2[i, [0, 1, 11]], "same t:", o1[i, 1] == o2[i, 1])
            for node in (int(o1[i, 11]), int(o2[i, 11])):
                pass
print("differing finite rays:", n)
