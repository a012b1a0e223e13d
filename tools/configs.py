"""The BASELINE.json configurations C1..C4 as builders for the product (rtamd) and the oracle, shared by tools/config_bench.py,
tools/config_run.py (PMC passes) and tests/golden/make_alg_bytes.py.  (C5 -- motion blur + Perlin noise -- has no code in the
reference: ray.rs:3-6 has no time, there is no noise texture.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "tests", "golden", "scenes")
for p in (os.path.join(ROOT, "rust-raytracer_amd"),):
    if p not in sys.path:
        sys.path.insert(0, p)

CONFIGS = {
    # key: (label, width, height, spp of the config, spp the oracle counts with)
    "scene_10": ("C1 scene_10 400x225x100", 400, 225, 100, 16),
    "scene_500_c2": ("C2 scene_500 1200x800x500", 1200, 800, 500, 2),
    "scene_500": ("headline scene_500 1200x1200x1000", 1200, 1200, 1000, 2),
    "cornell": ("C3 cornell 800x800x2000 (BSDF sampling, integrator 0)", 800, 800, 2000, 4),
    "cornell_mix": ("C3 cornell 800x800x2000 (light / cosine mixture pdf, integrator 1)", 800, 800, 2000, 4),
    "c4": ("C4 cornell + 102,400-triangle torus 1200x1200x1000", 1200, 1200, 1000, 1),
    "c5": ("C5 book-2 final scene AS NAMED (motion blur + Perlin marble; book-2 extensions, the reference has no code for either) 1600x1600x4000", 1600, 1600, 4000, 1),
    "c5r": ("C5 reduced (book-2 final scene without motion blur / Perlin: 400 boxes, 2 media, image texture, 1000-sphere instance) 1600x1600x4000", 1600, 1600, 4000, 1),
}
INTEGRATOR = {"cornell_mix": 1}  # rt_params.integrator of a configuration (default 0)
SHUTTER = {"c5": (0.0, 1.0)}      # rt_params.time0 / time1 of a configuration (default: closed)
CORNELL_CAM = ((278, 278, -800), (278, 278, 278), (0, 1, 0), 50, 1.0, 0.0, 10.0)


def product(key):
    """-> (rtamd.World, rtamd.Camera)"""
    import rtamd
    from rtamd import shapes
    if key == "scene_10":
        w, c = rtamd.load_scene_file(os.path.join(SCENES, "scene_10.json"))
        return w, c.with_aspect(16 / 9)
    if key == "scene_500_c2":
        w, c = rtamd.load_scene_file(os.path.join(SCENES, "scene_500.json"))
        return w, c.with_aspect(1.5)
    if key == "scene_500":
        return rtamd.load_scene_file(os.path.join(SCENES, "scene_500.json"))
    if key in ("cornell", "cornell_mix"):
        return rtamd.select_scene(os.path.join(SCENES, "cube.obj"), 1.0, 1)
    if key == "c4":
        P, N, I = shapes.torus(160, 320)
        w = rtamd.World()
        w.new(shapes.cornell_with_mesh(w, P, N, I), bvh_seed=1)
        f, t, up, vfov, asp, ap, fd = CORNELL_CAM
        return w, rtamd.Camera((f, t), up, vfov, asp, ap, fd)
    if key == "c5":
        w = rtamd.World()
        w.new(shapes.final_scene(w), bvh_seed=3)
        f, t, up, vfov, asp, ap, fd = shapes.FINAL_SCENE_CAMERA
        return w, rtamd.Camera((f, t), up, vfov, asp, ap, fd)
    if key == "c5r":
        w = rtamd.World()
        w.new(shapes.final_scene_reduced(w), bvh_seed=3)
        f, t, up, vfov, asp, ap, fd = shapes.FINAL_SCENE_CAMERA
        return w, rtamd.Camera((f, t), up, vfov, asp, ap, fd)
    raise KeyError(key)


def oracle_scene(key):
    """-> oracle.Scene with its camera set (test infrastructure: used by make_alg_bytes.py only)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    from rtamd import shapes
    if key == "scene_10":
        return oracle.load_scene_file(os.path.join(SCENES, "scene_10.json"), aspect=16 / 9)
    if key == "scene_500_c2":
        return oracle.load_scene_file(os.path.join(SCENES, "scene_500.json"), aspect=1.5)
    if key == "scene_500":
        return oracle.load_scene_file(os.path.join(SCENES, "scene_500.json"))
    if key in ("cornell", "cornell_mix"):
        return oracle.cornell_box_scene(os.path.join(SCENES, "cube.obj"), 1.0, 1)
    if key == "c4":
        P, N, I = shapes.torus(160, 320)
        o = oracle.Scene()
        o.World(shapes.cornell_with_mesh(o, P, N, I), 1)
        o.Camera(*CORNELL_CAM)
        return o
    if key == "c5":
        o = oracle.Scene()
        o.World(shapes.final_scene(o), 3)
        o.Camera(*shapes.FINAL_SCENE_CAMERA)
        o.set_shutter(*shapes.FINAL_SCENE_SHUTTER)
        return o
    if key == "c5r":
        o = oracle.Scene()
        o.World(shapes.final_scene_reduced(o), 3)
        o.Camera(*shapes.FINAL_SCENE_CAMERA)
        return o
    raise KeyError(key)
