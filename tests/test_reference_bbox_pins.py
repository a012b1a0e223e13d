"""Reference-held pin: every `BVHNode.bounding_box` stored in the reference's own scene files.

data/scene_10.json and data/scene_500.json (copied as fixtures under tests/golden/scenes/) were written by the upstream
scene generator and carry, for each of their 19 / 999 BVH nodes, the box that generator's code computed -- in f32, the
template's precision -- from Sphere::bounding_box (objects/sphere.rs:56-61: center -+ radius), AABB::surrounding_box
(objects/aabb.rs:33-45: component-wise min / max) and BVHNode::construct (objects/bvh.rs:47-58: surrounding_box of the two
children).  Nothing in the reference reads these files, but they are reference outputs for exactly those three functions and
for the tree wiring of the scene-file format.  Both loaders discard the stored box and
recompute it, so it is an independent check of:

  * the oracle:  Sphere / BVHNode_construct built node by node from the file, `bounding_box` of every node;
  * the product: the C++ loader (rt_scene_load_file), walked through rt_scene_root / rt_object_describe /
    rt_object_children, `rt_object_bounding_box` of every node; and the builder path (rt_object_sphere, rt_object_bvh_node).

Tolerance: the file's numbers are shortest-round-trip decimals of f32 values; oracle and product parse them as f64 (the
reference's Vec3 is f64).  (a) In f32 arithmetic on the f32 values the stored boxes are reproduced BIT FOR BIT
(center -+ radius rounded to f32, then exact min / max), along the PRODUCT's own wiring and leaf parameters.  (b) The f64 boxes
of oracle and product equal each other exactly and lie within two f32 ulps (of the largest operand) of the stored values:
each decimal (centre, radius, stored corner) is within half an f32 ulp of its f32 value, and the generator's f32 subtraction
rounded once more.
"""
import json

import numpy as np
import pytest

from conftest import scene_path

F32 = np.float32


def _v(d):
    return (float(d["x"]), float(d["y"]), float(d["z"]))


def _file_nodes(doc):
    """(path, node dict) of every node of the file in depth-first order."""
    out = []

    def walk(d, path):
        out.append((path, d))
        t = d["type"]
        if t == "HitableList":
            for i, it in enumerate(d["items"]):
                walk(it, path + (i,))
        elif t == "BVHNode":
            walk(d["left"], path + (0,))
            walk(d["right"], path + (1,))
    walk(doc["objects"], ())
    return out


def _stored_box(d):
    bb = d["bounding_box"]
    return np.array(_v(bb["min"]) + _v(bb["max"]))


def _ulp32(x):
    return float(np.spacing(F32(abs(x))))


def _build(sc, d, record, path=()):
    """Builds the file's tree node by node on a builder (oracle Scene or rtamd.World); records path -> object id."""
    t = d["type"]
    if t == "HitableList":
        oid = sc.HitableList([_build(sc, it, record, path + (i,)) for i, it in enumerate(d["items"])])
    elif t == "BVHNode":
        left = _build(sc, d["left"], record, path + (0,))
        right = _build(sc, d["right"], record, path + (1,))
        oid = sc.BVHNode_construct(left, right)
    else:
        oid = sc.Sphere(_v(d["center"]), d["radius"], record["mat"])
    record[path] = oid
    return oid


def _f32_box_along_product_wiring(w, oid, cache):
    """The box of product object `oid` recomputed in f32 arithmetic from the product's own leaves and wiring."""
    if oid in cache:
        return cache[oid]
    kind, d = w.describe(oid)
    if kind == "Sphere":
        c, r = np.array(d["v"][:3]).astype(F32), F32(d["v"][3])
        box = np.concatenate([c - r, c + r]).astype(F32)
    else:
        assert kind == "BVHNode" and len(d["children"]) == 2
        a = _f32_box_along_product_wiring(w, d["children"][0], cache)
        b = _f32_box_along_product_wiring(w, d["children"][1], cache)
        box = np.concatenate([np.minimum(a[:3], b[:3]), np.maximum(a[3:], b[3:])])
    cache[oid] = box
    return box


@pytest.mark.parametrize("name,n_bvh,n_spheres", [("scene_10.json", 19, 25), ("scene_500.json", 999, 1005)])
def test_stored_bvh_boxes_pin_oracle_and_product(name, n_bvh, n_spheres):
    import oracle
    import rtamd
    doc = json.load(open(scene_path(name)))
    nodes = _file_nodes(doc)
    bvh_nodes = [(p, d) for p, d in nodes if d["type"] == "BVHNode"]
    assert len(bvh_nodes) == n_bvh and sum(d["type"] == "Sphere" for _, d in nodes) == n_spheres

    # --- oracle and product builder path: the same tree, node by node ---
    o = oracle.Scene()
    rec_o = {"mat": o.Lambertian(o.ConstantTexture((0.5, 0.5, 0.5)))}
    _build(o, doc["objects"], rec_o)
    wb = rtamd.World()
    rec_b = {"mat": wb.Lambertian(wb.ConstantTexture((0.5, 0.5, 0.5)))}
    _build(wb, doc["objects"], rec_b)

    # --- product loader path: ids found by walking the loaded graph in the file's order ---
    wl, _ = rtamd.load_scene_file(scene_path(name))
    rec_l = {}

    def walk(oid, d, path):
        kind, desc = wl.describe(oid)
        rec_l[path] = oid
        if d["type"] == "HitableList":
            assert kind == "HitableList" and len(desc["children"]) == len(d["items"])
            for i, (c, it) in enumerate(zip(desc["children"], d["items"])):
                walk(c, it, path + (i,))
        elif d["type"] == "BVHNode":
            assert kind == "BVHNode" and len(desc["children"]) == 2          # left, right: the file's wiring
            walk(desc["children"][0], d["left"], path + (0,))
            walk(desc["children"][1], d["right"], path + (1,))
        else:
            assert kind == "Sphere"
            assert tuple(desc["v"][:3]) == _v(d["center"]) and desc["v"][3] == float(d["radius"])
    walk(wl.root(), doc["objects"], ())
    assert len(rec_l) == len(nodes)

    cache = {}
    worst = 0.0
    for path, d in bvh_nodes:
        stored = _stored_box(d)
        box_o = o.bounding_box(rec_o[path])
        box_b = wb.bounding_box(rec_b[path])
        box_l = wl.bounding_box(rec_l[path])
        # (b) oracle == product (builder) == product (loader), exactly, in f64
        assert np.array_equal(box_o, box_b) and np.array_equal(box_o, box_l), path
        # ... and within two f32 ulps of the stored f32 value (the operands, centre and radius, set the scale)
        scale = max(abs(stored).max(), 1e-30)
        assert np.all(np.abs(box_o - stored) <= 2 * _ulp32(scale)), (path, box_o, stored)
        worst = max(worst, float(np.abs(box_o - stored).max() / _ulp32(scale)))
        # (a) f32 arithmetic along the product's wiring reproduces the stored box bit for bit
        box32 = _f32_box_along_product_wiring(wl, rec_l[path], cache)
        assert np.array_equal(box32, stored.astype(F32)), (path, box32, stored)
    assert worst <= 2.0
    # the stored values really are f32 numbers (shortest round-trip decimals): parsing as f64 and rounding to f32 is lossless
    for _, d in bvh_nodes[:50]:
        s = _stored_box(d)
        assert np.array_equal(s.astype(F32).astype(np.float64).astype(F32), s.astype(F32))


def test_left_right_wiring_matters_to_the_pin():
    """The pin is sensitive: swapping two leaves of scene_10 changes some recomputed box."""
    doc = json.load(open(scene_path("scene_10.json")))
    bvh = [d for _, d in _file_nodes(doc) if d["type"] == "BVHNode"]
    leafy = [d for d in bvh if d["left"]["type"] == "Sphere" and d["right"]["type"] == "Sphere"]
    a, b = leafy[0], leafy[-1]
    a["left"], b["left"] = b["left"], a["left"]
    import oracle
    o = oracle.Scene()
    rec = {"mat": o.Lambertian(o.ConstantTexture((0.5, 0.5, 0.5)))}
    _build(o, doc["objects"], rec)
    changed = 0
    for path, d in _file_nodes(doc):
        if d["type"] == "BVHNode":
            stored = _stored_box(d)
            if np.any(np.abs(o.bounding_box(rec[path]) - stored) > 2 * _ulp32(abs(stored).max())):
                changed += 1
    assert changed >= 2
