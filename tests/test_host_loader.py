"""Host-side scene front end (no GPU): file loaders (JSON, YAML), OBJ reader, builders, bounding
boxes and the flattener, cross-checked against the oracle's independent Python loaders."""
import json
import os

import numpy as np
import pytest

from conftest import scene_path


def test_scene_file_counts():
    import rtamd
    exp = {"scene_10.json": (25, 19), "scene_10.yaml": (25, 19), "scene_200_no_bvh.json": (405, 0), "scene_500.json": (1005, 999)}
    for name, (ns, nb) in exp.items():
        w, cam = rtamd.load_scene_file(scene_path(name))
        info = w.info()
        assert (info["n_spheres"], info["n_boxes"]) == (ns, nb), name
        assert info["n_nodes"] == ns + nb and info["committed"] == 1
        assert tuple(cam.c.look_from) == (-6.0, 2.0, -6.0) and cam.c.vfov == 45.0 and cam.c.focus_dist == 8.062258


def test_yaml_and_json_flatten_identically(tmp_path):
    import rtamd
    a, _ = rtamd.load_scene_file(scene_path("scene_10.json"))
    b, _ = rtamd.load_scene_file(scene_path("scene_10.yaml"))
    assert a.info() == b.info()
    # pretty-printed JSON (the reference's own formatting) and flow-style YAML parse to the same scene
    doc = json.load(open(scene_path("scene_10.json")))
    p = tmp_path / "pretty.json"
    p.write_text(json.dumps(doc, indent=4))
    c, _ = rtamd.load_scene_file(str(p))
    assert c.info() == a.info()
    import yaml
    q = tmp_path / "flow.yaml"
    q.write_text(yaml.safe_dump(doc, default_flow_style=True))
    d, _ = rtamd.load_scene_file(str(q))
    assert d.info() == a.info()


def test_loader_errors():
    import rtamd
    with pytest.raises(rtamd.RtError) as e:
        rtamd.load_scene_file(scene_path("test.json"))      # older schema: Sphere without material
    assert e.value.code == -6
    with pytest.raises(rtamd.RtError) as e:
        rtamd.load_scene_file(scene_path("does_not_exist.json"))
    assert e.value.code == -5


@pytest.mark.parametrize("bad", ['{"objects": {"type": "Blob"}, "camera": {}}', '{"objects": {"type": "Sphere"}}', "[1,2", '{"camera": {}}'])
def test_malformed_scene_files(tmp_path, bad):
    import rtamd
    p = tmp_path / "bad.json"
    p.write_text(bad)
    with pytest.raises(rtamd.RtError) as e:
        rtamd.load_scene_file(str(p))
    assert e.value.code == -6


def test_builders_and_bounding_boxes_match_oracle():
    import oracle
    import rtamd
    w, o = rtamd.World(), oracle.Scene()
    mw = w.Lambertian(w.ConstantTexture((0.5, 0.5, 0.5)))
    mo = o.Lambertian(o.ConstantTexture((0.5, 0.5, 0.5)))
    pairs = [
        (w.Sphere((1, 2, 3), 0.5, mw), o.Sphere((1, 2, 3), 0.5, mo)),
        (w.XYRectangle((0, 1), (2, 3), 4, mw), o.XYRectangle((0, 1), (2, 3), 4, mo)),
        (w.XZRectangle((0, 1), (2, 3), 4, mw), o.XZRectangle((0, 1), (2, 3), 4, mo)),
        (w.YZRectangle((0, 1), (2, 3), 4, mw), o.YZRectangle((0, 1), (2, 3), 4, mo)),
        (w.Cube((0, 0, 0), (1, 2, 3), mw), o.Cube((0, 0, 0), (1, 2, 3), mo)),
    ]
    P, N, I = oracle.load_obj(scene_path("cube.obj"))
    mesh_w, mesh_o = w.Mesh(P, N, I, mw, bvh_seed=7), o.Mesh(P, N, I, mo, 7)
    pairs.append((mesh_w, mesh_o))
    pairs.append((w.Transform((10, 20, 30), (2, 3, 4), (5, 6, 7), mesh_w), o.Transform((10, 20, 30), (2, 3, 4), (5, 6, 7), mesh_o)))
    pairs.append((w.HitableList([p[0] for p in pairs[:3]]), o.HitableList([p[1] for p in pairs[:3]])))
    pairs.append((w.BVHNode_construct(pairs[0][0], pairs[4][0]), o.BVHNode_construct(pairs[0][1], pairs[4][1])))
    pairs.append((w.BVHNode_new([p[0] for p in pairs[:5]], bvh_seed=3), o.BVHNode_new([p[1] for p in pairs[:5]], 3)))
    for a, b in pairs:
        assert np.array_equal(w.bounding_box(a), o.bounding_box(b))


def test_obj_reader_matches_tobj_semantics():
    import oracle
    import rtamd
    w = rtamd.World()
    m = w.Lambertian(w.ConstantTexture((1, 1, 1)))
    mesh = w.Mesh_load_obj(scene_path("cube.obj"), m)
    w.set_root(mesh)
    info = w.info()
    P, N, I = oracle.load_obj(scene_path("cube.obj"))
    assert info["n_tris"] == 12 == len(I) and info["n_verts"] == len(P) == 24   # single_index: one vertex per (v,vt,vn) triple
    assert np.allclose(np.linalg.norm(N, axis=1), 1.0)
    # f32 coordinates widened to f64 (mesh.rs:160-172)
    assert float(np.float32(-0.999999)) in P[:, 2]


def test_mesh_without_normals_is_an_error_unless_synthesized():
    import oracle
    import rtamd
    w = rtamd.World()
    m = w.Lambertian(w.ConstantTexture((1, 1, 1)))
    with pytest.raises(rtamd.RtError) as e:
        w.Mesh_load_obj(scene_path("bun315.obj"), m)          # the reference would panic at mesh.rs:62
    assert e.value.code == -7
    mesh = w.Mesh_load_obj(scene_path("bun315.obj"), m, synthesize_normals=True)
    w.set_root(mesh)
    assert w.info()["n_tris"] == 4968
    P, N, I = oracle.load_obj(scene_path("bun315.obj"))
    assert N is None and len(I) == 4968 and len(P) == 2503


def test_error_codes_of_builders():
    import rtamd
    w = rtamd.World()
    m = w.Lambertian(w.ConstantTexture((1, 1, 1)))
    s = w.Sphere((0, 0, 0), 1, m)
    with pytest.raises(rtamd.RtError) as e:
        w.Sphere((0, 0, 0), 1, 99)
    assert e.value.code == -1
    with pytest.raises(rtamd.RtError) as e:
        w.Transform((0, 0, 0), (0, 1, 1), (0, 0, 0), s)       # singular: "Invalid transform matrix" (transform.rs:146)
    assert e.value.code == -4
    empty = w.HitableList([])
    with pytest.raises(rtamd.RtError) as e:
        w.BVHNode_construct(s, empty)                          # "No bounding box in bvh_node constructor." (bvh.rs:57)
    assert e.value.code == -3
    with pytest.raises(rtamd.RtError) as e:
        w.CheckerTexture(w.ConstantTexture((1, 1, 1)), w.CheckerTexture(w.ConstantTexture((0, 0, 0)), w.ConstantTexture((1, 1, 1))))
    assert e.value.code == -1
    w.set_root(s)
    with pytest.raises(rtamd.RtError):
        w.Sphere((0, 0, 0), 1, m)                              # immutable after commit
    with pytest.raises(rtamd.RtError) as e:
        rtamd.Camera(((0, 0, 0), (0, 0, 0)), (0, 1, 0), 40, 1, 0, 1).capture_image(w, 8, 8, 1)
    assert e.value.code in (-2, -9)                            # zero-length view vector, or no device on a CPU box


def test_cornell_box_scene_structure():
    import rtamd
    w, cam = rtamd.select_scene(scene_path("cube.obj"))
    info = w.info()
    # scene.rs:16-112: 5 walls + light = 6 rects, 1 Cube (one record, its 6 sides are scanned by the kernel), 2 spheres, 12 mesh triangles in one Transform
    assert (info["n_rects"], info["n_cubes"], info["n_spheres"], info["n_tris"], info["n_xforms"]) == (6, 1, 2, 12, 1)
    assert tuple(cam.c.look_from) == (278.0, 278.0, -800.0) and cam.c.vfov == 50.0 and cam.c.aperture == 0.0


def test_png_writer(tmp_path):
    import rtamd
    from PIL import Image
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(37, 301, 3), dtype=np.uint8)     # > 64 KiB raw: several stored deflate blocks
    p = str(tmp_path / "x.png")
    rtamd.write_png(p, img)
    assert np.array_equal(np.asarray(Image.open(p)), img)


def test_non_finite_geometry_is_an_argument_error():
    import rtamd
    w = rtamd.World()
    m = w.Lambertian(w.ConstantTexture((1, 1, 1)))
    nan, inf = float("nan"), float("inf")
    for make in (lambda: w.Sphere((nan, 0, 0), 1, m), lambda: w.Sphere((0, 0, 0), inf, m),
                 lambda: w.XZRectangle((0, 0), (1, nan), 2, m), lambda: w.Cube((0, 0, 0), (1, inf, 1), m),
                 lambda: w.Mesh([[0, 0, 0], [1, 0, 0], [0, nan, 0]], [[0, 0, 1]] * 3, [[0, 1, 2]], m),
                 lambda: w.Mesh([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 0, 1], [0, nan, 1], [0, 0, 1]], [[0, 1, 2]], m),
                 lambda: w.Transform((0, 0, nan), (1, 1, 1), (0, 0, 0), w.Sphere((0, 0, 0), 1, m))):
        with pytest.raises(rtamd.RtError) as e:
            make()
        assert e.value.code == -1 and "finite" in str(e.value)


def test_graph_introspection_matches_what_was_built():
    import rtamd
    w = rtamd.World()
    m = w.Metal(w.ConstantTexture((1, 1, 1)), 0.25)
    s = w.Sphere((1, 2, 3), 0.5, m)
    r = w.XZRectangle((0, 1), (2, 3), 4, m)
    c = w.Cube((0, 0, 0), (1, 2, 3), m)
    lst = w.HitableList([s, r, c])
    b = w.BVHNode_construct(s, c)
    assert w.describe(s) == ("Sphere", {"material": m, "axis": 0, "v": [1, 2, 3, 0.5, 0, 0, 0, 0], "children": []})
    kind, d = w.describe(r)
    assert kind == "Rect" and d["axis"] == 1 and d["v"][:5] == [0, 1, 2, 3, 4] and d["material"] == m
    kind, d = w.describe(c)
    assert kind == "Cube" and len(d["children"]) == 6 and all(w.describe(k)[0] == "Rect" for k in d["children"])
    assert w.describe(lst) == ("HitableList", {"material": -1, "axis": 0, "v": [0.0] * 8, "children": [s, r, c]})
    assert w.describe(b)[1]["children"] == [s, c]
    with pytest.raises(rtamd.RtError):
        w.root()
    w.set_root(lst)
    assert w.root() == lst
    with pytest.raises(rtamd.RtError):
        w.describe(10 ** 6)


def test_host_graph_entry_points_for_stored_fields_match_the_constructors():
    """A host that already OWNS the reference's objects hands over what they store: Transform's composed matrix (transform.rs:9-14),
    Triangles on shared vertex arrays under its own BVHNode tree (mesh.rs:8-16,144-146), Camera's derived frame (camera.rs:12-21).
    Built that way, a scene flattens to the same bytes as through the constructor-style entry points."""
    import oracle
    import rtamd
    P, N, I = oracle.load_obj(scene_path("cube.obj"))

    def build(via_stored_fields):
        w = rtamd.World()
        m = w.Lambertian(w.ConstantTexture((0.7, 0.7, 0.2)))
        mesh = w.Mesh(P, N, I, m, bvh_seed=4)
        if not via_stored_fields:
            t = w.Transform((20.0, 35.0, 10.0), (0.7, 1.1, 0.5), (2.5, 1.2, 2.0), mesh)
            root = w.HitableList([t, w.Sphere((0, 0, 0), 1, m)])
            return w, w.set_root(root) and root
        # walk the mesh's own BVH as a Describe visitor would and rebuild it from explicit triangles on a registered vertex array
        md = w.MeshData(P, N)

        seen = {}     # a shared node (Arc clone: BVHNode::new puts a lone object into BOTH children, Q14) is emitted once

        def clone(o):
            if o in seen:
                return seen[o]
            kind, d = w.describe(o)
            if kind == "Triangle":
                a, b, c = (int(x) for x in d["v"][:3])
                seen[o] = w.Triangle(md, a, b, c, d["material"])
            else:
                assert kind == "BVHNode"
                seen[o] = w.BVHNode_construct(clone(d["children"][0]), clone(d["children"][1]))
            return seen[o]
        kind, d = w.describe(mesh)
        assert kind == "Mesh"
        inner = clone(d["children"][0])
        # the composed matrix T*S*Rx*Ry*Rz of the constructor-built Transform, read back through its bounding box is not enough:
        # compose it here exactly as transform.rs:28-106 does (k-ascending products)
        import math
        rx, ry, rz = (math.radians(v) for v in (20.0, 35.0, 10.0))
        T = np.array([[1, 0, 0, 2.5], [0, 1, 0, 1.2], [0, 0, 1, 2.0], [0, 0, 0, 1.0]])
        S = np.diag([0.7, 1.1, 0.5, 1.0])
        RX = np.array([[1, 0, 0, 0], [0, math.cos(rx), -math.sin(rx), 0], [0, math.sin(rx), math.cos(rx), 0], [0, 0, 0, 1.0]])
        RY = np.array([[math.cos(ry), 0, math.sin(ry), 0], [0, 1, 0, 0], [-math.sin(ry), 0, math.cos(ry), 0], [0, 0, 0, 1.0]])
        RZ = np.array([[math.cos(rz), -math.sin(rz), 0, 0], [math.sin(rz), math.cos(rz), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])

        def mm(a, b):
            c = np.zeros((4, 4))
            for i in range(4):
                for j in range(4):
                    acc = a[i, 0] * b[0, j]
                    for k in range(1, 4):
                        acc = acc + a[i, k] * b[k, j]
                    c[i, j] = acc
            return c
        M = mm(mm(mm(mm(T, S), RX), RY), RZ)
        t = w.Transform_from_matrix(M, inner)
        root = w.HitableList([t, w.Sphere((0, 0, 0), 1, m)])
        w.set_root(root)
        return w, root

    wa, ra = build(False)
    wb, rb = build(True)
    ka, da = wa.describe(wa.describe(ra)[1]["children"][0])
    kb, db = wb.describe(wb.describe(rb)[1]["children"][0])
    assert ka == kb == "Transform"
    assert np.array_equal(wa.bounding_box(wa.describe(ra)[1]["children"][0]), wb.bounding_box(wb.describe(rb)[1]["children"][0]))
    ia, ib = wa.info(), wb.info()
    for k in ("n_nodes", "n_boxes", "n_tris", "n_xforms", "accel_nodes", "accel_items", "accel_instances"):
        assert ia[k] == ib[k], k
    # Camera: the stored frame equals Camera::new's, field by field (oracle's camera_basis is the independent restatement)
    cam = rtamd.Camera(((13.0, 2.0, 3.0), (0.0, 0.5, 0.0)), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)
    f = cam.frame()
    o = oracle.Scene()
    o.Camera((13.0, 2.0, 3.0), (0.0, 0.5, 0.0), (0, 1, 0), 20.0, 1.5, 0.1, 10.0)
    basis = o.camera_basis()      # origin, lower_left_corner, horizontal, vertical, u, v, w (3 each), lens_radius
    for i, name in enumerate(("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w")):
        assert tuple(getattr(f, name)) == tuple(basis[3 * i:3 * i + 3]), name
    assert f.lens_radius == basis[21] == 0.05
    with pytest.raises(rtamd.RtError) as e:
        wa2 = rtamd.World()
        wa2.Transform_from_matrix(np.zeros((4, 4)), wa2.Sphere((0, 0, 0), 1, wa2.Lambertian(wa2.ConstantTexture((1, 1, 1)))))
    assert e.value.code == -4


def test_small_bvhs_have_no_empty_children_and_do_not_disable_the_compact_instance_data():
    """Round-2 advice: a BVH whose items all fit one leaf used to get an inner root with an "empty" second child (lo = +inf,
    hi = -inf), which the slab test of the kernels always passes and which the 16-bit quantisation of kernel 5's NodeQ cannot
    represent -- one tiny instance switched the compact data (and with it kernels 5 / 6) off for the whole scene.  Now the root
    of a BVH with >= 2 items always splits into two real children and a single item gets a zero-size second box."""
    import rtamd
    from rtamd import shapes
    w = rtamd.World()
    white = w.Lambertian(w.ConstantTexture((0.7, 0.7, 0.7)))
    P, N, I = shapes.torus(20, 40)                       # 1 600 triangles: a large instance
    big = w.Transform((0.0, 0.0, 0.0), (50.0, 50.0, 50.0), (278.0, 200.0, 278.0), w.Mesh(P, N, I, white, bvh_seed=1))
    tri = w.Mesh(np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]), np.array([[0.0, 0.0, 1.0]] * 3), np.array([[0, 1, 2]], dtype=np.uint32), white)
    one = w.Transform((0.0, 10.0, 0.0), (30.0, 30.0, 30.0), (100.0, 100.0, 100.0), tri)      # a ONE-triangle instance
    P3, N3, I3 = shapes.torus(2, 2)                      # 8 triangles... (2 x 2 x 2): everything in at most two leaves
    few = w.Transform((0.0, 0.0, 0.0), (20.0, 20.0, 20.0), (400.0, 100.0, 100.0), w.Mesh(P3, N3, I3, white, bvh_seed=2))
    w.new([w.XZRectangle((0.0, 0.0), (555.0, 555.0), 0.0, white), big, one, few], bvh_seed=1)
    info = w.info()
    assert info["accel_ok"] == 1 and info["accel_instances"] == 3
    assert info["accel_compact"] == 1                    # was 0 with the empty-child wrappers
    w2 = rtamd.World()                                   # a world of one item: one inner node, two finite boxes
    white = w2.Lambertian(w2.ConstantTexture((0.7, 0.7, 0.7)))
    w2.new([w2.Sphere((0.0, 0.0, 0.0), 1.0, white)], bvh_seed=1)
    assert w2.info()["accel_ok"] == 1 and w2.info()["accel_nodes"] == 1
