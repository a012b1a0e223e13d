"""The C-ABI library loads on a GPU-less box and exports every symbol include/rtamd.h declares;
host-only entry points work without a device and the render entry points fail loudly."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, scene_path

HEADER = os.path.join(ROOT, "include", "rtamd.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    import rtamd
    assert sorted(rtamd.ABI_SYMBOLS) == declared_symbols()


def test_rust_binding_declares_every_symbol_and_describes_every_reference_type():
    """rust-raytracer_amd/rust/rtamd_ffi.rs cannot be compiled here (no rustc): at least its extern block must list exactly the
    header's symbols with the header's argument counts, its #[repr(C)] structs the header's fields in order, and every Hitable /
    Material / Texture of the reference must have its Describe impl."""
    rs = open(os.path.join(ROOT, "rust-raytracer_amd", "rust", "rtamd_ffi.rs")).read()
    ext = rs[rs.index('extern "C" {'):]
    ext = ext[:ext.index("\n}\n")]
    rust_fns = dict((m.group(1), m.group(2)) for m in re.finditer(r"pub fn (rt_[a-z0-9_]+)\((.*?)\)", ext, flags=re.S))
    assert sorted(rust_fns) == declared_symbols()
    header = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, args in rust_fns.items():
        m = re.search(r"\b%s\s*\((.*?)\)\s*;" % name, header, flags=re.S)
        c_args = [a for a in m.group(1).split(",") if a.strip() not in ("", "void")]
        r_args = [a for a in args.split(",") if a.strip()]
        assert len(c_args) == len(r_args), (name, c_args, r_args)

    def c_fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), header, flags=re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.sub(r"\[.*?\]", "", part.strip().split()[-1]).lstrip("*"))
        return names

    def rs_fields(struct):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % struct, rs, flags=re.S).group(1)
        return [m.group(1).rstrip("_") for m in re.finditer(r"pub ([a-z_0-9]+):", body)]
    for st in ("rt_camera", "rt_camera_frame", "rt_params", "rt_stats", "rt_sppm_config", "rt_tuning", "rt_object_desc", "rt_scene_info"):
        assert rs_fields(st) == c_fields(st), st
    for ty in ("ConstantTexture", "CheckerTexture", "ImageTexture"):
        assert "impl DescribeTexture for %s" % ty in rs
    for ty in ("Lambertian", "Metal", "Dielectric", "DiffuseLight"):
        assert re.search(r"impl<T: Texture \+ 'static> DescribeMaterial for %s<T>" % ty, rs)
    for ty in ("Sphere", "XYRectangle", "XZRectangle", "YZRectangle", "Cube", "Vec<Arc<dyn Hitable>>", "BVHNode", "Triangle", "Mesh", "Transform",
               "ConstantMedium", "SphereDiffuseLight", "XZRectLight"):
        assert "impl DescribeHitable for %s" % ty in rs, ty
    builder = rs[rs.index("impl SceneBuilder {"):rs.index("impl Drop for SceneBuilder")]
    for sym in declared_symbols():
        if sym.startswith(("rt_texture_", "rt_material_", "rt_object_")) and sym not in ("rt_object_describe", "rt_object_children"):
            assert sym + "(" in builder, "SceneBuilder has no method over %s" % sym


def test_library_exports_every_declared_symbol():
    import rtamd
    L = ctypes.CDLL(rtamd.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), "librtamd.so does not export %s" % name
    assert rtamd.lib().rt_abi_version() == 2


def test_struct_layouts_match_header_sizes():
    import rtamd
    assert ctypes.sizeof(rtamd.rt_camera) == 13 * 8
    assert ctypes.sizeof(rtamd.rt_params) == 4 * 4 + 8 + 8 + 6 * 4 + 2 * 8
    assert ctypes.sizeof(rtamd.rt_stats) == 3 * 8 + 8 + 6 * 4 + 8 + 4 * 8 + 5 * 8   # ABI version 2: five f64 behind reserved[]
    p = rtamd.default_params()
    assert (p.width, p.height, p.spp, p.max_depth, p.t_min, p.world) == (800, 800, 256, 50, 0.001, 1)   # main.rs:34-45, camera.rs:73


def test_no_cpu_fallback_without_a_device():
    import rtamd
    if rtamd.device_count() > 0:
        pytest.skip("a HIP device is present")
    world, cam = rtamd.load_scene_file(scene_path("scene_10.json"))
    with pytest.raises(rtamd.RtError) as e:
        world.render(cam, width=8, height=8, spp=1)
    assert e.value.code == -9   # RT_ERR_NO_DEVICE
    with pytest.raises(rtamd.RtError):
        rtamd.debug_rng(1, 0, 0, 4, device=True)


def test_product_never_touches_the_oracle():
    """the product tree must not import, link or open anything under oracle/ (checker != product)."""
    pkg = os.path.join(ROOT, "rust-raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".h", ".hip", ".hpp", ".rs", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "librt_oracle" not in text and "import oracle" not in text and "oracle/" not in text.replace("test oracle", ""), \
                    "%s references the oracle" % os.path.join(dirpath, fn)


def test_host_rng_matches_oracle_spec():
    import oracle
    import rtamd
    for key in [(1, 0, 0), (42, 1439999, 999), (2**63 + 5, 2**40, 3)]:
        assert rtamd.debug_rng(*key, 32, device=False) == oracle.rng_u64(*key, 32)


def test_rng_golden_vector():
    """tests/golden/rng_kat.json (from a pure-Python restatement) pins spec rtamd-rng-3 -- the integer stream and rand 0.8.4's two float
    conversions -- for both C++ restatements."""
    import json
    import oracle
    import rtamd
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "rng_kat.json")))
    for case in kat["cases"]:
        key = (case["seed"], case["pixel"], case["sample"])
        exp = [int(x, 16) for x in case["u64_hex"]]
        assert oracle.rng_u64(*key, len(exp)) == exp
        assert rtamd.debug_rng(*key, len(exp), device=False) == exp
        assert oracle.rng_f64(*key, 4) == case["f64_first4"]
        assert oracle.rng_range(*key, 4, -1.0, 1.0) == case["range_m1_1_first4"]
        assert oracle.rng_range(*key, 4, 0.0, 7.0) == case["range_0_7_first4"]
        g, r = rtamd.debug_rng_floats(*key, 4, -1.0, 1.0, device=False)
        assert g == case["f64_first4"] and r == case["range_m1_1_first4"]
        assert rtamd.debug_rng_floats(*key, 4, 0.0, 7.0, device=False)[1] == case["range_0_7_first4"]


def test_float_draws_have_the_resolution_of_rand_0_8_4():
    """gen::<f64>() = k * 2^-53 (53 bits, Standard), gen_range = 52 mantissa bits scaled: both below the open end, also for the
    all-ones draw (the retry of UniformFloat::sample_single cannot happen for lo in {-1, 0}); product host code == oracle on many streams."""
    import oracle
    import rtamd
    low_bits = 0
    for key in [(3, p, s) for p in range(40) for s in range(5)]:
        g, r = rtamd.debug_rng_floats(*key, 8, -1.0, 1.0, device=False)
        assert g == oracle.rng_f64(*key, 8) and r == oracle.rng_range(*key, 8, -1.0, 1.0)
        for x in g:
            k = x * 2.0 ** 53
            assert 0.0 <= x < 1.0 and k == int(k)
            low_bits |= int(k) & 0x1FFFFF                      # the 21 bits a 32-bit draw does not have
        assert all(-1.0 <= x < 1.0 for x in r)
    assert low_bits == 0x1FFFFF
    top = (2 ** 52 - 1) * 2.0 ** -52                           # the largest value of from_bits(0x3FF0.. | u64 >> 12) - 1
    assert top * 2.0 + -1.0 < 1.0 and top * 1.0 + 0.0 < 1.0 and top * 7.25 + 0.0 < 7.25


def test_tonemap_host_matches_oracle():
    import oracle
    import rtamd
    x = np.concatenate([np.linspace(-0.5, 1.5, 4001), [np.nan, np.inf, -np.inf, 0.0, 1.0, (254.9999 / 255) ** 2]])
    assert np.array_equal(rtamd.tonemap_u8(x), oracle.tonemap_u8(x))


def test_render_sppm_rejects_a_tile_partition():
    """rt_render_sppm is 'one GPU, whole frame' (rtamd.h): world > 1 is an argument error, checked before any device work."""
    import ctypes as C
    import rtamd
    w, cam = rtamd.select_scene(scene_path("cube.obj"), 1.0, 1)
    p = rtamd.default_params(width=16, height=16, spp=1, rank=1, world=2)
    cfg = rtamd.rt_sppm_config()
    w.L.rt_default_sppm_config(C.byref(cfg))
    out = np.zeros((16, 16, 3))
    rc = w.L.rt_render_sppm(w.h, C.byref(cam.c), C.byref(p), C.byref(cfg), out.ctypes.data_as(C.POINTER(C.c_double)), None, None, None)
    assert rc == -1 and b"world must be 1" in w.L.rt_last_error()


def _rust_describe_impls():
    """{reference type: [builder methods its describe_* function calls, in order]} parsed out of rtamd_ffi.rs (uncompiled here: no rustc)"""
    rs = open(os.path.join(ROOT, "rust-raytracer_amd", "rust", "rtamd_ffi.rs")).read()
    rs = "\n".join(line for line in rs.split("\n") if not line.lstrip().startswith("//"))   # (the commented-out Isotropic impl)
    impls = {}
    for m in re.finditer(r"impl(?:<[^>]*>)? Describe(?:Texture|Material|Hitable) for ([A-Za-z_<>: ]+?) \{\n", rs):
        ty = m.group(1).strip()
        body = rs[m.end():]
        body = body[:body.index("\n}\n")]
        fn = re.search(r"fn describe_(?:texture|material|hitable)\(&self, b: &mut SceneBuilder\)[^{]*\{\n(.*?)\n    \}", body, flags=re.S).group(1)
        impls[ty] = re.findall(r"\bb\.([a-z_0-9]+)\(", fn), fn
    return impls


def _integration_builder_table():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2a. Which builder each `Describe` impl calls"):text.index("## 2. Entry point")]
    table = {}
    for line in sec.split("\n"):
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(cells) == 3 and cells[0].startswith("`") and not cells[0].startswith("`reference"):
            table[cells[0].strip("`")] = (re.findall(r"`([a-z_0-9]+)`", cells[1]), cells[2].strip("`"))
    return table


def test_rust_describe_impls_call_the_builders_of_this_table():
    """Which builder entry point each reference type goes through -- the semantic half of the binding that neither a Rust compiler nor the
    symbol / arity / field checks above can see (round 4: `Cube` went over as `b.list(six rectangles)`, whose bounding box is not
    Cube::bounding_box's and whose six items bypass the one-record cube of the device).  INTEGRATION.md s2a is the table."""
    impls = _rust_describe_impls()
    table = _integration_builder_table()
    norm = lambda ty: re.sub(r"<T>$", "", ty)
    assert sorted(norm(t) for t in impls) == sorted(norm(t) for t in table), (sorted(impls), sorted(table))
    header = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    rs = open(os.path.join(ROOT, "rust-raytracer_amd", "rust", "rtamd_ffi.rs")).read()
    for ty, (calls, fn) in impls.items():
        want, abi = table[[t for t in table if norm(t) == norm(ty)][0]]
        if norm(ty) == "Mesh":
            assert "self.bvh.describe_hitable(b)" in fn and calls == []
            continue
        # loops push one id per element: the table lists a repeated call once
        dedup = [c for i, c in enumerate(calls) if not ("for " in fn and i > 0 and calls[i - 1] == c)]
        assert dedup == want, (ty, dedup, want)
        # ... and the builder method of that name really wraps the entry point the table names
        method = re.search(r"pub fn %s\(&mut self.*?\n    \}" % want[-1], rs, flags=re.S).group(0)
        assert abi in method and re.search(r"\b%s\s*\(" % abi, header), (ty, want[-1], abi)
    # the impl that was wrong: a Cube is ONE rt_object_cube with Cube::new's arguments, its material read from the first side
    cube_calls, cube_fn = impls["Cube"]
    assert cube_calls == ["cube"] and "self.box_min" in cube_fn and "self.box_max" in cube_fn and "describe_own_material" in cube_fn and "list" not in cube_calls
    for prim in ("Sphere", "XYRectangle", "XZRectangle", "YZRectangle", "Triangle"):   # every primitive answers the accessor the Cube impl asks
        body = rs[rs.index("impl DescribeHitable for %s {" % prim):]
        assert "fn describe_own_material" in body[:body.index("\n}\n")], prim


def test_cube_built_the_rust_way_has_exactly_cube_bounding_box():
    """the call sequence of the Rust `impl DescribeHitable for Cube` through ctypes: material of the first side, then rt_object_cube(box_min,
    box_max, m).  Its box is (box_min, box_max) bit for bit (Cube::bounding_box, cube.rs:67-69); the six-rectangle list of round 4's impl is
    not (each side padded by 1e-4 along its normal, rectangle.rs:36,74,111) -- and the reference's BVH culls by that box."""
    import rtamd
    w = rtamd.World()
    m = w.Lambertian(w.ConstantTexture((0.73, 0.73, 0.73)))
    mn, mx = (265.0, 0.0, 295.0), (430.0, 330.0, 460.0)      # scene.rs:93-97's tall box
    cube = w.Cube(mn, mx, m)
    assert np.array_equal(w.bounding_box(cube), np.array(mn + mx))
    sides = [w.XYRectangle((mn[0], mn[1]), (mx[0], mx[1]), mn[2], m), w.XYRectangle((mn[0], mn[1]), (mx[0], mx[1]), mx[2], m),
             w.XZRectangle((mn[0], mn[2]), (mx[0], mx[2]), mn[1], m), w.XZRectangle((mn[0], mn[2]), (mx[0], mx[2]), mx[1], m),
             w.YZRectangle((mn[1], mn[2]), (mx[1], mx[2]), mn[0], m), w.YZRectangle((mn[1], mn[2]), (mx[1], mx[2]), mx[0], m)]
    as_list = w.bounding_box(w.HitableList(sides))
    assert not np.array_equal(as_list, np.array(mn + mx)) and np.allclose(as_list, np.array(mn + mx), atol=2e-4)
    # the committed scene holds ONE cube record and no rectangle
    w.set_root(w.HitableList([cube]))
    w.commit()
    info = w.info()
    assert info["n_cubes"] == 1 and info["n_rects"] == 0
