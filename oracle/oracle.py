"""ctypes harness for the CPU oracle (oracle/rt_oracle.cpp).

ORACLE -- TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.  The product never does.

The scene-file reader here (Python json / PyYAML -> oracle builder calls) is
deliberately independent of the product's C++ loader, so a parity test of a
file-loaded scene also cross-checks the two loaders.
Schema: SURVEY.md sA.1 (derived from /root/reference/data/*.json|yaml).
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = C.POINTER(C.c_double)
c_u64_p = C.POINTER(C.c_uint64)
c_u32_p = C.POINTER(C.c_uint32)
c_u8_p = C.POINTER(C.c_uint8)
c_int_p = C.POINTER(C.c_int)


def build(force=False):
    so = os.path.join(_HERE, "librt_oracle.so")
    src = os.path.join(_HERE, "rt_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "librt_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(os.environ.get("ORACLE_LIB") or build())  # ORACLE_LIB: e.g. the -fsanitize=address,undefined build
    L.orc_scene_new.restype = C.c_void_p
    L.orc_scene_free.argtypes = [C.c_void_p]
    L.orc_last_error.restype = C.c_char_p
    L.orc_last_error.argtypes = [C.c_void_p]
    L.orc_tex_constant.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
    L.orc_tex_checker.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_tex_image.argtypes = [C.c_void_p, C.c_int, C.c_int, c_u8_p]
    L.orc_mat_lambertian.argtypes = [C.c_void_p, C.c_int]
    L.orc_mat_metal.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.orc_mat_dielectric.argtypes = [C.c_void_p, C.c_double, C.c_int]
    L.orc_mat_diffuse_light.argtypes = [C.c_void_p, C.c_int]
    L.orc_mat_isotropic.argtypes = [C.c_void_p, C.c_int]
    L.orc_constant_medium.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int]
    L.orc_hit_rng.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, C.c_double, C.c_double, C.c_uint64, C.c_uint64, C.c_uint64,
                              c_double_p, C.POINTER(C.c_int)]
    L.orc_det_ln.argtypes = [C.c_double]
    L.orc_det_ln.restype = C.c_double
    L.orc_sphere.argtypes = [C.c_void_p] + [C.c_double] * 4 + [C.c_int]
    L.orc_rect.argtypes = [C.c_void_p, C.c_int] + [C.c_double] * 5 + [C.c_int]
    L.orc_cube.argtypes = [C.c_void_p, c_double_p, c_double_p, C.c_int]
    L.orc_list.argtypes = [C.c_void_p, C.c_int, c_int_p]
    L.orc_bvh_construct.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_bvh_new.argtypes = [C.c_void_p, C.c_int, c_int_p, C.c_uint64]
    L.orc_mesh.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, C.c_int, c_u32_p, C.c_int, C.c_uint64]
    L.orc_transform.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p, C.c_int]
    L.orc_set_camera.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p] + [C.c_double] * 4
    L.orc_get_camera.argtypes = [C.c_void_p, c_double_p]
    L.orc_set_root.argtypes = [C.c_void_p, C.c_int]
    L.orc_bounding_box.argtypes = [C.c_void_p, C.c_int, c_double_p]
    L.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64,
                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_double_p, c_u64_p, C.c_int]
    L.orc_set_lights.argtypes = [C.c_void_p, C.c_int, c_int_p, c_double_p, c_double_p]
    L.orc_render_sppm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64, c_int_p, C.c_double, C.c_int,
                                  c_double_p, c_double_p, c_u64_p]
    L.orc_hit.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, C.c_double, C.c_double, c_double_p]
    L.orc_aabb_hit.argtypes = [c_double_p, c_double_p, c_double_p, C.c_double, C.c_double]
    L.orc_scatter.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, C.c_int, C.c_double, C.c_double,
                              C.c_uint64, C.c_uint64, C.c_uint64, c_double_p]
    L.orc_camera_ray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, c_double_p]
    L.orc_tonemap_u8.argtypes = [c_double_p, C.c_size_t, c_u8_p]
    L.orc_rng_stream.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, c_u64_p]
    L.orc_rng_f64.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, c_double_p]
    L.orc_det_sin.restype = C.c_double
    L.orc_det_sin.argtypes = [C.c_double]
    L.orc_tex_noise.argtypes = [C.c_void_p, C.c_double, C.c_uint64]
    L.orc_noise_value.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p]
    L.orc_moving_sphere.argtypes = [C.c_void_p, c_double_p, c_double_p, C.c_double, C.c_double, C.c_double, C.c_int]
    L.orc_set_shutter.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.orc_rng_range.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_double, C.c_double, c_double_p]
    L.orc_sample_helper.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, c_double_p, c_double_p]
    L.orc_vec3_op.argtypes = [C.c_int, c_double_p, c_double_p, C.c_double, c_double_p]
    L.orc_schlick.restype = C.c_double
    L.orc_schlick.argtypes = [C.c_double, C.c_double]
    _LIB = L
    return L


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


class OracleError(RuntimeError):
    pass


COUNTER_NAMES = ("n_aabb", "n_sphere", "n_rect", "n_tri", "n_xform", "n_segments", "n_samples")
# SURVEY.md s8d byte weights (reference-precision payload per test)
B_AABB, B_SPHERE, B_RECT, B_TRI, B_XFORM = 56, 36, 44, 160, 256


def algorithmic_bytes(counters, spp):
    """Mean algorithmic bytes per sample (SURVEY.md s8d contract figure)."""
    n = counters["n_samples"]
    tot = (B_AABB * counters["n_aabb"] + B_SPHERE * counters["n_sphere"] + B_RECT * counters["n_rect"]
           + B_TRI * counters["n_tri"] + B_XFORM * counters["n_xform"])
    return tot / n + 24.0 / spp


class Scene:
    """Builder mirroring the reference constructors (names follow the Rust types)."""

    def __init__(self):
        self.L = lib()
        self.h = self.L.orc_scene_new()
        self.camera = None

    def __del__(self):
        try:
            if self.h:
                self.L.orc_scene_free(self.h)
                self.h = None
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc < 0:
            raise OracleError("%s failed: rc=%d %s" % (what, rc, self.L.orc_last_error(self.h).decode()))
        return rc

    # textures / materials
    def ConstantTexture(self, c):
        return self._chk(self.L.orc_tex_constant(self.h, *[float(x) for x in c]), "ConstantTexture")

    def CheckerTexture(self, t0, t1):
        return self._chk(self.L.orc_tex_checker(self.h, t0, t1), "CheckerTexture")

    def NoiseTexture(self, scale, seed=1):
        """D9 (book 2): Perlin marble texture 0.5 (1 + sin(scale z + 10 turb(p))); tables from the stream (seed, PERLIN_KEY, 0)"""
        return self._chk(self.L.orc_tex_noise(self.h, float(scale), int(seed)), "NoiseTexture")

    def noise_value(self, tex, p):
        out = (C.c_double * 3)()
        self._chk(self.L.orc_noise_value(self.h, tex, _d3(p), out), "noise_value")
        return float(out[0]), float(out[1]), float(out[2])

    def ImageTexture(self, rgb_u8):
        a = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
        h, w, _ = a.shape
        return self._chk(self.L.orc_tex_image(self.h, w, h, a.ctypes.data_as(c_u8_p)), "ImageTexture")

    def Lambertian(self, tex):
        return self._chk(self.L.orc_mat_lambertian(self.h, tex), "Lambertian")

    def Metal(self, tex, fuzz):
        return self._chk(self.L.orc_mat_metal(self.h, tex, float(fuzz)), "Metal")

    def Dielectric(self, ir, tex):
        return self._chk(self.L.orc_mat_dielectric(self.h, float(ir), tex), "Dielectric")

    def DiffuseLight(self, tex):
        return self._chk(self.L.orc_mat_diffuse_light(self.h, tex), "DiffuseLight")

    def Isotropic(self, tex):
        return self._chk(self.L.orc_mat_isotropic(self.h, tex), "Isotropic")

    def ConstantMedium(self, density, boundary, phase_function):
        return self._chk(self.L.orc_constant_medium(self.h, float(density), boundary, phase_function), "ConstantMedium")

    # hitables
    def Sphere(self, center, radius, mat):
        return self._chk(self.L.orc_sphere(self.h, float(center[0]), float(center[1]), float(center[2]), float(radius), mat), "Sphere")

    def MovingSphere(self, center0, center1, time0, time1, radius, mat):
        """D9 (book 2): a sphere whose centre moves linearly from center0 at time0 to center1 at time1"""
        return self._chk(self.L.orc_moving_sphere(self.h, _d3(center0), _d3(center1), float(time0), float(time1), float(radius), mat), "MovingSphere")

    def XYRectangle(self, xy0, xy1, z, mat):
        return self._chk(self.L.orc_rect(self.h, 2, float(xy0[0]), float(xy0[1]), float(xy1[0]), float(xy1[1]), float(z), mat), "XYRectangle")

    def XZRectangle(self, xz0, xz1, y, mat):
        return self._chk(self.L.orc_rect(self.h, 1, float(xz0[0]), float(xz0[1]), float(xz1[0]), float(xz1[1]), float(y), mat), "XZRectangle")

    def YZRectangle(self, yz0, yz1, x, mat):
        return self._chk(self.L.orc_rect(self.h, 0, float(yz0[0]), float(yz0[1]), float(yz1[0]), float(yz1[1]), float(x), mat), "YZRectangle")

    def Cube(self, box_min, box_max, mat):
        return self._chk(self.L.orc_cube(self.h, _d3(box_min), _d3(box_max), mat), "Cube")

    def HitableList(self, ids):
        arr = (C.c_int * len(ids))(*ids)
        return self._chk(self.L.orc_list(self.h, len(ids), arr), "HitableList")

    def BVHNode_construct(self, left, right):
        return self._chk(self.L.orc_bvh_construct(self.h, left, right), "BVHNode::construct")

    def BVHNode_new(self, ids, seed):
        arr = (C.c_int * len(ids))(*ids)
        return self._chk(self.L.orc_bvh_new(self.h, len(ids), arr, int(seed)), "BVHNode::new")

    def Mesh(self, positions, normals, indices, mat, seed):
        p = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
        n = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
        i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        assert p.shape == n.shape
        return self._chk(self.L.orc_mesh(self.h, p.shape[0], p.ctypes.data_as(c_double_p), n.ctypes.data_as(c_double_p),
                                         i.shape[0], i.ctypes.data_as(c_u32_p), mat, int(seed)), "Mesh")

    def Transform(self, rotate_in_degree, scale, translate, obj):
        return self._chk(self.L.orc_transform(self.h, _d3(rotate_in_degree), _d3(scale), _d3(translate), obj), "Transform")

    def Camera(self, look_from, look_at, vup, vfov, aspect_ratio, aperture, focus_dist):
        self.camera = dict(look_from=tuple(look_from), look_at=tuple(look_at), vup=tuple(vup), vfov=vfov,
                           aspect=aspect_ratio, aperture=aperture, focus_dist=focus_dist)
        self._chk(self.L.orc_set_camera(self.h, _d3(look_from), _d3(look_at), _d3(vup), float(vfov), float(aspect_ratio),
                                        float(aperture), float(focus_dist)), "Camera::new")

    def set_shutter(self, time0, time1):
        """D9 (book 2): the camera draws a time in [time0, time1) per sample (after the lens sample) when time1 > time0"""
        self._chk(self.L.orc_set_shutter(self.h, float(time0), float(time1)), "shutter")

    def camera_basis(self):
        out = (C.c_double * 22)()
        self._chk(self.L.orc_get_camera(self.h, out), "camera basis")
        return np.array(out[:])

    def set_root(self, obj):
        self._chk(self.L.orc_set_root(self.h, obj), "set_root")

    def World(self, ids, seed):
        """World::new (world.rs:15-25): the whole list goes through BVHNode::new."""
        root = self.BVHNode_new(ids, seed)
        self.set_root(root)
        return root

    def bounding_box(self, obj):
        out = (C.c_double * 6)()
        self._chk(self.L.orc_bounding_box(self.h, obj, out), "bounding_box")
        return np.array(out[:])

    # queries
    def set_lights(self, ids, flux=None, scale=None):
        """World::new's `lights` (world.rs:18).  flux/scale: XZRectLight / SphereDiffuseLight fields (photon power =
        flux * scale, SPPM only); default flux (1,1,1), scale 1."""
        arr = (C.c_int * len(ids))(*ids)
        fl = np.ascontiguousarray(flux if flux is not None else np.ones((len(ids), 3)), dtype=np.float64).reshape(-1)
        sc = np.ascontiguousarray(scale if scale is not None else np.ones(len(ids)), dtype=np.float64).reshape(-1)
        self._chk(self.L.orc_set_lights(self.h, len(ids), arr, fl.ctypes.data_as(c_double_p), sc.ctypes.data_as(c_double_p)), "set_lights")

    def render_sppm(self, width, height, spp, iterations=50, photons_per_iter=500000, alpha=0.7, k_global=100, k_caustic=50,
                    max_bounces=4096, max_depth=50, t_min=1e-3, seed=1, n_workers=None):
        """SPPMIntegrator::new + capture_image (main.rs:52-54): returns (radiance [H,W,3], stats [H,W,10], (n_global, n_caustic))."""
        if n_workers is None:
            n_workers = os.cpu_count() or 1
        cfg = (C.c_int * 5)(iterations, photons_per_iter, k_global, k_caustic, max_bounces)
        out = np.zeros((height, width, 3), dtype=np.float64)
        stats = np.zeros((height, width, 10), dtype=np.float64)
        tot = (C.c_uint64 * 2)()
        self._chk(self.L.orc_render_sppm(self.h, width, height, spp, max_depth, float(t_min), int(seed), cfg, float(alpha), n_workers,
                                         out.ctypes.data_as(c_double_p), stats.ctypes.data_as(c_double_p), tot), "render_sppm")
        return out, stats, (int(tot[0]), int(tot[1]))

    def render(self, width, height, spp, max_depth=50, t_min=1e-3, seed=1, window=None, n_jobs=64, n_workers=None, integrator=0):
        """Camera::capture_image: returns (radiance f64 [wh,ww,3], counters dict).
        integrator 0 = sample_ray (BSDF sampling), 1 = light/cosine mixture pdf."""
        if window is None:
            window = (0, 0, width, height)
        x0, y0, x1, y1 = window
        if n_workers is None:
            n_workers = os.cpu_count() or 1
        out = np.zeros((y1 - y0, x1 - x0, 3), dtype=np.float64)
        cnt = (C.c_uint64 * 7)()
        rc = self.L.orc_render(self.h, width, height, spp, max_depth, float(t_min), int(seed), x0, y0, x1, y1,
                               n_jobs, n_workers, out.ctypes.data_as(c_double_p), cnt, int(integrator))
        self._chk(rc, "render")
        return out, dict(zip(COUNTER_NAMES, [int(c) for c in cnt]))

    def hit(self, orig, direction, t_min=1e-3, t_max=float("inf"), obj=-1, key=None):
        """closest hit of `obj` (default: the root).  key = (seed, pixel, sample): the RNG stream a ConstantMedium draws from;
        self.last_draws then holds how many numbers the query consumed."""
        out = (C.c_double * 12)()
        if key is None:
            self._chk(self.L.orc_hit(self.h, obj, _d3(orig), _d3(direction), float(t_min), float(t_max), out), "hit")
        else:
            n = C.c_int(0)
            self._chk(self.L.orc_hit_rng(self.h, obj, _d3(orig), _d3(direction), float(t_min), float(t_max), int(key[0]), int(key[1]),
                                         int(key[2]), out, C.byref(n)), "hit")
            self.last_draws = n.value
        if out[0] == 0.0:
            return None
        return dict(t=out[1], p=np.array(out[2:5]), normal=np.array(out[5:8]), front_face=bool(out[8]), uv=(out[9], out[10]),
                    prim_id=int(out[11]))

    def scatter(self, mat, ray_o, ray_d, p, normal, front_face, uv=(0.0, 0.0), key=(1, 0, 0)):
        out = (C.c_double * 14)()
        ray6 = (C.c_double * 6)(*[float(x) for x in list(ray_o) + list(ray_d)])
        self._chk(self.L.orc_scatter(self.h, mat, ray6, _d3(p), _d3(normal), int(front_face), float(uv[0]), float(uv[1]),
                                     key[0], key[1], key[2], out), "scatter")
        return dict(kind=int(out[0]), scattered=bool(out[1]), orig=np.array(out[2:5]), dir=np.array(out[5:8]),
                    attenuation=np.array(out[8:11]), emitted=np.array(out[11:14]))

    def camera_ray(self, width, height, x, y, seed=1, sample=0):
        out = (C.c_double * 6)()
        self._chk(self.L.orc_camera_ray(self.h, width, height, x, y, int(seed), int(sample), out), "camera_ray")
        return np.array(out[:3]), np.array(out[3:])


# ---------------------------------------------------------------------------
# free helpers
# ---------------------------------------------------------------------------
def tonemap_u8(rgb):
    a = np.ascontiguousarray(rgb, dtype=np.float64)
    out = np.zeros(a.shape, dtype=np.uint8)
    lib().orc_tonemap_u8(a.ctypes.data_as(c_double_p), a.size, out.ctypes.data_as(c_u8_p))
    return out


def rng_u64(seed, pixel, sample, n):
    out = (C.c_uint64 * n)()
    lib().orc_rng_stream(seed, pixel, sample, n, out)
    return [int(x) for x in out]


def rng_f64(seed, pixel, sample, n):
    out = (C.c_double * n)()
    lib().orc_rng_f64(seed, pixel, sample, n, out)
    return [float(x) for x in out]


def rng_range(seed, pixel, sample, n, lo, hi):
    out = (C.c_double * n)()
    lib().orc_rng_range(seed, pixel, sample, n, lo, hi, out)
    return [float(x) for x in out]


def sample_helper(which, key=(1, 0, 0), normal=(0.0, 0.0, 1.0)):
    out = (C.c_double * 3)()
    rc = lib().orc_sample_helper(which, key[0], key[1], key[2], _d3(normal), out)
    if rc < 0:
        raise OracleError("sample_helper rc=%d" % rc)
    return np.array(out[:])


def vec3_op(op, a, b=None, s=0.0):
    out = (C.c_double * 3)()
    rc = lib().orc_vec3_op(op, _d3(a), _d3(b) if b is not None else None, float(s), out)
    if rc < 0:
        raise OracleError("vec3_op rc=%d" % rc)
    return np.array(out[:])


def aabb_hit(box_min, box_max, orig, direction, t_min, t_max):
    box = (C.c_double * 6)(*[float(x) for x in list(box_min) + list(box_max)])
    return bool(lib().orc_aabb_hit(box, _d3(orig), _d3(direction), float(t_min), float(t_max)))


def det_ln(x):
    return lib().orc_det_ln(float(x))


def schlick(cosine, ref_idx):
    return lib().orc_schlick(float(cosine), float(ref_idx))


# ---------------------------------------------------------------------------
# scene files (schema: SURVEY.md sA.1)
# ---------------------------------------------------------------------------
def _v(d):
    return (float(d["x"]), float(d["y"]), float(d["z"]))


class SchemaError(ValueError):
    pass


def _texture(sc, d):
    t = d.get("type")
    if t == "ConstantTexture":
        return sc.ConstantTexture(_v(d["color"]))
    if t == "CheckerTexture":
        return sc.CheckerTexture(_texture(sc, d["t0"]), _texture(sc, d["t1"]))
    raise SchemaError("unknown texture type %r" % t)


def _material(sc, d):
    t = d.get("type")
    if t == "Lambertian":
        return sc.Lambertian(_texture(sc, d["albedo"]))
    if t == "Metal":  # albedo is a bare vec, not a texture
        return sc.Metal(sc.ConstantTexture(_v(d["albedo"])), d["fuzz"])
    if t == "Dielectric":  # ref_idx only; albedo defaults to (1,1,1)
        return sc.Dielectric(d["ref_idx"], sc.ConstantTexture((1.0, 1.0, 1.0)))
    if t == "DiffuseLight":
        return sc.DiffuseLight(_texture(sc, d["emit"]))
    raise SchemaError("unknown material type %r" % t)


def _object(sc, d):
    t = d.get("type", d.get("object_type"))
    if t == "HitableList":
        return sc.HitableList([_object(sc, it) for it in d["items"]])
    if t == "BVHNode":  # file topology honoured verbatim; bounding_box is redundant and recomputed
        left = _object(sc, d["left"])
        right = _object(sc, d["right"])
        return sc.BVHNode_construct(left, right)
    if t == "Sphere":
        if "material" not in d:
            raise SchemaError("Sphere without material")
        return sc.Sphere(_v(d["center"]), d["radius"], _material(sc, d["material"]))
    raise SchemaError("unknown object type %r" % t)


def load_scene_dict(doc, aspect=None):
    sc = Scene()
    objs = doc["objects"]
    if isinstance(objs, list):  # data/test.json's older schema: bare array
        objs = {"type": "HitableList", "items": objs}
    sc.set_root(_object(sc, objs))
    cam = doc["camera"]
    sc.Camera(_v(cam["look_from"]), _v(cam["look_at"]), _v(cam["vup"]), cam["vfov"],
              cam["aspect"] if aspect is None else aspect, cam["aperture"], cam["focus_dist"])
    return sc


def load_scene_file(path, aspect=None):
    with open(path, "r") as f:
        text = f.read()
    if path.endswith((".yaml", ".yml")):
        import yaml
        doc = yaml.safe_load(text)
    else:
        doc = json.loads(text)
    return load_scene_dict(doc, aspect)


def load_obj(path):
    """tobj::load_obj{single_index, triangulate} (mesh.rs:150-158): f32 coordinates widened to
    f64; one index per unique (v,vt,vn) triple in first-appearance order; fan triangulation.
    Returns (positions [n,3] f64, normals [n,3] f64 or None, indices [m,3] u32) of models[0]."""
    vs, vns = [], []
    uniq = {}
    pos, nrm, idx = [], [], []
    have_n = True
    with open(path, "r") as f:
        for line in f:
            tok = line.split()
            if not tok or tok[0].startswith("#"):
                continue
            if tok[0] == "v":
                vs.append([float(np.float32(float(x))) for x in tok[1:4]])
            elif tok[0] == "vn":
                vns.append([float(np.float32(float(x))) for x in tok[1:4]])
            elif tok[0] == "f":
                face = []
                for ft in tok[1:]:
                    parts = ft.split("/")
                    vi = int(parts[0])
                    vi = vi - 1 if vi > 0 else len(vs) + vi
                    ti = parts[1] if len(parts) > 1 and parts[1] else None
                    ni = None
                    if len(parts) > 2 and parts[2]:
                        ni = int(parts[2])
                        ni = ni - 1 if ni > 0 else len(vns) + ni
                    key = (vi, ti, ni)
                    if key not in uniq:
                        uniq[key] = len(pos)
                        pos.append(vs[vi])
                        if ni is None:
                            have_n = False
                            nrm.append([0.0, 0.0, 0.0])
                        else:
                            nrm.append(vns[ni])
                    face.append(uniq[key])
                for k in range(1, len(face) - 1):
                    idx.append([face[0], face[k], face[k + 1]])
    P = np.array(pos, dtype=np.float64).reshape(-1, 3)
    N = np.array(nrm, dtype=np.float64).reshape(-1, 3) if have_n else None
    return P, N, np.array(idx, dtype=np.uint32).reshape(-1, 3)


def synthesize_normals(P, I):
    """Area-weighted smooth vertex normals for a mesh that ships none (bun315.obj) -- the product's documented extension
    (include/rtamd.h, rt_object_mesh: synthesize_normals), restated independently in plain Python floats: per triangle, in
    index order, n = (pb - pa) x (pc - pa) is added to its three vertices (a, b, c in that order); each sum is divided by its
    length (0 -> (0, 1, 0)).  The reference itself would panic on such a mesh (mesh.rs:62)."""
    import math
    acc = [[0.0, 0.0, 0.0] for _ in range(len(P))]
    Pl = [[float(x) for x in p] for p in P]
    for tri in I:
        a, b, c = int(tri[0]), int(tri[1]), int(tri[2])
        e0 = [Pl[b][i] - Pl[a][i] for i in range(3)]
        e1 = [Pl[c][i] - Pl[a][i] for i in range(3)]
        n = [e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2], e0[0] * e1[1] - e0[1] * e1[0]]
        for v in (a, b, c):
            for i in range(3):
                acc[v][i] += n[i]
    out = np.zeros((len(P), 3))
    for v, n in enumerate(acc):
        l = math.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2])
        out[v] = (0.0, 1.0, 0.0) if l == 0.0 else (n[0] / l, n[1] / l, n[2] / l)
    return out


def cornell_box_scene(cube_obj_path, aspect_ratio=1.0, seed=1):
    """scene.rs:16-112 (cornell_box_scene), numbers verbatim."""
    sc = Scene()
    red = sc.Lambertian(sc.ConstantTexture((0.75, 0.25, 0.25)))
    white = sc.Lambertian(sc.ConstantTexture((0.75, 0.75, 0.75)))
    blue = sc.Lambertian(sc.ConstantTexture((0.25, 0.25, 0.75)))
    # XZRectLight::new (light.rs:134-146): XZRectangle + DiffuseLight(ConstantTexture(flux)); scale is photon-only
    light_mat = sc.DiffuseLight(sc.ConstantTexture((1.0, 1.0, 1.0)))
    P, N, I = load_obj(cube_obj_path)
    light = sc.XZRectangle((213.0, 227.0), (343.0, 332.0), 554.0, light_mat)
    items = [
        sc.YZRectangle((0.0, 0.0), (555.0, 555.0), 555.0, red),
        sc.YZRectangle((0.0, 0.0), (555.0, 555.0), 0.0, blue),
        sc.XZRectangle((0.0, 0.0), (555.0, 555.0), 0.0, white),
        sc.XZRectangle((0.0, 0.0), (555.0, 555.0), 555.0, white),
        sc.XYRectangle((0.0, 0.0), (555.0, 555.0), 555.0, white),
        sc.Sphere((140.0, 100.0, 240.0), 100.0, sc.Dielectric(1.5, sc.ConstantTexture((0.999, 0.999, 0.999)))),
        sc.Sphere((400.0, 100.0, 360.0), 100.0, sc.Metal(sc.ConstantTexture((0.999, 0.999, 0.999)), 0.0)),
        light,
        sc.Transform((0.0, 0.0, 0.0), (50.0, 50.0, 50.0), (100.0, 50.0, 100.0), sc.Mesh(P, N, I, white, seed)),
        sc.Cube((300.0, 0.0, 100.0), (380.0, 100.0, 180.0), white),
    ]
    sc.World(items, seed)
    sc.set_lights([light], flux=[(1.0, 1.0, 1.0)], scale=[1000000.0])  # scene.rs:26-32,110: XZRectLight(.., flux (1,1,1), scale 1e6)
    sc.Camera((278.0, 278.0, -800.0), (278.0, 278.0, 278.0), (0.0, 1.0, 0.0), 50.0, aspect_ratio, 0.0, 10.0)
    return sc
