"""rtamd -- Python host binding of librtamd.so (the C ABI in include/rtamd.h).

Mirrors the reference's host-side interface for the radiance path (names follow
/root/reference/raytracer/src): World / Hitable constructors (Sphere, XYRectangle,
..., BVHNode, Transform, Mesh, Cube), Material / Texture constructors, Camera and
Camera.capture_image.  Everything here is plumbing over ctypes; all intersection
and shading happens in the HIP kernels.  There is no CPU fallback: rendering
without the built library or without a HIP device raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("RTAMD_LIB") or os.path.join(_ROOT, "librtamd.so")  # RTAMD_LIB: A/B builds only

RT_OK = 0
ERR_NAMES = {
    -1: "RT_ERR_ARG", -2: "RT_ERR_UNIT_ZERO", -3: "RT_ERR_NO_BBOX", -4: "RT_ERR_SINGULAR", -5: "RT_ERR_IO",
    -6: "RT_ERR_SCHEMA", -7: "RT_ERR_NO_NORMALS", -8: "RT_ERR_NOT_COMMITTED", -9: "RT_ERR_NO_DEVICE",
    -10: "RT_ERR_UNSUPPORTED", -11: "RT_ERR_HIP", -12: "RT_ERR_INTERNAL",
}


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "RT_ERR"), code, msg))
        self.code = code


class rt_camera(C.Structure):
    _fields_ = [("look_from", C.c_double * 3), ("look_at", C.c_double * 3), ("vup", C.c_double * 3), ("vfov", C.c_double),
                ("aspect", C.c_double), ("aperture", C.c_double), ("focus_dist", C.c_double)]


class rt_camera_frame(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("lower_left_corner", C.c_double * 3), ("horizontal", C.c_double * 3),
                ("vertical", C.c_double * 3), ("u", C.c_double * 3), ("v", C.c_double * 3), ("w", C.c_double * 3), ("lens_radius", C.c_double)]


class rt_params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32), ("t_min", C.c_double),
                ("seed", C.c_uint64), ("rank", C.c_int32), ("world", C.c_int32), ("spp_chunk", C.c_int32), ("kernel", C.c_int32),
                ("device", C.c_int32), ("integrator", C.c_int32), ("time0", C.c_double), ("time1", C.c_double)]


class rt_stats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("kernel_ms", C.c_double), ("reduce_ms", C.c_double), ("samples", C.c_uint64),
                ("launches", C.c_int32), ("kernel_used", C.c_int32), ("scene_in_lds", C.c_int32), ("block_threads", C.c_int32),
                ("grid_blocks", C.c_int32), ("spp_chunk", C.c_int32), ("scene_bytes", C.c_uint64), ("reserved", C.c_uint64 * 4),
                ("upload_ms", C.c_double), ("posted_ms", C.c_double), ("stitch_copy_ms", C.c_double), ("comm_init_ms", C.c_double),
                ("exchange_ms", C.c_double)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}
        d["workspace_bytes"] = int(self.reserved[1])
        return d


class rt_sppm_config(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("photons_per_iter", C.c_int32), ("k_global", C.c_int32), ("k_caustic", C.c_int32),
                ("max_bounces", C.c_int32), ("reserved", C.c_int32), ("alpha", C.c_double)]


class rt_tuning(C.Structure):
    _fields_ = [("no_lds", C.c_int32), ("top_nodes", C.c_int32), ("sub_spp", C.c_int32), ("coop_pool", C.c_int32),
                ("max_leaf", C.c_int32), ("sppm_photon_capacity", C.c_int32), ("sppm_knn_candidates", C.c_int32),
                ("multi_force_rccl", C.c_int32), ("wf_workspace_mb", C.c_int32), ("reserved", C.c_int32), ("sah_box_cost", C.c_double)]


class rt_object_desc(C.Structure):
    _fields_ = [("type", C.c_int32), ("material", C.c_int32), ("n_children", C.c_int32), ("axis", C.c_int32), ("v", C.c_double * 8)]


OBJECT_TYPES = ("Sphere", "Rect", "Cube", "Triangle", "Mesh", "Transform", "HitableList", "BVHNode", "ConstantMedium", "MovingSphere")


class rt_scene_info(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_nodes", "n_boxes", "n_spheres", "n_rects", "n_tris", "n_xforms", "n_materials",
                                         "n_textures", "n_verts", "max_depth", "committed", "n_cubes")] + [("bytes", C.c_uint64)] + \
               [(n, C.c_int32) for n in ("accel_ok", "accel_nodes", "accel_items", "accel_instances", "accel_stack", "accel_compact")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("reserved")}


_d3 = C.c_double * 3
_dp = C.POINTER(C.c_double)
_LIB = None

# every symbol include/rtamd.h declares: (name, restype, argtypes)
_SIGS = [
    ("rt_abi_version", C.c_int, []),
    ("rt_last_error", C.c_char_p, []),
    ("rt_default_params", None, [C.POINTER(rt_params)]),
    ("rt_device_count", C.c_int, []),
    ("rt_tuning_default", None, [C.POINTER(rt_tuning)]),
    ("rt_tuning_set", C.c_int, [C.POINTER(rt_tuning)]),
    ("rt_release_workspaces", C.c_int64, []),
    ("rt_scene_create", C.c_int, [C.POINTER(C.c_void_p)]),
    ("rt_scene_destroy", None, [C.c_void_p]),
    ("rt_texture_constant", C.c_int, [C.c_void_p, _d3]),
    ("rt_texture_checker", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("rt_texture_image", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint8)]),
    ("rt_texture_noise", C.c_int, [C.c_void_p, C.c_double, C.c_uint64]),
    ("rt_material_lambertian", C.c_int, [C.c_void_p, C.c_int]),
    ("rt_material_metal", C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    ("rt_material_dielectric", C.c_int, [C.c_void_p, C.c_double, C.c_int]),
    ("rt_material_diffuse_light", C.c_int, [C.c_void_p, C.c_int]),
    ("rt_material_isotropic", C.c_int, [C.c_void_p, C.c_int]),
    ("rt_object_constant_medium", C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_int]),
    ("rt_object_sphere", C.c_int, [C.c_void_p, _d3, C.c_double, C.c_int]),
    ("rt_object_moving_sphere", C.c_int, [C.c_void_p, _d3, _d3, C.c_double, C.c_double, C.c_double, C.c_int]),
    ("rt_object_rect_xy", C.c_int, [C.c_void_p] + [C.c_double] * 5 + [C.c_int]),
    ("rt_object_rect_xz", C.c_int, [C.c_void_p] + [C.c_double] * 5 + [C.c_int]),
    ("rt_object_rect_yz", C.c_int, [C.c_void_p] + [C.c_double] * 5 + [C.c_int]),
    ("rt_object_cube", C.c_int, [C.c_void_p, _d3, _d3, C.c_int]),
    ("rt_object_sphere_light", C.c_int, [C.c_void_p, _d3, C.c_double, _d3, C.c_double]),
    ("rt_object_xz_rect_light", C.c_int, [C.c_void_p] + [C.c_double] * 5 + [_d3, C.c_double]),
    ("rt_object_mesh", C.c_int, [C.c_void_p, C.c_int, _dp, _dp, C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_uint64]),
    ("rt_object_mesh_obj", C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_uint64]),
    ("rt_object_transform", C.c_int, [C.c_void_p, _d3, _d3, _d3, C.c_int]),
    ("rt_object_transform_matrix", C.c_int, [C.c_void_p, C.c_double * 16, C.POINTER(C.c_double), C.c_int]),
    ("rt_mesh_data", C.c_int, [C.c_void_p, C.c_int, _dp, _dp]),
    ("rt_object_triangle", C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]),
    ("rt_object_list", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    ("rt_object_bvh_node", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("rt_object_bvh_build", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_uint64]),
    ("rt_object_bounding_box", C.c_int, [C.c_void_p, C.c_int, C.c_double * 6]),
    ("rt_scene_root", C.c_int, [C.c_void_p]),
    ("rt_object_describe", C.c_int, [C.c_void_p, C.c_int, C.POINTER(rt_object_desc)]),
    ("rt_object_children", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    ("rt_world_new", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_uint64]),
    ("rt_scene_set_root", C.c_int, [C.c_void_p, C.c_int]),
    ("rt_scene_set_lights", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    ("rt_scene_cornell_box", C.c_int, [C.c_void_p, C.c_char_p, C.c_double, C.c_uint64, C.POINTER(rt_camera)]),
    ("rt_scene_load_file", C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(rt_camera)]),
    ("rt_scene_commit", C.c_int, [C.c_void_p]),
    ("rt_scene_info_get", C.c_int, [C.c_void_p, C.POINTER(rt_scene_info)]),
    ("rt_scene_fingerprint", C.c_uint64, [C.c_void_p]),
    ("rt_spec_version", C.c_char_p, []),
    ("rt_render", C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), _dp, C.POINTER(rt_stats)]),
    ("rt_render_camera_frame", C.c_int, [C.c_void_p, C.POINTER(rt_camera_frame), C.POINTER(rt_params), _dp, C.POINTER(rt_stats)]),
    ("rt_camera_frame_from", C.c_int, [C.POINTER(rt_camera), C.POINTER(rt_camera_frame)]),
    ("rt_default_sppm_config", None, [C.POINTER(rt_sppm_config)]),
    ("rt_render_sppm", C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.POINTER(rt_sppm_config), _dp, _dp,
                                 C.POINTER(C.c_uint64), C.POINTER(rt_stats)]),
    ("rt_render_multi", C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.c_int, C.POINTER(C.c_int), _dp, C.POINTER(rt_stats)]),
    ("rt_render_multi_camera_frame", C.c_int, [C.c_void_p, C.POINTER(rt_camera_frame), C.POINTER(rt_params), C.c_int, C.POINTER(C.c_int), _dp,
                                      C.POINTER(rt_stats)]),
    ("rt_render_sppm_multi", C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.POINTER(rt_sppm_config), C.c_int, C.POINTER(C.c_int), _dp,
                              C.POINTER(rt_stats)]),
    ("rt_rccl_version", C.c_int, []),
    ("rt_render_tiles_device", C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.c_void_p, C.c_void_p,
                                         C.POINTER(rt_stats)]),
    ("rt_render_sppm_tiles_device", C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.POINTER(rt_sppm_config), C.c_void_p,
                                              C.c_void_p, C.POINTER(rt_stats)]),
    ("rt_render_accumulate_device", C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                              C.POINTER(rt_stats)]),
    ("rt_accum_finalize_device", C.c_int, [C.POINTER(rt_params), C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rt_accum_state_doubles", C.c_int64, [C.POINTER(rt_params)]),
    ("rt_render_accumulate", C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.c_int32, C.c_int32, _dp, C.POINTER(rt_stats)]),
    ("rt_accum_finalize", C.c_int, [C.POINTER(rt_params), _dp, _dp]),
    ("rt_tiles_total", C.c_int64, [C.POINTER(rt_params)]),
    ("rt_tiles_owned", C.c_int64, [C.POINTER(rt_params)]),
    ("rt_assemble_frame_device", C.c_int, [C.POINTER(rt_params), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("rt_tonemap_u8", C.c_int, [_dp, C.c_size_t, C.POINTER(C.c_uint8)]),
    ("rt_write_png", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint8)]),
    ("rt_debug_rng_device", C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]),
    ("rt_debug_rng_host", C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]),
    ("rt_debug_rng_floats", C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("rt_debug_math_device", C.c_int, [C.c_int, C.c_size_t, _dp, _dp, _dp]),
    ("rt_debug_hit_device", C.c_int, [C.c_void_p, C.c_int, C.c_size_t, _dp, C.c_double, C.c_double, _dp]),
    ("rt_debug_schedule", C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
]
ABI_SYMBOLS = [s[0] for s in _SIGS]


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP/HSA runtime and ONE RCCL.  PyTorch-ROCm bundles its own libamdhip64.so.7 and librccl.so.1
    (same SONAMEs as /opt/rocm's, which librtamd.so names as its dependencies): whichever copy is loaded first serves both.  If torch is
    installed but not imported yet, import it FIRST, so that a later `import torch` in the same process (bench.py, the tests) does not
    find the system's runtime under its own kernels -- "No HIP GPUs are available" -- or the system's RCCL under its `nccl` backend.
    (Pre-loading torch's libraries one by one instead is not safe: with librccl.so loaded before torch's own loader asks for it the
    process ends in `double free or corruption` at exit.)  RTAMD_HIP_RUNTIME=system skips this: /opt/rocm's runtime and RCCL, for hosts
    that never import torch."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("RTAMD_HIP_RUNTIME", "") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None:
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def lib():
    """Load librtamd.so.  Raises (never falls back) when the HIP extension has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise ImportError("librtamd.so is missing at %s -- run `make -C rust-raytracer_amd` (or __graft_entry__.build()); "
                          "there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, res, args in _SIGS:
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def _chk(rc):
    if rc < 0:
        raise RtError(rc, lib().rt_last_error().decode("utf-8", "replace"))
    return rc


def device_count():
    return lib().rt_device_count()


def default_params(**kw):
    p = rt_params()
    lib().rt_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _arr3(v):
    return _d3(*[float(x) for x in v])


class Camera:
    """Camera::new (camera.rs:24-55) arguments; capture_image renders through a World."""

    def __init__(self, look_from_to, vup, vfov, aspect_ratio, aperture, focus_dist):
        look_from, look_at = look_from_to
        self.c = rt_camera(_arr3(look_from), _arr3(look_at), _arr3(vup), float(vfov), float(aspect_ratio), float(aperture),
                           float(focus_dist))

    @classmethod
    def from_struct(cls, c):
        self = cls.__new__(cls)
        self.c = c
        return self

    def frame(self):
        """the Camera struct's stored fields (camera.rs:12-21) as an rt_camera_frame: rt_camera_frame_from = Camera::new"""
        f = rt_camera_frame()
        _chk(lib().rt_camera_frame_from(C.byref(self.c), C.byref(f)))
        return f

    def with_aspect(self, aspect):
        c = rt_camera.from_buffer_copy(self.c)
        c.aspect = float(aspect)
        return Camera.from_struct(c)

    def capture_image(self, world, width=800, height=800, sample_per_pixel=256, **kw):
        """Camera::capture_image (camera.rs:66-128): returns an RgbImage as uint8 [H,W,3]."""
        rad, _ = world.render(self, width=width, height=height, spp=sample_per_pixel, **kw)
        return tonemap_u8(rad)


class World:
    """World (world.rs:8-30) + the Hitable/Material/Texture constructors that populate it."""

    def __init__(self, handle=None):
        self.L = lib()
        if handle is None:
            h = C.c_void_p()
            _chk(self.L.rt_scene_create(C.byref(h)))
            handle = h
        self.h = handle

    def __del__(self):
        try:
            if self.h:
                self.L.rt_scene_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # --- textures (material.rs:48-50) ---
    def ConstantTexture(self, color):
        return _chk(self.L.rt_texture_constant(self.h, _arr3(color)))

    def CheckerTexture(self, t0, t1):
        return _chk(self.L.rt_texture_checker(self.h, t0, t1))

    def ImageTexture(self, rgb_u8):
        a = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
        return _chk(self.L.rt_texture_image(self.h, a.shape[1], a.shape[0], a.ctypes.data_as(C.POINTER(C.c_uint8))))

    def NoiseTexture(self, scale, seed=1):
        """book-2 extension (no reference code): Perlin marble texture 0.5 (1 + sin(scale z + 10 turb(p)))"""
        return _chk(self.L.rt_texture_noise(self.h, float(scale), int(seed)))

    # --- materials (material.rs:88-212) ---
    def Lambertian(self, albedo):
        return _chk(self.L.rt_material_lambertian(self.h, albedo))

    def Metal(self, albedo, fuzz):
        return _chk(self.L.rt_material_metal(self.h, albedo, float(fuzz)))

    def Dielectric(self, ir, albedo):
        return _chk(self.L.rt_material_dielectric(self.h, float(ir), albedo))

    def DiffuseLight(self, emit):
        return _chk(self.L.rt_material_diffuse_light(self.h, emit))

    def Isotropic(self, albedo):
        return _chk(self.L.rt_material_isotropic(self.h, albedo))

    # --- hitables (objects/*.rs, light.rs) ---
    def Sphere(self, center, radius, material):
        return _chk(self.L.rt_object_sphere(self.h, _arr3(center), float(radius), material))

    def MovingSphere(self, center0, center1, time0, time1, radius, material):
        """book-2 extension (no reference code): a sphere whose centre moves linearly from center0 at time0 to center1 at time1"""
        return _chk(self.L.rt_object_moving_sphere(self.h, _arr3(center0), _arr3(center1), float(time0), float(time1), float(radius), material))

    def XYRectangle(self, xy0, xy1, z, material):
        return _chk(self.L.rt_object_rect_xy(self.h, float(xy0[0]), float(xy0[1]), float(xy1[0]), float(xy1[1]), float(z), material))

    def XZRectangle(self, xz0, xz1, y, material):
        return _chk(self.L.rt_object_rect_xz(self.h, float(xz0[0]), float(xz0[1]), float(xz1[0]), float(xz1[1]), float(y), material))

    def YZRectangle(self, yz0, yz1, x, material):
        return _chk(self.L.rt_object_rect_yz(self.h, float(yz0[0]), float(yz0[1]), float(yz1[0]), float(yz1[1]), float(x), material))

    def Cube(self, box_min, box_max, material):
        return _chk(self.L.rt_object_cube(self.h, _arr3(box_min), _arr3(box_max), material))

    def SphereDiffuseLight(self, center, radius, flux, scale=1.0):
        return _chk(self.L.rt_object_sphere_light(self.h, _arr3(center), float(radius), _arr3(flux), float(scale)))

    def XZRectLight(self, xz0, xz1, y, flux, scale=1.0):
        return _chk(self.L.rt_object_xz_rect_light(self.h, float(xz0[0]), float(xz0[1]), float(xz1[0]), float(xz1[1]), float(y),
                                                   _arr3(flux), float(scale)))

    def ConstantMedium(self, density, boundary, phase_function):
        return _chk(self.L.rt_object_constant_medium(self.h, float(density), boundary, phase_function))

    def Mesh(self, positions, normals, indices, material, synthesize_normals=False, bvh_seed=1):
        p = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
        i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        n_ptr = None
        if normals is not None:
            n = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
            assert n.shape == p.shape
            n_ptr = n.ctypes.data_as(_dp)
        return _chk(self.L.rt_object_mesh(self.h, p.shape[0], p.ctypes.data_as(_dp), n_ptr, i.shape[0],
                                          i.ctypes.data_as(C.POINTER(C.c_uint32)), material, int(synthesize_normals), int(bvh_seed)))

    def Mesh_load_obj(self, obj_file, material, synthesize_normals=False, bvh_seed=1):
        return _chk(self.L.rt_object_mesh_obj(self.h, os.fsencode(obj_file), material, int(synthesize_normals), int(bvh_seed)))

    def Transform(self, rotate_in_degree, scale, translate, obj):
        return _chk(self.L.rt_object_transform(self.h, _arr3(rotate_in_degree), _arr3(scale), _arr3(translate), obj))

    def Transform_from_matrix(self, trans, obj, inverse_trans=None):
        """Transform as the reference stores it: the composed 4x4 (row-major) and optionally its inverse."""
        m = (C.c_double * 16)(*[float(x) for x in np.asarray(trans, dtype=np.float64).reshape(16)])
        inv = None
        if inverse_trans is not None:
            inv = (C.c_double * 16)(*[float(x) for x in np.asarray(inverse_trans, dtype=np.float64).reshape(16)])
        return _chk(self.L.rt_object_transform_matrix(self.h, m, inv, obj))

    def MeshData(self, positions, normals):
        """the shared vertex arrays of a mesh (returns a mesh id for Triangle)"""
        p = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
        n = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
        assert n.shape == p.shape
        return _chk(self.L.rt_mesh_data(self.h, p.shape[0], p.ctypes.data_as(_dp), n.ctypes.data_as(_dp)))

    def Triangle(self, mesh, a, b, c, material):
        return _chk(self.L.rt_object_triangle(self.h, mesh, int(a), int(b), int(c), material))

    def HitableList(self, objects):
        arr = (C.c_int * len(objects))(*objects)
        return _chk(self.L.rt_object_list(self.h, len(objects), arr))

    def BVHNode_construct(self, left, right):
        return _chk(self.L.rt_object_bvh_node(self.h, left, right))

    def BVHNode_new(self, objects, bvh_seed=1):
        arr = (C.c_int * len(objects))(*objects)
        return _chk(self.L.rt_object_bvh_build(self.h, len(objects), arr, int(bvh_seed)))

    def bounding_box(self, obj):
        out = (C.c_double * 6)()
        _chk(self.L.rt_object_bounding_box(self.h, obj, out))
        return np.array(out[:])

    # --- graph introspection ---
    def root(self):
        return _chk(self.L.rt_scene_root(self.h))

    def describe(self, obj):
        """(type name, dict) of an object id: material, axis, parameters v, children ids."""
        d = rt_object_desc()
        _chk(self.L.rt_object_describe(self.h, obj, C.byref(d)))
        kids = (C.c_int * max(1, d.n_children))()
        _chk(self.L.rt_object_children(self.h, obj, d.n_children, kids))
        return OBJECT_TYPES[d.type], {"material": d.material, "axis": d.axis, "v": list(d.v), "children": list(kids[:d.n_children])}

    # --- World::new / commit ---
    def new(self, hitable_list, lights=(), bvh_seed=1):
        """World::new(hitable_list, cam, lights) (world.rs:15-25): root = BVHNode::new(hitable_list); then commit."""
        arr = (C.c_int * len(hitable_list))(*hitable_list)
        _chk(self.L.rt_world_new(self.h, len(hitable_list), arr, int(bvh_seed)))
        if len(lights):
            self.set_lights(list(lights))
        return self.commit()

    def set_lights(self, lights):
        arr = (C.c_int * len(lights))(*lights)
        _chk(self.L.rt_scene_set_lights(self.h, len(lights), arr))

    def set_root(self, obj):
        _chk(self.L.rt_scene_set_root(self.h, obj))
        return self.commit()

    def commit(self):
        _chk(self.L.rt_scene_commit(self.h))
        return self

    def info(self):
        out = rt_scene_info()
        _chk(self.L.rt_scene_info_get(self.h, C.byref(out)))
        return out.as_dict()

    # --- the hot path ---
    def render(self, camera, width=800, height=800, spp=256, max_depth=50, t_min=1e-3, seed=1, rank=0, world=1, spp_chunk=0,
               kernel=0, device=-1, integrator=0, shutter=(0.0, 0.0)):
        """rt_render: linear radiance f64 [H,W,3] on the host + stats dict.  shutter = (time0, time1): the book-2 camera shutter."""
        p = default_params(width=width, height=height, spp=spp, max_depth=max_depth, t_min=t_min, seed=seed, rank=rank, world=world,
                           spp_chunk=spp_chunk, kernel=kernel, device=device, integrator=integrator, time0=float(shutter[0]), time1=float(shutter[1]))
        out = np.zeros((height, width, 3), dtype=np.float64)
        st = rt_stats()
        _chk(self.L.rt_render(self.h, C.byref(camera.c), C.byref(p), out.ctypes.data_as(_dp), C.byref(st)))
        return out, st.as_dict()

    def render_multi(self, camera, devices=None, gpus=0, width=800, height=800, spp=256, max_depth=50, t_min=1e-3, seed=1, spp_chunk=0,
                     kernel=0, integrator=0, shutter=(0.0, 0.0)):
        """rt_render_multi: the frame across the GPUs of this node in ONE call (one host thread per rank inside the library, rows
        gathered on devices[0] through RCCL).  devices = HIP ordinals, one per rank (may repeat); or gpus = N for devices 0..N-1
        (0 = all visible).  Returns (radiance [H,W,3], [per-rank stats dicts]); stats[0] also carries 'exchange_seconds' and
        'rows_through_rccl'."""
        p = default_params(width=width, height=height, spp=spp, max_depth=max_depth, t_min=t_min, seed=seed, spp_chunk=spp_chunk,
                           kernel=kernel, integrator=integrator, time0=float(shutter[0]), time1=float(shutter[1]))
        n = len(devices) if devices is not None else int(gpus)
        ids = (C.c_int * n)(*[int(d) for d in devices]) if devices is not None else None
        n_st = n if n > 0 else max(1, device_count())
        st = (rt_stats * n_st)()
        out = np.zeros((height, width, 3), dtype=np.float64)
        _chk(self.L.rt_render_multi(self.h, C.byref(camera.c), C.byref(p), n, ids, out.ctypes.data_as(_dp), st))
        ds = [s.as_dict() for s in st]
        ds[0]["exchange_seconds"] = st[0].reserved[2] * 1e-6
        ds[0]["rows_through_rccl"] = int(st[0].reserved[3])
        return out, ds

    def render_sppm_multi(self, camera, devices=None, gpus=0, width=800, height=800, spp=256, max_depth=50, t_min=1e-3, seed=1, kernel=0, **sppm):
        """rt_render_sppm_multi: main.rs:52-54 across GPUs (every rank repeats the SPPM pre-pass, renders its tiles)."""
        p = default_params(width=width, height=height, spp=spp, max_depth=max_depth, t_min=t_min, seed=seed, kernel=kernel)
        cfg = rt_sppm_config()
        self.L.rt_default_sppm_config(C.byref(cfg))
        for k, v in sppm.items():
            if not hasattr(cfg, k):
                raise TypeError("unknown SPPM setting %r" % k)
            setattr(cfg, k, v)
        n = len(devices) if devices is not None else int(gpus)
        ids = (C.c_int * n)(*[int(d) for d in devices]) if devices is not None else None
        st = (rt_stats * (n if n > 0 else max(1, device_count())))()
        out = np.zeros((height, width, 3), dtype=np.float64)
        _chk(self.L.rt_render_sppm_multi(self.h, C.byref(camera.c), C.byref(p), C.byref(cfg), n, ids, out.ctypes.data_as(_dp), st))
        return out, [s.as_dict() for s in st]

    def render_camera_frame(self, frame, **kw):
        """rt_render_camera_frame: `frame` is an rt_camera_frame (the Camera struct's stored fields)."""
        width, height = kw.get("width", 800), kw.get("height", 800)
        p = default_params(**kw)
        out = np.zeros((height, width, 3), dtype=np.float64)
        st = rt_stats()
        _chk(self.L.rt_render_camera_frame(self.h, C.byref(frame), C.byref(p), out.ctypes.data_as(_dp), C.byref(st)))
        return out, st.as_dict()

    def render_sppm(self, camera, width=800, height=800, spp=256, iterations=50, photons_per_iter=500000, alpha=0.7, k_global=100,
                    k_caustic=50, max_bounces=4096, max_depth=50, t_min=1e-3, seed=1, kernel=0, device=-1):
        """rt_render_sppm = SPPMIntegrator::new + capture_image (main.rs:52-54).
        Returns (radiance [H,W,3], per-pixel stats [H,W,10], (photons in the global maps, in the caustic maps), stats dict)."""
        p = default_params(width=width, height=height, spp=spp, max_depth=max_depth, t_min=t_min, seed=seed, kernel=kernel, device=device)
        cfg = rt_sppm_config()
        self.L.rt_default_sppm_config(C.byref(cfg))
        cfg.iterations, cfg.photons_per_iter, cfg.k_global, cfg.k_caustic = iterations, photons_per_iter, k_global, k_caustic
        cfg.max_bounces, cfg.alpha = max_bounces, alpha
        out = np.zeros((height, width, 3), dtype=np.float64)
        stats = np.zeros((height, width, 10), dtype=np.float64)
        tot = (C.c_uint64 * 2)()
        st = rt_stats()
        _chk(self.L.rt_render_sppm(self.h, C.byref(camera.c), C.byref(p), C.byref(cfg), out.ctypes.data_as(_dp), stats.ctypes.data_as(_dp), tot,
                                   C.byref(st)))
        d = st.as_dict()
        d["prepass_seconds"] = st.reserved[0] * 1e-6
        return out, stats, (int(tot[0]), int(tot[1])), d

    def render_tiles_device(self, camera, params, d_tiles_ptr, stream_ptr=None):
        """rt_render_tiles_device: d_tiles_ptr is a raw device pointer (e.g. torch tensor .data_ptr())."""
        st = rt_stats()
        _chk(self.L.rt_render_tiles_device(self.h, C.byref(camera.c), C.byref(params), C.c_void_p(d_tiles_ptr),
                                           C.c_void_p(stream_ptr or 0), C.byref(st)))
        return st.as_dict()

    def render_accumulate(self, camera, params, sample_begin, sample_end, state=None):
        """rt_render_accumulate: samples [sample_begin, sample_end) added to the host-held accumulator state (numpy f64, created when None);
        returns (state, stats).  accum_finalize(params, state) gives the frame once every sample is in."""
        n = int(lib().rt_accum_state_doubles(C.byref(params)))
        if n <= 0:
            raise RtError(n, "bad image size or partition")
        if state is None:
            state = np.zeros(n, dtype=np.float64)
        assert state.dtype == np.float64 and state.size == n and state.flags["C_CONTIGUOUS"]
        st = rt_stats()
        _chk(self.L.rt_render_accumulate(self.h, C.byref(camera.c), C.byref(params), int(sample_begin), int(sample_end),
                                         state.ctypes.data_as(_dp), C.byref(st)))
        return state, st.as_dict()

    def render_accumulate_device(self, camera, params, sample_begin, sample_end, d_accum_ptr, stream_ptr=None):
        """rt_render_accumulate_device: samples [sample_begin, sample_end) of this rank's tiles are added, in index order, to the caller's
        accumulator (raw device pointer, rt_tiles_owned * 64 * 3 f64; sample_begin == 0 initialises it) -- resumable rendering."""
        st = rt_stats()
        _chk(self.L.rt_render_accumulate_device(self.h, C.byref(camera.c), C.byref(params), int(sample_begin), int(sample_end), C.c_void_p(d_accum_ptr),
                                                C.c_void_p(stream_ptr or 0), C.byref(st)))
        return st.as_dict()

    def render_sppm_tiles_device(self, camera, params, d_tiles_ptr, stream_ptr=None, **sppm):
        """rt_render_sppm_tiles_device: the (replicated, deterministic) SPPM pre-pass, then this rank's tiles of the final pass.
        sppm: iterations, photons_per_iter, alpha, k_global, k_caustic, max_bounces (defaults = the reference's constants)."""
        cfg = rt_sppm_config()
        self.L.rt_default_sppm_config(C.byref(cfg))
        for k, v in sppm.items():
            if not hasattr(cfg, k):
                raise TypeError("unknown SPPM setting %r" % k)
            setattr(cfg, k, v)
        st = rt_stats()
        _chk(self.L.rt_render_sppm_tiles_device(self.h, C.byref(camera.c), C.byref(params), C.byref(cfg), C.c_void_p(d_tiles_ptr),
                                                C.c_void_p(stream_ptr or 0), C.byref(st)))
        d = st.as_dict()
        d["prepass_seconds"] = st.reserved[0] * 1e-6
        return d

    def debug_hit(self, rays, t_min=1e-3, t_max=float("inf"), kernel=1):
        r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        out = np.zeros((r.shape[0], 12), dtype=np.float64)
        _chk(self.L.rt_debug_hit_device(self.h, int(kernel), r.shape[0], r.ctypes.data_as(_dp), float(t_min), float(t_max),
                                        out.ctypes.data_as(_dp)))
        return out


def load_scene_file(path):
    """rt_scene_load_file: returns (World, Camera) for data/<name>.json|.yaml."""
    L = lib()
    h = C.c_void_p()
    cam = rt_camera()
    _chk(L.rt_scene_load_file(os.fsencode(path), C.byref(h), C.byref(cam)))
    return World(h), Camera.from_struct(cam)


def select_scene(cube_obj_path, aspect_ratio=1.0, bvh_seed=1):
    """scene.rs:114-116 select_scene(_index) == cornell_box_scene(): returns (World, Camera)."""
    w = World()
    cam = rt_camera()
    _chk(w.L.rt_scene_cornell_box(w.h, os.fsencode(cube_obj_path), float(aspect_ratio), int(bvh_seed), C.byref(cam)))
    w.commit()
    return w, Camera.from_struct(cam)


def set_tuning(**fields):
    """rt_tuning_set: measurement / test hooks (process-wide; the library reads no environment variable).
    set_tuning() with no arguments restores the automatic defaults."""
    t = rt_tuning()
    lib().rt_tuning_default(C.byref(t))
    for k, v in fields.items():
        if not hasattr(t, k):
            raise TypeError("unknown rt_tuning field %r" % k)
        setattr(t, k, v)
    _chk(lib().rt_tuning_set(C.byref(t)))


def release_workspaces():
    """rt_release_workspaces: free the idle per-device render workspaces; returns the bytes released."""
    return int(lib().rt_release_workspaces())


def debug_schedule(tiles_owned, n_waves, s_begin, s_end, sub_spp=8, job_units=2):
    """rt_debug_schedule: (rounds, levels) with levels = [(first round, first unit, first sample, samples per unit, units per job), ...] and the
    closing row (rounds, units, s_end, 0, 0) last.  Host logic only."""
    out = (C.c_int * 25)()
    rounds = _chk(lib().rt_debug_schedule(int(tiles_owned), int(n_waves), int(s_begin), int(s_end), int(sub_spp), int(job_units), out))
    rows = [tuple(out[5 * i:5 * i + 5]) for i in range(5)]
    n = next(i for i, r in enumerate(rows) if r[3] == 0)
    return rounds, rows[:n + 1]


def tiles_owned(params):
    return int(lib().rt_tiles_owned(C.byref(params)))


def tiles_total(params):
    return int(lib().rt_tiles_total(C.byref(params)))


def assemble_frame_device(params, d_gathered_ptr, tiles_per_rank_stride, d_frame_ptr, stream_ptr=None):
    _chk(lib().rt_assemble_frame_device(C.byref(params), C.c_void_p(d_gathered_ptr), int(tiles_per_rank_stride),
                                        C.c_void_p(d_frame_ptr), C.c_void_p(stream_ptr or 0)))


def accum_finalize(params, state):
    """rt_accum_finalize: the frame [H, W, 3] of a complete host-held accumulator state (World.render_accumulate)"""
    out = np.zeros((params.height, params.width, 3), dtype=np.float64)
    _chk(lib().rt_accum_finalize(C.byref(params), state.ctypes.data_as(_dp), out.ctypes.data_as(_dp)))
    return out


def accum_finalize_device(params, d_accum_ptr, d_tiles_ptr, stream_ptr=None):
    """rt_accum_finalize_device: d_tiles = d_accum / params.spp (the end of a resumable render, World.render_accumulate_device)"""
    _chk(lib().rt_accum_finalize_device(C.byref(params), C.c_void_p(d_accum_ptr), C.c_void_p(d_tiles_ptr), C.c_void_p(stream_ptr or 0)))


def tonemap_u8(rgb):
    a = np.ascontiguousarray(rgb, dtype=np.float64)
    out = np.zeros(a.shape, dtype=np.uint8)
    _chk(lib().rt_tonemap_u8(a.ctypes.data_as(_dp), a.size, out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out


def write_png(path, rgb_u8):
    a = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
    _chk(lib().rt_write_png(os.fsencode(path), a.shape[1], a.shape[0], a.ctypes.data_as(C.POINTER(C.c_uint8))))


def debug_rng(seed, pixel, sample, n, device=True):
    out = (C.c_uint64 * n)()
    fn = lib().rt_debug_rng_device if device else lib().rt_debug_rng_host
    _chk(fn(int(seed), int(pixel), int(sample), n, out))
    return [int(x) for x in out]


def debug_rng_floats(seed, pixel, sample, n, lo=-1.0, hi=1.0, device=True):
    """(first n gen::<f64>(), first n gen_range(lo..hi)) of stream (seed, pixel, sample), computed by the device or the host code"""
    g, r = (C.c_double * n)(), (C.c_double * n)()
    _chk(lib().rt_debug_rng_floats(int(seed), int(pixel), int(sample), n, float(lo), float(hi), 1 if device else 0, g, r))
    return [float(x) for x in g], [float(x) for x in r]


def debug_math(op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.zeros_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.float64)
        bp = b.ctypes.data_as(_dp)
    _chk(lib().rt_debug_math_device(op, a.size, a.ctypes.data_as(_dp), bp, out.ctypes.data_as(_dp)))
    return out
