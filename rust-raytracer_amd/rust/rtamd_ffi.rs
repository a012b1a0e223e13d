//! rtamd_ffi.rs -- the reference-side binding a maintainer of BlackCloud37/rust-raytracer adds to
//! route `Camera::capture_image` through librtamd.so (C ABI: include/rtamd.h).
//!
//! STATUS: authored, NOT compiled in this repository's build image (no rustc/cargo there, see
//! SURVEY.md s8c).  It is the Rust spelling of what rust-raytracer_amd/host_cpp/rtamd.hpp (C++,
//! compiled and tested) and rust-raytracer_amd/rtamd/__init__.py (ctypes, tested) do; the `extern`
//! block below is checked against include/rtamd.h symbol by symbol (tests/test_abi_symbols.py).
//!
//! How it plugs in (INTEGRATION.md s1 has the same list with line numbers):
//!   1. copy this file to raytracer/src/rtamd_ffi.rs, `mod rtamd_ffi;` in main.rs,
//!      `println!("cargo:rustc-link-lib=dylib=rtamd");` in build.rs;
//!   2. make the three description traits supertraits of the reference's own traits, so that every
//!      `Arc<dyn Hitable>` / `Arc<dyn Material>` / texture can describe itself:
//!        objects/hit.rs:51   pub trait Hitable: Sync + Send + crate::rtamd_ffi::DescribeHitable
//!        material.rs:21      pub trait Material: Send + Sync + crate::rtamd_ffi::DescribeMaterial
//!        material.rs:18      pub trait Texture: Send + Sync + crate::rtamd_ffi::DescribeTexture
//!   3. open five private fields to the crate (`pub(crate)`): Transform.{obj, trans, inverse_trans}
//!      (transform.rs:10-12), Mesh.bvh (mesh.rs:145), SphereDiffuseLight.sphere (light.rs:69),
//!      XZRectLight.area (light.rs:129), AllLights.lights (light.rs:197);
//!   4. main.rs:52-54 becomes
//!        let result = rtamd_ffi::capture_image(&world, &rtamd_ffi::RenderConfig::from_configs(&CONFIGS))?;
//!      (or `capture_image_sppm` for the reference's own SPPM integrator).
//! Nothing else in the crate changes: scenes are still built with the reference's constructors,
//! the object graph is walked ONCE per frame, and no `hit()` of the crate runs any more.
#![allow(non_camel_case_types, dead_code, clippy::too_many_arguments)]
use std::collections::HashMap;
use std::ffi::{c_void, CStr, CString};
use std::os::raw::{c_char, c_double, c_int};
use std::sync::Arc;

use crate::camera::Camera;
use crate::light::{Light, SphereDiffuseLight, XZRectLight};
use crate::material::{CheckerTexture, ConstantTexture, Dielectric, DiffuseLight, ImageTexture, Lambertian, Material, Metal, Texture};
use crate::objects::bvh::BVHNode;
use crate::objects::cube::Cube;
use crate::objects::hit::Hitable;
use crate::objects::medium::ConstantMedium;
use crate::objects::mesh::{Mesh, Triangle};
use crate::objects::rectangle::{XYRectangle, XZRectangle, YZRectangle};
use crate::objects::sphere::Sphere;
use crate::objects::transform::Transform;
use crate::vec3::Vec3;
use crate::world::World;

// ------------------------------------------------------------------ C ABI (include/rtamd.h) ----
#[repr(C)]
pub struct rt_scene {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_camera {
    pub look_from: [c_double; 3],
    pub look_at: [c_double; 3],
    pub vup: [c_double; 3],
    pub vfov: c_double,
    pub aspect: c_double,
    pub aperture: c_double,
    pub focus_dist: c_double,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_camera_frame {
    pub origin: [c_double; 3],
    pub lower_left_corner: [c_double; 3],
    pub horizontal: [c_double; 3],
    pub vertical: [c_double; 3],
    pub u: [c_double; 3],
    pub v: [c_double; 3],
    pub w: [c_double; 3],
    pub lens_radius: c_double,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_params {
    pub width: i32,
    pub height: i32,
    pub spp: i32,
    pub max_depth: i32,
    pub t_min: c_double,
    pub seed: u64,
    pub rank: i32,
    pub world: i32,
    pub spp_chunk: i32,
    pub kernel: i32,
    pub device: i32,
    pub integrator: i32,
    pub time0: c_double,
    pub time1: c_double,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_stats {
    pub seconds: c_double,
    pub kernel_ms: c_double,
    pub reduce_ms: c_double,
    pub samples: u64,
    pub launches: i32,
    pub kernel_used: i32,
    pub scene_in_lds: i32,
    pub block_threads: i32,
    pub grid_blocks: i32,
    pub spp_chunk: i32,
    pub scene_bytes: u64,
    pub reserved: [u64; 4],
    pub upload_ms: c_double,
    pub posted_ms: c_double,
    pub stitch_copy_ms: c_double,
    pub comm_init_ms: c_double,
    pub exchange_ms: c_double,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_sppm_config {
    pub iterations: i32,
    pub photons_per_iter: i32,
    pub k_global: i32,
    pub k_caustic: i32,
    pub max_bounces: i32,
    pub reserved: i32,
    pub alpha: c_double,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_tuning {
    pub no_lds: i32,
    pub top_nodes: i32,
    pub sub_spp: i32,
    pub coop_pool: i32,
    pub max_leaf: i32,
    pub sppm_photon_capacity: i32,
    pub sppm_knn_candidates: i32,
    pub multi_force_rccl: i32,
    pub wf_workspace_mb: i32,
    pub reserved: i32,
    pub sah_box_cost: c_double,
}

/// The library's defaults (`rt_tuning_default`): "automatic" is -1 for `top_nodes` and `sppm_knn_candidates`, NOT zero, so
/// this cannot be derived -- an all-zero block would cap the LDS node cache at 0 and force the out-of-LDS k-nearest selection.
impl Default for rt_tuning {
    fn default() -> Self {
        let mut t = rt_tuning { no_lds: 0, top_nodes: -1, sub_spp: 0, coop_pool: 0, max_leaf: 0, sppm_photon_capacity: 0,
                                sppm_knn_candidates: -1, multi_force_rccl: 0, wf_workspace_mb: 0, reserved: 0, sah_box_cost: 0.0 };
        unsafe { rt_tuning_default(&mut t) };
        t
    }
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_object_desc {
    pub type_: i32,
    pub material: i32,
    pub n_children: i32,
    pub axis: i32,
    pub v: [c_double; 8],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_scene_info {
    pub n_nodes: i32,
    pub n_boxes: i32,
    pub n_spheres: i32,
    pub n_rects: i32,
    pub n_tris: i32,
    pub n_xforms: i32,
    pub n_materials: i32,
    pub n_textures: i32,
    pub n_verts: i32,
    pub max_depth: i32,
    pub committed: i32,
    pub n_cubes: i32,
    pub bytes: u64,
    pub accel_ok: i32,
    pub accel_nodes: i32,
    pub accel_items: i32,
    pub accel_instances: i32,
    pub accel_stack: i32,
    pub accel_compact: i32,
}

#[link(name = "rtamd")]
extern "C" {
    // library
    pub fn rt_abi_version() -> c_int;
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_default_params(p: *mut rt_params);
    pub fn rt_device_count() -> c_int;
    pub fn rt_tuning_default(t: *mut rt_tuning);
    pub fn rt_tuning_set(t: *const rt_tuning) -> c_int;
    pub fn rt_release_workspaces() -> i64;
    // scene graph builders
    pub fn rt_scene_create(out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(s: *mut rt_scene);
    pub fn rt_texture_constant(s: *mut rt_scene, color: *const c_double) -> c_int;
    pub fn rt_texture_checker(s: *mut rt_scene, t0: c_int, t1: c_int) -> c_int;
    pub fn rt_texture_image(s: *mut rt_scene, width: c_int, height: c_int, rgb: *const u8) -> c_int;
    pub fn rt_texture_noise(s: *mut rt_scene, scale: c_double, seed: u64) -> c_int;
    pub fn rt_material_lambertian(s: *mut rt_scene, albedo_tex: c_int) -> c_int;
    pub fn rt_material_metal(s: *mut rt_scene, albedo_tex: c_int, fuzz: c_double) -> c_int;
    pub fn rt_material_dielectric(s: *mut rt_scene, ir: c_double, albedo_tex: c_int) -> c_int;
    pub fn rt_material_diffuse_light(s: *mut rt_scene, emit_tex: c_int) -> c_int;
    pub fn rt_material_isotropic(s: *mut rt_scene, albedo_tex: c_int) -> c_int;
    pub fn rt_object_sphere(s: *mut rt_scene, center: *const c_double, radius: c_double, material: c_int) -> c_int;
    pub fn rt_object_moving_sphere(s: *mut rt_scene, center0: *const c_double, center1: *const c_double, time0: c_double, time1: c_double, radius: c_double, material: c_int) -> c_int;
    pub fn rt_object_rect_xy(s: *mut rt_scene, x0: c_double, y0: c_double, x1: c_double, y1: c_double, z: c_double, material: c_int) -> c_int;
    pub fn rt_object_rect_xz(s: *mut rt_scene, x0: c_double, z0: c_double, x1: c_double, z1: c_double, y: c_double, material: c_int) -> c_int;
    pub fn rt_object_rect_yz(s: *mut rt_scene, y0: c_double, z0: c_double, y1: c_double, z1: c_double, x: c_double, material: c_int) -> c_int;
    pub fn rt_object_cube(s: *mut rt_scene, box_min: *const c_double, box_max: *const c_double, material: c_int) -> c_int;
    pub fn rt_object_sphere_light(s: *mut rt_scene, center: *const c_double, radius: c_double, flux: *const c_double, scale: c_double) -> c_int;
    pub fn rt_object_xz_rect_light(s: *mut rt_scene, x0: c_double, z0: c_double, x1: c_double, z1: c_double, y: c_double, flux: *const c_double, scale: c_double) -> c_int;
    pub fn rt_object_constant_medium(s: *mut rt_scene, density: c_double, boundary: c_int, phase_material: c_int) -> c_int;
    pub fn rt_object_mesh(s: *mut rt_scene, n_vert: c_int, positions: *const c_double, normals: *const c_double, n_tri: c_int, indices: *const u32, material: c_int, synthesize_normals: c_int, bvh_seed: u64) -> c_int;
    pub fn rt_object_mesh_obj(s: *mut rt_scene, obj_path: *const c_char, material: c_int, synthesize_normals: c_int, bvh_seed: u64) -> c_int;
    pub fn rt_object_transform(s: *mut rt_scene, rotate_deg: *const c_double, scale: *const c_double, translate: *const c_double, object: c_int) -> c_int;
    pub fn rt_object_transform_matrix(s: *mut rt_scene, trans: *const c_double, inverse_trans: *const c_double, object: c_int) -> c_int;
    pub fn rt_mesh_data(s: *mut rt_scene, n_vert: c_int, positions: *const c_double, normals: *const c_double) -> c_int;
    pub fn rt_object_triangle(s: *mut rt_scene, mesh: c_int, a: u32, b: u32, c: u32, material: c_int) -> c_int;
    pub fn rt_object_list(s: *mut rt_scene, n: c_int, objects: *const c_int) -> c_int;
    pub fn rt_object_bvh_node(s: *mut rt_scene, left: c_int, right: c_int) -> c_int;
    pub fn rt_object_bvh_build(s: *mut rt_scene, n: c_int, objects: *const c_int, bvh_seed: u64) -> c_int;
    pub fn rt_object_bounding_box(s: *const rt_scene, object: c_int, out_min_max: *mut c_double) -> c_int;
    pub fn rt_scene_root(s: *const rt_scene) -> c_int;
    pub fn rt_object_describe(s: *const rt_scene, object: c_int, out: *mut rt_object_desc) -> c_int;
    pub fn rt_object_children(s: *const rt_scene, object: c_int, capacity: c_int, out: *mut c_int) -> c_int;
    pub fn rt_world_new(s: *mut rt_scene, n: c_int, objects: *const c_int, bvh_seed: u64) -> c_int;
    pub fn rt_scene_set_lights(s: *mut rt_scene, n: c_int, objects: *const c_int) -> c_int;
    pub fn rt_scene_set_root(s: *mut rt_scene, object: c_int) -> c_int;
    pub fn rt_scene_cornell_box(s: *mut rt_scene, cube_obj_path: *const c_char, aspect_ratio: c_double, bvh_seed: u64, cam_out: *mut rt_camera) -> c_int;
    pub fn rt_scene_load_file(path: *const c_char, out: *mut *mut rt_scene, cam_out: *mut rt_camera) -> c_int;
    pub fn rt_scene_commit(s: *mut rt_scene) -> c_int;
    pub fn rt_scene_info_get(s: *const rt_scene, out: *mut rt_scene_info) -> c_int;
    pub fn rt_scene_fingerprint(s: *const rt_scene) -> u64;
    pub fn rt_spec_version() -> *const c_char;
    // the hot path
    pub fn rt_render(s: *const rt_scene, cam: *const rt_camera, p: *const rt_params, out_rgb: *mut c_double, stats: *mut rt_stats) -> c_int;
    pub fn rt_render_camera_frame(s: *const rt_scene, frame: *const rt_camera_frame, p: *const rt_params, out_rgb: *mut c_double, stats: *mut rt_stats) -> c_int;
    pub fn rt_camera_frame_from(cam: *const rt_camera, out: *mut rt_camera_frame) -> c_int;
    pub fn rt_default_sppm_config(c: *mut rt_sppm_config);
    pub fn rt_render_multi(s: *const rt_scene, cam: *const rt_camera, p: *const rt_params, n_devices: c_int, device_ids: *const c_int, out_rgb: *mut c_double, stats: *mut rt_stats) -> c_int;
    pub fn rt_render_multi_camera_frame(s: *const rt_scene, frame: *const rt_camera_frame, p: *const rt_params, n_devices: c_int, device_ids: *const c_int, out_rgb: *mut c_double, stats: *mut rt_stats) -> c_int;
    pub fn rt_render_sppm_multi(s: *const rt_scene, cam: *const rt_camera, p: *const rt_params, cfg: *const rt_sppm_config, n_devices: c_int, device_ids: *const c_int, out_rgb: *mut c_double, stats: *mut rt_stats) -> c_int;
    pub fn rt_rccl_version() -> c_int;
    pub fn rt_render_sppm(s: *const rt_scene, cam: *const rt_camera, p: *const rt_params, cfg: *const rt_sppm_config, out_rgb: *mut c_double, stats_out: *mut c_double, photons_stored: *mut u64, stats: *mut rt_stats) -> c_int;
    pub fn rt_render_tiles_device(s: *const rt_scene, cam: *const rt_camera, p: *const rt_params, d_tiles: *mut c_double, hip_stream: *mut c_void, stats: *mut rt_stats) -> c_int;
    pub fn rt_render_sppm_tiles_device(s: *const rt_scene, cam: *const rt_camera, p: *const rt_params, cfg: *const rt_sppm_config, d_tiles: *mut c_double, hip_stream: *mut c_void, stats: *mut rt_stats) -> c_int;
    pub fn rt_render_accumulate_device(s: *const rt_scene, cam: *const rt_camera, p: *const rt_params, sample_begin: i32, sample_end: i32, d_accum: *mut c_double, hip_stream: *mut c_void, stats: *mut rt_stats) -> c_int;
    pub fn rt_accum_finalize_device(p: *const rt_params, d_accum: *const c_double, d_tiles: *mut c_double, hip_stream: *mut c_void) -> c_int;
    pub fn rt_accum_state_doubles(p: *const rt_params) -> i64;
    pub fn rt_render_accumulate(s: *const rt_scene, cam: *const rt_camera, p: *const rt_params, sample_begin: i32, sample_end: i32, accum_state: *mut c_double, stats: *mut rt_stats) -> c_int;
    pub fn rt_accum_finalize(p: *const rt_params, accum_state: *const c_double, out_rgb: *mut c_double) -> c_int;
    pub fn rt_tiles_total(p: *const rt_params) -> i64;
    pub fn rt_tiles_owned(p: *const rt_params) -> i64;
    pub fn rt_assemble_frame_device(p: *const rt_params, d_gathered: *const c_double, tiles_per_rank_stride: i64, d_frame: *mut c_double, hip_stream: *mut c_void) -> c_int;
    pub fn rt_tonemap_u8(rgb: *const c_double, n_channels: usize, out: *mut u8) -> c_int;
    pub fn rt_write_png(path: *const c_char, width: c_int, height: c_int, rgb: *const u8) -> c_int;
    // diagnostics used by the parity tests
    pub fn rt_debug_rng_device(seed: u64, pixel: u64, sample: u64, n: c_int, out_host: *mut u64) -> c_int;
    pub fn rt_debug_rng_host(seed: u64, pixel: u64, sample: u64, n: c_int, out_host: *mut u64) -> c_int;
    pub fn rt_debug_rng_floats(seed: u64, pixel: u64, sample: u64, n: c_int, lo: f64, hi: f64, on_device: c_int, out_gen: *mut f64, out_range: *mut f64) -> c_int;
    pub fn rt_debug_math_device(op: c_int, n: usize, a_host: *const c_double, b_host: *const c_double, out_host: *mut c_double) -> c_int;
    pub fn rt_debug_hit_device(s: *const rt_scene, kernel: c_int, n: usize, rays_host: *const c_double, t_min: c_double, t_max: c_double, out_host: *mut c_double) -> c_int;
    pub fn rt_debug_schedule(tiles_owned: i64, n_waves: c_int, s_begin: c_int, s_end: c_int, sub_spp: c_int, job_units: c_int, out25: *mut c_int) -> c_int;
}

// ------------------------------------------------------------------ errors ----
/// A negative rt_status plus rt_last_error(); stands where the reference panics (vec3.rs:88,
/// bvh.rs:43,57, transform.rs:146, mesh.rs:62,158).
#[derive(Debug)]
pub struct RtError {
    pub code: i32,
    pub message: String,
}
pub type Id = i32;
fn check(rc: c_int) -> Result<Id, RtError> {
    if rc >= 0 {
        return Ok(rc);
    }
    let message = unsafe { CStr::from_ptr(rt_last_error()) }.to_string_lossy().into_owned();
    Err(RtError { code: rc, message })
}
fn v3(v: &Vec3) -> [c_double; 3] {
    [v.x, v.y, v.z]
}

// ------------------------------------------------------------------ the walk ----
/// Owns an rt_scene while the host's object graph is lowered onto it.  Shared nodes (`Arc` clones --
/// BVHNode::new puts a lone object into both children, bvh.rs:66; one material on many objects) are
/// emitted once: ids are cached by the address of the described value.
pub struct SceneBuilder {
    raw: *mut rt_scene,
    seen: HashMap<usize, Id>,
    meshes: HashMap<usize, Id>, // Arc<Vec<Vec3>> positions pointer -> mesh id
}
impl SceneBuilder {
    pub fn new() -> Result<Self, RtError> {
        let mut raw: *mut rt_scene = std::ptr::null_mut();
        check(unsafe { rt_scene_create(&mut raw) })?;
        Ok(Self { raw, seen: HashMap::new(), meshes: HashMap::new() })
    }
    fn key<T: ?Sized>(x: &T) -> usize {
        x as *const T as *const () as usize
    }
    fn once<T: ?Sized>(&mut self, x: &T, emit: impl FnOnce(&mut Self) -> Result<Id, RtError>) -> Result<Id, RtError> {
        let k = Self::key(x);
        if let Some(id) = self.seen.get(&k) {
            return Ok(*id);
        }
        let id = emit(self)?;
        self.seen.insert(k, id);
        Ok(id)
    }
    pub fn texture(&mut self, t: &dyn Texture) -> Result<Id, RtError> {
        self.once(t, |b| t.describe_texture(b))
    }
    pub fn material(&mut self, m: &dyn Material) -> Result<Id, RtError> {
        self.once(m, |b| m.describe_material(b))
    }
    pub fn hitable(&mut self, h: &dyn Hitable) -> Result<Id, RtError> {
        self.once(h, |b| h.describe_hitable(b))
    }
    // one method per rt_* builder -------------------------------------------------------------
    pub fn constant_texture(&mut self, color: &Vec3) -> Result<Id, RtError> {
        check(unsafe { rt_texture_constant(self.raw, v3(color).as_ptr()) })
    }
    pub fn checker_texture(&mut self, t0: Id, t1: Id) -> Result<Id, RtError> {
        check(unsafe { rt_texture_checker(self.raw, t0, t1) })
    }
    pub fn image_texture(&mut self, width: u32, height: u32, rgb8: &[u8]) -> Result<Id, RtError> {
        assert_eq!(rgb8.len(), (width * height * 3) as usize);
        check(unsafe { rt_texture_image(self.raw, width as c_int, height as c_int, rgb8.as_ptr()) })
    }
    /// book-2 extension (the reference has no noise texture): Perlin marble, tables from the stream (seed, "perlin" key, 0)
    pub fn noise_texture(&mut self, scale: f64, seed: u64) -> Result<Id, RtError> {
        check(unsafe { rt_texture_noise(self.raw, scale, seed) })
    }
    pub fn lambertian(&mut self, albedo: Id) -> Result<Id, RtError> {
        check(unsafe { rt_material_lambertian(self.raw, albedo) })
    }
    pub fn metal(&mut self, albedo: Id, fuzz: f64) -> Result<Id, RtError> {
        check(unsafe { rt_material_metal(self.raw, albedo, fuzz) })
    }
    pub fn dielectric(&mut self, ir: f64, albedo: Id) -> Result<Id, RtError> {
        check(unsafe { rt_material_dielectric(self.raw, ir, albedo) })
    }
    pub fn diffuse_light(&mut self, emit: Id) -> Result<Id, RtError> {
        check(unsafe { rt_material_diffuse_light(self.raw, emit) })
    }
    pub fn isotropic(&mut self, albedo: Id) -> Result<Id, RtError> {
        check(unsafe { rt_material_isotropic(self.raw, albedo) })
    }
    pub fn sphere(&mut self, center: &Vec3, radius: f64, material: Id) -> Result<Id, RtError> {
        check(unsafe { rt_object_sphere(self.raw, v3(center).as_ptr(), radius, material) })
    }
    /// book-2 extension (the reference's Ray has no time): a sphere whose centre moves linearly during the shutter
    pub fn moving_sphere(&mut self, center0: &Vec3, center1: &Vec3, time0: f64, time1: f64, radius: f64, material: Id) -> Result<Id, RtError> {
        check(unsafe { rt_object_moving_sphere(self.raw, v3(center0).as_ptr(), v3(center1).as_ptr(), time0, time1, radius, material) })
    }
    pub fn rect_xy(&mut self, xy0: (f64, f64), xy1: (f64, f64), z: f64, material: Id) -> Result<Id, RtError> {
        check(unsafe { rt_object_rect_xy(self.raw, xy0.0, xy0.1, xy1.0, xy1.1, z, material) })
    }
    pub fn rect_xz(&mut self, xz0: (f64, f64), xz1: (f64, f64), y: f64, material: Id) -> Result<Id, RtError> {
        check(unsafe { rt_object_rect_xz(self.raw, xz0.0, xz0.1, xz1.0, xz1.1, y, material) })
    }
    pub fn rect_yz(&mut self, yz0: (f64, f64), yz1: (f64, f64), x: f64, material: Id) -> Result<Id, RtError> {
        check(unsafe { rt_object_rect_yz(self.raw, yz0.0, yz0.1, yz1.0, yz1.1, x, material) })
    }
    pub fn cube(&mut self, box_min: &Vec3, box_max: &Vec3, material: Id) -> Result<Id, RtError> {
        check(unsafe { rt_object_cube(self.raw, v3(box_min).as_ptr(), v3(box_max).as_ptr(), material) })
    }
    pub fn sphere_light(&mut self, center: &Vec3, radius: f64, flux: &Vec3, scale: f64) -> Result<Id, RtError> {
        check(unsafe { rt_object_sphere_light(self.raw, v3(center).as_ptr(), radius, v3(flux).as_ptr(), scale) })
    }
    pub fn xz_rect_light(&mut self, xz0: (f64, f64), xz1: (f64, f64), y: f64, flux: &Vec3, scale: f64) -> Result<Id, RtError> {
        check(unsafe { rt_object_xz_rect_light(self.raw, xz0.0, xz0.1, xz1.0, xz1.1, y, v3(flux).as_ptr(), scale) })
    }
    pub fn constant_medium(&mut self, density: f64, boundary: Id, phase_function: Id) -> Result<Id, RtError> {
        check(unsafe { rt_object_constant_medium(self.raw, density, boundary, phase_function) })
    }
    pub fn mesh_obj(&mut self, obj_file: &str, material: Id, bvh_seed: u64) -> Result<Id, RtError> {
        let c = CString::new(obj_file).unwrap();
        check(unsafe { rt_object_mesh_obj(self.raw, c.as_ptr(), material, 0, bvh_seed) })
    }
    pub fn mesh_arrays(&mut self, positions: &[Vec3], normals: &[Vec3], indices: &[u32], material: Id, bvh_seed: u64) -> Result<Id, RtError> {
        let p: Vec<f64> = positions.iter().flat_map(|v| [v.x, v.y, v.z]).collect();
        let n: Vec<f64> = normals.iter().flat_map(|v| [v.x, v.y, v.z]).collect();
        check(unsafe {
            rt_object_mesh(self.raw, positions.len() as c_int, p.as_ptr(), n.as_ptr(), (indices.len() / 3) as c_int, indices.as_ptr(), material, 0, bvh_seed)
        })
    }
    /// the vertex arrays a mesh's triangles share (mesh.rs:12-13); cached by the Arc's address
    pub fn mesh_data(&mut self, positions: &Arc<Vec<Vec3>>, normals: &Arc<Vec<Vec3>>) -> Result<Id, RtError> {
        let k = Arc::as_ptr(positions) as usize;
        if let Some(id) = self.meshes.get(&k) {
            return Ok(*id);
        }
        let p: Vec<f64> = positions.iter().flat_map(|v| [v.x, v.y, v.z]).collect();
        let n: Vec<f64> = normals.iter().flat_map(|v| [v.x, v.y, v.z]).collect();
        let id = check(unsafe { rt_mesh_data(self.raw, positions.len() as c_int, p.as_ptr(), if n.is_empty() { std::ptr::null() } else { n.as_ptr() }) })?;
        self.meshes.insert(k, id);
        Ok(id)
    }
    pub fn triangle(&mut self, mesh: Id, a: usize, b: usize, c: usize, material: Id) -> Result<Id, RtError> {
        check(unsafe { rt_object_triangle(self.raw, mesh, a as u32, b as u32, c as u32, material) })
    }
    pub fn transform(&mut self, rotate_in_degree: &Vec3, scale: &Vec3, translate: &Vec3, obj: Id) -> Result<Id, RtError> {
        check(unsafe { rt_object_transform(self.raw, v3(rotate_in_degree).as_ptr(), v3(scale).as_ptr(), v3(translate).as_ptr(), obj) })
    }
    /// row-major 4x4, as `nalgebra::Matrix4` indexes (m[(row, col)])
    pub fn transform_matrix(&mut self, trans: &[f64; 16], inverse_trans: Option<&[f64; 16]>, obj: Id) -> Result<Id, RtError> {
        let inv = inverse_trans.map_or(std::ptr::null(), |m| m.as_ptr());
        check(unsafe { rt_object_transform_matrix(self.raw, trans.as_ptr(), inv, obj) })
    }
    pub fn list(&mut self, items: &[Id]) -> Result<Id, RtError> {
        check(unsafe { rt_object_list(self.raw, items.len() as c_int, items.as_ptr()) })
    }
    pub fn bvh_node(&mut self, left: Id, right: Id) -> Result<Id, RtError> {
        check(unsafe { rt_object_bvh_node(self.raw, left, right) })
    }
    pub fn bvh_build(&mut self, objects: &[Id], bvh_seed: u64) -> Result<Id, RtError> {
        check(unsafe { rt_object_bvh_build(self.raw, objects.len() as c_int, objects.as_ptr(), bvh_seed) })
    }
    pub fn bounding_box(&self, object: Id) -> Result<([f64; 3], [f64; 3]), RtError> {
        let mut b = [0.0f64; 6];
        check(unsafe { rt_object_bounding_box(self.raw, object, b.as_mut_ptr()) })?;
        Ok(([b[0], b[1], b[2]], [b[3], b[4], b[5]]))
    }
    /// World::new(hitable_list, cam, lights): root = BVHNode::new(hitable_list) with seeded split axes
    pub fn world_new(&mut self, hitable_list: &[Id], bvh_seed: u64) -> Result<(), RtError> {
        check(unsafe { rt_world_new(self.raw, hitable_list.len() as c_int, hitable_list.as_ptr(), bvh_seed) }).map(|_| ())
    }
    pub fn set_root(&mut self, object: Id) -> Result<(), RtError> {
        check(unsafe { rt_scene_set_root(self.raw, object) }).map(|_| ())
    }
    pub fn set_lights(&mut self, lights: &[Id]) -> Result<(), RtError> {
        check(unsafe { rt_scene_set_lights(self.raw, lights.len() as c_int, lights.as_ptr()) }).map(|_| ())
    }
    pub fn commit(self) -> Result<Scene, RtError> {
        check(unsafe { rt_scene_commit(self.raw) })?;
        let raw = self.raw;
        std::mem::forget(self);
        Ok(Scene { raw })
    }
}
impl Drop for SceneBuilder {
    fn drop(&mut self) {
        unsafe { rt_scene_destroy(self.raw) }
    }
}

/// A committed (immutable, device-resident on first render) scene.
pub struct Scene {
    raw: *mut rt_scene,
}
unsafe impl Send for Scene {}
unsafe impl Sync for Scene {} // rtamd.h: an rt_scene is immutable after commit and may be rendered concurrently
impl Drop for Scene {
    fn drop(&mut self) {
        unsafe { rt_scene_destroy(self.raw) }
    }
}
impl Scene {
    pub fn load_file(path: &str) -> Result<(Scene, rt_camera), RtError> {
        let c = CString::new(path).unwrap();
        let mut raw: *mut rt_scene = std::ptr::null_mut();
        let mut cam = rt_camera::default();
        check(unsafe { rt_scene_load_file(c.as_ptr(), &mut raw, &mut cam) })?;
        Ok((Scene { raw }, cam))
    }
    pub fn info(&self) -> Result<rt_scene_info, RtError> {
        let mut i = rt_scene_info::default();
        check(unsafe { rt_scene_info_get(self.raw, &mut i) })?;
        Ok(i)
    }
    pub fn root(&self) -> Result<Id, RtError> {
        check(unsafe { rt_scene_root(self.raw) })
    }
    pub fn describe(&self, object: Id) -> Result<(rt_object_desc, Vec<Id>), RtError> {
        let mut d = rt_object_desc::default();
        check(unsafe { rt_object_describe(self.raw, object, &mut d) })?;
        let mut kids = vec![0 as c_int; d.n_children.max(0) as usize];
        check(unsafe { rt_object_children(self.raw, object, kids.len() as c_int, kids.as_mut_ptr()) })?;
        Ok((d, kids))
    }
    /// linear radiance (sum / spp), f64 RGB, row-major, y down: capture_image minus the u8 conversion
    pub fn render(&self, frame: &rt_camera_frame, p: &rt_params) -> Result<(Vec<f64>, rt_stats), RtError> {
        let mut out = vec![0.0f64; (p.width as usize) * (p.height as usize) * 3];
        let mut st = rt_stats::default();
        check(unsafe { rt_render_camera_frame(self.raw, frame, p, out.as_mut_ptr(), &mut st) })?;
        Ok((out, st))
    }
    /// The frame across `gpus` GPUs of this node in ONE call (0 = every visible GPU): the fan-out over the devices, the RCCL
    /// gather of the ranks' tile rows and the stitch happen inside the library, as the pool, the channel and the stitch happen
    /// inside `capture_image` (camera.rs:74-126).  Per-rank statistics come back in the vector.
    pub fn render_multi(&self, frame: &rt_camera_frame, p: &rt_params, gpus: usize) -> Result<(Vec<f64>, Vec<rt_stats>), RtError> {
        let n = if gpus == 0 { (unsafe { rt_device_count() }).max(1) as usize } else { gpus };
        let mut out = vec![0.0f64; (p.width as usize) * (p.height as usize) * 3];
        let mut st = vec![rt_stats::default(); n];
        check(unsafe { rt_render_multi_camera_frame(self.raw, frame, p, n as c_int, std::ptr::null(), out.as_mut_ptr(), st.as_mut_ptr()) })?;
        Ok((out, st))
    }
    /// capture_image in instalments: the sample indices `[begin, end)` of every pixel are added, in index order, to the running sums in
    /// `state` (an empty vector on the first call, `begin == 0`; opaque: write it to disk to checkpoint).  `finish_accumulated` turns a
    /// complete state into the frame `render` would have returned, bit for bit -- the RNG is keyed by (seed, pixel, sample), so the
    /// state and the next sample index are all there is to save.
    pub fn render_accumulate(&self, cam: &rt_camera, p: &rt_params, begin: i32, end: i32, state: &mut Vec<f64>) -> Result<rt_stats, RtError> {
        let n64 = unsafe { rt_accum_state_doubles(p) };
        if n64 <= 0 {
            return Err(RtError { code: n64 as c_int, message: "bad image size or partition".to_string() });
        }
        let n = n64 as usize;
        if state.len() != n {
            if begin > 0 {
                // a state of another size cannot be the sums of samples [0, begin) of THIS frame: refusing beats silently starting over
                return Err(RtError { code: -1, message: format!("render_accumulate: state has {} values, this frame needs {} (begin = {})", state.len(), n, begin) });
            }
            state.clear();
            state.resize(n, 0.0);
        }
        let mut st = rt_stats::default();
        check(unsafe { rt_render_accumulate(self.raw, cam, p, begin, end, state.as_mut_ptr(), &mut st) })?;
        Ok(st)
    }
    pub fn finish_accumulated(p: &rt_params, state: &[f64]) -> Result<Vec<f64>, RtError> {
        let mut out = vec![0.0f64; (p.width as usize) * (p.height as usize) * 3];
        check(unsafe { rt_accum_finalize(p, state.as_ptr(), out.as_mut_ptr()) })?;
        Ok(out)
    }
    pub fn render_sppm_multi(&self, cam: &rt_camera, p: &rt_params, cfg: &rt_sppm_config, gpus: usize) -> Result<(Vec<f64>, Vec<rt_stats>), RtError> {
        let n = if gpus == 0 { (unsafe { rt_device_count() }).max(1) as usize } else { gpus };
        let mut out = vec![0.0f64; (p.width as usize) * (p.height as usize) * 3];
        let mut st = vec![rt_stats::default(); n];
        check(unsafe { rt_render_sppm_multi(self.raw, cam, p, cfg, n as c_int, std::ptr::null(), out.as_mut_ptr(), st.as_mut_ptr()) })?;
        Ok((out, st))
    }
    pub fn render_sppm(&self, cam: &rt_camera, p: &rt_params, cfg: &rt_sppm_config) -> Result<(Vec<f64>, rt_stats), RtError> {
        let mut out = vec![0.0f64; (p.width as usize) * (p.height as usize) * 3];
        let mut st = rt_stats::default();
        let mut stored = [0u64; 2];
        check(unsafe { rt_render_sppm(self.raw, cam, p, cfg, out.as_mut_ptr(), std::ptr::null_mut(), stored.as_mut_ptr(), &mut st) })?;
        Ok((out, st))
    }
}

// ------------------------------------------------------------------ Describe traits ----
/// Implemented below for every Texture / Material / Hitable of the reference; each impl reads only
/// what the type stores and calls the builder entry point that stands for its constructor.
pub trait DescribeTexture {
    fn describe_texture(&self, b: &mut SceneBuilder) -> Result<Id, RtError>;
}
pub trait DescribeMaterial {
    fn describe_material(&self, b: &mut SceneBuilder) -> Result<Id, RtError>;
}
pub trait DescribeHitable {
    fn describe_hitable(&self, b: &mut SceneBuilder) -> Result<Id, RtError>;
    /// The one material of a primitive (a sphere, a rectangle, a triangle), `None` for containers.  `Cube` keeps its six sides as
    /// `Arc<dyn Hitable>` and no material of its own (cube.rs:9-13): its impl asks the first side through this accessor.
    fn describe_own_material(&self, _b: &mut SceneBuilder) -> Result<Option<Id>, RtError> {
        Ok(None)
    }
}

// material.rs:48-84
impl DescribeTexture for ConstantTexture {
    fn describe_texture(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        b.constant_texture(&self.0)
    }
}
impl DescribeTexture for CheckerTexture {
    fn describe_texture(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        let t0 = b.constant_texture(&(self.0).0)?; // .0 is used when sines < 0 (material.rs:62-66)
        let t1 = b.constant_texture(&(self.1).0)?;
        b.checker_texture(t0, t1)
    }
}
impl DescribeTexture for ImageTexture {
    fn describe_texture(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        let rgb = self.0.to_rgb8(); // material.rs:72-81 reads it through GenericImageView::get_pixel, top row first
        b.image_texture(rgb.width(), rgb.height(), rgb.as_raw())
    }
}
// material.rs:88-212
impl<T: Texture + 'static> DescribeMaterial for Lambertian<T> {
    fn describe_material(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        let t = b.texture(&self.albedo)?;
        b.lambertian(t)
    }
}
impl<T: Texture + 'static> DescribeMaterial for Metal<T> {
    fn describe_material(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        let t = b.texture(&self.albedo)?;
        b.metal(t, self.fuzz)
    }
}
impl<T: Texture + 'static> DescribeMaterial for Dielectric<T> {
    fn describe_material(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        let t = b.texture(&self.albedo)?;
        b.dielectric(self.ir, t)
    }
}
impl<T: Texture + 'static> DescribeMaterial for DiffuseLight<T> {
    fn describe_material(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        let t = b.texture(&self.emit)?;
        b.diffuse_light(t)
    }
}
// material.rs:213-231, once un-commented (and given the current `scatter` signature):
// impl<T: Texture + 'static> DescribeMaterial for Isotropic<T> {
//     fn describe_material(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
//         let t = b.texture(&self.albedo)?;
//         b.isotropic(t)
//     }
// }

// objects/sphere.rs:9-13
impl DescribeHitable for Sphere {
    fn describe_hitable(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        let m = b.material(self.material.as_ref())?;
        b.sphere(&self.center, self.radius, m)
    }
    fn describe_own_material(&self, b: &mut SceneBuilder) -> Result<Option<Id>, RtError> {
        Ok(Some(b.material(self.material.as_ref())?))
    }
}
// objects/rectangle.rs:7-12, 44-49, 82-87
impl DescribeHitable for XYRectangle {
    fn describe_hitable(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        let m = b.material(self.material.as_ref())?;
        b.rect_xy(self.xy0, self.xy1, self.z, m)
    }
    fn describe_own_material(&self, b: &mut SceneBuilder) -> Result<Option<Id>, RtError> {
        Ok(Some(b.material(self.material.as_ref())?))
    }
}
impl DescribeHitable for XZRectangle {
    fn describe_hitable(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        let m = b.material(self.material.as_ref())?;
        b.rect_xz(self.xz0, self.xz1, self.y, m)
    }
    fn describe_own_material(&self, b: &mut SceneBuilder) -> Result<Option<Id>, RtError> {
        Ok(Some(b.material(self.material.as_ref())?))
    }
}
impl DescribeHitable for YZRectangle {
    fn describe_hitable(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        let m = b.material(self.material.as_ref())?;
        b.rect_yz(self.yz0, self.yz1, self.x, m)
    }
    fn describe_own_material(&self, b: &mut SceneBuilder) -> Result<Option<Id>, RtError> {
        Ok(Some(b.material(self.material.as_ref())?))
    }
}
// objects/cube.rs:9-70.  A Cube goes over as Cube::new's arguments -- rt_object_cube -- NOT as the list of its six sides: its
// bounding_box is exactly (box_min, box_max) (cube.rs:67-69), while a list of six rectangles has the union of six boxes that are
// padded by 1e-4 along their normals (rectangle.rs:36,74,111) -- and the reference's BVHNode culls by that box (bvh.rs:88: a ray that has
// hit something at exactly the box's entry never visits the cube, aabb.rs:28-30), so the box is part of what is rendered.  The sides
// share one material (Cube::new clones `mat` into each, cube.rs:17-54); the struct does not keep it, the first side does.
impl DescribeHitable for Cube {
    fn describe_hitable(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        let m = match self.sides.first() {
            Some(side) => side.describe_own_material(b)?,
            None => None,
        };
        let m = m.ok_or_else(|| RtError { code: -1, message: "Cube without sides: no material to describe".to_string() })?;
        b.cube(&self.box_min, &self.box_max, m) // Cube::hit = the scan over the six sides in Cube::new's order (cube.rs:64-66): cube_hit on the device
    }
}
// impl Hitable for Vec<Arc<dyn Hitable>>, objects/hit.rs:56-93
impl DescribeHitable for Vec<Arc<dyn Hitable>> {
    fn describe_hitable(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        let mut ids = Vec::with_capacity(self.len());
        for h in self.iter() {
            ids.push(b.hitable(h.as_ref())?);
        }
        b.list(&ids)
    }
}
// objects/bvh.rs:29-33: the tree AS BUILT (random split axes and all) is kept; the stored bounding_box is recomputed
impl DescribeHitable for BVHNode {
    fn describe_hitable(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        let l = b.hitable(self.left.as_ref())?;
        let r = b.hitable(self.right.as_ref())?;
        b.bvh_node(l, r)
    }
}
// objects/mesh.rs:8-16: a triangle on the shared vertex arrays
impl DescribeHitable for Triangle {
    fn describe_hitable(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        let mesh = b.mesh_data(&self.positions, &self.normals)?;
        let m = b.material(self.material.as_ref())?;
        b.triangle(mesh, self.a, self.b, self.c, m)
    }
    fn describe_own_material(&self, b: &mut SceneBuilder) -> Result<Option<Id>, RtError> {
        Ok(Some(b.material(self.material.as_ref())?))
    }
}
// objects/mesh.rs:144-146 (needs `pub(crate) bvh`)
impl DescribeHitable for Mesh {
    fn describe_hitable(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        self.bvh.describe_hitable(b) // Mesh::hit is `self.bvh.hit(...)` (mesh.rs:201-203)
    }
}
// objects/transform.rs:9-14 (needs `pub(crate)` on obj, trans, inverse_trans): the stored matrices go over as they are
impl DescribeHitable for Transform {
    fn describe_hitable(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        let inner = b.hitable(self.obj.as_ref())?;
        let mut m = [0.0f64; 16];
        let mut inv = [0.0f64; 16];
        for r in 0..4 {
            for c in 0..4 {
                m[4 * r + c] = self.trans[(r, c)];
                inv[4 * r + c] = self.inverse_trans[(r, c)];
            }
        }
        b.transform_matrix(&m, Some(&inv), inner)
    }
}
// objects/medium.rs:9-13
impl DescribeHitable for ConstantMedium {
    fn describe_hitable(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        let boundary = b.hitable(self.boundary.as_ref())?;
        let phase = b.material(self.phase_function.as_ref())?;
        b.constant_medium(-1.0 / self.neg_inv_density, boundary, phase)
    }
}
// light.rs:67-72, 127-132 (need `pub(crate)` on sphere / area): lights are Hitables with a flux for the SPPM pre-pass
impl DescribeHitable for SphereDiffuseLight {
    fn describe_hitable(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        b.sphere_light(&self.sphere.center, self.sphere.radius, &self.flux, self.scale)
    }
}
impl DescribeHitable for XZRectLight {
    fn describe_hitable(&self, b: &mut SceneBuilder) -> Result<Id, RtError> {
        b.xz_rect_light(self.area.xz0, self.area.xz1, self.area.y, &self.flux, self.scale)
    }
}

// ------------------------------------------------------------------ capture_image ----
/// camera.rs:12-21: the stored frame of a constructed Camera
pub fn camera_frame(cam: &Camera) -> rt_camera_frame {
    rt_camera_frame {
        origin: v3(&cam.origin),
        lower_left_corner: v3(&cam.lower_left_corner),
        horizontal: v3(&cam.horizontal),
        vertical: v3(&cam.vertical),
        u: v3(&cam.u),
        v: v3(&cam.v),
        w: v3(&cam.w),
        lens_radius: cam.lens_radius,
    }
}

/// The compile-time constants of main.rs:34-45, camera.rs:73 and photon_mapper.rs:334-335 as values.
pub struct RenderConfig {
    pub width: usize,
    pub height: usize,
    pub sample_per_pixel: usize,
    pub max_depth: i32,
    pub t_min: f64,
    pub seed: u64,
    /// 0 = sample_ray with the Diffuse continuation (photon_mapper.rs:346-347); 1 = light/cosine mixture pdf
    pub integrator: i32,
    /// GPUs of this node the frame is spread over (image tiles dealt round-robin, RCCL gather); 0 = every visible GPU; default 1 (as the
    /// C++ host's Config) until a run over two or more distinct devices is on record.  The image does not depend on it.
    pub gpus: usize,
}
impl RenderConfig {
    pub fn new(width: usize, height: usize) -> Self {
        Self { width, height, sample_per_pixel: 256, max_depth: 50, t_min: 0.001, seed: 1, integrator: 0, gpus: 1 }
    }
    fn params(&self) -> rt_params {
        let mut p = rt_params::default();
        unsafe { rt_default_params(&mut p) };
        p.width = self.width as i32;
        p.height = self.height as i32;
        p.spp = self.sample_per_pixel as i32;
        p.max_depth = self.max_depth;
        p.t_min = self.t_min;
        p.seed = self.seed;
        p.integrator = self.integrator;
        p
    }
}

/// Lowers a `World` (world.rs:8-12): its root BVHNode exactly as `World::new` built it, and its lights
/// (needs `pub(crate) lights` on AllLights, light.rs:197).
pub fn describe_world(world: &World) -> Result<Scene, RtError> {
    let mut b = SceneBuilder::new()?;
    let root = world.bvh.describe_hitable(&mut b)?;
    b.set_root(root)?;
    let mut lights = Vec::new();
    for l in world.lights.lights.iter() {
        let h: &dyn Hitable = l.as_ref(); // trait Light: Hitable (light.rs:61)
        lights.push(b.hitable(h)?); // a light that is also in the hitable list keeps its id (emitted once)
    }
    if !lights.is_empty() {
        b.set_lights(&lights)?;
    }
    b.commit()
}

/// `world.cam.capture_image(integrator)` (camera.rs:66-128, main.rs:54): radiance through the HIP path, then
/// From<Vec3> for Rgb<u8> (vec3.rs:223-231) inside the library.
pub fn capture_image(world: &World, cfg: &RenderConfig) -> Result<image::RgbImage, RtError> {
    let scene = describe_world(world)?;
    let (radiance, _stats) = scene.render_multi(&camera_frame(&world.cam), &cfg.params(), cfg.gpus)?; // one call, all GPUs of the node
    let mut px = vec![0u8; radiance.len()];
    check(unsafe { rt_tonemap_u8(radiance.as_ptr(), radiance.len(), px.as_mut_ptr()) })?;
    Ok(image::RgbImage::from_raw(cfg.width as u32, cfg.height as u32, px).expect("buffer size"))
}

/// main.rs:52-54 as the reference really runs it: SPPMIntegrator::new(world) (photon_mapper.rs:139-233) followed by
/// capture_image with the SPPM sample_ray.  rt_render_sppm takes Camera::new's arguments, so the caller passes them.
pub fn capture_image_sppm(world: &World, cam_args: &rt_camera, cfg: &RenderConfig) -> Result<image::RgbImage, RtError> {
    let scene = describe_world(world)?;
    let mut sc = rt_sppm_config::default();
    unsafe { rt_default_sppm_config(&mut sc) };
    let (radiance, _stats) = scene.render_sppm_multi(cam_args, &cfg.params(), &sc, cfg.gpus)?;
    let mut px = vec![0u8; radiance.len()];
    check(unsafe { rt_tonemap_u8(radiance.as_ptr(), radiance.len(), px.as_mut_ptr()) })?;
    Ok(image::RgbImage::from_raw(cfg.width as u32, cfg.height as u32, px).expect("buffer size"))
}
