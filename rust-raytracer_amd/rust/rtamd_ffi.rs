//! rtamd_ffi.rs -- the reference-side binding a maintainer of BlackCloud37/rust-raytracer adds to
//! route `Camera::capture_image` through librtamd.so (C ABI: include/rtamd.h).
//!
//! STATUS: authored, NOT compiled in this repository's build image (no rustc/cargo there, see
//! SURVEY.md s8c).  It is the Rust spelling of what rust-raytracer_amd/host_cpp/rtamd.hpp (C++,
//! compiled and tested) and rust-raytracer_amd/rtamd/__init__.py (ctypes, tested) do.
//!
//! Drop into raytracer/src/rtamd_ffi.rs, add `mod rtamd_ffi;` to main.rs, link with
//! `cargo:rustc-link-lib=dylib=rtamd` (build.rs), and replace main.rs:52-54 by
//!     let result = rtamd_ffi::capture_image(&scene_desc, &world.cam_desc, &CONFIGS)?;
#![allow(non_camel_case_types, dead_code)]
use std::ffi::{c_void, CStr, CString};
use std::os::raw::{c_char, c_double, c_int};

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

#[link(name = "rtamd")]
extern "C" {
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_default_params(p: *mut rt_params);
    pub fn rt_scene_create(out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(s: *mut rt_scene);
    pub fn rt_texture_constant(s: *mut rt_scene, color: *const c_double) -> c_int;
    pub fn rt_texture_checker(s: *mut rt_scene, t0: c_int, t1: c_int) -> c_int;
    pub fn rt_texture_image(s: *mut rt_scene, w: c_int, h: c_int, rgb: *const u8) -> c_int;
    pub fn rt_material_lambertian(s: *mut rt_scene, tex: c_int) -> c_int;
    pub fn rt_material_metal(s: *mut rt_scene, tex: c_int, fuzz: c_double) -> c_int;
    pub fn rt_material_dielectric(s: *mut rt_scene, ir: c_double, tex: c_int) -> c_int;
    pub fn rt_material_diffuse_light(s: *mut rt_scene, tex: c_int) -> c_int;
    pub fn rt_object_sphere(s: *mut rt_scene, c: *const c_double, r: c_double, m: c_int) -> c_int;
    pub fn rt_object_rect_xy(s: *mut rt_scene, x0: c_double, y0: c_double, x1: c_double, y1: c_double, z: c_double, m: c_int) -> c_int;
    pub fn rt_object_rect_xz(s: *mut rt_scene, x0: c_double, z0: c_double, x1: c_double, z1: c_double, y: c_double, m: c_int) -> c_int;
    pub fn rt_object_rect_yz(s: *mut rt_scene, y0: c_double, z0: c_double, y1: c_double, z1: c_double, x: c_double, m: c_int) -> c_int;
    pub fn rt_object_cube(s: *mut rt_scene, mn: *const c_double, mx: *const c_double, m: c_int) -> c_int;
    pub fn rt_object_xz_rect_light(s: *mut rt_scene, x0: c_double, z0: c_double, x1: c_double, z1: c_double, y: c_double, flux: *const c_double, scale: c_double) -> c_int;
    pub fn rt_object_sphere_light(s: *mut rt_scene, c: *const c_double, r: c_double, flux: *const c_double, scale: c_double) -> c_int;
    pub fn rt_object_mesh(s: *mut rt_scene, n_vert: c_int, pos: *const c_double, nrm: *const c_double, n_tri: c_int, idx: *const u32,
                          m: c_int, synth_normals: c_int, bvh_seed: u64) -> c_int;
    pub fn rt_object_mesh_obj(s: *mut rt_scene, path: *const c_char, m: c_int, synth_normals: c_int, bvh_seed: u64) -> c_int;
    pub fn rt_object_transform(s: *mut rt_scene, rot: *const c_double, scale: *const c_double, translate: *const c_double, obj: c_int) -> c_int;
    pub fn rt_object_list(s: *mut rt_scene, n: c_int, objs: *const c_int) -> c_int;
    pub fn rt_object_bvh_node(s: *mut rt_scene, left: c_int, right: c_int) -> c_int;
    pub fn rt_object_bvh_build(s: *mut rt_scene, n: c_int, objs: *const c_int, bvh_seed: u64) -> c_int;
    pub fn rt_world_new(s: *mut rt_scene, n: c_int, objs: *const c_int, bvh_seed: u64) -> c_int;
    pub fn rt_scene_set_lights(s: *mut rt_scene, n: c_int, objs: *const c_int) -> c_int;
    pub fn rt_scene_load_file(path: *const c_char, out: *mut *mut rt_scene, cam: *mut rt_camera) -> c_int;
    pub fn rt_scene_commit(s: *mut rt_scene) -> c_int;
    pub fn rt_render(s: *const rt_scene, cam: *const rt_camera, p: *const rt_params, out_rgb: *mut c_double, stats: *mut rt_stats) -> c_int;
    pub fn rt_render_tiles_device(s: *const rt_scene, cam: *const rt_camera, p: *const rt_params, d_tiles: *mut c_double,
                                  hip_stream: *mut c_void, stats: *mut rt_stats) -> c_int;
    pub fn rt_default_sppm_config(c: *mut rt_sppm_config);
    pub fn rt_render_sppm(s: *const rt_scene, cam: *const rt_camera, p: *const rt_params, cfg: *const rt_sppm_config, out_rgb: *mut c_double,
                          stats_out: *mut c_double, photons_stored: *mut u64, stats: *mut rt_stats) -> c_int;
    pub fn rt_render_sppm_tiles_device(s: *const rt_scene, cam: *const rt_camera, p: *const rt_params, cfg: *const rt_sppm_config,
                                       d_tiles: *mut c_double, hip_stream: *mut c_void, stats: *mut rt_stats) -> c_int;
    pub fn rt_tonemap_u8(rgb: *const c_double, n: usize, out: *mut u8) -> c_int;
    pub fn rt_write_png(path: *const c_char, w: c_int, h: c_int, rgb: *const u8) -> c_int;
}

#[derive(Debug)]
pub struct RtError(pub i32, pub String);

fn check(rc: c_int) -> Result<c_int, RtError> {
    if rc < 0 {
        let msg = unsafe { CStr::from_ptr(rt_last_error()) }.to_string_lossy().into_owned();
        Err(RtError(rc, msg))
    } else {
        Ok(rc)
    }
}

/// What each `Hitable` / `Material` / `Texture` impl of the reference adds: a method that describes
/// itself to the flattener.  (`hit`, `scatter`, `get_color` stay for the CPU path.)
pub trait Describe {
    fn describe(&self, s: &mut SceneBuilder) -> Result<c_int, RtError>;
}

pub struct SceneBuilder {
    pub raw: *mut rt_scene,
    pub bvh_seed: u64,
}

impl SceneBuilder {
    pub fn new(bvh_seed: u64) -> Result<Self, RtError> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { rt_scene_create(&mut raw) })?;
        Ok(Self { raw, bvh_seed })
    }
    pub fn constant_texture(&mut self, c: [f64; 3]) -> Result<c_int, RtError> {
        check(unsafe { rt_texture_constant(self.raw, c.as_ptr()) })
    }
    pub fn lambertian(&mut self, tex: c_int) -> Result<c_int, RtError> {
        check(unsafe { rt_material_lambertian(self.raw, tex) })
    }
    pub fn sphere(&mut self, center: [f64; 3], radius: f64, mat: c_int) -> Result<c_int, RtError> {
        check(unsafe { rt_object_sphere(self.raw, center.as_ptr(), radius, mat) })
    }
    // ... one thin wrapper per rt_* builder, omitted for brevity: identical pattern ...

    /// World::new(hitable_list, cam, lights) (world.rs:15-25)
    pub fn world_new(&mut self, objects: &[c_int]) -> Result<(), RtError> {
        check(unsafe { rt_world_new(self.raw, objects.len() as c_int, objects.as_ptr(), self.bvh_seed) })?;
        check(unsafe { rt_scene_commit(self.raw) })?;
        Ok(())
    }
}

impl Drop for SceneBuilder {
    fn drop(&mut self) {
        unsafe { rt_scene_destroy(self.raw) }
    }
}

// Example `Describe` impl for objects/sphere.rs:
//
// impl Describe for Sphere {
//     fn describe(&self, s: &mut SceneBuilder) -> Result<c_int, RtError> {
//         let m = self.material.describe(s)?;          // Material: Describe
//         s.sphere([self.center.x, self.center.y, self.center.z], self.radius, m)
//     }
// }

/// Camera::capture_image (camera.rs:66-128) through the GPU path: linear radiance -> Rgb<u8>.
pub fn capture_image(scene: &SceneBuilder, cam: &rt_camera, width: usize, height: usize, sample_per_pixel: usize, seed: u64)
                     -> Result<Vec<u8>, RtError> {
    let mut p = rt_params::default();
    unsafe { rt_default_params(&mut p) };
    p.width = width as i32;
    p.height = height as i32;
    p.spp = sample_per_pixel as i32;
    p.seed = seed;
    let mut rad = vec![0f64; width * height * 3];
    let mut st = rt_stats::default();
    check(unsafe { rt_render(scene.raw, cam, &p, rad.as_mut_ptr(), &mut st) })?;
    let mut rgb = vec![0u8; rad.len()];
    check(unsafe { rt_tonemap_u8(rad.as_ptr(), rad.len(), rgb.as_mut_ptr()) })?;
    Ok(rgb) // image::RgbImage::from_raw(width as u32, height as u32, rgb)
}

pub fn load_scene_file(path: &str) -> Result<(*mut rt_scene, rt_camera), RtError> {
    let c = CString::new(path).map_err(|_| RtError(-1, "path contains NUL".into()))?;
    let mut raw = std::ptr::null_mut();
    let mut cam = rt_camera::default();
    check(unsafe { rt_scene_load_file(c.as_ptr(), &mut raw, &mut cam) })?;
    Ok((raw, cam))
}
