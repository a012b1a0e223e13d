/*
 * rtamd.h -- C ABI of the MI355X-native radiance path (librtamd.so).
 *
 * Drop-in boundary for ONE hot path of BlackCloud37/rust-raytracer: the
 * per-pixel radiance loop  Camera::capture_image -> Integrator::sample_ray ->
 * World::hit -> {BVHNode,AABB,Sphere,Rect,Triangle,Transform}::hit ->
 * Material::{emitted,scatter}.  The reference has no FFI of its own (SURVEY.md
 * s8b), so every entry point below names the Rust item it stands in for
 * (paths relative to /root/reference/raytracer/src).  The cut is at
 * capture_image granularity: the host walks its Hitable/Material/Texture graph
 * once through the rt_texture_* / rt_material_* / rt_object_* builders (one per
 * reference constructor), commits, and calls rt_render.
 *
 * Conventions
 *   - plain C, pointers + sizes only; no C++/torch types cross this boundary.
 *   - every function returns RT_OK (0) / a non-negative id, or a negative
 *     rt_status; rt_last_error() gives the message (thread-local).  Nothing
 *     aborts or unwinds across the boundary (the reference's panic sites map
 *     to error codes, listed per function).
 *   - the library owns what it copies at build/commit; the caller owns every
 *     buffer it passes in.  An rt_scene is immutable after rt_scene_commit and
 *     may then be rendered from several host threads concurrently.
 *   - all geometry and radiance is f64, as in the reference (vec3.rs:15-19).
 *   - there is NO CPU fallback: without a HIP device the render entry points
 *     fail with RT_ERR_NO_DEVICE.
 */
#ifndef RTAMD_H
#define RTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTAMD_ABI_VERSION 2   /* 2 (round 5): rt_stats grew by five f64 fields behind reserved[] */

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_ARG = -1,          /* bad argument / unknown id */
    RT_ERR_UNIT_ZERO = -2,    /* Vec3::unit of a zero vector      (vec3.rs:88 panic)      */
    RT_ERR_NO_BBOX = -3,      /* "No bounding box in bvh_node constructor" (bvh.rs:43,57) */
    RT_ERR_SINGULAR = -4,     /* "Invalid transform matrix"       (transform.rs:146)      */
    RT_ERR_IO = -5,           /* file cannot be read ("Failed to load OBJ file", mesh.rs:158) */
    RT_ERR_SCHEMA = -6,       /* scene file does not follow the schema of data/ (.json/.yaml) */
    RT_ERR_NO_NORMALS = -7,   /* mesh without vertex normals      (mesh.rs:62 index panic) */
    RT_ERR_NOT_COMMITTED = -8,
    RT_ERR_NO_DEVICE = -9,    /* no HIP device / HIP runtime error: there is no CPU fallback */
    RT_ERR_UNSUPPORTED = -10,
    RT_ERR_HIP = -11,
    RT_ERR_INTERNAL = -12     /* a device-side invariant failed (an instance below an instance reached the serving waves; a request ring overran) */
} rt_status;

typedef struct rt_scene rt_scene; /* opaque: World's object graph + its flattened device form */

/* Camera::new arguments (camera.rs:24-32); data/<scene>.json "camera" block. */
typedef struct rt_camera {
    double look_from[3];
    double look_at[3];
    double vup[3];
    double vfov;       /* degrees */
    double aspect;     /* CONFIGS.aspect_ratio (main.rs:35) */
    double aperture;
    double focus_dist;
} rt_camera;

/* The Camera struct as the reference STORES it (camera.rs:12-21: the derived frame, not the constructor arguments): a host
 * that already owns a `Camera` hands these eight fields over as they are. */
typedef struct rt_camera_frame {
    double origin[3];
    double lower_left_corner[3];
    double horizontal[3];
    double vertical[3];
    double u[3], v[3], w[3];
    double lens_radius;
} rt_camera_frame;

/* capture_image's compile-time constants made run-time (SURVEY.md s5 "Config / flags"). */
typedef struct rt_params {
    int32_t width;       /* CONFIGS.width  (main.rs:34) default 800 */
    int32_t height;      /* CONFIGS.height (main.rs:45) default 800 */
    int32_t spp;         /* sample_per_pixel (camera.rs:73) default 256 */
    int32_t max_depth;   /* photon_mapper.rs:334 default 50 */
    double t_min;        /* photon_mapper.rs:335 default 0.001 */
    uint64_t seed;       /* rtamd-rng-3 stream seed (replaces thread_rng) */
    int32_t rank;        /* image-tile partition: this call renders tiles t with t % world == rank */
    int32_t world;       /* 1 = whole image */
    int32_t spp_chunk;   /* samples per pixel per kernel launch; 0 = all in one launch */
    int32_t kernel;      /* 0 = auto; 1 = reference-order stackless traversal; 2 = SAH-BVH2 accel traversal (auto whenever the
                            scene has a usable accel); 5 = kernel 2's BVH with the cooperative instance service: the walks
                            through mesh instances are queued per workgroup and served by full waves (auto when an
                            instance's object-space BVH has >= 64 nodes; needs 1..64 instances, at least one of f32-vertex triangles);
                            6 = the same service across the whole GPU and across launches (parked paths in HBM pools, a
                            dedicated walk launch per cycle; 1..64 instances; by request only -- it is slower than
                            kernel 5; workspace within rt_tuning.wf_workspace_mb, default 1.9 GB).
                            All give bit-identical images (same f64 primitive tests, same tie rule: when two objects share the
                            closest t EXACTLY the later-visited one wins unless the reference's own BVHNode box test would have
                            culled it -- bvh.rs:88, aabb.rs:28-30 -- in every kernel). */
    int32_t device;      /* HIP device ordinal; -1 = current */
    int32_t integrator;  /* 0 = sample_ray as the reference structures it (BSDF sampling only; default);
                            1 = light importance sampling: on Diffuse hits the direction is drawn from the
                                0.5*lights + 0.5*cosine mixture pdf (book-3 MixturePDF semantics; needs rt_scene_set_lights);
                            2 = the reference's SPPM sample_ray (only through rt_render_sppm) */
    double time0, time1; /* book-2 extension (the reference has no code for it: its Ray has no time, ray.rs:3-6): the camera's shutter.
                            With time1 > time0 every sample draws a time in [time0, time1] (gen_range's product may round up to time1) right after its lens sample and moving spheres
                            (rt_object_moving_sphere) are where they are at that time; default 0, 0 = no draw */
} rt_params;

typedef struct rt_stats {
    double seconds;          /* wall time of the call (host clock) */
    double kernel_ms;        /* sum of path-trace kernel launch durations (HIP events on the launch stream) */
    double reduce_ms;        /* 0: the ordered reduction happens inside the path-trace kernel */
    uint64_t samples;        /* pixel-samples traced by this call (this rank) */
    int32_t launches;        /* path-trace kernel launches */
    int32_t kernel_used;     /* which kernel ran */
    int32_t scene_in_lds;    /* 1 if the flattened scene was staged into LDS */
    int32_t block_threads;
    int32_t grid_blocks;
    int32_t spp_chunk;
    uint64_t scene_bytes;    /* flattened scene size */
    uint64_t reserved[4];    /* [0] SPPM pre-pass microseconds, [1] render workspace bytes on the device; rt_render_multi, entry 0: [2] microseconds from
                                the last rank's finish until every row had arrived on the root device (the exchange proper: no stitch, no host copy),
                                [3] rows that travelled through RCCL */
    /* ABI version 2: */
    double upload_ms;        /* copying the scene to this call's device (0 when a copy was already there: uploads happen once per scene and device) */
    double posted_ms;        /* rt_render_multi, entry i: when rank i's rows were handed to the exchange, ms after the call began (its own finish time;
                                0 for rows rendered in place on the root device) */
    double stitch_copy_ms;   /* rt_render_multi, entry 0: the stitch on the root device and the copy of the frame to out_rgb */
    double comm_init_ms;     /* rt_render_multi, entry 0: creating the RCCL communicators of this device list (0 when they were cached) */
    double exchange_ms;      /* rt_render_multi, entry 0: reserved[2] in ms */
} rt_stats;

/* ---- library ------------------------------------------------------------ */
int rt_abi_version(void);
const char* rt_last_error(void);
void rt_default_params(rt_params* p);           /* the reference's constants, see rt_params */
int rt_device_count(void);                      /* HIP devices visible; 0 without a GPU */

/* Measurement and test hooks.  The library reads NO environment variable: everything that changes how (never what) it
 * renders is set here, process-wide, before the calls it should affect.  0 / negative = automatic (the default). */
typedef struct rt_tuning {
    int32_t no_lds;                /* 1: keep scene tables in global memory even when they fit LDS (A/B runs)            */
    int32_t top_nodes;             /* >= 0: cap on the BVH nodes cached in LDS when the scene lives in L2/HBM; -1 auto  */
    int32_t sub_spp;               /* 1..8: sample indices per work unit (a wave's pool = 64 px x sub_spp); 0 auto      */
    int32_t coop_pool;             /* 1..1024: parked-path slots per workgroup in kernel 5 (small values force its in-lane fallback); 0 auto */
    int32_t max_leaf;              /* 1..4: accel builder, items per leaf; 0 auto (read at rt_scene_commit)             */
    int32_t sppm_photon_capacity;  /* > 0: initial photon-buffer capacity (forces the grow-and-retry path); 0 auto      */
    int32_t sppm_knn_candidates;   /* >= 0: k-nearest candidates kept in LDS (0 forces the out-of-LDS selection); -1 auto */
    int32_t multi_force_rccl;      /* 1: rt_render_multi sends EVERY rank's rows through the RCCL communicator, also those that already sit on
                                      the root device (a rank then sends to itself): exercises the exchange on a one-GPU box; 0 auto      */
    int32_t wf_workspace_mb;       /* > 0: kernel 6's workspace budget in MB (unit buffers + record pools; default 1900); 8600 = what round 3 took */
    int32_t reserved;
    double sah_box_cost;           /* > 0: accel builder, SAH cost of a box-pair test relative to 2.0 per primitive; 0 auto */
} rt_tuning;
void rt_tuning_default(rt_tuning* t);
int rt_tuning_set(const rt_tuning* t);
/* The render entry points keep a workspace per device for the life of the process (unit rings of the resident waves, ~0.3 GB,
 * accumulator, tickets: no hipMalloc on the hot path), rt_render_multi also its RCCL communicators and its frame-sized buffers (gathered
 * rows, stitched frame, the ranks' rows: up to 4 idle ones per device and size and at most 1 GiB of idle buffers in all -- beyond that
 * the least recently returned ones are freed).  This frees the idle ones;
 * returns the bytes released. */
int64_t rt_release_workspaces(void);

/* ---- scene graph builders (one per reference constructor) ---------------- */
int rt_scene_create(rt_scene** out);
void rt_scene_destroy(rt_scene* s);

/* material.rs:48-50  ConstantTexture(Vec3) / CheckerTexture(t0,t1) / ImageTexture(rgb8, row-major, top row first) */
int rt_texture_constant(rt_scene* s, const double color[3]);
int rt_texture_checker(rt_scene* s, int t0, int t1);
int rt_texture_image(rt_scene* s, int width, int height, const uint8_t* rgb);
/* Book-2 extension, no reference counterpart (BASELINE config C5 names it): noise_texture(scale), the marble texture
 * 0.5 (1 + sin(scale p.z + 10 turb(p))) over Perlin noise with 7 octaves of turbulence; the 256 gradient vectors and the three
 * permutations come from the RNG stream (seed, "perlin" key, 0); sin is the deterministic rtamd-sin-1 (csrc/common/detsin.h) */
int rt_texture_noise(rt_scene* s, double scale, uint64_t seed);
/* material.rs:88-212  Lambertian::new / Metal::new / Dielectric::new / DiffuseLight::new */
int rt_material_lambertian(rt_scene* s, int albedo_tex);
int rt_material_metal(rt_scene* s, int albedo_tex, double fuzz);
int rt_material_dielectric(rt_scene* s, double ir, int albedo_tex);
int rt_material_diffuse_light(rt_scene* s, int emit_tex);
/* material.rs:213-231  Isotropic::new(albedo) -- commented out in the reference; the phase function of a ConstantMedium:
 * scatter = (albedo, Ray(p, random_in_unit_sphere())); treated as a pass-through (Specular) interaction by every integrator */
int rt_material_isotropic(rt_scene* s, int albedo_tex);

/* objects/sphere.rs:9-13 */
int rt_object_sphere(rt_scene* s, const double center[3], double radius, int material);
/* Book-2 extension, no reference counterpart: moving_sphere(center0, center1, time0, time1, radius, material) -- Sphere::hit around
 * the centre center0 + (center1 - center0) (ray.time - time0) / (time1 - time0); box = the union of the boxes at both times.
 * Rendered by kernels 1 and 2 with integrator 0; needs rt_params.time1 > rt_params.time0 to move.  The shutter [rt_params.time0,
 * rt_params.time1] must lie inside [time0, time1] of every moving sphere of the scene (their boxes are the union of the boxes at those two
 * times): a render with a shutter outside that range is refused with RT_ERR_ARG */
int rt_object_moving_sphere(rt_scene* s, const double center0[3], const double center1[3], double time0, double time1, double radius, int material);
/* objects/rectangle.rs:7-12,44-49,82-87 : {a0,b0}=xy0|xz0|yz0, {a1,b1}=xy1|xz1|yz1, k = z|y|x */
int rt_object_rect_xy(rt_scene* s, double x0, double y0, double x1, double y1, double z, int material);
int rt_object_rect_xz(rt_scene* s, double x0, double z0, double x1, double z1, double y, int material);
int rt_object_rect_yz(rt_scene* s, double y0, double z0, double y1, double z1, double x, int material);
/* objects/cube.rs:16 Cube::new(box_min, box_max, mat) */
int rt_object_cube(rt_scene* s, const double box_min[3], const double box_max[3], int material);
/* SphereDiffuseLight::new(center, radius, flux, scale) light.rs:74-86 / XZRectLight::new(xz0, xz1, y, flux, scale)
 * light.rs:134-146 : as a Hitable the light is its primitive + DiffuseLight(ConstantTexture(flux)); `scale` only feeds
 * the photon power (flux * scale) of the SPPM pre-pass */
int rt_object_sphere_light(rt_scene* s, const double center[3], double radius, const double flux[3], double scale);
int rt_object_xz_rect_light(rt_scene* s, double x0, double z0, double x1, double z1, double y, const double flux[3], double scale);
/* objects/medium.rs:16 ConstantMedium::new(d, boundary, phase_function): constant-density participating medium inside
 * `boundary` (any Hitable with a box, not itself a medium).  Its hit() draws a random number (medium.rs:37-38), so scenes with a
 * medium are rendered by the reference-order kernel only (kernel 2 and the SPPM pre-pass refuse them); the logarithm is the
 * deterministic rtamd-ln-1 (csrc/common/detlog.h, < 1 ulp from libm). */
int rt_object_constant_medium(rt_scene* s, double density, int boundary, int phase_material);
/* objects/mesh.rs:149 Mesh::load_obj given parsed arrays: positions/normals n_vert*3, indices n_tri*3.
 * normals == NULL -> RT_ERR_NO_NORMALS unless synthesize_normals != 0 (area-weighted smooth normals). */
int rt_object_mesh(rt_scene* s, int n_vert, const double* positions, const double* normals, int n_tri, const uint32_t* indices,
                   int material, int synthesize_normals, uint64_t bvh_seed);
/* Mesh::load_obj(path, material): tobj{single_index, triangulate} semantics, models[0] */
int rt_object_mesh_obj(rt_scene* s, const char* obj_path, int material, int synthesize_normals, uint64_t bvh_seed);
/* objects/transform.rs:17 Transform::new(rotate_in_degree, scale, translate, obj): M = T*S*Rx*Ry*Rz */
int rt_object_transform(rt_scene* s, const double rotate_deg[3], const double scale[3], const double translate[3], int object);
/* Transform as the reference stores it (transform.rs:9-14: obj, trans, inverse_trans; row-major 4x4).  inverse_trans may be
 * NULL (computed as try_inverse does; RT_ERR_SINGULAR if there is none). */
int rt_object_transform_matrix(rt_scene* s, const double trans[16], const double* inverse_trans, int object);
/* Triangle::new on shared vertex arrays (mesh.rs:8-53: a, b, c index Arc<Vec<Vec3>> positions / normals): rt_mesh_data registers
 * the arrays once (returns a mesh id, not an object id), rt_object_triangle one triangle; the host keeps its own BVHNode tree
 * over them (rt_object_bvh_node), e.g. Mesh.bvh as BVHNode::new really built it. */
int rt_mesh_data(rt_scene* s, int n_vert, const double* positions, const double* normals);
int rt_object_triangle(rt_scene* s, int mesh, uint32_t a, uint32_t b, uint32_t c, int material);
/* impl Hitable for Vec<Arc<dyn Hitable>> (objects/hit.rs:56-93) */
int rt_object_list(rt_scene* s, int n, const int* objects);
/* BVHNode::construct(left,right) (bvh.rs:47-58) and BVHNode::new(src_objects) (bvh.rs:60-83; split axes from
 * the seeded rtamd-rng-3 "bvh" stream instead of thread_rng) */
int rt_object_bvh_node(rt_scene* s, int left, int right);
int rt_object_bvh_build(rt_scene* s, int n, const int* objects, uint64_t bvh_seed);
/* Hitable::bounding_box (objects/hit.rs:53): out = min[3], max[3]; RT_ERR_NO_BBOX for None */
int rt_object_bounding_box(const rt_scene* s, int object, double out_min_max[6]);

/* Introspection of the host-side object graph (the walk a `Describe` visitor of the reference's trait objects does the
 * other way round): what an object id stands for, its parameters and its children.
 *   sphere:    v = {center[3], radius}                      (objects/sphere.rs:9-13)
 *   moving sphere: v = {center0[3], radius, center1[3]} (its times are not reported)
 *   rect:      v = {a0, b0, a1, b1, k}, axis = constant axis (0: YZRectangle x=k, 1: XZRectangle y=k, 2: XYRectangle z=k)
 *   triangle:  v = {ia, ib, ic} vertex indices of its mesh  (objects/mesh.rs:8-14)
 *   medium:    v = {density}, material = phase function, children = {boundary}   (objects/medium.rs:9-13)
 *   cube / list / mesh / transform / bvh: children only (cube: its 6 sides; mesh: its inner BVHNode; bvh: {left, right}) */
typedef enum rt_object_type {
    RT_OBJ_SPHERE = 0, RT_OBJ_RECT = 1, RT_OBJ_CUBE = 2, RT_OBJ_TRIANGLE = 3, RT_OBJ_MESH = 4, RT_OBJ_TRANSFORM = 5,
    RT_OBJ_LIST = 6, RT_OBJ_BVH = 7, RT_OBJ_MEDIUM = 8, RT_OBJ_MOVING_SPHERE = 9
} rt_object_type;
typedef struct rt_object_desc {
    int32_t type;        /* rt_object_type */
    int32_t material;    /* material id, -1 for containers */
    int32_t n_children;
    int32_t axis;        /* rect only */
    double v[8];
} rt_object_desc;
int rt_scene_root(const rt_scene* s);            /* object id of the root (World.bvh / the file's top-level list), RT_ERR_ARG if unset */
int rt_object_describe(const rt_scene* s, int object, rt_object_desc* out);
int rt_object_children(const rt_scene* s, int object, int capacity, int* out);  /* writes min(capacity, n) ids, returns n */

/* World::new(hitable_list, cam, lights) (world.rs:15-25): root = BVHNode::new(list) */
int rt_world_new(rt_scene* s, int n, const int* objects, uint64_t bvh_seed);
/* World::new's `lights: Vec<Arc<dyn Light>>` (world.rs:18; scene.rs:110 passes the XZRectLight): the objects that the
 * mixture-pdf integrator samples.  Each must be a sphere or an XZ rectangle in world space (the reference's two Light
 * impls, light.rs:67-86,127-146); RT_ERR_UNSUPPORTED for lights under a Transform. */
int rt_scene_set_lights(rt_scene* s, int n, const int* objects);
/* root = an existing object (e.g. the HitableList of a scene file) */
int rt_scene_set_root(rt_scene* s, int object);
/* scene.rs:16-112 cornell_box_scene(): the reference's only built-in scene, numbers verbatim.
 * cube_obj_path = "data/mesh/cube.obj" of the reference. */
int rt_scene_cornell_box(rt_scene* s, const char* cube_obj_path, double aspect_ratio, uint64_t bvh_seed, rt_camera* cam_out);
/* data/<name>.json|.yaml loader (schema SURVEY.md sA.1; README.md:86-89 Track 5). File BVH topology is kept
 * verbatim, the redundant "bounding_box" is recomputed as BVHNode::construct does. Creates AND commits. */
int rt_scene_load_file(const char* path, rt_scene** out, rt_camera* cam_out);
/* flatten the graph into the linear device form (DFS pre-order program, SoA tables) */
int rt_scene_commit(rt_scene* s);

/* introspection of the flattened form (tests, INTEGRATION) */
typedef struct rt_scene_info {
    int32_t n_nodes, n_boxes, n_spheres, n_rects, n_tris, n_xforms, n_materials, n_textures;
    int32_t n_verts, max_depth, committed, n_cubes;   /* a Cube is one record and one node (six sides scanned by the kernel's cube_hit) */
    uint64_t bytes;
    int32_t accel_ok, accel_nodes, accel_items, accel_instances, accel_stack, accel_compact;  /* accel_compact: 1 if the compact object-space copies kernel 5 needs were built */
} rt_scene_info;
int rt_scene_info_get(const rt_scene* s, rt_scene_info* out);
/* What a saved accumulator state (rt_render_accumulate) belongs to, beside its rt_params: a 64-bit fingerprint of the committed scene (FNV-1a
 * over the flattened blob: geometry, materials, textures, BVH wiring; 0 for a scene that is not committed) and the version of everything
 * that decides the bits of an image -- RNG, deterministic ln / sin, sampling order (a string such as "rtamd-image-5 rng-3 ln-1 sin-1").
 * A checkpoint written by another build or for another scene must not be resumed: the host compares both (host_cpp/rtamd.hpp does). */
uint64_t rt_scene_fingerprint(const rt_scene* s);
const char* rt_spec_version(void);

/* ---- the hot path -------------------------------------------------------- */
/* Camera::capture_image (camera.rs:66-128) minus the u8 conversion: linear radiance (sum/spp), f64 RGB,
 * row-major, y down, into caller-owned HOST memory out_rgb[height*width*3].  world > 1 renders only this
 * rank's tiles (others left 0). */
int rt_render(const rt_scene* s, const rt_camera* cam, const rt_params* p, double* out_rgb, rt_stats* stats);
/* the same for a host that owns a constructed Camera (its stored frame); rt_camera_frame_from = Camera::new (camera.rs:24-55) */
int rt_render_camera_frame(const rt_scene* s, const rt_camera_frame* frame, const rt_params* p, double* out_rgb, rt_stats* stats);
int rt_camera_frame_from(const rt_camera* cam, rt_camera_frame* out);

/* The reference's main.rs:52-54 as it really is: SPPMIntegrator::new(world) (photon_mapper.rs:139-233: `iterations` x
 * {photons_per_iter photon paths -> global + caustic photon maps; one eye ray per pixel; progressive radius update}) followed
 * by capture_image with SPPMIntegrator::sample_ray (photon_mapper.rs:327-365: the first Diffuse hit adds the pixel's estimates
 * and ends the path).  Needs rt_scene_set_lights with lights made by rt_object_*_light.  One GPU, whole frame:
 * p->world != 1 is RT_ERR_ARG.
 * stats_out (optional, HOST, height*width*10 f64): per pixel {global: flux[3], radius2, photons; caustic: flux[3], radius2, photons}.
 * spp == 0 runs the pre-pass only (out_rgb may be NULL).  rt_stats.reserved[0] = pre-pass time in microseconds. */
typedef struct rt_sppm_config {
    int32_t iterations;        /* max_iter_cnt      photon_mapper.rs:148  default 50     */
    int32_t photons_per_iter;  /* photon_per_iter   photon_mapper.rs:149  default 500000 */
    int32_t k_global;          /* GLOBAL_INIT_PHOTONS  :18  default 100 */
    int32_t k_caustic;         /* CAUSTIC_INIT_PHOTONS :19  default 50  */
    int32_t max_bounces;       /* cap on photon / eye path length (the reference loops until absorbed); default 4096 */
    int32_t reserved;
    double alpha;              /* ALPHA :17  default 0.7 */
} rt_sppm_config;
void rt_default_sppm_config(rt_sppm_config* c);
int rt_render_sppm(const rt_scene* s, const rt_camera* cam, const rt_params* p, const rt_sppm_config* cfg, double* out_rgb,
                   double* stats_out, uint64_t photons_stored[2], rt_stats* stats);

/* Same, device-resident: renders this rank's 8x8 tiles into d_tiles (DEVICE memory,
 * rt_tiles_owned(p)*64*3 f64, tile-major) on `hip_stream` (hipStream_t as void*, NULL = default stream).
 * The call returns after the work has completed on that stream. */
int rt_render_tiles_device(const rt_scene* s, const rt_camera* cam, const rt_params* p, double* d_tiles, void* hip_stream,
                           rt_stats* stats);
/* Resumable / progressive rendering (checkpoint and restart of a long frame).  capture_image adds a pixel's samples in index order
 * and divides once (camera.rs:96-102); here the running sums are the CALLER's: rt_render_accumulate_device traces the sample indices
 * [sample_begin, sample_end) of every pixel of this rank's tiles (0 <= begin < end <= p->spp) and adds them, in index order, to d_accum
 * (DEVICE memory, rt_tiles_owned(p)*64*3 f64; sample_begin == 0 initialises it, no zeroing needed); rt_accum_finalize_device writes
 * d_tiles = d_accum / p->spp (0 outside the image) as rt_render_tiles_device would have.  Calls with consecutive ranges, in order, give
 * that function's result bit for bit however the range is cut (the additions are the same sequence), so the state of an interrupted
 * frame is d_accum and the next sample index -- the RNG is keyed by (seed, pixel, sample) and has no state to save.  p must be the
 * same in every call (spp = the frame's total).  Kernels 1 / 2 / 5, integrators 0 / 1. */
int rt_render_accumulate_device(const rt_scene* s, const rt_camera* cam, const rt_params* p, int32_t sample_begin, int32_t sample_end,
                                double* d_accum, void* hip_stream, rt_stats* stats);
int rt_accum_finalize_device(const rt_params* p, const double* d_accum, double* d_tiles, void* hip_stream);
/* The same for a host that holds no device memory (the reference's Rust host): the running sums travel in `accum_state`, HOST memory,
 * rt_accum_state_doubles(p) f64, opaque to the caller (it is this rank's tile-major accumulator) -- write it to disk to checkpoint, read it
 * back to resume.  rt_accum_finalize turns a complete state (all p->spp samples) into the frame out_rgb[H][W][3] that rt_render would
 * have returned (world == 1), bit for bit.  Each call copies the state to the device and back (24 B per pixel). */
int64_t rt_accum_state_doubles(const rt_params* p);
int rt_render_accumulate(const rt_scene* s, const rt_camera* cam, const rt_params* p, int32_t sample_begin, int32_t sample_end,
                         double* accum_state, rt_stats* stats);
int rt_accum_finalize(const rt_params* p, const double* accum_state, double* out_rgb);
/* SPPM across GPUs: every rank runs the same deterministic pre-pass (photon maps + per-pixel statistics of the WHOLE
 * frame: ~0.13 s for the reference's 50 x 500 000 photons) and renders only its own tiles; the buffers are gathered and
 * stitched exactly like rt_render_tiles_device's.  (The per-pixel pre-pass statistics are only returned by rt_render_sppm.) */
int rt_render_sppm_tiles_device(const rt_scene* s, const rt_camera* cam, const rt_params* p, const rt_sppm_config* cfg, double* d_tiles,
                                void* hip_stream, rt_stats* stats);
int64_t rt_tiles_total(const rt_params* p);   /* ceil(W/8)*ceil(H/8) */
int64_t rt_tiles_owned(const rt_params* p);   /* tiles t in [0,total) with t % world == rank */
/* the stitch of camera.rs:115-123: scatter gathered tile-major buffers (rank-major: rank 0's tiles, rank 1's, ...
 * each padded to rt_tiles_owned of rank 0) into a row-major frame; both pointers DEVICE memory. */
int rt_assemble_frame_device(const rt_params* p, const double* d_gathered, int64_t tiles_per_rank_stride, double* d_frame,
                             void* hip_stream);

/* ---- the frame across the GPUs of one node ------------------------------------ */
/* Camera::capture_image as the reference structures it (camera.rs:74-126: the worker pool, the jobs, the channel and the stitch are
 * all INSIDE the call; main.rs:52-54 makes one call): one process, one host thread per logical rank.  The committed scene is
 * replicated on every device; rank i renders the 8x8 tiles t with t % n_devices == i on device_ids[i]; the ranks' tile-major rows
 * are gathered on device_ids[0] -- grouped ncclSend / ncclRecv over a communicator from ncclCommInitAll (RCCL, xGMI), cached per
 * device list for the life of the process; rows of ranks that share the root's device are rendered in place -- stitched there
 * (camera.rs:115-123) and copied once to out_rgb (HOST, height*width*3 f64).  The image is bit-identical to rt_render's for every
 * device list (the RNG is keyed by pixel and sample).
 *   device_ids: HIP ordinals, one per rank; an ordinal may repeat (its ranks share that device).  NULL = 0 .. n_devices-1.
 *   n_devices:  number of ranks; 0 = one per visible device.
 *   p->rank / p->world must be 0 / 1 (the call partitions the frame itself); p->device is ignored.
 *   stats:      NULL or n_devices entries, entry i = rank i's render (kernel_ms, samples, upload_ms, posted_ms ...).  stats[0].seconds = wall
 *               time of the whole call, stats[0].exchange_ms (= reserved[2] in microseconds) = from the last rank's finish until every row
 *               had arrived on the root device, stats[0].stitch_copy_ms = stitch + copy to the host, stats[0].comm_init_ms = creating the
 *               communicators (first call with a device list), stats[0].reserved[3] = number of rows that travelled through RCCL.
 *               A rank's rows are handed to RCCL (ncclSend on its device, the matching ncclRecv on the root's) by the rank's own host thread
 *               the moment its render has finished -- not after all ranks have joined -- so the rows of early ranks travel while the others
 *               still render.
 * Errors of any rank come back as that rank's rt_status (first failing rank wins); nothing aborts. */
int rt_render_multi(const rt_scene* s, const rt_camera* cam, const rt_params* p, int n_devices, const int* device_ids, double* out_rgb,
                    rt_stats* stats);
/* the same for a host that owns a constructed Camera (its stored frame, camera.rs:12-21), as rt_render_camera_frame */
int rt_render_multi_camera_frame(const rt_scene* s, const rt_camera_frame* frame, const rt_params* p, int n_devices, const int* device_ids,
                                 double* out_rgb, rt_stats* stats);
/* main.rs:52-54 as the reference really runs it, across GPUs: every rank repeats the deterministic SPPM pre-pass and renders its own
 * tiles (rt_render_sppm_tiles_device), gathered and stitched as above. */
int rt_render_sppm_multi(const rt_scene* s, const rt_camera* cam, const rt_params* p, const rt_sppm_config* cfg, int n_devices,
                         const int* device_ids, double* out_rgb, rt_stats* stats);
/* ncclGetVersion of the RCCL the library is linked with (e.g. 22606), 0 if it cannot be asked */
int rt_rccl_version(void);

/* From<Vec3> for Rgb<u8> (vec3.rs:223-231): floor(clamp(sqrt(c),0,1)*255), NaN -> 0.  Host buffers. */
int rt_tonemap_u8(const double* rgb, size_t n_channels, uint8_t* out);
/* RgbImage::save("output/test.png") (main.rs:55): 8-bit RGB PNG */
int rt_write_png(const char* path, int width, int height, const uint8_t* rgb);

/* ---- diagnostics used by the parity tests ------------------------------- */
/* rtamd-rng-3 (csrc/common/rng.h): first n u64 draws of stream (seed, pixel, sample) computed ON THE DEVICE */
int rt_debug_rng_device(uint64_t seed, uint64_t pixel, uint64_t sample, int n, uint64_t* out_host);
/* host-side restatement of the same stream (used by BVHNode::new's axis draws) */
int rt_debug_rng_host(uint64_t seed, uint64_t pixel, uint64_t sample, int n, uint64_t* out_host);
/* the float conversions of the spec (rand 0.8.4's): out_gen[i] = the i-th gen::<f64>() of a fresh stream (seed, pixel, sample),
 * out_range[i] = the i-th gen_range(lo..hi) of another fresh stream of the same key; on_device != 0 computes them in a kernel */
int rt_debug_rng_floats(uint64_t seed, uint64_t pixel, uint64_t sample, int n, double lo, double hi, int on_device, double* out_gen,
                        double* out_range);
/* device f64 sqrt / divide / rtamd-ln-1 / rtamd-sin-1, element-wise: op 0 = sqrt(a), 1 = a/b, 2 = det_ln(a), 3 = det_sin(a) */
int rt_debug_math_device(int op, size_t n, const double* a_host, const double* b_host, double* out_host);
/* closest hit of explicit world-space rays through device traversal `kernel` (1, 2, or 3 = kernel 2's LDS node table "NodeW" with
 * its own box test, which pt_kernel uses when the scene is LDS-resident; 5 / 6 = the walks of the instance service in one lane: the
 * world-space walk with the large instances deferred, then their object-space walks over the Node2 / item records (5: the in-lane
 * fallback of kernel 5) or over the compact NodeQ / Tri32 copies (6: what the serving waves of kernels 5 / 6 walk), with the exact-tie rule): rays n*6 (orig,dir);
 * out n*12 = {hit, t, p[3], normal[3], front_face, u, v, leaf index in the reference-order program} */
int rt_debug_hit_device(const rt_scene* s, int kernel, size_t n, const double* rays_host, double t_min, double t_max, double* out_host);
/* the per-tile job sequence of one launch over samples [s_begin, s_end) for a rank that owns `tiles_owned` tiles on a GPU with `n_waves`
 * resident waves (host logic only, no device needed): out[25] = 5 rows {first round, first unit, first sample, samples per unit, units per
 * job}, one per level of decreasing unit / job size, the row behind the last level = {rounds, units, s_end, 0, 0}.  Every tile's samples
 * are cut the same way; jobs are dealt round by round, so the small jobs of the last levels are what runs when the queue runs dry.
 * Returns the number of rounds (jobs per tile) or a negative rt_status. */
int rt_debug_schedule(int64_t tiles_owned, int n_waves, int s_begin, int s_end, int sub_spp, int job_units, int* out25);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_H */
