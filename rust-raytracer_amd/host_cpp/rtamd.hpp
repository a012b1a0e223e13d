// rtamd.hpp -- C++ host side ABOVE the C ABI (include/rtamd.h), mirroring the reference's Rust
// interface for the radiance path: same type names, constructor arguments and error behaviour
// (a Rust panic becomes a C++ exception carrying the rt_status).
//
// The host owns an object graph exactly like the reference's (Arc<dyn Hitable> -> shared_ptr<Hitable>,
// Arc<dyn Material>, Texture); World::World walks it ONCE and emits it through the rt_* builders --
// the same walk a Rust `impl` over the reference's traits performs.
// Nothing here intersects rays: Camera::capture_image is rt_render.
//   raytracer/src/vec3.rs        -> Vec3, RgbImage conversion
//   raytracer/src/material.rs    -> Texture / Material types
//   raytracer/src/objects/*.rs   -> Hitable types
//   raytracer/src/world.rs       -> World
//   raytracer/src/camera.rs      -> Camera
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include <unistd.h>  // fsync: a checkpoint is on disk before it replaces the previous one

#include "rtamd.h"

namespace rtamd_host {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline int check(int rc) {
    if (rc < 0) throw Error(rc, rt_last_error());
    return rc;
}

// vec3.rs:14-19 (operators: `*` between two Vec3 is the DOT product, vec3.rs:335-341)
struct Vec3 {
    double x = 0, y = 0, z = 0;
    Vec3() {}
    Vec3(double a, double b, double c) : x(a), y(b), z(c) {}
    static Vec3 ones() { return Vec3(1, 1, 1); }
    static Vec3 zero() { return Vec3(0, 0, 0); }
    static Vec3 all(double v) { return Vec3(v, v, v); }
    double squared_length() const { return x * x + y * y + z * z; }
    double length() const { return std::sqrt(squared_length()); }
    Vec3 unit() const {
        double l = length();
        if (l == 0.) throw Error(RT_ERR_UNIT_ZERO, "unitizing zero vector");
        return Vec3(x / l, y / l, z / l);
    }
    static Vec3 elemul(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
    static Vec3 cross(Vec3 a, Vec3 b) { return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
    const double* data() const { return &x; }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator+(Vec3 a, double s) { return Vec3(a.x + s, a.y + s, a.z + s); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator-(Vec3 a, double s) { return Vec3(a.x - s, a.y - s, a.z - s); }
inline Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }
inline double operator*(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 operator*(Vec3 a, double s) { return Vec3(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator*(double s, Vec3 a) { return Vec3(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator/(Vec3 a, double s) { return Vec3(a.x / s, a.y / s, a.z / s); }
inline bool operator==(Vec3 a, Vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

class Emitter;  // the walk that lowers the graph onto the C ABI

// material.rs:18-20, 48-84
struct Texture {
    virtual ~Texture() {}
    virtual int emit(Emitter& e) const = 0;
};
struct ConstantTexture : Texture {
    Vec3 color;
    explicit ConstantTexture(Vec3 c) : color(c) {}
    int emit(Emitter& e) const override;
};
struct CheckerTexture : Texture {
    ConstantTexture t0, t1;
    CheckerTexture(ConstantTexture a, ConstantTexture b) : t0(a), t1(b) {}
    int emit(Emitter& e) const override;
};
struct ImageTexture : Texture {
    int width, height;
    std::vector<uint8_t> rgb;
    ImageTexture(int w, int h, std::vector<uint8_t> px) : width(w), height(h), rgb(std::move(px)) {}
    int emit(Emitter& e) const override;
};
// material.rs:21-46, 88-212
struct Material {
    virtual ~Material() {}
    virtual int emit(Emitter& e) const = 0;
};
using TexturePtr = std::shared_ptr<const Texture>;
using MaterialPtr = std::shared_ptr<const Material>;
struct Lambertian : Material {
    TexturePtr albedo;
    explicit Lambertian(TexturePtr a) : albedo(std::move(a)) {}
    int emit(Emitter& e) const override;
};
struct Metal : Material {
    TexturePtr albedo;
    double fuzz;
    Metal(TexturePtr a, double f) : albedo(std::move(a)), fuzz(f) {}
    int emit(Emitter& e) const override;
};
struct Dielectric : Material {
    double ir;
    TexturePtr albedo;
    Dielectric(double i, TexturePtr a) : ir(i), albedo(std::move(a)) {}
    int emit(Emitter& e) const override;
};
struct DiffuseLight : Material {
    TexturePtr emit_tex;
    explicit DiffuseLight(TexturePtr t) : emit_tex(std::move(t)) {}
    int emit(Emitter& e) const override;
};
struct Isotropic : Material {  // material.rs:213-231 (commented out in the reference): phase function of a ConstantMedium
    TexturePtr albedo;
    explicit Isotropic(TexturePtr a) : albedo(std::move(a)) {}
    int emit(Emitter& e) const override;
};

// objects/hit.rs:51-54
struct Hitable {
    virtual ~Hitable() {}
    virtual int emit(Emitter& e) const = 0;
};
using HitablePtr = std::shared_ptr<const Hitable>;
using HitableList = std::vector<HitablePtr>;  // impl Hitable for Vec<Arc<dyn Hitable>>, hit.rs:56

struct Sphere : Hitable {  // sphere.rs:9-13
    Vec3 center;
    double radius;
    MaterialPtr material;
    Sphere(Vec3 c, double r, MaterialPtr m) : center(c), radius(r), material(std::move(m)) {}
    int emit(Emitter& e) const override;
};
struct XYRectangle : Hitable {  // rectangle.rs:7-12
    std::pair<double, double> xy0, xy1;
    double z;
    MaterialPtr material;
    XYRectangle(std::pair<double, double> a, std::pair<double, double> b, double k, MaterialPtr m) : xy0(a), xy1(b), z(k), material(std::move(m)) {}
    int emit(Emitter& e) const override;
};
struct XZRectangle : Hitable {  // rectangle.rs:44-49
    std::pair<double, double> xz0, xz1;
    double y;
    MaterialPtr material;
    XZRectangle(std::pair<double, double> a, std::pair<double, double> b, double k, MaterialPtr m) : xz0(a), xz1(b), y(k), material(std::move(m)) {}
    int emit(Emitter& e) const override;
};
struct YZRectangle : Hitable {  // rectangle.rs:82-87
    std::pair<double, double> yz0, yz1;
    double x;
    MaterialPtr material;
    YZRectangle(std::pair<double, double> a, std::pair<double, double> b, double k, MaterialPtr m) : yz0(a), yz1(b), x(k), material(std::move(m)) {}
    int emit(Emitter& e) const override;
};
struct Cube : Hitable {  // cube.rs:9-62
    Vec3 box_min, box_max;
    MaterialPtr material;
    Cube(Vec3 mn, Vec3 mx, MaterialPtr m) : box_min(mn), box_max(mx), material(std::move(m)) {}
    int emit(Emitter& e) const override;
};
struct List : Hitable {  // a Vec<Arc<dyn Hitable>> used as a Hitable
    HitableList items;
    explicit List(HitableList l) : items(std::move(l)) {}
    int emit(Emitter& e) const override;
};
struct BVHNode : Hitable {  // bvh.rs:29-83
    HitablePtr left, right;   // BVHNode::construct
    HitableList src_objects;  // BVHNode::new (when left/right are null)
    static std::shared_ptr<BVHNode> construct(HitablePtr l, HitablePtr r) {
        auto n = std::make_shared<BVHNode>();
        n->left = std::move(l);
        n->right = std::move(r);
        return n;
    }
    static std::shared_ptr<BVHNode> new_(HitableList objs) {
        auto n = std::make_shared<BVHNode>();
        n->src_objects = std::move(objs);
        return n;
    }
    int emit(Emitter& e) const override;
};
struct Mesh : Hitable {  // mesh.rs:144-198
    std::string obj_file;
    MaterialPtr material;
    bool synthesize_normals = false;
    static std::shared_ptr<Mesh> load_obj(std::string file, MaterialPtr m, bool synth = false) {
        auto me = std::make_shared<Mesh>();
        me->obj_file = std::move(file);
        me->material = std::move(m);
        me->synthesize_normals = synth;
        return me;
    }
    int emit(Emitter& e) const override;
};
struct Transform : Hitable {  // transform.rs:17-22
    Vec3 rotate_in_degree, scale, translate;
    HitablePtr obj;
    Transform(Vec3 r, Vec3 s, Vec3 t, HitablePtr o) : rotate_in_degree(r), scale(s), translate(t), obj(std::move(o)) {}
    int emit(Emitter& e) const override;
};
struct ConstantMedium : Hitable {  // medium.rs:9-22
    double density;
    HitablePtr boundary;
    MaterialPtr phase_function;
    ConstantMedium(double d, HitablePtr b, MaterialPtr p) : density(d), boundary(std::move(b)), phase_function(std::move(p)) {}
    int emit(Emitter& e) const override;
};
struct XZRectLight : Hitable {  // light.rs:127-146 (as a Hitable; `scale` only feeds SPPM photon power)
    std::pair<double, double> xz0, xz1;
    double y;
    Vec3 flux;
    double scale;
    XZRectLight(std::pair<double, double> a, std::pair<double, double> b, double k, Vec3 f, double s) : xz0(a), xz1(b), y(k), flux(f), scale(s) {}
    int emit(Emitter& e) const override;
};
struct SphereDiffuseLight : Hitable {  // light.rs:67-86
    Vec3 center;
    double radius;
    Vec3 flux;
    double scale;
    SphereDiffuseLight(Vec3 c, double r, Vec3 f, double s) : center(c), radius(r), flux(f), scale(s) {}
    int emit(Emitter& e) const override;
};

class Emitter {
   public:
    rt_scene* s;
    uint64_t bvh_seed;
    std::map<const void*, int> seen;  // shared nodes (Arc clones) are emitted once
    Emitter(rt_scene* sc, uint64_t seed) : s(sc), bvh_seed(seed) {}
    template <class T>
    int once(const T* node) {
        auto it = seen.find(node);
        if (it != seen.end()) return it->second;
        int id = node->emit(*this);
        seen[node] = id;
        return id;
    }
};
inline int ConstantTexture::emit(Emitter& e) const { return check(rt_texture_constant(e.s, color.data())); }
inline int CheckerTexture::emit(Emitter& e) const { return check(rt_texture_checker(e.s, t0.emit(e), t1.emit(e))); }
inline int ImageTexture::emit(Emitter& e) const { return check(rt_texture_image(e.s, width, height, rgb.data())); }
inline int Lambertian::emit(Emitter& e) const { return check(rt_material_lambertian(e.s, e.once(albedo.get()))); }
inline int Metal::emit(Emitter& e) const { return check(rt_material_metal(e.s, e.once(albedo.get()), fuzz)); }
inline int Dielectric::emit(Emitter& e) const { return check(rt_material_dielectric(e.s, ir, e.once(albedo.get()))); }
inline int DiffuseLight::emit(Emitter& e) const { return check(rt_material_diffuse_light(e.s, e.once(emit_tex.get()))); }
inline int Isotropic::emit(Emitter& e) const { return check(rt_material_isotropic(e.s, e.once(albedo.get()))); }
inline int ConstantMedium::emit(Emitter& e) const {
    return check(rt_object_constant_medium(e.s, density, e.once(boundary.get()), e.once(phase_function.get())));
}
inline int Sphere::emit(Emitter& e) const { return check(rt_object_sphere(e.s, center.data(), radius, e.once(material.get()))); }
inline int XYRectangle::emit(Emitter& e) const {
    return check(rt_object_rect_xy(e.s, xy0.first, xy0.second, xy1.first, xy1.second, z, e.once(material.get())));
}
inline int XZRectangle::emit(Emitter& e) const {
    return check(rt_object_rect_xz(e.s, xz0.first, xz0.second, xz1.first, xz1.second, y, e.once(material.get())));
}
inline int YZRectangle::emit(Emitter& e) const {
    return check(rt_object_rect_yz(e.s, yz0.first, yz0.second, yz1.first, yz1.second, x, e.once(material.get())));
}
inline int Cube::emit(Emitter& e) const { return check(rt_object_cube(e.s, box_min.data(), box_max.data(), e.once(material.get()))); }
inline int List::emit(Emitter& e) const {
    std::vector<int> ids;
    for (auto& h : items) ids.push_back(e.once(h.get()));
    return check(rt_object_list(e.s, (int)ids.size(), ids.data()));
}
inline int BVHNode::emit(Emitter& e) const {
    if (left && right) return check(rt_object_bvh_node(e.s, e.once(left.get()), e.once(right.get())));
    std::vector<int> ids;
    for (auto& h : src_objects) ids.push_back(e.once(h.get()));
    return check(rt_object_bvh_build(e.s, (int)ids.size(), ids.data(), e.bvh_seed));
}
inline int Mesh::emit(Emitter& e) const {
    return check(rt_object_mesh_obj(e.s, obj_file.c_str(), e.once(material.get()), synthesize_normals ? 1 : 0, e.bvh_seed));
}
inline int Transform::emit(Emitter& e) const {
    return check(rt_object_transform(e.s, rotate_in_degree.data(), scale.data(), translate.data(), e.once(obj.get())));
}
inline int XZRectLight::emit(Emitter& e) const {
    return check(rt_object_xz_rect_light(e.s, xz0.first, xz0.second, xz1.first, xz1.second, y, flux.data(), scale));
}
inline int SphereDiffuseLight::emit(Emitter& e) const { return check(rt_object_sphere_light(e.s, center.data(), radius, flux.data(), scale)); }

// image::RgbImage stand-in (camera.rs:88,114)
struct RgbImage {
    int width = 0, height = 0;
    std::vector<uint8_t> data;  // row-major RGB8
    void save(const std::string& path) const { check(rt_write_png(path.c_str(), width, height, data.data())); }
};

// camera.rs:11-64
struct Camera {
    rt_camera c{};
    Camera() {}
    Camera(std::pair<Vec3, Vec3> look_from_to, Vec3 vup, double vfov, double aspect_ratio, double aperture, double focus_dist) {
        for (int i = 0; i < 3; i++) {
            c.look_from[i] = look_from_to.first.data()[i];
            c.look_at[i] = look_from_to.second.data()[i];
            c.vup[i] = vup.data()[i];
        }
        c.vfov = vfov;
        c.aspect = aspect_ratio;
        c.aperture = aperture;
        c.focus_dist = focus_dist;
    }
};

// main.rs:26-47 GlobalConfig + the constants capture_image / sample_ray hard-code
struct Config {
    int width = 800, height = 800;  // main.rs:34-45
    int sample_per_pixel = 256;     // camera.rs:73
    int max_depth = 50;             // photon_mapper.rs:334
    double t_min = 0.001;           // photon_mapper.rs:335
    uint64_t seed = 1;
    int integrator = 0;             // 0 BSDF sampling (the reference's structure); 1 light/cosine mixture pdf
    // SPPMIntegrator::new's constants (photon_mapper.rs:17-19,148-149); sppm_iterations == 0: no SPPM pre-pass
    int sppm_iterations = 0, sppm_photons_per_iter = 500000;
    // GPUs of this node the frame is spread over (rt_render_multi: image tiles dealt round-robin, one host thread per GPU inside the
    // library, RCCL gather of the rows); 1 = rt_render on the current device; 0 = every visible GPU.  The image does not depend on it.
    int gpus = 1;
    std::vector<int> devices;       // explicit HIP ordinals, one per rank (overrides gpus; an ordinal may repeat)
};

// world.rs:8-30.  World::new(hitable_list, cam, lights): root = BVHNode::new(hitable_list).
class World {
   public:
    Camera cam;
    World(const HitableList& hitable_list, Camera camera, const HitableList& lights = {}, uint64_t bvh_seed = 1) : cam(camera) {
        check(rt_scene_create(&s_));
        try {
            Emitter e(s_, bvh_seed);
            std::vector<int> ids, lids;
            for (auto& h : hitable_list) ids.push_back(e.once(h.get()));
            for (auto& l : lights) lids.push_back(e.once(l.get()));  // a light shared with the hitable list is emitted once
            check(rt_world_new(s_, (int)ids.size(), ids.data(), bvh_seed));
            if (!lids.empty()) check(rt_scene_set_lights(s_, (int)lids.size(), lids.data()));
            check(rt_scene_commit(s_));
        } catch (...) {
            rt_scene_destroy(s_);
            throw;
        }
    }
    // a scene file of the reference's data/ directory (README.md Track 5)
    explicit World(const std::string& scene_file) {
        check(rt_scene_load_file(scene_file.c_str(), &s_, &cam.c));
    }
    ~World() { rt_scene_destroy(s_); }
    World(const World&) = delete;
    World& operator=(const World&) = delete;
    const rt_scene* handle() const { return s_; }

    // Camera::capture_image (camera.rs:66-128): radiance via the HIP path, then From<Vec3> for Rgb<u8>
    RgbImage capture_image(const Config& cfg = Config(), rt_stats* stats = nullptr, std::vector<double>* radiance = nullptr,
                           std::vector<rt_stats>* per_rank = nullptr) const {
        const bool multi = cfg.gpus != 1 || !cfg.devices.empty();
        const int n_ranks = !cfg.devices.empty() ? (int)cfg.devices.size() : cfg.gpus == 0 ? std::max(1, rt_device_count()) : cfg.gpus;
        const int* ids = cfg.devices.empty() ? nullptr : cfg.devices.data();
        std::vector<rt_stats> rank_stats(multi ? (size_t)n_ranks : 0);
        rt_params p;
        rt_default_params(&p);
        p.width = cfg.width; p.height = cfg.height; p.spp = cfg.sample_per_pixel; p.max_depth = cfg.max_depth;
        p.t_min = cfg.t_min; p.seed = cfg.seed; p.integrator = cfg.integrator;
        std::vector<double> rad((size_t)cfg.width * cfg.height * 3);
        if (cfg.sppm_iterations > 0) {  // main.rs:52-54: SPPMIntegrator::new(world) then capture_image(integrator)
            rt_sppm_config sc;
            rt_default_sppm_config(&sc);
            sc.iterations = cfg.sppm_iterations;
            sc.photons_per_iter = cfg.sppm_photons_per_iter;
            if (multi) check(rt_render_sppm_multi(s_, &cam.c, &p, &sc, n_ranks, ids, rad.data(), rank_stats.data()));
            else check(rt_render_sppm(s_, &cam.c, &p, &sc, rad.data(), nullptr, nullptr, stats));
        } else {
            if (multi) check(rt_render_multi(s_, &cam.c, &p, n_ranks, ids, rad.data(), rank_stats.data()));
            else check(rt_render(s_, &cam.c, &p, rad.data(), stats));
        }
        if (multi && stats) {  // the frame's totals: rank 0's record with the samples and kernel time of all ranks (max over ranks: they run side by side)
            *stats = rank_stats[0];
            for (size_t i = 1; i < rank_stats.size(); i++) {
                stats->samples += rank_stats[i].samples;
                stats->kernel_ms = std::max(stats->kernel_ms, rank_stats[i].kernel_ms);
            }
        }
        if (per_rank) *per_rank = rank_stats;
        RgbImage img;
        img.width = cfg.width;
        img.height = cfg.height;
        img.data.resize(rad.size());
        check(rt_tonemap_u8(rad.data(), rad.size(), img.data.data()));
        if (radiance) *radiance = std::move(rad);
        return img;
    }

    // capture_image in instalments (rt_render_accumulate / rt_accum_finalize): every call traces the next `run_samples` sample indices of
    // every pixel into the accumulator kept in `state_file` (created on the first call; its header pins the frame it belongs to) and
    // returns true, with the finished image, once all cfg.sample_per_pixel samples are in.  Killing the process between calls loses at
    // most one instalment; the finished frame equals capture_image's bit for bit (the RNG is keyed by pixel and sample: no other state).
    bool capture_image_resumable(const Config& cfg, const std::string& state_file, int run_samples, RgbImage* out, int* done_samples = nullptr) const {
        rt_params p;
        rt_default_params(&p);
        p.width = cfg.width; p.height = cfg.height; p.spp = cfg.sample_per_pixel; p.max_depth = cfg.max_depth;
        p.t_min = cfg.t_min; p.seed = cfg.seed; p.integrator = cfg.integrator;
        // what the state belongs to: the frame (rt_params), the scene and the camera (fingerprints), and the library's image spec
        // (RNG / ln / sin / sampling-order version) -- a state of another scene or build must not be resumed under a bit-identity claim
        struct Header {
            char magic[8];
            int32_t width, height, spp, max_depth, integrator, next_sample;
            uint64_t seed;
            double t_min;
            int64_t n_doubles;
            uint64_t scene_fingerprint, camera_fingerprint;
            char spec[48];
        } want = {{'R', 'T', 'A', 'M', 'D', 'C', 'K', '2'}, p.width, p.height, p.spp, p.max_depth, p.integrator, 0, p.seed, p.t_min, rt_accum_state_doubles(&p),
                  rt_scene_fingerprint(s_), 0, {0}};
        {
            uint64_t h = 1469598103934665603ull;  // FNV-1a over Camera::new's arguments
            const unsigned char* cb = (const unsigned char*)&cam.c;
            for (size_t i = 0; i < sizeof(cam.c); i++) h = (h ^ cb[i]) * 1099511628211ull;
            want.camera_fingerprint = h;
            std::strncpy(want.spec, rt_spec_version(), sizeof(want.spec) - 1);
        }
        if (want.n_doubles <= 0 || run_samples < 1) throw Error(RT_ERR_ARG, "capture_image_resumable: bad frame size or instalment");
        std::vector<double> state((size_t)want.n_doubles, 0.);
        Header h = want;
        if (FILE* f = std::fopen(state_file.c_str(), "rb")) {
            const bool ok = std::fread(&h, sizeof(h), 1, f) == 1 && std::fread(state.data(), sizeof(double), state.size(), f) == state.size();
            std::fclose(f);
            Header cmp = h;
            cmp.next_sample = 0;
            if (!ok || std::memcmp(&cmp, &want, sizeof(Header)) != 0 || h.next_sample < 0 || h.next_sample > p.spp)
                throw Error(RT_ERR_ARG, "capture_image_resumable: " + state_file + " is not the state of this frame (frame parameters, scene, camera or library spec \"" +
                                            std::string(want.spec) + "\" differ)");
        }
        if (h.next_sample < p.spp) {
            const int end = std::min(p.spp, h.next_sample + run_samples);
            check(rt_render_accumulate(s_, &cam.c, &p, h.next_sample, end, state.data(), nullptr));
            h.next_sample = end;
            const std::string tmp = state_file + ".tmp";  // written beside, then renamed: a kill never leaves half a state
            FILE* f = std::fopen(tmp.c_str(), "wb");
            if (!f) throw Error(RT_ERR_IO, "cannot write " + tmp);
            bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1 && std::fwrite(state.data(), sizeof(double), state.size(), f) == state.size();
            ok = ok && std::fflush(f) == 0 && ::fsync(::fileno(f)) == 0;  // on disk before it takes the old state's name
            if (std::fclose(f) != 0 || !ok || std::rename(tmp.c_str(), state_file.c_str()) != 0) throw Error(RT_ERR_IO, "cannot write " + state_file);
        }
        if (done_samples) *done_samples = h.next_sample;
        if (h.next_sample < p.spp) return false;
        std::vector<double> rad((size_t)cfg.width * cfg.height * 3);
        check(rt_accum_finalize(&p, state.data(), rad.data()));
        out->width = cfg.width;
        out->height = cfg.height;
        out->data.resize(rad.size());
        check(rt_tonemap_u8(rad.data(), rad.size(), out->data.data()));
        return true;
    }

   private:
    rt_scene* s_ = nullptr;
};

// scene.rs:16-112 cornell_box_scene(), written against the mirrored types exactly as the reference writes it
inline std::unique_ptr<World> cornell_box_scene(const std::string& cube_obj, double aspect_ratio = 1.0, uint64_t bvh_seed = 1) {
    auto tex = [](double r, double g, double b) { return std::make_shared<ConstantTexture>(Vec3(r, g, b)); };
    MaterialPtr red = std::make_shared<Lambertian>(tex(0.75, 0.25, 0.25));
    MaterialPtr white = std::make_shared<Lambertian>(tex(0.75, 0.75, 0.75));
    MaterialPtr blue = std::make_shared<Lambertian>(tex(0.25, 0.25, 0.75));
    auto light = std::make_shared<XZRectLight>(std::make_pair(213., 227.), std::make_pair(343., 332.), 554., Vec3(1., 1., 1.), 1000000.);
    HitableList hitable_list = {
        std::make_shared<YZRectangle>(std::make_pair(0.0, 0.0), std::make_pair(555.0, 555.0), 555., red),
        std::make_shared<YZRectangle>(std::make_pair(0., 0.), std::make_pair(555., 555.), 0., blue),
        std::make_shared<XZRectangle>(std::make_pair(0., 0.), std::make_pair(555., 555.), 0., white),
        std::make_shared<XZRectangle>(std::make_pair(0., 0.), std::make_pair(555., 555.), 555., white),
        std::make_shared<XYRectangle>(std::make_pair(0., 0.), std::make_pair(555., 555.), 555., white),
        std::make_shared<Sphere>(Vec3(140., 100., 240.), 100., std::make_shared<Dielectric>(1.5, tex(0.999, 0.999, 0.999))),
        std::make_shared<Sphere>(Vec3(400., 100., 360.), 100., std::make_shared<Metal>(tex(0.999, 0.999, 0.999), 0.)),
        light,
        std::make_shared<Transform>(Vec3::zero(), Vec3::ones() * 50., Vec3(100., 50., 100.), Mesh::load_obj(cube_obj, white)),
        std::make_shared<Cube>(Vec3(300., 0., 100.), Vec3(380., 100., 180.), white),
    };
    Camera cam({Vec3(278., 278., -800.), Vec3(278., 278., 278.)}, Vec3(0., 1., 0.), 50., aspect_ratio, 0.0, 10.0);
    return std::make_unique<World>(hitable_list, cam, HitableList{light}, bvh_seed);  // scene.rs:100-111: lights = vec![light]
}

}  // namespace rtamd_host
