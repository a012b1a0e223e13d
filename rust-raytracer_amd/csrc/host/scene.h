// Host-side scene description: the reference's World / Hitable / Material /
// Texture object graph (world.rs, objects/*.rs, material.rs) kept as tagged
// records.  This is a DESCRIPTION only -- there is no host-side hit(): the only
// thing that intersects rays in the product is the HIP kernel.
#pragma once
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../common/flat.h"
#include "rtamd.h"

namespace rtamd {

struct RtError {
    int code;
    std::string msg;
    RtError(int c, std::string m) : code(c), msg(std::move(m)) {}
};

struct Box {  // objects/aabb.rs:6-9
    double mn[3], mx[3];
};

enum TexType { TEX_CONSTANT = 0, TEX_CHECKER = 1, TEX_IMAGE = 2, TEX_NOISE = 3 };
struct TextureRec {
    int type = TEX_CONSTANT;
    double color[3] = {0, 0, 0};  // TEX_NOISE: color[0] = scale
    int t0 = -1, t1 = -1;
    int w = 0, h = 0;
    std::vector<uint8_t> rgb;     // TEX_NOISE: the Perlin tables, 256 x 3 f64 unit vectors then 3 x 256 bytes of permutations (6 912 bytes)
};
enum MatType { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_DIFFUSE_LIGHT = 3, MAT_ISOTROPIC = 4 };
struct MaterialRec {
    int type = MAT_LAMBERTIAN;
    int tex = -1;
    double param = 0;
};

enum ObjType { OBJ_SPHERE, OBJ_RECT, OBJ_CUBE, OBJ_TRIANGLE, OBJ_MESH, OBJ_TRANSFORM, OBJ_LIST, OBJ_BVH, OBJ_MEDIUM, OBJ_MOVING_SPHERE };
struct MeshData {
    std::vector<double> pos, nrm;  // 3 per vertex
};
struct ObjectRec {
    int type = OBJ_SPHERE;
    int material = -1;
    double c[3] = {0, 0, 0}, r = 0;             // sphere; moving sphere: center0
    double c1[3] = {0, 0, 0}, time0 = 0, time1 = 0;  // moving sphere (D9): center1, the times the centre is at center0 / center1
    int axis = 0;                               // rect: constant axis (0 YZ, 1 XZ, 2 XY)
    double a0 = 0, b0 = 0, a1 = 0, b1 = 0, k = 0;
    std::vector<int> children;                  // cube: 6 rects; list: items; bvh: {left,right}; mesh: {bvh}; transform: {obj}; medium: {boundary}
    double density = 0;                         // medium: d of ConstantMedium::new (neg_inv_density = -1 / d); material = phase function
    int mesh = -1;                              // triangle / mesh: index into Scene::meshes
    uint32_t ia = 0, ib = 0, ic = 0;            // triangle vertex indices
    double M[16], Minv[16];                     // transform (row-major)
    bool has_box = false;
    Box box;
    // XZRectLight / SphereDiffuseLight fields (light.rs:67-72,128-132): photon power = flux * scale (SPPM only)
    double light_flux[3] = {1., 1., 1.};
    double light_scale = 1.;
};

struct FlatScene {
    std::vector<char> blob;
    FlatView view{};  // base == nullptr on the host copy
    rt_scene_info info{};
    // D9: the time range every moving sphere of the scene is defined on = [latest time0, earliest time1] (their boxes are the unions of the
    // boxes at those two times, scene.cpp: a ray time outside it would move a centre out of its committed box); render_tiles checks the shutter
    double msph_t0_max = -1e300, msph_t1_min = 1e300;
};

struct DeviceCopy {
    int device = -1;
    void* d_blob = nullptr;
};

}  // namespace rtamd

struct rt_scene {
    std::vector<rtamd::TextureRec> textures;
    std::vector<rtamd::MaterialRec> materials;
    std::vector<rtamd::ObjectRec> objects;
    std::vector<std::unique_ptr<rtamd::MeshData>> meshes;
    int root = -1;
    std::vector<int> lights;  // World::new's lights (object ids)
    bool committed = false;
    rtamd::FlatScene flat;
    // device copies of the blob, one per HIP device, created lazily by the render entry points
    mutable std::mutex dev_mu;
    mutable std::vector<rtamd::DeviceCopy> dev;
    ~rt_scene();
};

namespace rtamd {

// rt_tuning_set's process-wide state (abi.cpp): the only thing besides the arguments that influences a render
struct Tuning {
    int no_lds = 0, n_top = -1, sub_spp = 0, max_leaf = 0, sppm_cap = 0, knn_cand = -1, coop_pool = 0, multi_force_rccl = 0, wf_workspace_mb = 0;
    double c_box = 0.;
};
Tuning tuning();  // a snapshot (copied under a mutex): take one per API call

// builders (throw RtError)
int add_texture_constant(rt_scene& s, const double c[3]);
int add_texture_checker(rt_scene& s, int t0, int t1);
int add_texture_image(rt_scene& s, int w, int h, const uint8_t* rgb);
int add_material(rt_scene& s, int type, int tex, double param);
int add_sphere(rt_scene& s, const double c[3], double r, int mat);
int add_moving_sphere(rt_scene& s, const double c0[3], const double c1[3], double time0, double time1, double r, int mat);
int add_texture_noise(rt_scene& s, double scale, uint64_t seed);
int add_rect(rt_scene& s, int axis, double a0, double b0, double a1, double b1, double k, int mat);
int add_cube(rt_scene& s, const double mn[3], const double mx[3], int mat);
int add_mesh(rt_scene& s, int n_vert, const double* pos, const double* nrm, int n_tri, const uint32_t* idx, int mat,
             bool synth_normals, uint64_t bvh_seed);
int add_transform(rt_scene& s, const double rot_deg[3], const double scale[3], const double translate[3], int obj);
int add_transform_matrix(rt_scene& s, const double trans[16], const double* inverse_trans, int obj);
int add_mesh_data(rt_scene& s, int n_vert, const double* pos, const double* nrm);
int add_triangle(rt_scene& s, int mesh, uint32_t a, uint32_t b, uint32_t c, int mat);
int add_list(rt_scene& s, int n, const int* objs);
int add_medium(rt_scene& s, double density, int boundary, int phase_material);
int add_bvh_node(rt_scene& s, int left, int right);
int add_bvh_build(rt_scene& s, std::vector<int> objs, uint64_t bvh_seed);
void check_obj(const rt_scene& s, int o);
void check_mat(const rt_scene& s, int m);
bool bounding_box(const rt_scene& s, int o, Box& out);

// flatten.cpp
void flatten(rt_scene& s);
// loader.cpp
rt_scene* load_scene_file(const char* path, rt_camera* cam);
// obj.cpp
struct ObjMesh {
    std::vector<double> pos, nrm;
    std::vector<uint32_t> idx;
    bool has_normals = false;
};
ObjMesh load_obj_file(const char* path);
void synthesize_normals(int n_vert, const double* pos, int n_tri, const uint32_t* idx, std::vector<double>& out);
// png.cpp
uint8_t tonemap_channel(double c);
void write_png(const char* path, int w, int h, const uint8_t* rgb);

// camera.cpp : Camera::new (camera.rs:24-55)
struct CameraDev {
    double origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    double lens_radius;
};
CameraDev make_camera(const rt_camera& c);

}  // namespace rtamd
