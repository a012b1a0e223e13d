// extern "C" entry points of librtamd.so (declared in include/rtamd.h).
// Every function converts C++ exceptions into rt_status codes; nothing unwinds
// across the boundary.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "common/rng.h"
#include "common/schedule.h"
#include "device/device.h"
#include "host/scene.h"
#include "rtamd.h"

using namespace rtamd;

static thread_local std::string g_err;
static Tuning g_tuning;
static std::mutex g_tuning_mu;
namespace rtamd {
// a consistent snapshot: callers take ONE copy per API call (render_tiles, render_sppm, accel_build_bvh, make_plan) and pass
// it down, so a concurrent rt_tuning_set can never make one render see two different settings
Tuning tuning() {
    std::lock_guard<std::mutex> g(g_tuning_mu);
    return g_tuning;
}
}  // namespace rtamd

rt_scene::~rt_scene() { free_device_copies(*this); }

template <class F>
static int guard(F&& f) {
    try {
        return f();
    } catch (const RtError& e) {
        g_err = e.msg;
        return e.code;
    } catch (const std::bad_alloc&) {
        g_err = "out of memory";
        return RT_ERR_ARG;
    } catch (const std::exception& e) {
        g_err = e.what();
        return RT_ERR_ARG;
    } catch (...) {
        g_err = "unknown error";
        return RT_ERR_ARG;
    }
}
#define REQUIRE(c, msg) \
    if (!(c)) throw RtError(RT_ERR_ARG, msg)

static void not_committed_only(const rt_scene* s) {
    REQUIRE(s, "null scene");
    if (s->committed) throw RtError(RT_ERR_ARG, "scene is immutable after rt_scene_commit");
}

extern "C" {

int rt_abi_version(void) { return RTAMD_ABI_VERSION; }
const char* rt_last_error(void) { return g_err.c_str(); }

void rt_default_params(rt_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->width = 800;      // main.rs:34
    p->height = 800;     // main.rs:45 (aspect 1)
    p->spp = 256;        // camera.rs:73
    p->max_depth = 50;   // photon_mapper.rs:334
    p->t_min = 0.001;    // photon_mapper.rs:335
    p->seed = 1;
    p->rank = 0;
    p->world = 1;
    p->spp_chunk = 0;
    p->kernel = 0;
    p->device = -1;
    p->integrator = 0;
    p->time0 = 0.;  // (book-2 extension: shutter closed = no time draw)
    p->time1 = 0.;
}
int rt_device_count(void) { return device_count(); }
// Frame-sized device buffers of rt_render_multi (the gathered rows, the stitched frame, the ranks' own rows) are kept between calls:
// allocating and freeing them cost every frame about a millisecond (hipFree waits for the device).  Per (device, size): at most
// FRAME_POOL_KEEP idle buffers, and FRAME_POOL_BYTES of idle buffers in all -- a host that renders many resolutions or rank counts does
// not pile up a frame per shape: beyond the cap the least recently returned buffers are freed; rt_release_workspaces frees them all.
namespace {
const size_t FRAME_POOL_KEEP = 4;
const size_t FRAME_POOL_BYTES = size_t(1) << 30;
struct IdleFrame {
    int dev;
    size_t bytes;
    void* p;
};
std::mutex g_frame_pool_mu;
std::vector<IdleFrame> g_frame_pool;  // in the order they were returned: the front is the least recently used
void* frame_pool_take(int dev, size_t bytes) {
    {
        std::lock_guard<std::mutex> g(g_frame_pool_mu);
        for (size_t i = g_frame_pool.size(); i-- > 0;)
            if (g_frame_pool[i].dev == dev && g_frame_pool[i].bytes == bytes) {
                void* p = g_frame_pool[i].p;
                g_frame_pool.erase(g_frame_pool.begin() + (ptrdiff_t)i);
                return p;
            }
    }
    return dev_alloc(bytes);  // (the caller has made `dev` current)
}
void frame_pool_give(int dev, size_t bytes, void* p) {  // the caller has made `dev` current
    std::vector<IdleFrame> drop;
    {
        std::lock_guard<std::mutex> g(g_frame_pool_mu);
        size_t same = 0, total = bytes;
        for (const IdleFrame& f : g_frame_pool) {
            same += (f.dev == dev && f.bytes == bytes) ? 1 : 0;
            total += f.bytes;
        }
        if (same >= FRAME_POOL_KEEP || bytes > FRAME_POOL_BYTES) {
            drop.push_back(IdleFrame{dev, bytes, p});
        } else {
            g_frame_pool.push_back(IdleFrame{dev, bytes, p});
            while (total > FRAME_POOL_BYTES && g_frame_pool.size() > 1) {  // evict the least recently returned
                total -= g_frame_pool.front().bytes;
                drop.push_back(g_frame_pool.front());
                g_frame_pool.erase(g_frame_pool.begin());
            }
        }
    }
    for (const IdleFrame& f : drop) {
        try {
            if (f.dev != dev) dev_set_device(f.dev);
            dev_free(f.p);
        } catch (...) {
        }
    }
    if (!drop.empty()) {
        try {
            dev_set_device(dev);
        } catch (...) {
        }
    }
}
size_t frame_pool_release() {
    std::vector<IdleFrame> all;
    {
        std::lock_guard<std::mutex> g(g_frame_pool_mu);
        all.swap(g_frame_pool);
    }
    size_t freed = 0;
    for (const IdleFrame& f : all) {
        try {
            dev_set_device(f.dev);
            dev_free(f.p);
            freed += f.bytes;
        } catch (...) {
        }
    }
    return freed;
}
// rt_params.device >= 0 makes that device current for the call only: the caller's current device comes back
struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(int want) {
        if (want < 0) return;
        prev = dev_get_device();
        if (prev != want) dev_set_device(want);
        else prev = -1;
    }
    ~DeviceScope() {
        if (prev < 0) return;
        try {
            dev_set_device(prev);
        } catch (...) {
        }
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};
}  // namespace

int64_t rt_release_workspaces(void) {
    int64_t n = 0;
    guard([&] {
        int cur = -1;
        try {
            cur = dev_get_device();
        } catch (...) {
        }
        n = (int64_t)release_workspaces();
        n += (int64_t)frame_pool_release();
        if (cur >= 0) dev_set_device(cur);
        exchange_release_idle();
        return (int)RT_OK;
    });
    return n;
}
void rt_tuning_default(rt_tuning* t) {
    if (!t) return;
    std::memset(t, 0, sizeof(*t));
    t->top_nodes = -1;
    t->sppm_knn_candidates = -1;
}
int rt_tuning_set(const rt_tuning* t) {
    return guard([&] {
        REQUIRE(t, "null argument");
        REQUIRE(t->max_leaf >= 0 && t->max_leaf <= 4, "max_leaf must be 0 (auto) or 1..4");
        REQUIRE(t->sah_box_cost >= 0. && t->sah_box_cost < 1e6, "sah_box_cost out of range");
        Tuning n;
        n.no_lds = t->no_lds != 0;
        n.n_top = t->top_nodes < 0 ? -1 : t->top_nodes;
        n.sub_spp = std::max(0, t->sub_spp);
        n.max_leaf = t->max_leaf;
        n.coop_pool = std::max(0, t->coop_pool);
        n.sppm_cap = std::max(0, t->sppm_photon_capacity);
        n.knn_cand = t->sppm_knn_candidates < 0 ? -1 : t->sppm_knn_candidates;
        n.multi_force_rccl = t->multi_force_rccl != 0;
        n.wf_workspace_mb = std::max(0, t->wf_workspace_mb);
        n.c_box = t->sah_box_cost;
        {
            std::lock_guard<std::mutex> g(g_tuning_mu);
            g_tuning = n;
        }
        return (int)RT_OK;
    });
}

int rt_scene_create(rt_scene** out) {
    return guard([&] {
        REQUIRE(out, "null out pointer");
        *out = new rt_scene();
        return (int)RT_OK;
    });
}
void rt_scene_destroy(rt_scene* s) { delete s; }

int rt_texture_constant(rt_scene* s, const double color[3]) {
    return guard([&] {
        not_committed_only(s);
        REQUIRE(color, "null color");
        return add_texture_constant(*s, color);
    });
}
int rt_texture_checker(rt_scene* s, int t0, int t1) {
    return guard([&] {
        not_committed_only(s);
        return add_texture_checker(*s, t0, t1);
    });
}
int rt_texture_image(rt_scene* s, int width, int height, const uint8_t* rgb) {
    return guard([&] {
        not_committed_only(s);
        return add_texture_image(*s, width, height, rgb);
    });
}
int rt_texture_noise(rt_scene* s, double scale, uint64_t seed) {
    return guard([&] {
        not_committed_only(s);
        return add_texture_noise(*s, scale, seed);
    });
}
int rt_material_lambertian(rt_scene* s, int tex) {
    return guard([&] {
        not_committed_only(s);
        return add_material(*s, MAT_LAMBERTIAN, tex, 0.);
    });
}
int rt_material_metal(rt_scene* s, int tex, double fuzz) {
    return guard([&] {
        not_committed_only(s);
        return add_material(*s, MAT_METAL, tex, fuzz);
    });
}
int rt_material_dielectric(rt_scene* s, double ir, int tex) {
    return guard([&] {
        not_committed_only(s);
        return add_material(*s, MAT_DIELECTRIC, tex, ir);
    });
}
int rt_material_isotropic(rt_scene* s, int tex) {
    return guard([&] {
        not_committed_only(s);
        return add_material(*s, MAT_ISOTROPIC, tex, 0.);
    });
}
int rt_object_constant_medium(rt_scene* s, double density, int boundary, int phase_material) {
    return guard([&] {
        not_committed_only(s);
        return add_medium(*s, density, boundary, phase_material);
    });
}
int rt_material_diffuse_light(rt_scene* s, int tex) {
    return guard([&] {
        not_committed_only(s);
        return add_material(*s, MAT_DIFFUSE_LIGHT, tex, 0.);
    });
}

int rt_object_sphere(rt_scene* s, const double center[3], double radius, int material) {
    return guard([&] {
        not_committed_only(s);
        REQUIRE(center, "null center");
        return add_sphere(*s, center, radius, material);
    });
}
int rt_object_moving_sphere(rt_scene* s, const double center0[3], const double center1[3], double time0, double time1, double radius, int material) {
    return guard([&] {
        not_committed_only(s);
        REQUIRE(center0 && center1, "null center");
        return add_moving_sphere(*s, center0, center1, time0, time1, radius, material);
    });
}
int rt_object_rect_xy(rt_scene* s, double x0, double y0, double x1, double y1, double z, int material) {
    return guard([&] {
        not_committed_only(s);
        return add_rect(*s, 2, x0, y0, x1, y1, z, material);
    });
}
int rt_object_rect_xz(rt_scene* s, double x0, double z0, double x1, double z1, double y, int material) {
    return guard([&] {
        not_committed_only(s);
        return add_rect(*s, 1, x0, z0, x1, z1, y, material);
    });
}
int rt_object_rect_yz(rt_scene* s, double y0, double z0, double y1, double z1, double x, int material) {
    return guard([&] {
        not_committed_only(s);
        return add_rect(*s, 0, y0, z0, y1, z1, x, material);
    });
}
int rt_object_cube(rt_scene* s, const double box_min[3], const double box_max[3], int material) {
    return guard([&] {
        not_committed_only(s);
        REQUIRE(box_min && box_max, "null box");
        return add_cube(*s, box_min, box_max, material);
    });
}
static int mark_light(rt_scene* s, int obj, const double flux[3], double scale) {
    ObjectRec& o = s->objects[obj];
    for (int i = 0; i < 3; i++) o.light_flux[i] = flux[i];
    o.light_scale = scale;
    return obj;
}
int rt_object_sphere_light(rt_scene* s, const double center[3], double radius, const double flux[3], double scale) {
    return guard([&] {  // SphereDiffuseLight::new, light.rs:74-86
        not_committed_only(s);
        REQUIRE(center && flux, "null argument");
        int m = add_material(*s, MAT_DIFFUSE_LIGHT, add_texture_constant(*s, flux), 0.);
        return mark_light(s, add_sphere(*s, center, radius, m), flux, scale);
    });
}
int rt_object_xz_rect_light(rt_scene* s, double x0, double z0, double x1, double z1, double y, const double flux[3], double scale) {
    return guard([&] {  // XZRectLight::new, light.rs:134-146 (scale only feeds the photon power)
        not_committed_only(s);
        REQUIRE(flux, "null flux");
        int m = add_material(*s, MAT_DIFFUSE_LIGHT, add_texture_constant(*s, flux), 0.);
        return mark_light(s, add_rect(*s, 1, x0, z0, x1, z1, y, m), flux, scale);
    });
}
int rt_object_mesh(rt_scene* s, int n_vert, const double* positions, const double* normals, int n_tri, const uint32_t* indices,
                   int material, int synthesize_normals_flag, uint64_t bvh_seed) {
    return guard([&] {
        not_committed_only(s);
        return add_mesh(*s, n_vert, positions, normals, n_tri, indices, material, synthesize_normals_flag != 0, bvh_seed);
    });
}
int rt_object_mesh_obj(rt_scene* s, const char* obj_path, int material, int synthesize_normals_flag, uint64_t bvh_seed) {
    return guard([&] {
        not_committed_only(s);
        REQUIRE(obj_path, "null path");
        ObjMesh m = load_obj_file(obj_path);
        return add_mesh(*s, (int)(m.pos.size() / 3), m.pos.data(), m.has_normals ? m.nrm.data() : nullptr, (int)(m.idx.size() / 3),
                        m.idx.data(), material, synthesize_normals_flag != 0, bvh_seed);
    });
}
int rt_object_transform(rt_scene* s, const double rotate_deg[3], const double scale[3], const double translate[3], int object) {
    return guard([&] {
        not_committed_only(s);
        REQUIRE(rotate_deg && scale && translate, "null argument");
        return add_transform(*s, rotate_deg, scale, translate, object);
    });
}
int rt_object_transform_matrix(rt_scene* s, const double trans[16], const double* inverse_trans, int object) {
    return guard([&] {
        not_committed_only(s);
        REQUIRE(trans, "null matrix");
        return add_transform_matrix(*s, trans, inverse_trans, object);
    });
}
int rt_mesh_data(rt_scene* s, int n_vert, const double* positions, const double* normals) {
    return guard([&] {
        not_committed_only(s);
        return add_mesh_data(*s, n_vert, positions, normals);
    });
}
int rt_object_triangle(rt_scene* s, int mesh, uint32_t a, uint32_t b, uint32_t c, int material) {
    return guard([&] {
        not_committed_only(s);
        return add_triangle(*s, mesh, a, b, c, material);
    });
}
int rt_object_list(rt_scene* s, int n, const int* objects) {
    return guard([&] {
        not_committed_only(s);
        REQUIRE(n >= 0 && (n == 0 || objects), "bad list");
        return add_list(*s, n, objects);
    });
}
int rt_object_bvh_node(rt_scene* s, int left, int right) {
    return guard([&] {
        not_committed_only(s);
        check_obj(*s, left);
        check_obj(*s, right);
        return add_bvh_node(*s, left, right);
    });
}
int rt_object_bvh_build(rt_scene* s, int n, const int* objects, uint64_t bvh_seed) {
    return guard([&] {
        not_committed_only(s);
        REQUIRE(n > 0 && objects, "BVHNode::new needs a non-empty list");
        return add_bvh_build(*s, std::vector<int>(objects, objects + n), bvh_seed);
    });
}
int rt_object_bounding_box(const rt_scene* s, int object, double out_min_max[6]) {
    return guard([&] {
        REQUIRE(s && out_min_max, "null argument");
        Box b;
        if (!bounding_box(*s, object, b)) throw RtError(RT_ERR_NO_BBOX, "object has no bounding box");
        for (int i = 0; i < 3; i++) {
            out_min_max[i] = b.mn[i];
            out_min_max[3 + i] = b.mx[i];
        }
        return (int)RT_OK;
    });
}
int rt_scene_root(const rt_scene* s) {
    return guard([&] {
        REQUIRE(s, "null scene");
        REQUIRE(s->root >= 0, "scene has no root");
        return s->root;
    });
}
int rt_object_describe(const rt_scene* s, int object, rt_object_desc* out) {
    return guard([&] {
        REQUIRE(s && out, "null argument");
        check_obj(*s, object);
        const ObjectRec& o = s->objects[object];
        static_assert((int)OBJ_SPHERE == RT_OBJ_SPHERE && (int)OBJ_RECT == RT_OBJ_RECT && (int)OBJ_CUBE == RT_OBJ_CUBE &&
                      (int)OBJ_TRIANGLE == RT_OBJ_TRIANGLE && (int)OBJ_MESH == RT_OBJ_MESH && (int)OBJ_TRANSFORM == RT_OBJ_TRANSFORM &&
                      (int)OBJ_LIST == RT_OBJ_LIST && (int)OBJ_BVH == RT_OBJ_BVH && (int)OBJ_MEDIUM == RT_OBJ_MEDIUM &&
                      (int)OBJ_MOVING_SPHERE == RT_OBJ_MOVING_SPHERE, "rt_object_type mirrors ObjType");
        std::memset(out, 0, sizeof(*out));
        out->type = o.type;
        out->material = (o.type == OBJ_SPHERE || o.type == OBJ_RECT || o.type == OBJ_TRIANGLE || o.type == OBJ_MOVING_SPHERE) ? o.material : -1;
        out->n_children = (int32_t)o.children.size();
        if (o.type == OBJ_SPHERE) {
            out->v[0] = o.c[0]; out->v[1] = o.c[1]; out->v[2] = o.c[2]; out->v[3] = o.r;
        } else if (o.type == OBJ_MOVING_SPHERE) {
            out->v[0] = o.c[0]; out->v[1] = o.c[1]; out->v[2] = o.c[2]; out->v[3] = o.r;
            out->v[4] = o.c1[0]; out->v[5] = o.c1[1]; out->v[6] = o.c1[2];
        } else if (o.type == OBJ_RECT) {
            out->axis = o.axis;
            out->v[0] = o.a0; out->v[1] = o.b0; out->v[2] = o.a1; out->v[3] = o.b1; out->v[4] = o.k;
        } else if (o.type == OBJ_TRIANGLE) {
            out->v[0] = (double)o.ia; out->v[1] = (double)o.ib; out->v[2] = (double)o.ic;
        } else if (o.type == OBJ_MEDIUM) {
            out->material = o.material;
            out->v[0] = o.density;
        }
        return (int)RT_OK;
    });
}
int rt_object_children(const rt_scene* s, int object, int capacity, int* out) {
    return guard([&] {
        REQUIRE(s && (out || capacity <= 0), "null argument");
        check_obj(*s, object);
        const ObjectRec& o = s->objects[object];
        for (int i = 0; i < capacity && i < (int)o.children.size(); i++) out[i] = o.children[i];
        return (int)o.children.size();
    });
}
int rt_world_new(rt_scene* s, int n, const int* objects, uint64_t bvh_seed) {
    return guard([&] {
        not_committed_only(s);
        REQUIRE(n > 0 && objects, "World::new needs a non-empty list");
        s->root = add_bvh_build(*s, std::vector<int>(objects, objects + n), bvh_seed);
        return s->root;
    });
}
int rt_scene_set_lights(rt_scene* s, int n, const int* objects) {
    return guard([&] {
        not_committed_only(s);
        REQUIRE(n >= 0 && (n == 0 || objects), "bad light list");
        std::vector<int> v;
        for (int i = 0; i < n; i++) {
            check_obj(*s, objects[i]);
            const ObjectRec& o = s->objects[objects[i]];
            if (!(o.type == OBJ_SPHERE || (o.type == OBJ_RECT && o.axis == 1)))
                throw RtError(RT_ERR_ARG, "a light must be a sphere or an XZ rectangle (light.rs:67-86,127-146)");
            v.push_back(objects[i]);
        }
        s->lights = v;
        return (int)RT_OK;
    });
}
int rt_scene_set_root(rt_scene* s, int object) {
    return guard([&] {
        not_committed_only(s);
        check_obj(*s, object);
        s->root = object;
        return (int)RT_OK;
    });
}
int rt_scene_cornell_box(rt_scene* s, const char* cube_obj_path, double aspect_ratio, uint64_t bvh_seed, rt_camera* cam_out) {
    return guard([&] {  // scene.rs:16-112, numbers verbatim
        not_committed_only(s);
        REQUIRE(cube_obj_path, "null path");
        auto ctex = [&](double r, double g, double b) {
            const double c[3] = {r, g, b};
            return add_texture_constant(*s, c);
        };
        int red = add_material(*s, MAT_LAMBERTIAN, ctex(0.75, 0.25, 0.25), 0.);
        int white = add_material(*s, MAT_LAMBERTIAN, ctex(0.75, 0.75, 0.75), 0.);
        int blue = add_material(*s, MAT_LAMBERTIAN, ctex(0.25, 0.25, 0.75), 0.);
        int light = add_material(*s, MAT_DIFFUSE_LIGHT, ctex(1., 1., 1.), 0.);  // XZRectLight::new(.., flux (1,1,1), 1e6)
        std::vector<int> items;
        items.push_back(add_rect(*s, 0, 0.0, 0.0, 555.0, 555.0, 555., red));
        items.push_back(add_rect(*s, 0, 0., 0., 555., 555., 0., blue));
        items.push_back(add_rect(*s, 1, 0., 0., 555., 555., 0., white));
        items.push_back(add_rect(*s, 1, 0., 0., 555., 555., 555., white));
        items.push_back(add_rect(*s, 2, 0., 0., 555., 555., 555., white));
        const double c1[3] = {140., 100., 240.}, c2[3] = {400., 100., 360.};
        items.push_back(add_sphere(*s, c1, 100., add_material(*s, MAT_DIELECTRIC, ctex(0.999, 0.999, 0.999), 1.5)));
        items.push_back(add_sphere(*s, c2, 100., add_material(*s, MAT_METAL, ctex(0.999, 0.999, 0.999), 0.)));
        const int light_obj = add_rect(*s, 1, 213., 227., 343., 332., 554., light);
        const double one[3] = {1., 1., 1.};
        mark_light(s, light_obj, one, 1000000.);  // XZRectLight::new((213,227),(343,332),554, flux (1,1,1), scale 1e6), scene.rs:26-32
        items.push_back(light_obj);
        s->lights = {light_obj};  // scene.rs:110 vec![Arc::new(light)]
        ObjMesh m = load_obj_file(cube_obj_path);
        if (!m.has_normals) throw RtError(RT_ERR_NO_NORMALS, "cube.obj without normals");
        int mesh = add_mesh(*s, (int)(m.pos.size() / 3), m.pos.data(), m.nrm.data(), (int)(m.idx.size() / 3), m.idx.data(), white, false,
                            bvh_seed);
        const double rot[3] = {0., 0., 0.}, sc[3] = {1. * 50., 1. * 50., 1. * 50.}, tr[3] = {100., 50., 100.};
        items.push_back(add_transform(*s, rot, sc, tr, mesh));
        const double bmin[3] = {300., 0., 100.}, bmax[3] = {380., 100., 180.};
        items.push_back(add_cube(*s, bmin, bmax, white));
        s->root = add_bvh_build(*s, items, bvh_seed);
        if (cam_out) {
            rt_camera c = {{278., 278., -800.}, {278., 278., 278.}, {0., 1., 0.}, 50., aspect_ratio, 0.0, 10.0};
            *cam_out = c;
        }
        return (int)RT_OK;
    });
}
int rt_scene_load_file(const char* path, rt_scene** out, rt_camera* cam_out) {
    return guard([&] {
        REQUIRE(path && out, "null argument");
        *out = load_scene_file(path, cam_out);
        return (int)RT_OK;
    });
}
int rt_scene_commit(rt_scene* s) {
    return guard([&] {
        REQUIRE(s, "null scene");
        if (s->committed) return (int)RT_OK;
        flatten(*s);
        return (int)RT_OK;
    });
}
uint64_t rt_scene_fingerprint(const rt_scene* s) {
    if (!s || !s->committed) return 0;
    uint64_t h = 1469598103934665603ull;  // FNV-1a, 64 bit
    for (unsigned char c : s->flat.blob) {
        h ^= (uint64_t)c;
        h *= 1099511628211ull;
    }
    return h ? h : 1;
}
// bumped whenever an image's bits may change for unchanged inputs (round 5: near ties go to the reference-order walk; rng-3 since round 4)
const char* rt_spec_version(void) { return "rtamd-image-5 rng-3 ln-1 sin-1"; }
int rt_scene_info_get(const rt_scene* s, rt_scene_info* out) {
    return guard([&] {
        REQUIRE(s && out, "null argument");
        *out = s->flat.info;
        out->committed = s->committed ? 1 : 0;
        return (int)RT_OK;
    });
}

// ---- render --------------------------------------------------------------
static RenderPlan make_plan(const rt_params* p) {
    REQUIRE(p, "null params");
    REQUIRE(p->width > 0 && p->height > 0, "width/height must be positive");
    REQUIRE(p->spp > 0, "spp must be positive");
    REQUIRE(p->max_depth >= 0, "max_depth must be >= 0");
    REQUIRE(p->world >= 1 && p->rank >= 0 && p->rank < p->world, "bad rank/world");
    REQUIRE((p->kernel >= 0 && p->kernel <= 2) || p->kernel == 5 || p->kernel == 6, "unknown kernel id (0 auto, 1, 2, 5, 6)");
    REQUIRE(p->integrator >= 0 && p->integrator <= 2, "unknown integrator id");
    REQUIRE(std::isfinite(p->time0) && std::isfinite(p->time1) && p->time1 >= p->time0, "shutter: time1 must be >= time0 and both finite");
    RenderPlan pl;
    pl.width = p->width; pl.height = p->height; pl.spp = p->spp; pl.max_depth = p->max_depth;
    pl.t_min = p->t_min; pl.seed = p->seed; pl.rank = p->rank; pl.world = p->world;
    pl.tiles_x = (p->width + TILE_W - 1) / TILE_W;
    pl.tiles_y = (p->height + TILE_H - 1) / TILE_H;
    pl.tiles_total = (int64_t)pl.tiles_x * pl.tiles_y;
    pl.tiles_owned = (pl.tiles_total - p->rank + p->world - 1) / p->world;
    if (pl.tiles_owned < 0) pl.tiles_owned = 0;
    pl.kernel = p->kernel;
    pl.integrator = p->integrator;
    pl.time0 = p->time0;
    pl.time1 = p->time1;
    // One launch renders all sample indices unless the caller splits them (rt_params.spp_chunk): samples are folded into the
    // accumulator inside the kernel, unit by unit, so no per-launch sample buffer bounds the launch size.
    int chunk = p->spp_chunk > 0 ? p->spp_chunk : p->spp;
    if (chunk > p->spp) chunk = p->spp;
    // one work unit = 64 pixels x sub_spp samples (a wave works through it with in-wave regeneration and fetches the next
    // one as soon as its pool is empty).  Many units balance the 4096 resident waves at the end of a launch: aim at
    // >= ~12 units per wave, 4 <= sub_spp <= 8 (measured on the headline workload: 4: 2144, 8: 2166, 16: 2104 Msamples/s).
    int64_t want_units = 12 * 4096;
    int64_t subs = std::max<int64_t>(1, want_units / std::max<int64_t>(1, pl.tiles_owned));
    int sub = (int)((chunk + subs - 1) / subs);
    pl.sub_spp = std::max(std::min(chunk, 4), std::min(sub, 8));
    const int tun_sub_spp = tuning().sub_spp;
    if (tun_sub_spp > 0) pl.sub_spp = std::max(1, std::min(std::min(chunk, 8), tun_sub_spp));  // rt_tuning (A/B runs)
    {   // at most 2^31 jobs per launch; 2^30 uniform units leave room for the tapered end of the schedule (host/schedule.cpp: up to
        // 1.2x as many units as a uniform cut, single-unit jobs)
        const int64_t max_chunk = (int64_t(1) << 30) / std::max<int64_t>(1, pl.tiles_owned) * pl.sub_spp;
        if (max_chunk < 1) throw RtError(RT_ERR_UNSUPPORTED, "image too large for one rank");
        if (chunk > max_chunk) chunk = (int)max_chunk;
    }
    pl.spp_chunk = chunk;
    return pl;
}

int64_t rt_tiles_total(const rt_params* p) {
    if (!p || p->width <= 0 || p->height <= 0) return RT_ERR_ARG;
    return (int64_t)((p->width + TILE_W - 1) / TILE_W) * ((p->height + TILE_H - 1) / TILE_H);
}
int64_t rt_tiles_owned(const rt_params* p) {
    if (!p || p->width <= 0 || p->height <= 0 || p->world < 1 || p->rank < 0 || p->rank >= p->world) return RT_ERR_ARG;
    int64_t total = rt_tiles_total(p);
    return (total - p->rank + p->world - 1) / p->world;
}

int rt_render_tiles_device(const rt_scene* s, const rt_camera* cam, const rt_params* p, double* d_tiles, void* hip_stream,
                           rt_stats* stats) {
    return guard([&] {
        REQUIRE(s && cam && d_tiles, "null argument");
        if (!s->committed) throw RtError(RT_ERR_NOT_COMMITTED, "rt_scene_commit has not been called");
        if (device_count() < 1) throw RtError(RT_ERR_NO_DEVICE, "no HIP device: librtamd has no CPU fallback");
        auto t0 = std::chrono::steady_clock::now();
        RenderPlan pl = make_plan(p);
        DeviceScope dev_scope(p->device);  // (d_tiles and hip_stream must belong to it; the caller's current device is restored)
        CameraDev cd = make_camera(*cam);
        if (stats) std::memset(stats, 0, sizeof(*stats));
        render_tiles(*s, cd, pl, d_tiles, hip_stream, stats);
        if (stats) {
            stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            // pixels actually inside the image among this rank's tiles
            uint64_t px = 0;
            for (int64_t lt = 0; lt < pl.tiles_owned; lt++) {
                int64_t t = lt * pl.world + pl.rank;
                int tx = (int)(t % pl.tiles_x), ty = (int)(t / pl.tiles_x);
                int w = std::min(TILE_W, pl.width - tx * TILE_W), h = std::min(TILE_H, pl.height - ty * TILE_H);
                px += (uint64_t)w * h;
            }
            stats->samples = px * (uint64_t)pl.spp;
        }
        return (int)RT_OK;
    });
}

int rt_render_accumulate_device(const rt_scene* s, const rt_camera* cam, const rt_params* p, int32_t sample_begin, int32_t sample_end,
                                double* d_accum, void* hip_stream, rt_stats* stats) {
    return guard([&] {
        REQUIRE(s && cam && p && d_accum, "null argument");
        if (!s->committed) throw RtError(RT_ERR_NOT_COMMITTED, "rt_scene_commit has not been called");
        if (device_count() < 1) throw RtError(RT_ERR_NO_DEVICE, "no HIP device: librtamd has no CPU fallback");
        auto t0 = std::chrono::steady_clock::now();
        RenderPlan pl = make_plan(p);
        REQUIRE(sample_begin >= 0 && sample_begin < sample_end && sample_end <= pl.spp, "sample range: 0 <= begin < end <= spp");
        REQUIRE(pl.integrator == 0 || pl.integrator == 1, "resumable rendering: integrators 0 and 1");
        pl.s_first = sample_begin;
        pl.s_last = sample_end;
        pl.ext_accum = d_accum;
        DeviceScope dev_scope(p->device);  // (d_accum and hip_stream must belong to it)
        CameraDev cd = make_camera(*cam);
        if (stats) std::memset(stats, 0, sizeof(*stats));
        render_tiles(*s, cd, pl, nullptr, hip_stream, stats);
        if (stats) {
            stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            uint64_t px = 0;
            for (int64_t lt = 0; lt < pl.tiles_owned; lt++) {
                int64_t t = lt * pl.world + pl.rank;
                int tx = (int)(t % pl.tiles_x), ty = (int)(t / pl.tiles_x);
                px += (uint64_t)std::min(TILE_W, pl.width - tx * TILE_W) * std::min(TILE_H, pl.height - ty * TILE_H);
            }
            stats->samples = px * (uint64_t)(sample_end - sample_begin);
        }
        return (int)RT_OK;
    });
}
int rt_accum_finalize_device(const rt_params* p, const double* d_accum, double* d_tiles, void* hip_stream) {
    return guard([&] {
        REQUIRE(p && d_accum && d_tiles, "null argument");
        if (device_count() < 1) throw RtError(RT_ERR_NO_DEVICE, "no HIP device: librtamd has no CPU fallback");
        const RenderPlan pl = make_plan(p);
        DeviceScope dev_scope(p->device);
        finalize_tiles(pl, d_accum, d_tiles, hip_stream);
        return (int)RT_OK;
    });
}

int64_t rt_accum_state_doubles(const rt_params* p) {
    const int64_t n = rt_tiles_owned(p);
    return n < 0 ? n : std::max<int64_t>(1, n) * TILE_PIX * 3;
}
namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) dev_free(p);
    }
};
}  // namespace
int rt_render_accumulate(const rt_scene* s, const rt_camera* cam, const rt_params* p, int32_t sample_begin, int32_t sample_end,
                         double* accum_state, rt_stats* stats) {
    return guard([&] {
        REQUIRE(s && cam && p && accum_state, "null argument");
        if (device_count() < 1) throw RtError(RT_ERR_NO_DEVICE, "no HIP device: librtamd has no CPU fallback");
        const int64_t n = rt_accum_state_doubles(p);
        REQUIRE(n > 0, "bad image size or partition");
        DeviceScope dev_scope(p->device);
        DevBuf acc;
        acc.p = dev_alloc((size_t)n * sizeof(double));
        // the state always goes to the device: a rank that owns no tile (more ranks than tiles) launches nothing, and what comes back must
        // be what went in -- zeros at sample_begin == 0 -- not uninitialised device memory
        if (sample_begin == 0) std::memset(accum_state, 0, (size_t)n * sizeof(double));
        dev_copy_to_device(acc.p, accum_state, (size_t)n * sizeof(double));
        const int rc = rt_render_accumulate_device(s, cam, p, sample_begin, sample_end, (double*)acc.p, nullptr, stats);
        if (rc != RT_OK) return rc;
        dev_copy_to_host(accum_state, acc.p, (size_t)n * sizeof(double));
        return (int)RT_OK;
    });
}
int rt_accum_finalize(const rt_params* p, const double* accum_state, double* out_rgb) {
    return guard([&] {
        REQUIRE(p && accum_state && out_rgb, "null argument");
        REQUIRE(p->world == 1 && p->rank == 0, "rt_accum_finalize stitches a whole frame: rank / world must be 0 / 1");
        if (device_count() < 1) throw RtError(RT_ERR_NO_DEVICE, "no HIP device: librtamd has no CPU fallback");
        const RenderPlan pl = make_plan(p);
        const size_t n = (size_t)std::max<int64_t>(1, pl.tiles_owned) * TILE_PIX * 3, frame_bytes = (size_t)pl.width * pl.height * 3 * sizeof(double);
        DeviceScope dev_scope(p->device);
        DevBuf acc, tiles, frame;
        acc.p = dev_alloc(n * sizeof(double));
        tiles.p = dev_alloc(n * sizeof(double));
        frame.p = dev_alloc(frame_bytes);
        dev_copy_to_device(acc.p, accum_state, n * sizeof(double));
        finalize_tiles(pl, (const double*)acc.p, (double*)tiles.p, nullptr);
        assemble_frame(pl, (const double*)tiles.p, pl.tiles_owned, (double*)frame.p, nullptr);
        dev_copy_to_host(out_rgb, frame.p, frame_bytes);
        return (int)RT_OK;
    });
}

int rt_render_sppm_tiles_device(const rt_scene* s, const rt_camera* cam, const rt_params* p, const rt_sppm_config* cfg, double* d_tiles,
                                void* hip_stream, rt_stats* stats) {
    return guard([&] {
        REQUIRE(s && cam && p && cfg && d_tiles, "null argument");
        REQUIRE(p->spp > 0, "spp must be positive");
        if (!s->committed) throw RtError(RT_ERR_NOT_COMMITTED, "rt_scene_commit has not been called");
        if (device_count() < 1) throw RtError(RT_ERR_NO_DEVICE, "no HIP device: librtamd has no CPU fallback");
        auto t0 = std::chrono::steady_clock::now();
        rt_params q = *p;
        q.integrator = 0;
        RenderPlan pl = make_plan(&q);
        DeviceScope dev_scope(p->device);
        CameraDev cd = make_camera(*cam);
        rt_stats st{};
        render_sppm(*s, cd, pl, *cfg, d_tiles, nullptr, hip_stream, &st, nullptr);
        if (stats) {
            *stats = st;
            stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        return (int)RT_OK;
    });
}

int rt_assemble_frame_device(const rt_params* p, const double* d_gathered, int64_t tiles_per_rank_stride, double* d_frame,
                             void* hip_stream) {
    return guard([&] {
        REQUIRE(d_gathered && d_frame, "null argument");
        RenderPlan pl = make_plan(p);
        assemble_frame(pl, d_gathered, tiles_per_rank_stride, d_frame, hip_stream);
        return (int)RT_OK;
    });
}

static CameraDev camera_from_frame(const rt_camera_frame& f) {
    CameraDev d;
    for (int i = 0; i < 3; i++) {
        d.origin[i] = f.origin[i]; d.llc[i] = f.lower_left_corner[i]; d.horizontal[i] = f.horizontal[i];
        d.vertical[i] = f.vertical[i]; d.u[i] = f.u[i]; d.v[i] = f.v[i]; d.w[i] = f.w[i];
    }
    d.lens_radius = f.lens_radius;
    return d;
}
int rt_camera_frame_from(const rt_camera* cam, rt_camera_frame* out) {
    return guard([&] {
        REQUIRE(cam && out, "null argument");
        const CameraDev d = make_camera(*cam);
        for (int i = 0; i < 3; i++) {
            out->origin[i] = d.origin[i]; out->lower_left_corner[i] = d.llc[i]; out->horizontal[i] = d.horizontal[i];
            out->vertical[i] = d.vertical[i]; out->u[i] = d.u[i]; out->v[i] = d.v[i]; out->w[i] = d.w[i];
        }
        out->lens_radius = d.lens_radius;
        return (int)RT_OK;
    });
}
static int render_host(const rt_scene* s, const CameraDev& cd, const rt_params* p, double* out_rgb, rt_stats* stats);
int rt_render_camera_frame(const rt_scene* s, const rt_camera_frame* frame, const rt_params* p, double* out_rgb, rt_stats* stats) {
    return guard([&] {
        REQUIRE(s && frame && p && out_rgb, "null argument");
        for (const double* v : {frame->origin, frame->lower_left_corner, frame->horizontal, frame->vertical, frame->u, frame->v, frame->w})
            for (int i = 0; i < 3; i++) REQUIRE(std::isfinite(v[i]), "camera frame must be finite");
        REQUIRE(std::isfinite(frame->lens_radius), "camera frame must be finite");
        return render_host(s, camera_from_frame(*frame), p, out_rgb, stats);
    });
}
int rt_render(const rt_scene* s, const rt_camera* cam, const rt_params* p, double* out_rgb, rt_stats* stats) {
    return guard([&] {
        REQUIRE(s && cam && p && out_rgb, "null argument");
        return render_host(s, make_camera(*cam), p, out_rgb, stats);
    });
}
static int render_host(const rt_scene* s, const CameraDev& cd, const rt_params* p, double* out_rgb, rt_stats* stats) {
    {
        if (!s->committed) throw RtError(RT_ERR_NOT_COMMITTED, "rt_scene_commit has not been called");
        if (device_count() < 1) throw RtError(RT_ERR_NO_DEVICE, "no HIP device: librtamd has no CPU fallback");
        auto t0 = std::chrono::steady_clock::now();
        DeviceScope dev_scope(p->device);
        RenderPlan pl = make_plan(p);
        if (stats) std::memset(stats, 0, sizeof(*stats));
        struct Buf {
            void* p = nullptr;
            ~Buf() {
                if (p) dev_free(p);
            }
        } tiles, frame;
        size_t tile_bytes = (size_t)std::max<int64_t>(1, pl.tiles_owned) * TILE_PIX * 3 * sizeof(double);
        size_t frame_bytes = (size_t)pl.width * pl.height * 3 * sizeof(double);
        tiles.p = dev_alloc(tile_bytes);
        frame.p = dev_alloc(frame_bytes);
        rt_stats st{};
        render_tiles(*s, cd, pl, (double*)tiles.p, nullptr, &st);
        if (pl.world == 1) {
            assemble_frame(pl, (const double*)tiles.p, pl.tiles_owned, (double*)frame.p, nullptr);
            dev_copy_to_host(out_rgb, frame.p, frame_bytes);
        } else {
            // only this rank's tiles: others stay 0 in the caller's frame
            std::vector<double> h((size_t)pl.tiles_owned * TILE_PIX * 3);
            dev_copy_to_host(h.data(), tiles.p, h.size() * sizeof(double));
            std::memset(out_rgb, 0, frame_bytes);
            for (int64_t lt = 0; lt < pl.tiles_owned; lt++) {
                int64_t t = lt * pl.world + pl.rank;
                int tx = (int)(t % pl.tiles_x), ty = (int)(t / pl.tiles_x);
                for (int pix = 0; pix < TILE_PIX; pix++) {
                    int x = tx * TILE_W + (pix & 7), y = ty * TILE_H + (pix >> 3);
                    if (x >= pl.width || y >= pl.height) continue;
                    for (int c = 0; c < 3; c++) out_rgb[((size_t)y * pl.width + x) * 3 + c] = h[((size_t)lt * TILE_PIX + pix) * 3 + c];
                }
            }
        }
        if (stats) {
            *stats = st;
            stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            uint64_t px = 0;
            for (int64_t lt = 0; lt < pl.tiles_owned; lt++) {
                int64_t t = lt * pl.world + pl.rank;
                int tx = (int)(t % pl.tiles_x), ty = (int)(t / pl.tiles_x);
                px += (uint64_t)std::min(TILE_W, pl.width - tx * TILE_W) * std::min(TILE_H, pl.height - ty * TILE_H);
            }
            stats->samples = px * (uint64_t)pl.spp;
        }
        return (int)RT_OK;
    }
}

}  // extern "C"

// ---- the frame across the GPUs of one node (camera.rs:74-126 has the fan-out and the stitch inside capture_image) ----
namespace {
struct DevMem {  // device memory that remembers which device it lives on
    void* p = nullptr;
    int device = -1;
    DevMem() = default;
    DevMem(const DevMem&) = delete;
    DevMem& operator=(const DevMem&) = delete;
    size_t size = 0;
    void alloc(int dev, size_t bytes) {
        dev_set_device(dev);
        p = frame_pool_take(dev, bytes);
        device = dev;
        size = bytes;
    }
    ~DevMem() {
        if (!p) return;
        try {
            dev_set_device(device);
            frame_pool_give(device, size, p);
        } catch (...) {
        }
    }
};
struct ExchangeLease {
    Exchange* e = nullptr;
    bool failed = false;  // an exchange call threw: the communicators are destroyed instead of going back into the cache
    ~ExchangeLease() { exchange_close(e, failed); }
};
}  // namespace

// per_rank(rank plan, destination rows on the rank's device, stats) renders one rank's tiles; it runs on its own host thread with
// the rank's device current.
template <class F>
static int render_fanout(const rt_params* p, int n_devices, const int* device_ids, double* out_rgb, rt_stats* stats, F&& per_rank) {
    REQUIRE(p && out_rgb, "null argument");
    REQUIRE(p->world == 1 && p->rank == 0, "rt_render_multi partitions the frame itself: rank / world must be 0 / 1");
    const int visible = device_count();
    if (visible < 1) throw RtError(RT_ERR_NO_DEVICE, "no HIP device: librtamd has no CPU fallback");
    REQUIRE(n_devices >= 0 && n_devices <= 4096, "n_devices out of range");
    const int n = n_devices == 0 ? visible : n_devices;
    std::vector<int> ids((size_t)n);
    for (int i = 0; i < n; i++) {
        ids[(size_t)i] = device_ids ? device_ids[i] : i;
        if (ids[(size_t)i] < 0 || ids[(size_t)i] >= visible)
            throw RtError(RT_ERR_NO_DEVICE, "device ordinal " + std::to_string(ids[(size_t)i]) + " of rank " + std::to_string(i) + " is not visible (" +
                                                std::to_string(visible) + " device(s))");
    }
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(t - t0).count(); };
    const int caller_device = dev_get_device();
    struct Restore {
        int d;
        ~Restore() {
            try {
                dev_set_device(d);
            } catch (...) {
            }
        }
    } restore{caller_device};
    const bool force = tuning().multi_force_rccl != 0;
    const int root = ids[0];
    std::vector<RenderPlan> plans((size_t)n);
    for (int i = 0; i < n; i++) {
        rt_params q = *p;
        q.rank = i;
        q.world = n;
        q.device = -1;
        plans[(size_t)i] = make_plan(&q);
    }
    const int64_t stride = plans[0].tiles_owned;  // rank 0 owns the most tiles: every rank's rows are padded to it
    const size_t row_doubles = (size_t)std::max<int64_t>(1, stride) * TILE_PIX * 3;
    const size_t frame_bytes = (size_t)p->width * p->height * 3 * sizeof(double);
    DevMem gathered, frame;
    gathered.alloc(root, (size_t)n * row_doubles * sizeof(double));
    frame.alloc(root, frame_bytes);
    // a rank on the root's device renders straight into its slot of the gathered buffer; the others (every rank when the
    // exchange is forced) into a row of their own on their device, which then travels
    std::vector<DevMem> rows((size_t)n);
    std::vector<double*> dst((size_t)n);
    std::vector<int> uniq;  // comm rank r = uniq[r]; the root's device first
    auto comm_rank = [&](int dev) {
        for (size_t r = 0; r < uniq.size(); r++)
            if (uniq[r] == dev) return (int)r;
        uniq.push_back(dev);
        return (int)uniq.size() - 1;
    };
    comm_rank(root);
    std::vector<RowMove> moves((size_t)n, RowMove{0, nullptr, 0, nullptr, 0});  // moves[i].count == 0: rank i's rows stay where they are rendered
    size_t n_moves = 0;
    for (int i = 0; i < n; i++) {
        double* slot = (double*)gathered.p + (size_t)i * row_doubles;
        if (ids[(size_t)i] == root && !force) {
            dst[(size_t)i] = slot;
        } else {
            rows[(size_t)i].alloc(ids[(size_t)i], row_doubles * sizeof(double));
            dst[(size_t)i] = (double*)rows[(size_t)i].p;
            const size_t count = (size_t)plans[(size_t)i].tiles_owned * TILE_PIX * 3;
            if (count) {
                moves[(size_t)i] = RowMove{comm_rank(ids[(size_t)i]), (const double*)rows[(size_t)i].p, 0, slot, count};
                n_moves++;
            }
        }
    }
    // the communicators BEFORE the renders start (creating them is the expensive part of a first call and no part of the exchange)
    ExchangeLease lease;
    double comm_init_ms = 0.;
    if (n_moves) {
        const auto tc = std::chrono::steady_clock::now();
        bool created = false;
        lease.e = exchange_open(uniq, &created);
        if (created) comm_init_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc).count();
    }
    std::mutex post_mu;  // one rank at a time drives the communicators (exchange_post)
    std::vector<rt_stats> st((size_t)n);
    std::vector<int> rc((size_t)n, (int)RT_OK);
    std::vector<std::string> msg((size_t)n);
    std::vector<double> posted_ms((size_t)n, 0.);
    auto work = [&](int i) {
        try {
            dev_set_device(ids[(size_t)i]);
            std::memset(&st[(size_t)i], 0, sizeof(rt_stats));
            per_rank(plans[(size_t)i], dst[(size_t)i], &st[(size_t)i]);
            if (moves[(size_t)i].count) {  // this rank's rows leave NOW: the ranks still rendering do not hold them up
                std::lock_guard<std::mutex> g(post_mu);
                try {
                    exchange_post(lease.e, moves[(size_t)i]);
                } catch (...) {
                    lease.failed = true;
                    throw;
                }
                posted_ms[(size_t)i] = ms_since(std::chrono::steady_clock::now());
            }
        } catch (const RtError& e) {
            rc[(size_t)i] = e.code;
            msg[(size_t)i] = e.msg;
        } catch (const std::exception& e) {
            rc[(size_t)i] = RT_ERR_INTERNAL;
            msg[(size_t)i] = e.what();
        } catch (...) {
            rc[(size_t)i] = RT_ERR_INTERNAL;
            msg[(size_t)i] = "unknown error";
        }
    };
    {
        std::vector<std::thread> th;
        for (int i = 1; i < n; i++) th.emplace_back(work, i);
        work(0);  // rank 0 on the calling thread, as the reference's caller blocks in capture_image
        for (auto& t : th) t.join();
    }
    const auto t1 = std::chrono::steady_clock::now();
    // ---- the exchange: every posted row has to arrive in its slot on the root device (also after a failed rank: nothing may still be in
    //      flight when the buffers go back to the pool)
    if (lease.e) {
        try {
            exchange_wait(lease.e);
        } catch (...) {
            lease.failed = true;
            throw;
        }
    }
    for (int i = 0; i < n; i++)
        if (rc[(size_t)i] != RT_OK) throw RtError(rc[(size_t)i], "rank " + std::to_string(i) + " (device " + std::to_string(ids[(size_t)i]) + "): " + msg[(size_t)i]);
    const auto t2 = std::chrono::steady_clock::now();
    dev_set_device(root);
    rt_params whole = *p;
    whole.rank = 0;
    whole.world = n;
    assemble_frame(make_plan(&whole), (const double*)gathered.p, stride, (double*)frame.p, nullptr);
    dev_copy_to_host(out_rgb, frame.p, frame_bytes);
    const auto t3 = std::chrono::steady_clock::now();
    if (stats) {
        for (int i = 0; i < n; i++) {
            stats[i] = st[(size_t)i];
            uint64_t px = 0;
            const RenderPlan& pl = plans[(size_t)i];
            for (int64_t lt = 0; lt < pl.tiles_owned; lt++) {
                const int64_t t = lt * pl.world + pl.rank;
                const int tx = (int)(t % pl.tiles_x), ty = (int)(t / pl.tiles_x);
                px += (uint64_t)std::min(TILE_W, pl.width - tx * TILE_W) * std::min(TILE_H, pl.height - ty * TILE_H);
            }
            stats[i].samples = px * (uint64_t)pl.spp;
            stats[i].posted_ms = posted_ms[(size_t)i];
        }
        stats[0].seconds = std::chrono::duration<double>(t3 - t0).count();
        stats[0].exchange_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();  // the last rank's finish (the join) -> all rows on the root
        stats[0].reserved[2] = (uint64_t)(stats[0].exchange_ms * 1e3);
        stats[0].reserved[3] = (uint64_t)n_moves;
        stats[0].stitch_copy_ms = std::chrono::duration<double, std::milli>(t3 - t2).count();
        stats[0].comm_init_ms = comm_init_ms;
    }
    return (int)RT_OK;
}

extern "C" {

int rt_render_multi(const rt_scene* s, const rt_camera* cam, const rt_params* p, int n_devices, const int* device_ids, double* out_rgb,
                    rt_stats* stats) {
    return guard([&] {
        REQUIRE(s && cam, "null argument");
        if (!s->committed) throw RtError(RT_ERR_NOT_COMMITTED, "rt_scene_commit has not been called");
        const CameraDev cd = make_camera(*cam);
        return render_fanout(p, n_devices, device_ids, out_rgb, stats,
                             [&](const RenderPlan& pl, double* d_rows, rt_stats* st) { render_tiles(*s, cd, pl, d_rows, nullptr, st); });
    });
}
int rt_render_multi_camera_frame(const rt_scene* s, const rt_camera_frame* frame, const rt_params* p, int n_devices, const int* device_ids,
                                 double* out_rgb, rt_stats* stats) {
    return guard([&] {
        REQUIRE(s && frame, "null argument");
        for (const double* v : {frame->origin, frame->lower_left_corner, frame->horizontal, frame->vertical, frame->u, frame->v, frame->w})
            for (int i = 0; i < 3; i++) REQUIRE(std::isfinite(v[i]), "camera frame must be finite");
        REQUIRE(std::isfinite(frame->lens_radius), "camera frame must be finite");
        if (!s->committed) throw RtError(RT_ERR_NOT_COMMITTED, "rt_scene_commit has not been called");
        const CameraDev cd = camera_from_frame(*frame);
        return render_fanout(p, n_devices, device_ids, out_rgb, stats,
                             [&](const RenderPlan& pl, double* d_rows, rt_stats* st) { render_tiles(*s, cd, pl, d_rows, nullptr, st); });
    });
}
int rt_render_sppm_multi(const rt_scene* s, const rt_camera* cam, const rt_params* p, const rt_sppm_config* cfg, int n_devices,
                         const int* device_ids, double* out_rgb, rt_stats* stats) {
    return guard([&] {
        REQUIRE(s && cam && cfg && p, "null argument");
        REQUIRE(p->spp > 0, "spp must be positive");
        if (!s->committed) throw RtError(RT_ERR_NOT_COMMITTED, "rt_scene_commit has not been called");
        const CameraDev cd = make_camera(*cam);
        rt_params q = *p;
        q.integrator = 0;  // as rt_render_sppm_tiles_device: the plan of the final pass; render_sppm switches the integrator
        return render_fanout(&q, n_devices, device_ids, out_rgb, stats, [&](const RenderPlan& pl, double* d_rows, rt_stats* st) {
            render_sppm(*s, cd, pl, *cfg, d_rows, nullptr, nullptr, st, nullptr);
        });
    });
}
int rt_rccl_version(void) { return exchange_library_version(); }

void rt_default_sppm_config(rt_sppm_config* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->iterations = 50;            // photon_mapper.rs:148
    c->photons_per_iter = 500000;  // photon_mapper.rs:149
    c->k_global = 100;             // GLOBAL_INIT_PHOTONS
    c->k_caustic = 50;             // CAUSTIC_INIT_PHOTONS
    c->max_bounces = 4096;
    c->alpha = 0.7;                // ALPHA
}
int rt_render_sppm(const rt_scene* s, const rt_camera* cam, const rt_params* p, const rt_sppm_config* cfg, double* out_rgb, double* stats_out,
                   uint64_t photons_stored[2], rt_stats* stats) {
    return guard([&] {
        REQUIRE(s && cam && p && cfg, "null argument");
        REQUIRE(p->spp == 0 || out_rgb, "null output buffer");
        REQUIRE(p->world == 1 && p->rank == 0, "rt_render_sppm renders the whole frame on one GPU (world must be 1); "
                                               "the tile-partitioned form is rt_render_sppm_tiles_device");
        if (!s->committed) throw RtError(RT_ERR_NOT_COMMITTED, "rt_scene_commit has not been called");
        if (device_count() < 1) throw RtError(RT_ERR_NO_DEVICE, "no HIP device: librtamd has no CPU fallback");
        auto t0 = std::chrono::steady_clock::now();
        DeviceScope dev_scope(p->device);
        rt_params q = *p;
        if (q.spp == 0) q.spp = 1;  // make_plan wants a positive spp; the pre-pass-only case never launches the render
        q.integrator = 0;
        RenderPlan pl = make_plan(&q);
        pl.spp = p->spp;
        CameraDev cd = make_camera(*cam);
        struct Buf {
            void* p = nullptr;
            ~Buf() {
                if (p) dev_free(p);
            }
        } tiles, frame;
        size_t frame_bytes = (size_t)pl.width * pl.height * 3 * sizeof(double);
        tiles.p = dev_alloc((size_t)std::max<int64_t>(1, pl.tiles_owned) * TILE_PIX * 3 * sizeof(double));
        frame.p = dev_alloc(frame_bytes);
        rt_stats st{};
        render_sppm(*s, cd, pl, *cfg, (double*)tiles.p, stats_out, nullptr, &st, photons_stored);
        if (p->spp > 0) {
            assemble_frame(pl, (const double*)tiles.p, pl.tiles_owned, (double*)frame.p, nullptr);
            dev_copy_to_host(out_rgb, frame.p, frame_bytes);
        }
        if (stats) {
            *stats = st;
            stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            stats->samples = (uint64_t)pl.width * pl.height * (uint64_t)p->spp;
        }
        return (int)RT_OK;
    });
}
int rt_tonemap_u8(const double* rgb, size_t n_channels, uint8_t* out) {
    return guard([&] {
        REQUIRE((rgb && out) || n_channels == 0, "null argument");
        for (size_t i = 0; i < n_channels; i++) out[i] = tonemap_channel(rgb[i]);
        return (int)RT_OK;
    });
}
int rt_write_png(const char* path, int width, int height, const uint8_t* rgb) {
    return guard([&] {
        REQUIRE(path, "null path");
        write_png(path, width, height, rgb);
        return (int)RT_OK;
    });
}

int rt_debug_rng_device(uint64_t seed, uint64_t pixel, uint64_t sample, int n, uint64_t* out_host) {
    return guard([&] {
        REQUIRE(n > 0 && out_host, "bad argument");
        if (device_count() < 1) throw RtError(RT_ERR_NO_DEVICE, "no HIP device");
        debug_rng_device(seed, pixel, sample, n, out_host);
        return (int)RT_OK;
    });
}
int rt_debug_rng_host(uint64_t seed, uint64_t pixel, uint64_t sample, int n, uint64_t* out_host) {
    return guard([&] {
        REQUIRE(n > 0 && out_host, "bad argument");
        Rng r;
        r.seed_stream(seed, pixel, sample);
        for (int i = 0; i < n; i++) out_host[i] = r.next_u64();
        return (int)RT_OK;
    });
}
int rt_debug_rng_floats(uint64_t seed, uint64_t pixel, uint64_t sample, int n, double lo, double hi, int on_device, double* out_gen,
                        double* out_range) {
    return guard([&] {
        REQUIRE(n > 0 && out_gen && out_range && lo < hi && std::isfinite(hi - lo), "bad argument");
        if (on_device) {
            if (device_count() < 1) throw RtError(RT_ERR_NO_DEVICE, "no HIP device");
            debug_rng_floats_device(seed, pixel, sample, n, lo, hi, out_gen, out_range);
            return (int)RT_OK;
        }
        Rng r;
        r.seed_stream(seed, pixel, sample);
        for (int i = 0; i < n; i++) out_gen[i] = r.gen_f64();
        r.seed_stream(seed, pixel, sample);
        for (int i = 0; i < n; i++) out_range[i] = (lo == -1. && hi == 1.) ? r.gen_range_pm1() : (lo == 0. && hi == 1.) ? r.gen_range_01() : r.gen_range(lo, hi);
        return (int)RT_OK;
    });
}
int rt_debug_math_device(int op, size_t n, const double* a_host, const double* b_host, double* out_host) {
    return guard([&] {
        REQUIRE(n > 0 && a_host && out_host && (op == 0 || op == 2 || op == 3 || (op == 1 && b_host)), "bad argument");
        if (device_count() < 1) throw RtError(RT_ERR_NO_DEVICE, "no HIP device");
        debug_math_device(op, n, a_host, b_host, out_host);
        return (int)RT_OK;
    });
}
int rt_debug_hit_device(const rt_scene* s, int kernel, size_t n, const double* rays_host, double t_min, double t_max, double* out_host) {
    return guard([&] {
        REQUIRE(s && n > 0 && rays_host && out_host && (kernel == 1 || kernel == 2 || kernel == 3 || kernel == 5 || kernel == 6), "bad argument");
        if (device_count() < 1) throw RtError(RT_ERR_NO_DEVICE, "no HIP device");
        debug_hit_device(*s, kernel, n, rays_host, t_min, t_max, out_host);
        return (int)RT_OK;
    });
}
int rt_debug_schedule(int64_t tiles_owned, int n_waves, int s_begin, int s_end, int sub_spp, int job_units, int* out25) {
    int rounds = 0;
    const int rc = guard([&] {
        REQUIRE(tiles_owned > 0 && n_waves > 0 && s_begin >= 0 && s_end > s_begin && sub_spp >= 1 && sub_spp <= 8 && job_units >= 1 && out25, "bad argument");
        Schedule sch;
        rounds = make_schedule(sch, (int)std::min<int64_t>(tiles_owned, 0x7fffffff), n_waves, s_begin, s_end, sub_spp, job_units);
        for (int i = 0; i < 25; i++) out25[i] = sch.lvl[i / 5][i % 5];
        return (int)RT_OK;
    });
    return rc < 0 ? rc : rounds;
}

}  // extern "C"
