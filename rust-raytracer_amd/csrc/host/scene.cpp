// Scene-graph builders: one per reference constructor.  Bounding boxes are
// computed once at construction (objects are immutable afterwards).
#include "scene.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>

#include "../common/rng.h"

namespace rtamd {

static const double PI = 3.14159265358979323846264338327950288;

void check_obj(const rt_scene& s, int o) {
    if (o < 0 || o >= (int)s.objects.size()) throw RtError(RT_ERR_ARG, "unknown object id " + std::to_string(o));
}
void check_mat(const rt_scene& s, int m) {
    if (m < 0 || m >= (int)s.materials.size()) throw RtError(RT_ERR_ARG, "unknown material id " + std::to_string(m));
}
static void check_tex(const rt_scene& s, int t) {
    if (t < 0 || t >= (int)s.textures.size()) throw RtError(RT_ERR_ARG, "unknown texture id " + std::to_string(t));
}

int add_texture_constant(rt_scene& s, const double c[3]) {
    TextureRec t;
    t.type = TEX_CONSTANT;
    for (int i = 0; i < 3; i++) t.color[i] = c[i];
    s.textures.push_back(std::move(t));
    return (int)s.textures.size() - 1;
}
// material.rs:49 CheckerTexture(pub ConstantTexture, pub ConstantTexture): both children are constants by type
int add_texture_checker(rt_scene& s, int t0, int t1) {
    check_tex(s, t0);
    check_tex(s, t1);
    if (s.textures[t0].type != TEX_CONSTANT || s.textures[t1].type != TEX_CONSTANT)
        throw RtError(RT_ERR_ARG, "CheckerTexture takes two ConstantTextures (material.rs:49)");
    TextureRec t;
    t.type = TEX_CHECKER;
    t.t0 = t0;
    t.t1 = t1;
    s.textures.push_back(std::move(t));
    return (int)s.textures.size() - 1;
}
int add_texture_image(rt_scene& s, int w, int h, const uint8_t* rgb) {
    if (w <= 0 || h <= 0 || !rgb) throw RtError(RT_ERR_ARG, "ImageTexture needs a non-empty image");
    TextureRec t;
    t.type = TEX_IMAGE;
    t.w = w;
    t.h = h;
    t.rgb.assign(rgb, rgb + (size_t)w * h * 3);
    s.textures.push_back(std::move(t));
    return (int)s.textures.size() - 1;
}
// D9 (book 2, no reference code): noise_texture(scale) = marble over Perlin noise.  The tables come from the stream
// (seed, RT_PERLIN_STREAM_KEY, 0): 256 x unit_vector(random(-1, 1)), then perm_x, perm_y, perm_z (identity permuted from the top with
// random_int(0, i) = (next_u32 * (i + 1)) >> 32).  Layout of the table blob: 768 f64, then 768 bytes.
int add_texture_noise(rt_scene& s, double scale, uint64_t seed) {
    if (!std::isfinite(scale)) throw RtError(RT_ERR_ARG, "noise scale must be finite");
    TextureRec t;
    t.type = TEX_NOISE;
    t.color[0] = scale;
    Rng rng;
    rng.seed_stream(seed, RT_PERLIN_STREAM_KEY, 0);
    std::vector<double> vec(768);
    for (int i = 0; i < 256; i++) {
        const double x = rng.gen_range(-1., 1.), y = rng.gen_range(-1., 1.), z = rng.gen_range(-1., 1.);
        const double len = std::sqrt(x * x + y * y + z * z);  // Vec3::unit, vec3.rs:85-90
        if (len == 0.) throw RtError(RT_ERR_UNIT_ZERO, "unitizing zero vector (noise table)");
        vec[3 * i] = x / len;
        vec[3 * i + 1] = y / len;
        vec[3 * i + 2] = z / len;
    }
    std::vector<uint8_t> perm(768);
    for (int a = 0; a < 3; a++) {
        uint8_t* p = perm.data() + 256 * a;
        for (int i = 0; i < 256; i++) p[i] = (uint8_t)i;
        for (int i = 255; i > 0; i--) {
            const int target = (int)(((uint64_t)rng.next_u32() * (uint64_t)(i + 1)) >> 32);
            std::swap(p[i], p[target]);
        }
    }
    t.rgb.resize(768 * sizeof(double) + 768);
    std::memcpy(t.rgb.data(), vec.data(), 768 * sizeof(double));
    std::memcpy(t.rgb.data() + 768 * sizeof(double), perm.data(), 768);
    s.textures.push_back(std::move(t));
    return (int)s.textures.size() - 1;
}
int add_material(rt_scene& s, int type, int tex, double param) {
    check_tex(s, tex);
    MaterialRec m;
    m.type = type;
    m.tex = tex;
    m.param = param;
    s.materials.push_back(m);
    return (int)s.materials.size() - 1;
}

// Geometry must be finite: the reference would sort NaN box corners with an inconsistent comparator (bvh.rs:34-44) and
// feed NaN slabs to every ray; here it is an argument error at the boundary.
static void require_finite(const double* v, size_t n, const char* what) {
    for (size_t i = 0; i < n; i++)
        if (!std::isfinite(v[i])) throw RtError(RT_ERR_ARG, std::string(what) + " must be finite");
}
static int push(rt_scene& s, ObjectRec&& o) {
    s.objects.push_back(std::move(o));
    return (int)s.objects.size() - 1;
}

// sphere.rs:56-61
int add_sphere(rt_scene& s, const double c[3], double r, int mat) {
    check_mat(s, mat);
    require_finite(c, 3, "sphere center");
    require_finite(&r, 1, "sphere radius");
    ObjectRec o;
    o.type = OBJ_SPHERE;
    o.material = mat;
    for (int i = 0; i < 3; i++) o.c[i] = c[i];
    o.r = r;
    o.has_box = true;
    for (int i = 0; i < 3; i++) {
        o.box.mn[i] = c[i] - r;
        o.box.mx[i] = c[i] + r;
    }
    return push(s, std::move(o));
}
// D9 (book 2, no reference code): moving_sphere -- the centre moves linearly from c0 at time0 to c1 at time1; box = the union of the boxes at both ends
int add_moving_sphere(rt_scene& s, const double c0[3], const double c1[3], double time0, double time1, double r, int mat) {
    check_mat(s, mat);
    require_finite(c0, 3, "sphere center");
    require_finite(c1, 3, "sphere center");
    require_finite(&r, 1, "sphere radius");
    if (!std::isfinite(time0) || !std::isfinite(time1) || !(time1 > time0)) throw RtError(RT_ERR_ARG, "a moving sphere needs time1 > time0");
    ObjectRec o;
    o.type = OBJ_MOVING_SPHERE;
    o.material = mat;
    for (int i = 0; i < 3; i++) { o.c[i] = c0[i]; o.c1[i] = c1[i]; }
    o.time0 = time0;
    o.time1 = time1;
    o.r = r;
    o.has_box = true;
    for (int i = 0; i < 3; i++) {
        o.box.mn[i] = std::fmin(c0[i] - r, c1[i] - r);
        o.box.mx[i] = std::fmax(c0[i] + r, c1[i] + r);
    }
    return push(s, std::move(o));
}
// rectangle.rs:35-41,73-79,110-116 : +-1e-4 thick box around the plane
int add_rect(rt_scene& s, int axis, double a0, double b0, double a1, double b1, double k, int mat) {
    check_mat(s, mat);
    if (axis < 0 || axis > 2) throw RtError(RT_ERR_ARG, "rect axis must be 0..2");
    {
        const double v[5] = {a0, b0, a1, b1, k};
        require_finite(v, 5, "rectangle coordinates");
    }
    ObjectRec o;
    o.type = OBJ_RECT;
    o.material = mat;
    o.axis = axis;
    o.a0 = a0; o.b0 = b0; o.a1 = a1; o.b1 = b1; o.k = k;
    const double BIAS = 0.0001;
    o.has_box = true;
    if (axis == 2) {
        o.box = Box{{a0, b0, k - BIAS}, {a1, b1, k + BIAS}};
    } else if (axis == 1) {
        o.box = Box{{a0, k - BIAS, b0}, {a1, k + BIAS, b1}};
    } else {
        o.box = Box{{k - BIAS, a0, b0}, {k + BIAS, a1, b1}};
    }
    return push(s, std::move(o));
}
// cube.rs:16-61 : sides in the reference's order XY(min.z) XY(max.z) XZ(min.y) XZ(max.y) YZ(min.x) YZ(max.x)
int add_cube(rt_scene& s, const double mn[3], const double mx[3], int mat) {
    check_mat(s, mat);
    ObjectRec c;
    c.type = OBJ_CUBE;
    c.material = mat;
    c.children.push_back(add_rect(s, 2, mn[0], mn[1], mx[0], mx[1], mn[2], mat));
    c.children.push_back(add_rect(s, 2, mn[0], mn[1], mx[0], mx[1], mx[2], mat));
    c.children.push_back(add_rect(s, 1, mn[0], mn[2], mx[0], mx[2], mn[1], mat));
    c.children.push_back(add_rect(s, 1, mn[0], mn[2], mx[0], mx[2], mx[1], mat));
    c.children.push_back(add_rect(s, 0, mn[1], mn[2], mx[1], mx[2], mn[0], mat));
    c.children.push_back(add_rect(s, 0, mn[1], mn[2], mx[1], mx[2], mx[0], mat));
    c.has_box = true;  // cube.rs:67-69: exactly (box_min, box_max)
    for (int i = 0; i < 3; i++) {
        c.box.mn[i] = mn[i];
        c.box.mx[i] = mx[i];
    }
    return push(s, std::move(c));
}

bool bounding_box(const rt_scene& s, int o, Box& out) {
    check_obj(s, o);
    if (!s.objects[o].has_box) return false;
    out = s.objects[o].box;
    return true;
}

// aabb.rs:33-45
static Box surrounding(const Box& a, const Box& b) {
    Box r;
    for (int i = 0; i < 3; i++) {
        r.mn[i] = std::fmin(a.mn[i], b.mn[i]);
        r.mx[i] = std::fmax(a.mx[i], b.mx[i]);
    }
    return r;
}

// hit.rs:69-92 : None for an empty list or any box-less item
int add_list(rt_scene& s, int n, const int* objs) {
    ObjectRec l;
    l.type = OBJ_LIST;
    bool ok = n > 0;
    Box acc{};
    for (int i = 0; i < n; i++) {
        check_obj(s, objs[i]);
        l.children.push_back(objs[i]);
        Box b;
        if (ok && bounding_box(s, objs[i], b)) acc = (i == 0) ? b : surrounding(acc, b);
        else ok = false;
    }
    l.has_box = ok;
    if (ok) l.box = acc;
    return push(s, std::move(l));
}

// ConstantMedium::new(d, boundary, phase_function), medium.rs:16-22 ; bounding_box = the boundary's (medium.rs:54-56)
int add_medium(rt_scene& s, double density, int boundary, int phase_material) {
    check_obj(s, boundary);
    check_mat(s, phase_material);
    require_finite(&density, 1, "medium density");
    if (!(density > 0.)) throw RtError(RT_ERR_ARG, "medium density must be positive");
    const int bt = s.objects[boundary].type;
    if (bt == OBJ_MEDIUM) throw RtError(RT_ERR_UNSUPPORTED, "a ConstantMedium as the boundary of a ConstantMedium is not supported");
    ObjectRec m;
    m.type = OBJ_MEDIUM;
    m.material = phase_material;
    m.density = density;
    m.children = {boundary};
    Box b;
    m.has_box = bounding_box(s, boundary, b);
    if (m.has_box) m.box = b;
    return push(s, std::move(m));
}

// BVHNode::construct, bvh.rs:47-58
int add_bvh_node(rt_scene& s, int left, int right) {
    Box bl, br;
    if (!bounding_box(s, left, bl) || !bounding_box(s, right, br))
        throw RtError(RT_ERR_NO_BBOX, "No bounding box in bvh_node constructor.");
    ObjectRec n;
    n.type = OBJ_BVH;
    n.children = {left, right};
    n.has_box = true;
    n.box = surrounding(bl, br);
    return push(s, std::move(n));
}

// BVHNode::new, bvh.rs:60-83.  One axis draw per call at entry (parent, then the
// left subtree, then the right); sort_by(box_compare) is a stable sort in which
// only "Less" (strict <) is observable; n==1 duplicates the object (Q14).
static int bvh_new_rec(rt_scene& s, std::vector<int> objs, Rng& rng) {
    int axis = (int)rng.gen_below3();
    auto less = [&s, axis](int a, int b) {
        Box ba, bb;
        if (!bounding_box(s, a, ba) || !bounding_box(s, b, bb))
            throw RtError(RT_ERR_NO_BBOX, "No bounding box in bvh_node constructor.");
        return ba.mn[axis] < bb.mn[axis];
    };
    size_t n = objs.size();
    if (n == 1) return add_bvh_node(s, objs[0], objs[0]);
    if (n == 2) {
        if (less(objs[0], objs[1])) return add_bvh_node(s, objs[0], objs[1]);
        return add_bvh_node(s, objs[1], objs[0]);
    }
    std::stable_sort(objs.begin(), objs.end(), less);
    size_t mid = n / 2;
    std::vector<int> lo(objs.begin(), objs.begin() + mid), hi(objs.begin() + mid, objs.end());
    int l = bvh_new_rec(s, std::move(lo), rng);
    int r = bvh_new_rec(s, std::move(hi), rng);
    return add_bvh_node(s, l, r);
}
int add_bvh_build(rt_scene& s, std::vector<int> objs, uint64_t bvh_seed) {
    if (objs.empty()) throw RtError(RT_ERR_ARG, "BVHNode::new on an empty list");
    for (int o : objs) check_obj(s, o);
    Rng rng;
    rng.seed_stream(bvh_seed, RT_BVH_STREAM_KEY, 0);
    return bvh_new_rec(s, std::move(objs), rng);
}

// area-weighted smooth vertex normals for meshes that ship none (bun315.obj);
// an extension: the reference indexes normals unconditionally and would panic (mesh.rs:62).
void synthesize_normals(int n_vert, const double* pos, int n_tri, const uint32_t* idx, std::vector<double>& out) {
    out.assign((size_t)n_vert * 3, 0.0);
    for (int t = 0; t < n_tri; t++) {
        uint32_t a = idx[3 * t], b = idx[3 * t + 1], c = idx[3 * t + 2];
        double e0[3], e1[3];
        for (int i = 0; i < 3; i++) {
            e0[i] = pos[3 * b + i] - pos[3 * a + i];
            e1[i] = pos[3 * c + i] - pos[3 * a + i];
        }
        double n[3] = {e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2], e0[0] * e1[1] - e0[1] * e1[0]};
        for (uint32_t v : {a, b, c})
            for (int i = 0; i < 3; i++) out[3 * v + i] += n[i];
    }
    for (int v = 0; v < n_vert; v++) {
        double* n = &out[3 * v];
        double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (l == 0.) {
            n[0] = 0.; n[1] = 1.; n[2] = 0.;
        } else {
            n[0] /= l; n[1] /= l; n[2] /= l;
        }
    }
}

// The vertex arrays a mesh's triangles share (Arc<Vec<Vec3>> positions / normals, mesh.rs:12-13): returns a mesh id
int add_mesh_data(rt_scene& s, int n_vert, const double* pos, const double* nrm) {
    if (n_vert <= 0 || !pos) throw RtError(RT_ERR_ARG, "empty vertex array");
    if (!nrm) throw RtError(RT_ERR_NO_NORMALS, "mesh has no vertex normals (the reference indexes normals[a] unconditionally, mesh.rs:62)");
    require_finite(pos, (size_t)n_vert * 3, "mesh positions");
    require_finite(nrm, (size_t)n_vert * 3, "mesh normals");
    auto md = std::make_unique<MeshData>();
    md->pos.assign(pos, pos + (size_t)n_vert * 3);
    md->nrm.assign(nrm, nrm + (size_t)n_vert * 3);
    s.meshes.push_back(std::move(md));
    return (int)s.meshes.size() - 1;
}
// Triangle::new (mesh.rs:19-53) on a mesh id: the host keeps its own BVHNode tree over the triangles (rt_object_bvh_node)
int add_triangle(rt_scene& s, int mesh, uint32_t a, uint32_t b, uint32_t c, int mat) {
    check_mat(s, mat);
    if (mesh < 0 || mesh >= (int)s.meshes.size()) throw RtError(RT_ERR_ARG, "unknown mesh id");
    const size_t nv = s.meshes[mesh]->pos.size() / 3;
    if (a >= nv || b >= nv || c >= nv) throw RtError(RT_ERR_ARG, "triangle index out of range");
    ObjectRec o;
    o.type = OBJ_TRIANGLE;
    o.material = mat;
    o.mesh = mesh;
    o.ia = a; o.ib = b; o.ic = c;
    const double* P = s.meshes[mesh]->pos.data();
    const double *pa = P + 3 * a, *pb = P + 3 * b, *pc = P + 3 * c;
    o.has_box = true;
    for (int i = 0; i < 3; i++) {  // mesh.rs:33-42 : +-0.1 in object space (Q9)
        o.box.mx[i] = std::fmax(std::fmax(pa[i], pb[i]), pc[i]) + 0.1;
        o.box.mn[i] = std::fmin(std::fmin(pa[i], pb[i]), pc[i]) - 0.1;
    }
    return push(s, std::move(o));
}

// Mesh::load_obj's construction half (mesh.rs:160-198) + Triangle::new (mesh.rs:19-53)
int add_mesh(rt_scene& s, int n_vert, const double* pos, const double* nrm, int n_tri, const uint32_t* idx, int mat,
             bool synth_normals, uint64_t bvh_seed) {
    check_mat(s, mat);
    if (n_vert <= 0 || n_tri <= 0 || !pos || !idx) throw RtError(RT_ERR_ARG, "empty mesh");
    require_finite(pos, (size_t)n_vert * 3, "mesh positions");
    if (nrm) require_finite(nrm, (size_t)n_vert * 3, "mesh normals");
    for (int i = 0; i < 3 * n_tri; i++)
        if (idx[i] >= (uint32_t)n_vert) throw RtError(RT_ERR_ARG, "triangle index out of range");
    auto md = std::make_unique<MeshData>();
    md->pos.assign(pos, pos + (size_t)n_vert * 3);
    if (nrm) {
        md->nrm.assign(nrm, nrm + (size_t)n_vert * 3);
    } else if (synth_normals) {
        synthesize_normals(n_vert, pos, n_tri, idx, md->nrm);
    } else {
        throw RtError(RT_ERR_NO_NORMALS, "mesh has no vertex normals (the reference indexes normals[a] unconditionally, mesh.rs:62)");
    }
    int mesh_id = (int)s.meshes.size();
    const double* P = md->pos.data();
    s.meshes.push_back(std::move(md));
    std::vector<int> tris;
    tris.reserve(n_tri);
    for (int t = 0; t < n_tri; t++) {
        ObjectRec o;
        o.type = OBJ_TRIANGLE;
        o.material = mat;
        o.mesh = mesh_id;
        o.ia = idx[3 * t]; o.ib = idx[3 * t + 1]; o.ic = idx[3 * t + 2];
        const double *pa = P + 3 * o.ia, *pb = P + 3 * o.ib, *pc = P + 3 * o.ic;
        o.has_box = true;
        for (int i = 0; i < 3; i++) {  // mesh.rs:33-42 : +-0.1 in object space (Q9)
            o.box.mx[i] = std::fmax(std::fmax(pa[i], pb[i]), pc[i]) + 0.1;
            o.box.mn[i] = std::fmin(std::fmin(pa[i], pb[i]), pc[i]) - 0.1;
        }
        tris.push_back(push(s, std::move(o)));
    }
    int bvh = add_bvh_build(s, std::move(tris), bvh_seed);
    ObjectRec m;
    m.type = OBJ_MESH;
    m.material = mat;
    m.mesh = mesh_id;
    m.children = {bvh};
    m.has_box = s.objects[bvh].has_box;
    m.box = s.objects[bvh].box;
    return push(s, std::move(m));
}

// ---- 4x4 helpers standing in for nalgebra::Matrix4<f64> (row-major) -------
static void mat_mul(const double* a, const double* b, double* c) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double acc = a[i * 4 + 0] * b[0 * 4 + j];
            for (int k = 1; k < 4; k++) acc = acc + a[i * 4 + k] * b[k * 4 + j];
            c[i * 4 + j] = acc;
        }
}
// cofactor (GLU-style) inverse, as nalgebra's 4x4 try_inverse specialisation; column-major scratch
static bool mat_inverse(const double* a, double* out) {
    double m[16], inv[16];
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) m[c * 4 + r] = a[r * 4 + c];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.) return false;
    double inv_det = 1.0 / det;
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) out[r * 4 + c] = inv[c * 4 + r] * inv_det;
    return true;
}
static void xf_point(const double* t, const double* p, double* o) {  // vec3.rs:174-178
    for (int i = 0; i < 3; i++) o[i] = t[i * 4 + 0] * p[0] + t[i * 4 + 1] * p[1] + t[i * 4 + 2] * p[2] + t[i * 4 + 3] * 1.;
}

static int finish_transform(rt_scene& s, ObjectRec&& t, int obj, const double* inverse_trans);
// Transform::new, transform.rs:17-148 : M = T*S*Rx*Ry*Rz ; box = 8 transformed corners
int add_transform(rt_scene& s, const double rot_deg[3], const double scale[3], const double translate[3], int obj) {
    check_obj(s, obj);
    require_finite(rot_deg, 3, "transform rotation");
    require_finite(scale, 3, "transform scale");
    require_finite(translate, 3, "transform translation");
    double rx = rot_deg[0] * PI / 180., ry = rot_deg[1] * PI / 180., rz = rot_deg[2] * PI / 180.;
    const double T[16] = {1., 0., 0., translate[0], 0., 1., 0., translate[1], 0., 0., 1., translate[2], 0., 0., 0., 1.};
    const double S[16] = {scale[0], 0., 0., 0., 0., scale[1], 0., 0., 0., 0., scale[2], 0., 0., 0., 0., 1.};
    const double RX[16] = {1., 0., 0., 0., 0., std::cos(rx), -std::sin(rx), 0., 0., std::sin(rx), std::cos(rx), 0., 0., 0., 0., 1.};
    const double RY[16] = {std::cos(ry), 0., std::sin(ry), 0., 0., 1., 0., 0., -std::sin(ry), 0., std::cos(ry), 0., 0., 0., 0., 1.};
    const double RZ[16] = {std::cos(rz), -std::sin(rz), 0., 0., std::sin(rz), std::cos(rz), 0., 0., 0., 0., 1., 0., 0., 0., 0., 1.};
    double a[16], b[16];
    ObjectRec t;
    t.type = OBJ_TRANSFORM;
    t.children = {obj};
    mat_mul(T, S, a);
    mat_mul(a, RX, b);
    mat_mul(b, RY, a);
    mat_mul(a, RZ, t.M);
    return finish_transform(s, std::move(t), obj, nullptr);
}
// Transform as the reference STORES it (transform.rs:9-14: obj, trans, inverse_trans): the composed matrix itself
int add_transform_matrix(rt_scene& s, const double trans[16], const double* inverse_trans, int obj) {
    check_obj(s, obj);
    require_finite(trans, 16, "transform matrix");
    if (inverse_trans) require_finite(inverse_trans, 16, "inverse transform matrix");
    ObjectRec t;
    t.type = OBJ_TRANSFORM;
    t.children = {obj};
    for (int i = 0; i < 16; i++) t.M[i] = trans[i];
    return finish_transform(s, std::move(t), obj, inverse_trans);
}
// box = the 8 transformed corners (transform.rs:110-136); inverse = try_inverse (transform.rs:138-146) unless the host hands its own
static int finish_transform(rt_scene& s, ObjectRec&& t, int obj, const double* inverse_trans) {
    Box bb;
    if (bounding_box(s, obj, bb)) {
        const double INF = std::numeric_limits<double>::infinity();
        double mn[3] = {INF, INF, INF}, mx[3] = {-INF, -INF, -INF};
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    double fi = i, fj = j, fk = k;
                    double corner[3] = {fi * bb.mx[0] + (1. - fi) * bb.mn[0], fj * bb.mx[1] + (1. - fj) * bb.mn[1],
                                        fk * bb.mx[2] + (1. - fk) * bb.mn[2]};
                    double w[3];
                    xf_point(t.M, corner, w);
                    for (int c = 0; c < 3; c++) {
                        mn[c] = std::fmin(mn[c], w[c]);
                        mx[c] = std::fmax(mx[c], w[c]);
                    }
                }
        t.has_box = true;
        for (int c = 0; c < 3; c++) {
            t.box.mn[c] = mn[c];
            t.box.mx[c] = mx[c];
        }
    }
    if (inverse_trans) {
        for (int i = 0; i < 16; i++) t.Minv[i] = inverse_trans[i];
    } else if (!mat_inverse(t.M, t.Minv)) {
        throw RtError(RT_ERR_SINGULAR, "Invalid transform matrix");
    }
    return push(s, std::move(t));
}

// Camera::new, camera.rs:24-55
static void v_unit(const double* a, double* o) {
    double l = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (l == 0.) throw RtError(RT_ERR_UNIT_ZERO, "unitizing zero vector (camera basis)");
    for (int i = 0; i < 3; i++) o[i] = a[i] / l;
}
static void v_cross(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
CameraDev make_camera(const rt_camera& c) {
    CameraDev d;
    double theta = c.vfov * PI / 180.;
    double h = std::tan(theta / 2.);
    double viewport_height = 2.0 * h;
    double viewport_width = c.aspect * viewport_height;
    double diff[3], cr[3];
    for (int i = 0; i < 3; i++) diff[i] = c.look_from[i] - c.look_at[i];
    v_unit(diff, d.w);
    v_cross(c.vup, d.w, cr);
    v_unit(cr, d.u);
    v_cross(d.w, d.u, d.v);
    double fh = c.focus_dist * viewport_width, fv = c.focus_dist * viewport_height;
    for (int i = 0; i < 3; i++) {
        d.origin[i] = c.look_from[i];
        d.horizontal[i] = d.u[i] * fh;
        d.vertical[i] = d.v[i] * fv;
        d.llc[i] = d.origin[i] - d.horizontal[i] / 2. - d.vertical[i] / 2. - d.w[i] * c.focus_dist;
    }
    d.lens_radius = c.aperture / 2.;
    return d;
}

}  // namespace rtamd
