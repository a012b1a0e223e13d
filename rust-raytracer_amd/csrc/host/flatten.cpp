// Flattener: object graph -> linear DFS pre-order program + SoA tables in one blob
// (layout: common/flat.h).  The emission order IS the reference's visit order:
//   BVHNode::hit   box, then left subtree, then right subtree   (bvh.rs:86-102)
//   Vec<..>::hit   items in insertion order                     (hit.rs:57-67)
//   Cube::hit      its 6 sides as a list -- ONE node, the scan is inside the kernel's cube_hit  (cube.rs:64-66)
//   Mesh::hit      its inner BVHNode                            (mesh.rs:201-203)
//   Transform::hit enter object space, inner object, leave      (transform.rs:152-165)
// so the kernel's closest-hit update ("accept when t <= best", later wins ties)
// reproduces the reference's result including its tie rule.
#include <cstring>
#include <map>

#include <algorithm>
#include <cmath>

#include "accel.h"
#include "scene.h"

namespace rtamd {

namespace {

struct Builder {
    rt_scene& s;
    std::vector<uint32_t> meta;
    std::vector<double> boxes, spheres, rects, xforms, vpos, vnrm;
    std::vector<int32_t> sphere_mat, rect_mat;
    std::vector<uint32_t> tris;
    std::vector<double> tripre;  // per triangle: pa, e0 = pb - pa, e1 = pc - pa, pad (10 doubles = 80 B)
    std::vector<MediumDev> media;
    std::vector<double> msph;  // moving spheres: {center0, center1, time0, time1, radius, material} (10 f64)
    std::map<int, uint32_t> msph_of;
    double media_extent = 0.;  // largest |coordinate| of the media's bounding boxes: scatter points inside a medium are ray origins too
    std::map<int, uint32_t> medium_of;
    int medium_depth = 0;
    std::map<int, uint32_t> sphere_of, rect_of, tri_of, xform_of;  // (rect_of: rectangles and cubes share the rect table)
    int n_cubes = 0;
    std::vector<uint32_t> mesh_base;
    uint32_t kinds = 0;
    int xf_depth = 0, depth = 0, max_depth = 0;
    std::map<int, bool> in_xform;  // object id -> emitted inside a Transform
    // accel (kernel 2) item collection: context 0 = world space, context 1+i = object space of instance i
    struct InstCtx {
        std::vector<AccelItem> items;
        std::map<int, size_t> of;  // object id -> item slot (a re-emitted object keeps one slot, latest order)
        uint32_t xform = 0;
        const double* Minv = nullptr;
        const double* M = nullptr;
    };
    std::vector<InstCtx> actx{1};
    std::vector<size_t> ctx_stack{0};
    std::map<int, size_t> ctx_of_xform;  // Transform object id -> its context
    bool accel_ok = true;

    void accel_item(int obj_id, const ObjectRec& o, uint32_t kp, uint32_t node_index, const Box* tight = nullptr) {
        if (medium_depth > 0) return;  // the boundary of a ConstantMedium is not a surface: only the medium's own two queries see it
        InstCtx& c = actx[ctx_stack.back()];
        if (!o.has_box) {
            accel_ok = false;
            return;
        }
        auto it = c.of.find(obj_id);
        if (it == c.of.end()) {
            c.of.emplace(obj_id, c.items.size());
            c.items.push_back(AccelItem{tight ? *tight : o.box, kp, (int32_t)node_index});
        } else {
            c.items[it->second].order = (int32_t)node_index;  // the later visit wins ties (Q5/Q14)
        }
    }

    explicit Builder(rt_scene& sc) : s(sc) {}

    uint32_t node(uint32_t kind, uint32_t payload, uint32_t skip = 0) {
        if (payload >= (1u << (32 - NK_BITS))) throw RtError(RT_ERR_UNSUPPORTED, "scene too large for 28-bit payload index");
        meta.push_back(kind | (payload << NK_BITS));
        meta.push_back(skip);
        kinds |= 1u << kind;
        return (uint32_t)(meta.size() / 2 - 1);
    }

    void emit(int id) {
        const ObjectRec& o = s.objects[id];
        if (xf_depth > 0) in_xform[id] = true;
        depth++;
        if (depth > max_depth) max_depth = depth;
        switch (o.type) {
            case OBJ_SPHERE: {
                auto it = sphere_of.find(id);
                if (it == sphere_of.end()) {
                    it = sphere_of.emplace(id, (uint32_t)sphere_mat.size()).first;
                    spheres.insert(spheres.end(), {o.c[0], o.c[1], o.c[2], o.r});
                    sphere_mat.push_back(o.material);
                }
                uint32_t n = node(NK_SPHERE, it->second);
                accel_item(id, o, NK_SPHERE | (it->second << NK_BITS), n);
                break;
            }
            case OBJ_MOVING_SPHERE: {  // D9
                auto it = msph_of.find(id);
                if (it == msph_of.end()) {
                    it = msph_of.emplace(id, (uint32_t)(msph.size() / 10)).first;
                    msph.insert(msph.end(), {o.c[0], o.c[1], o.c[2], o.c1[0], o.c1[1], o.c1[2], o.time0, o.time1, o.r, (double)o.material});
                }
                uint32_t n = node(NK_MSPHERE, it->second);
                accel_item(id, o, NK_MSPHERE | (it->second << NK_BITS), n);
                break;
            }
            case OBJ_RECT: {
                auto it = rect_of.find(id);
                if (it == rect_of.end()) {
                    it = rect_of.emplace(id, (uint32_t)rect_mat.size()).first;
                    rects.insert(rects.end(), {o.a0, o.b0, o.a1, o.b1, o.k, 0.0});
                    rect_mat.push_back(o.material);
                }
                uint32_t kind = o.axis == 0 ? NK_RECT_YZ : (o.axis == 1 ? NK_RECT_XZ : NK_RECT_XY);
                uint32_t n = node(kind, it->second);
                accel_item(id, o, kind | (it->second << NK_BITS), n);
                break;
            }
            case OBJ_TRIANGLE: {
                auto it = tri_of.find(id);
                if (it == tri_of.end()) {
                    it = tri_of.emplace(id, (uint32_t)(tris.size() / 4)).first;
                    uint32_t base = mesh_base[o.mesh];
                    tris.insert(tris.end(), {base + o.ia, base + o.ib, base + o.ic, (uint32_t)o.material});
                    // what Triangle::hit recomputes per call (mesh.rs:69): edge = [pb - pa, pc - pa]; same f64 subtractions
                    const double* P = s.meshes[o.mesh]->pos.data();
                    const double *pa = P + 3 * o.ia, *pb = P + 3 * o.ib, *pc = P + 3 * o.ic;
                    tripre.insert(tripre.end(), {pa[0], pa[1], pa[2], pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2], pc[0] - pa[0], pc[1] - pa[1],
                                                 pc[2] - pa[2], 0.0});
                }
                uint32_t n = node(NK_TRI, it->second);
                {
                    // accel: the TIGHT vertex box (the reference's +-0.1 object-space padding, mesh.rs:33-42, only serves its
                    // own BVH; any box containing the triangle prunes correctly)
                    const double* P = s.meshes[o.mesh]->pos.data();
                    const double *pa = P + 3 * o.ia, *pb = P + 3 * o.ib, *pc = P + 3 * o.ic;
                    Box tb;
                    for (int a = 0; a < 3; a++) {
                        tb.mn[a] = std::fmin(std::fmin(pa[a], pb[a]), pc[a]);
                        tb.mx[a] = std::fmax(std::fmax(pa[a], pb[a]), pc[a]);
                    }
                    accel_item(id, o, NK_TRI | (it->second << NK_BITS), n, &tb);
                }
                break;
            }
            case OBJ_CUBE: {  // one record in the rect table + one node (flat.h NK_CUBE); o.box is exactly (box_min, box_max), cube.rs:67-69
                auto it = rect_of.find(id);
                if (it == rect_of.end()) {
                    it = rect_of.emplace(id, (uint32_t)rect_mat.size()).first;
                    rects.insert(rects.end(), {o.box.mn[0], o.box.mn[1], o.box.mn[2], o.box.mx[0], o.box.mx[1], o.box.mx[2]});
                    rect_mat.push_back(o.material);
                    n_cubes++;
                }
                if (it->second >= (1u << (32 - NK_BITS - 3))) throw RtError(RT_ERR_UNSUPPORTED, "scene too large for the cube payload (record index * 8 + side)");
                uint32_t n = node(NK_CUBE, it->second * 8u);
                accel_item(id, o, NK_CUBE | ((it->second * 8u) << NK_BITS), n);
                break;
            }
            case OBJ_LIST:
            case OBJ_MESH:
                for (int c : o.children) emit(c);
                break;
            case OBJ_BVH: {
                uint32_t bi = (uint32_t)(boxes.size() / 6);
                boxes.insert(boxes.end(), {o.box.mn[0], o.box.mn[1], o.box.mn[2], o.box.mx[0], o.box.mx[1], o.box.mx[2]});
                uint32_t n = node(NK_BOX, bi);
                if (bi >= 0x7FFFFFFFu) throw RtError(RT_ERR_UNSUPPORTED, "too many BVH nodes");
                emit(o.children[0]);
                emit(o.children[1]);
                meta[2 * n + 1] = (uint32_t)(meta.size() / 2);
                break;
            }
            case OBJ_TRANSFORM: {
                if (xf_depth >= 1) throw RtError(RT_ERR_UNSUPPORTED, "nested Transform (depth > 1) is not supported by the device traversal yet");
                auto it = xform_of.find(id);
                if (it == xform_of.end()) {
                    it = xform_of.emplace(id, (uint32_t)(xforms.size() / 32)).first;
                    xforms.insert(xforms.end(), o.Minv, o.Minv + 16);
                    xforms.insert(xforms.end(), o.M, o.M + 16);
                }
                uint32_t n = node(NK_XFORM_BEGIN, it->second);
                // accel: the Transform is one item of the enclosing space; its subtree gets its own object-space BVH
                // (a Transform emitted twice -- BVHNode::new's 1-object leaf, Q14 -- re-enters its own context, so its
                // items take the later visit's indices exactly as a re-emitted primitive does)
                if (medium_depth > 0) {  // a Transform inside a medium's boundary: reference-order program only, no accel context
                    xf_depth++;
                    emit(o.children[0]);
                    xf_depth--;
                    node(NK_XFORM_END, it->second);
                    meta[2 * n + 1] = (uint32_t)(meta.size() / 2);
                    break;
                }
                auto ci = ctx_of_xform.find(id);
                if (ci == ctx_of_xform.end()) {
                    actx.emplace_back();
                    actx.back().xform = it->second;
                    actx.back().Minv = o.Minv;
                    actx.back().M = o.M;
                    ci = ctx_of_xform.emplace(id, actx.size() - 1).first;
                }
                uint32_t inst_index = (uint32_t)(ci->second - 1);
                accel_item(id, o, NK_INSTANCE | (inst_index << NK_BITS), n);
                ctx_stack.push_back(ci->second);
                xf_depth++;
                emit(o.children[0]);
                xf_depth--;
                ctx_stack.pop_back();
                node(NK_XFORM_END, it->second);
                meta[2 * n + 1] = (uint32_t)(meta.size() / 2);
                break;
            }
            case OBJ_MEDIUM: {
                // ConstantMedium::hit consumes a random number INSIDE hit (medium.rs:37-38), so what the path draws depends on
                // the order in which the reference visits objects.  Kernel 1 walks the reference-order program; the accel kernel
                // (kernel 2) reproduces the visit order for the media only (traverse2_media in kernels.hip), which needs every
                // medium in world space (not under a Transform).
                if (medium_depth > 0) throw RtError(RT_ERR_UNSUPPORTED, "a ConstantMedium inside the boundary of a ConstantMedium is not supported");
                // one MediumDev per VISIT: BVHNode::new duplicates a single object into both children (Q14), so the reference visits
                // such a medium twice, and each visit may draw
                const uint32_t mi = (uint32_t)media.size();
                media.push_back(MediumDev{-1. / o.density, o.material, 0, 0, 0, 0, 0});
                if (xf_depth > 0) accel_ok = false;  // a medium under a Transform: reference order only
                if (o.has_box)
                    for (int a = 0; a < 3; a++) media_extent = std::fmax(media_extent, std::fmax(std::fabs(o.box.mn[a]), std::fabs(o.box.mx[a])));
                else
                    accel_ok = false;
                medium_depth++;
                uint32_t beg = node(NK_MEDIUM_BEGIN, mi);
                emit(o.children[0]);
                uint32_t mid = node(NK_MEDIUM_MID, mi);
                emit(o.children[0]);
                uint32_t end = node(NK_MEDIUM_END, mi);
                meta[2 * mid + 1] = end;
                media[mi].n_begin = beg;
                media[mi].n_mid = mid;
                media[mi].n_end = end;
                {
                    const ObjectRec& bo = s.objects[o.children[0]];
                    auto si = sphere_of.find(o.children[0]);
                    if (bo.type == OBJ_SPHERE && xf_depth == 0 && si != sphere_of.end() && mid == beg + 2 && end == mid + 2)
                        media[mi].boundary_kp = NK_SPHERE | (si->second << NK_BITS);
                }
                medium_depth--;
                break;
            }
            default:
                throw RtError(RT_ERR_ARG, "unknown object type in flatten");
        }
        depth--;
    }
};

template <class T>
uint32_t append(std::vector<char>& blob, const std::vector<T>& v) {
    size_t off = (blob.size() + 15) & ~size_t(15);
    blob.resize(off);
    if (!v.empty()) {
        blob.resize(off + v.size() * sizeof(T));
        std::memcpy(blob.data() + off, v.data(), v.size() * sizeof(T));
    }
    if (blob.size() > 0xFFFFFFF0u) throw RtError(RT_ERR_UNSUPPORTED, "flattened scene exceeds 4 GiB");
    return (uint32_t)off;
}

}  // namespace

void flatten(rt_scene& s) {
    if (s.root < 0) throw RtError(RT_ERR_ARG, "scene has no root (rt_world_new / rt_scene_set_root)");
    Builder b(s);
    // global vertex table: meshes concatenated
    uint32_t nv = 0;
    for (auto& m : s.meshes) {
        b.mesh_base.push_back(nv);
        nv += (uint32_t)(m->pos.size() / 3);
        b.vpos.insert(b.vpos.end(), m->pos.begin(), m->pos.end());
        b.vnrm.insert(b.vnrm.end(), m->nrm.begin(), m->nrm.end());
    }
    b.emit(s.root);

    // World::new's lights -> {kind, payload} pairs addressing the sphere / rect tables
    std::vector<uint32_t> lights;
    for (int lid : s.lights) {
        const ObjectRec& o = s.objects[lid];
        if (b.in_xform.count(lid)) throw RtError(RT_ERR_UNSUPPORTED, "a light under a Transform is not supported");
        if (o.type == OBJ_SPHERE) {
            auto it = b.sphere_of.find(lid);
            if (it == b.sphere_of.end()) {  // a light that is not part of the hitable list: still addressable
                it = b.sphere_of.emplace(lid, (uint32_t)b.sphere_mat.size()).first;
                b.spheres.insert(b.spheres.end(), {o.c[0], o.c[1], o.c[2], o.r});
                b.sphere_mat.push_back(o.material);
            }
            lights.push_back(NK_SPHERE);
            lights.push_back(it->second);
        } else {
            auto it = b.rect_of.find(lid);
            if (it == b.rect_of.end()) {
                it = b.rect_of.emplace(lid, (uint32_t)b.rect_mat.size()).first;
                b.rects.insert(b.rects.end(), {o.a0, o.b0, o.a1, o.b1, o.k, 0.0});
                b.rect_mat.push_back(o.material);
            }
            lights.push_back(NK_RECT_XZ);
            lights.push_back(it->second);
        }
    }

    std::vector<MatDev> mats;
    for (auto& m : s.materials) {
        MatDev md{m.type, m.tex, m.param, 0., 0., 0.};
        if (m.type == MAT_DIELECTRIC) {
            md.inv_ir = 1.0 / m.param;
            const double qf = (1. - md.inv_ir) / (1. + md.inv_ir), qb = (1. - m.param) / (1. + m.param);
            md.r0_front = qf * qf;
            md.r0_back = qb * qb;
        }
        mats.push_back(md);
    }
    std::vector<TexDev> texs;
    std::vector<uint8_t> texels;
    for (auto& t : s.textures) {
        TexDev d{};
        d.type = t.type;
        d.t0 = t.t0;
        d.t1 = t.t1;
        d.w = t.w;
        d.h = t.h;
        if (t.type == TEX_NOISE) texels.resize((texels.size() + 7) & ~size_t(7));  // its f64 gradient vectors are read as doubles
        d.texel_off = (uint32_t)texels.size();
        for (int i = 0; i < 3; i++) d.color[i] = t.color[i];
        texels.insert(texels.end(), t.rgb.begin(), t.rgb.end());
        texs.push_back(d);
    }

    FlatScene& f = s.flat;
    f.blob.clear();
    FlatView v{};
    // ---- accel (kernel 2) ----
    AccelBuild ab;
    ab.ok = b.accel_ok;
    uint32_t root2 = REF_DONE, max_inst_nodes = 0, inst_depth = 0, n_world_items = 0, world_depth = 0, stack_inline = 0;
    std::vector<char> compact_cand;  // per instance: only triangles with f32 vertices (kernels 5 / 6 can defer it)
    std::vector<double> inst_oo;  // per instance: bound of |object-space ray origin|
    double origin_limit = 0.;
    // An instance's item in the enclosing space carries the Transform's own bounding box (the box of the 8 transformed corners
    // of the child's box, transform.rs:104-150): loose for a rotated mesh.  For culling, the union of the transformed boxes of the
    // instance's ITEMS is as valid (affine images of the items lie inside it; the f64 rounding of M * corner is orders of
    // magnitude below the pad added at build time) and tighter: fewer rays enter the object-space BVH for nothing.
    for (size_t i = 1; b.accel_ok && i < b.actx.size(); i++) {
        const auto& c = b.actx[i];
        if (c.items.empty() || !c.M) continue;
        Box tb;
        for (int a = 0; a < 3; a++) { tb.mn[a] = INFINITY; tb.mx[a] = -INFINITY; }
        for (const auto& it : c.items)
            for (int corner = 0; corner < 8; corner++) {
                const double x = (corner & 1) ? it.box.mx[0] : it.box.mn[0], y = (corner & 2) ? it.box.mx[1] : it.box.mn[1],
                             z = (corner & 4) ? it.box.mx[2] : it.box.mn[2];
                for (int a = 0; a < 3; a++) {
                    const double w = c.M[4 * a] * x + c.M[4 * a + 1] * y + c.M[4 * a + 2] * z + c.M[4 * a + 3];
                    tb.mn[a] = std::fmin(tb.mn[a], w);
                    tb.mx[a] = std::fmax(tb.mx[a], w);
                }
            }
        bool finite = true;
        for (int a = 0; a < 3; a++) finite = finite && std::isfinite(tb.mn[a]) && std::isfinite(tb.mx[a]);
        if (!finite) continue;
        const uint32_t want = NK_INSTANCE | ((uint32_t)(i - 1) << NK_BITS);
        for (auto& pc : b.actx)
            for (auto& it : pc.items)
                if (it.kp == want)
                    for (int a = 0; a < 3; a++) {  // never larger than the Transform's own box; a margin of 2^-40 of its size for the rounding
                        const double m = std::ldexp(std::fabs(tb.mx[a]) + std::fabs(tb.mn[a]), -40);
                        it.box.mn[a] = std::fmax(it.box.mn[a], tb.mn[a] - m);
                        it.box.mx[a] = std::fmin(it.box.mx[a], tb.mx[a] + m);
                    }
    }
    if (ab.ok && !b.actx[0].items.empty()) {
        // E_w: largest |coordinate| of the world items; boxes are padded so that rounding a ray origin with
        // max-abs coordinate <= 64*E_w to f32 (relative error 2^-24) can never make the f32 slab test cull a box
        // the exact test keeps: 4 * 2^-24 * |o|max covers of = fl32(o) and c = fl32(of * iv)  (derivation above box32
        // in csrc/device/kernels.hip, which also needs every coordinate below 2^36 in magnitude); the pad is THREE times that
        // (12 * 2^-24 * |o|max) since round 3 so that box32w, the test of the LDS-resident node table, needs no widening factor
        // on the far side (its proof, above box32w, uses the extra margin against the relative error of the slab parameters)
        double ew = b.media_extent;
        for (auto& it : b.actx[0].items)
            for (int a = 0; a < 3; a++) ew = std::fmax(ew, std::fmax(std::fabs(it.box.mn[a]), std::fabs(it.box.mx[a])));
        if (!(ew > 0.) || !std::isfinite(ew)) {
            ab.ok = false;
        } else {
            origin_limit = 64. * ew;
            const double pad_w = 3. * std::ldexp(origin_limit, -22);  // 12 * 2^-24 * |o|max
            if (!(origin_limit < 68719476736.)) ab.ok = false;  // 2^36
            root2 = accel_build_bvh(ab, b.actx[0].items, pad_w, 0);
            const int depth_tlas = ab.max_depth;
            world_depth = (uint32_t)depth_tlas;
            ab.inst.assign(2 * (b.actx.size() - 1), 0u);
            // Which instances can kernels 5 / 6 defer?  Those that hold nothing but triangles with f32 vertices (flat.h "Compact
            // instance data").  The others are entered in the lane (item kind NK_INSTANCE_INLINE); their BVHs are built FIRST so that
            // their leaves sit right behind the world's in the item array: kernels 5 / 6 stage that prefix in LDS.
            auto is_f32v = [](double x) { return (double)(float)x == x; };
            compact_cand.assign(b.actx.size() - 1, 1);
            inst_oo.assign(b.actx.size() - 1, 0.);
            for (size_t i = 1; ab.ok && i < b.actx.size(); i++) {
                auto& c = b.actx[i];
                // object-space origin bound: |M^-1 o| <= sum_b |Minv[a][b]| * |o|max + |Minv[a][3]|
                double oo = 0.;
                for (int a = 0; a < 3; a++)
                    oo = std::fmax(oo, (std::fabs(c.Minv[4 * a]) + std::fabs(c.Minv[4 * a + 1]) + std::fabs(c.Minv[4 * a + 2])) * origin_limit +
                                           std::fabs(c.Minv[4 * a + 3]));
                for (auto& it : c.items)  // hit points inside the instance also serve as origins of secondary rays (in world space only)
                    for (int a = 0; a < 3; a++) oo = std::fmax(oo, std::fmax(std::fabs(it.box.mn[a]), std::fabs(it.box.mx[a])));
                if (!(oo < 68719476736.)) { ab.ok = false; break; }
                inst_oo[i - 1] = oo;
                for (auto& it : c.items) {
                    if ((it.kp & NK_MASK) != NK_TRI) { compact_cand[i - 1] = 0; break; }
                    const uint32_t t = it.kp >> NK_BITS;
                    for (int c3 = 0; c3 < 3 && compact_cand[i - 1]; c3++)
                        for (int a = 0; a < 3; a++)
                            if (!is_f32v(b.vpos[3 * (size_t)b.tris[4 * (size_t)t + c3] + a])) compact_cand[i - 1] = 0;
                    if (!compact_cand[i - 1]) break;
                }
                if (!compact_cand[i - 1]) {
                    const uint32_t want = NK_INSTANCE | ((uint32_t)(i - 1) << NK_BITS);
                    for (auto& it : b.actx[0].items)
                        if (it.kp == want) it.kp = NK_INSTANCE_INLINE | ((uint32_t)(i - 1) << NK_BITS);
                }
            }
            // (the world BVH was built above with the items' kinds as they were: its leaf items are patched below, after the build)
            for (int pass = 0; pass < 2; pass++) {
                for (size_t i = 1; ab.ok && i < b.actx.size(); i++) {
                    if ((compact_cand[i - 1] != 0) != (pass == 1)) continue;  // pass 0: inline instances, pass 1: deferrable ones
                    auto& c = b.actx[i];
                    const size_t nodes_before = ab.nodes.size();
                    const int depth_before = ab.max_depth;
                    ab.max_depth = depth_tlas + 1;
                    uint32_t r = accel_build_bvh(ab, c.items, 3. * std::ldexp(inst_oo[i - 1], -22), depth_tlas + 1);
                    if (pass == 1) {
                        max_inst_nodes = std::max<uint32_t>(max_inst_nodes, (uint32_t)(ab.nodes.size() - nodes_before));
                        inst_depth = std::max<uint32_t>(inst_depth, (uint32_t)std::max(1, ab.max_depth - depth_tlas + 1));
                    }
                    ab.max_depth = std::max(ab.max_depth, depth_before);
                    ab.inst[2 * (i - 1)] = c.xform;
                    ab.inst[2 * (i - 1) + 1] = r;
                }
                if (pass == 0) {
                    n_world_items = (uint32_t)(ab.items.size() / 2);
                    stack_inline = (uint32_t)(ab.max_depth + 2);
                }
            }
            for (size_t j = 0; j < ab.items.size() / 2; j++) {  // the world leaves' instance items, as classified
                const uint32_t kp = ab.items[2 * j];
                if ((kp & NK_MASK) == NK_INSTANCE && !compact_cand[kp >> NK_BITS]) ab.items[2 * j] = NK_INSTANCE_INLINE | (kp & ~NK_MASK);
            }
        }
    } else {
        ab.ok = false;
    }
    if (ab.max_depth + 2 > ACCEL_MAX_STACK) ab.ok = false;
    std::vector<double> tripre2;  // triangle records in ACCEL ITEM order: a leaf's 1..4 triangles are contiguous
    if (ab.ok) {
        // (b) relabel the Node2 array by depth (all BVHs interleaved): the first K nodes are the K shallowest, which is
        // what the kernels cache in LDS when the whole scene does not fit
        const size_t nn = ab.nodes.size();
        std::vector<int> depth(nn, 0);
        std::vector<uint32_t> stack;
        auto walk = [&](uint32_t root) {
            if ((root >> REF_TAG_SHIFT) != 0u) return;
            depth[root] = 0;
            stack.assign(1, root);
            while (!stack.empty()) {
                uint32_t n = stack.back();
                stack.pop_back();
                for (int k = 0; k < 2; k++) {
                    uint32_t c = ab.nodes[n].child[k];
                    if ((c >> REF_TAG_SHIFT) == 0u) {
                        depth[c] = depth[n] + 1;
                        stack.push_back(c);
                    }
                }
            }
        };
        walk(root2);
        for (size_t i = 0; i + 1 < ab.inst.size(); i += 2) walk(ab.inst[i + 1]);
        std::vector<uint32_t> order(nn);
        for (size_t i = 0; i < nn; i++) order[i] = (uint32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t c) { return depth[a] < depth[c]; });
        std::vector<uint32_t> new_of(nn);
        for (size_t i = 0; i < nn; i++) new_of[order[i]] = (uint32_t)i;
        auto remap = [&](uint32_t r) { return ((r >> REF_TAG_SHIFT) == 0u) ? new_of[r] : r; };
        std::vector<Node2> sorted(nn);
        for (size_t i = 0; i < nn; i++) {
            Node2 nd = ab.nodes[order[i]];
            nd.child[0] = remap(nd.child[0]);
            nd.child[1] = remap(nd.child[1]);
            sorted[i] = nd;
        }
        ab.nodes.swap(sorted);
        root2 = remap(root2);
        for (size_t i = 0; i + 1 < ab.inst.size(); i += 2) ab.inst[i + 1] = remap(ab.inst[i + 1]);
        // (a) per item slot, the triangle's {pa, e0, e1} record (zeros for non-triangles)
        const size_t n_items = ab.items.size() / 2;
        if (!b.tripre.empty()) {
            tripre2.assign(n_items * 10, 0.0);
            for (size_t j = 0; j < n_items; j++) {
                uint32_t kp = ab.items[2 * j];
                if ((kp & NK_MASK) == NK_TRI) {
                    const double* src = &b.tripre[(size_t)(kp >> NK_BITS) * 10];
                    std::copy(src, src + 10, &tripre2[j * 10]);
                }
            }
        }
    }

    // ---- kernel 5: compact object-space data (common/flat.h "Compact instance data") ----
    std::vector<NodeQ> n2q;
    std::vector<Tri32> tri32;
    std::vector<QGrid> qgrid;
    bool coop_data = ab.ok && !ab.inst.empty();
    {
        bool any = false;
        for (char c : compact_cand) any = any || c != 0;
        coop_data = coop_data && any;
    }
    uint32_t world_top = 0;
    if (ab.ok) {
        // world-space nodes after the depth sort: new_of of the first n_world_nodes old indices -- recomputed from the roots
        std::vector<uint32_t> st;
        if ((root2 >> REF_TAG_SHIFT) == 0u) st.push_back(root2);
        while (!st.empty()) {
            const uint32_t n = st.back();
            st.pop_back();
            world_top = std::max(world_top, n + 1);
            for (int k = 0; k < 2; k++)
                if ((ab.nodes[n].child[k] >> REF_TAG_SHIFT) == 0u) st.push_back(ab.nodes[n].child[k]);
        }
    }
    if (coop_data) {
        n2q.assign(ab.nodes.size(), NodeQ{});
        tri32.assign(ab.items.size() / 2, Tri32{});
        qgrid.assign(ab.inst.size() / 2, QGrid{});
        auto is_f32 = [](double x) { return (double)(float)x == x; };
        for (size_t i = 0; coop_data && i < ab.inst.size() / 2; i++) {
            if (!compact_cand[i]) continue;  // an inline instance: no compact copy
            const uint32_t root = ab.inst[2 * i + 1];
            if ((root >> REF_TAG_SHIFT) != 0u) { coop_data = false; break; }
            const Node2& rn = ab.nodes[root];
            double mn[3], mx[3];
            const float* lo[3] = {rn.lo_x, rn.lo_y, rn.lo_z};
            const float* hi[3] = {rn.hi_x, rn.hi_y, rn.hi_z};
            for (int a = 0; a < 3; a++) {
                mn[a] = std::fmin((double)lo[a][0], (double)lo[a][1]);
                mx[a] = std::fmax((double)hi[a][0], (double)hi[a][1]);
            }
            // A flat instance (a single triangle, a planar mesh) has an extent of twice the pad along one axis: its grid scale would be
            // astronomical and the ray's grid coordinates (origin bound x scale) would leave the range box32 is proven for (2^35).  A
            // thin axis is therefore widened around its centre until  (origin bound) * QGRID_MAX / extent <= 2^33  -- the boxes of
            // that axis are then only rounded outward onto a coarser grid: still conservative.
            {
                double mab = 0.;
                for (int a = 0; a < 3; a++) mab = std::fmax(mab, std::fmax(std::fabs(mn[a]), std::fabs(mx[a])));
                const double emin = QGRID_MAX * (inst_oo[i] + mab) / 8589934592.;
                for (int a = 0; a < 3; a++)
                    if (mx[a] - mn[a] < emin) {
                        const double c = 0.5 * (mn[a] + mx[a]);
                        mn[a] = c - 0.5 * emin;
                        mx[a] = c + 0.5 * emin;
                    }
            }
            // grid: g(x) = (x - mn) * k + shift, shift = P + 1, (mx - mn) * k = QGRID_MAX - 2 P - 2
            double k0 = 0., mabs = 0.;
            for (int a = 0; a < 3; a++) {
                const double ext = mx[a] - mn[a];
                k0 = std::fmax(k0, ext > 0. ? QGRID_MAX / ext : 1.);
                mabs = std::fmax(mabs, std::fabs(mn[a]));
            }
            const double og = (inst_oo[i] + mabs) * k0 + 65536.;  // bound of |o_g|
            if (!(og < 34359738368.)) { coop_data = false; break; }  // 2^35: box32 needs coordinates below 2^36
            const double P = std::ceil(std::ldexp(og, -22)) + 2.;
            if (!(P <= 4096.)) { coop_data = false; break; }
            QGrid g{};
            for (int a = 0; a < 3; a++) {
                const double ext = mx[a] - mn[a];
                g.mn[a] = mn[a];
                g.k[a] = ext > 0. ? (QGRID_MAX - 2. * P - 2.) / ext : 1.;
            }
            g.shift = P + 1.;
            qgrid[i] = g;
            std::vector<uint32_t> st{root};
            while (coop_data && !st.empty()) {
                const uint32_t n = st.back();
                st.pop_back();
                const Node2& nd = ab.nodes[n];
                const float* l[3] = {nd.lo_x, nd.lo_y, nd.lo_z};
                const float* h[3] = {nd.hi_x, nd.hi_y, nd.hi_z};
                uint32_t ql[3], qh[3];
                for (int a = 0; a < 3; a++) {
                    uint32_t w_lo = 0, w_hi = 0;
                    for (int c = 0; c < 2; c++) {
                        const double a_lo = std::floor(((double)l[a][c] - g.mn[a]) * g.k[a]) + 1.;           // g(lo) - P, rounded down
                        const double a_hi = std::ceil(((double)h[a][c] - g.mn[a]) * g.k[a]) + 2. * P + 1.;   // g(hi) + P, rounded up
                        if (!(a_lo >= 0. && a_hi <= QGRID_MAX && a_lo <= a_hi)) coop_data = false;
                        w_lo |= (uint32_t)a_lo << (16 * c);
                        w_hi |= (uint32_t)a_hi << (16 * c);
                    }
                    ql[a] = w_lo;
                    qh[a] = w_hi;
                }
                NodeQ q{ql[0], ql[1], ql[2], qh[0], qh[1], qh[2], {nd.child[0], nd.child[1]}};
                n2q[n] = q;
                for (int c = 0; c < 2; c++) {
                    const uint32_t r = nd.child[c];
                    if ((r >> REF_TAG_SHIFT) == 0u) {
                        st.push_back(r);
                    } else if ((r >> REF_TAG_SHIFT) == 1u) {
                        const uint32_t first = r & REF_LEAF_FIRST_MASK, cnt = ((r >> REF_LEAF_COUNT_SHIFT) & 7u) + 1u;
                        for (uint32_t j = first; j < first + cnt; j++) {
                            const uint32_t kp = ab.items[2 * j];
                            if ((kp & NK_MASK) != NK_TRI) { coop_data = false; break; }
                            const uint32_t t = kp >> NK_BITS;
                            Tri32 tr{};
                            const double* pre = &b.tripre[(size_t)t * 10];
                            double v[3][3];
                            for (int c3 = 0; c3 < 3; c3++)
                                for (int a = 0; a < 3; a++) {
                                    v[c3][a] = b.vpos[3 * (size_t)b.tris[4 * (size_t)t + c3] + a];
                                    if (!is_f32(v[c3][a])) coop_data = false;
                                }
                            for (int a = 0; a < 3; a++) {
                                tr.pa[a] = (float)v[0][a];
                                tr.pb[a] = (float)v[1][a];
                                tr.pc[a] = (float)v[2][a];
                                // the lane forms pb - pa, pc - pa in f64: must be the hoisted record's values, bit for bit
                                if (v[0][a] != pre[a] || v[1][a] - v[0][a] != pre[3 + a] || v[2][a] - v[0][a] != pre[6 + a]) coop_data = false;
                            }
                            tr.order = ab.items[2 * j + 1];
                            tr.kp = kp;
                            {   // the largest edge component, rounded up to f32 (tri_miss32's error scale)
                                double me = 0.;
                                for (int a = 0; a < 3; a++) me = std::fmax(me, std::fmax(std::fabs(pre[3 + a]), std::fabs(pre[6 + a])));
                                float mf = (float)me;
                                if ((double)mf < me) mf = std::nextafterf(mf, INFINITY);
                                tr.me = mf;
                            }
                            tri32[j] = tr;
                        }
                    }
                }
            }
        }
    }
    if (!coop_data) {
        n2q.clear();
        tri32.clear();
        qgrid.clear();
    }

    // hot part (read once per visited node): candidates for LDS residency
    v.off_meta = append(f.blob, b.meta);
    v.off_boxes = append(f.blob, b.boxes);
    v.off_spheres = append(f.blob, b.spheres);
    v.stage2_begin = v.off_spheres;
    v.off_rects = append(f.blob, b.rects);
    v.off_tris = append(f.blob, b.tris);
    v.off_tripre = append(f.blob, b.tripre);
    v.off_xforms = append(f.blob, b.xforms);
    f.blob.resize((f.blob.size() + 15) & ~size_t(15));
    v.stage_bytes = (uint32_t)f.blob.size();
    if (!ab.ok) {
        ab.nodes.clear();
        ab.items.clear();
        ab.inst.clear();
    }
    // kernel 2 stages [spheres .. xforms | items2 | inst2 | tripre2 | n2] into LDS when it fits
    v.off_items2 = append(f.blob, ab.items);
    v.off_inst2 = append(f.blob, ab.inst);
    v.off_tripre2 = append(f.blob, tripre2);  // triangle records in item order (big meshes); staged with the rest when everything fits
    v.off_n2 = append(f.blob, ab.nodes);
    f.blob.resize((f.blob.size() + 15) & ~size_t(15));
    v.stage2_end = (uint32_t)f.blob.size();
    v.off_n2q = append(f.blob, n2q);  // kernel 5's compact object-space data: never staged as a whole
    v.off_tri32 = append(f.blob, tri32);
    v.off_qgrid = append(f.blob, qgrid);
    v.n_nodes2 = (uint32_t)ab.nodes.size();
    v.accel_ok = ab.ok ? 1u : 0u;
    v.root2 = root2;
    v.stack2 = (uint32_t)(ab.max_depth + 2);
    v.n_inst2 = (uint32_t)(ab.inst.size() / 2);
    v.max_inst_nodes2 = max_inst_nodes;
    v.inst_depth2 = inst_depth;
    v.n_world_items2 = n_world_items;
    v.stack2_inline = stack_inline;
    v.n_inline2 = 0;
    for (char c : compact_cand) v.n_inline2 += c ? 0u : 1u;
    v.coop_data_ok = coop_data ? 1u : 0u;
    v.world_top2 = world_top;
    v.world_depth2 = world_depth;
    v.origin_limit2 = origin_limit;
    // cold part (read once per path segment, by the winning leaf only): always global
    v.off_sphere_mat = append(f.blob, b.sphere_mat);
    v.off_rect_mat = append(f.blob, b.rect_mat);
    v.off_mats = append(f.blob, mats);
    v.off_texs = append(f.blob, texs);
    v.off_media = append(f.blob, b.media);
    v.n_media = (uint32_t)b.media.size();
    v.has_noise = 0;
    for (auto& tx : s.textures) v.has_noise |= (tx.type == TEX_NOISE) ? 1u : 0u;
    v.off_msph = append(f.blob, b.msph);
    v.n_msph = (uint32_t)(b.msph.size() / 10);
    f.msph_t0_max = -1e300;
    f.msph_t1_min = 1e300;
    for (size_t i = 0; i + 10 <= b.msph.size(); i += 10) {
        f.msph_t0_max = std::max(f.msph_t0_max, b.msph[i + 6]);
        f.msph_t1_min = std::min(f.msph_t1_min, b.msph[i + 7]);
    }
    {   // where the reference-order program lies in the blob, for the one walk that must read it from global memory whatever the
        // kernel staged (tie_resolve, kernels.hip); the last word is the record's own offset
        std::vector<uint32_t> tv = {v.off_meta, v.off_boxes, v.off_spheres, v.off_rects, v.off_tripre, v.off_xforms, 0u, (uint32_t)(b.meta.size() / 2)};
        f.blob.resize((f.blob.size() + 15) & ~size_t(15));
        tv[6] = (uint32_t)f.blob.size();
        v.off_tie_view = append(f.blob, tv);
    }
    v.off_lights = append(f.blob, lights);
    v.n_lights = (uint32_t)(lights.size() / 2);
    v.off_vpos = append(f.blob, b.vpos);  // kept for introspection; the kernels read tripre instead
    v.off_vnrm = append(f.blob, b.vnrm);
    v.off_texels = append(f.blob, texels);
    f.blob.resize((f.blob.size() + 15) & ~size_t(15));
    v.total_bytes = (uint32_t)f.blob.size();
    v.n_nodes = (uint32_t)(b.meta.size() / 2);
    v.kinds_mask = b.kinds;
    v.base = nullptr;
    f.view = v;

    rt_scene_info& in = f.info;
    in.n_nodes = (int32_t)v.n_nodes;
    in.n_boxes = (int32_t)(b.boxes.size() / 6);
    in.n_spheres = (int32_t)b.sphere_mat.size();
    in.n_rects = (int32_t)b.rect_mat.size() - b.n_cubes;
    in.n_cubes = b.n_cubes;
    in.n_tris = (int32_t)(b.tris.size() / 4);
    in.n_xforms = (int32_t)(b.xforms.size() / 32);
    in.n_materials = (int32_t)mats.size();
    in.n_textures = (int32_t)texs.size();
    in.n_verts = (int32_t)nv;
    in.max_depth = b.max_depth;
    in.committed = 1;
    in.bytes = f.blob.size();
    in.accel_ok = ab.ok ? 1 : 0;
    in.accel_nodes = (int32_t)ab.nodes.size();
    in.accel_items = (int32_t)(ab.items.size() / 2);
    in.accel_instances = (int32_t)(ab.inst.size() / 2);
    in.accel_stack = (int32_t)v.stack2;
    in.accel_compact = (int32_t)v.coop_data_ok;
    s.committed = true;
}

}  // namespace rtamd
