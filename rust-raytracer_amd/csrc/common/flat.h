// Flattened ("linear") scene shared by the host flattener and the HIP kernels.
//
// The reference's Arc<dyn Hitable> tree (objects/bvh.rs:29-33, objects/hit.rs:56,
// objects/transform.rs:9-14 ...) becomes ONE program of nodes in depth-first
// pre-order -- exactly the order the reference's recursion visits them (left child
// before right, list items in insertion order) -- so a node's first child is the
// next node and a failed box test jumps to `skip`.  No stack, no pointers.
//
// Everything lives in one blob so it can be staged into LDS with one copy:
//   hot  [meta | boxes | spheres | rects | tris | tripre | xforms | n2 | items2 | inst2]
//        kernel 1 stages [meta..xforms], kernel 2 stages [spheres..inst2] into LDS when it fits
//   cold [sphere_mat | rect_mat | mats | texs | vpos | vnrm | texels] <- read once per segment, stays global
// each section 16-byte aligned.  f64 payloads are the reference's own values.
#pragma once
#include <stdint.h>

namespace rtamd {

enum NodeKind : uint32_t {
    NK_BOX = 0,          // AABB test (BVHNode::hit's bounding_box.hit, bvh.rs:88); payload = box index; skip = node after subtree
    NK_SPHERE = 1,       // Sphere::hit; payload = sphere index
    NK_RECT_YZ = 2,      // rect with constant axis 0 (YZRectangle); payload = rect index
    NK_RECT_XZ = 3,      // constant axis 1 (XZRectangle)
    NK_RECT_XY = 4,      // constant axis 2 (XYRectangle)
    NK_TRI = 5,          // Triangle::hit; payload = triangle index
    NK_XFORM_BEGIN = 6,  // Transform::hit entry: ray -> object space (transform.rs:153-156); payload = xform index
    NK_XFORM_END = 7,    // leave the Transform: restore the world ray
    // 8 = NK_INSTANCE (accel items only, below)
    // ConstantMedium::hit (medium.rs:25-53) = two closest-hit queries on the boundary, then one random draw.  The boundary's
    // subtree is emitted TWICE between three brackets; payload = medium index:
    NK_MEDIUM_BEGIN = 9,  // save the outer best hit; query A: boundary.hit(r, -inf, +inf)
    NK_MEDIUM_MID = 10,   // A missed -> jump to END (skip link); else query B: boundary.hit(r, t_A + 0.0001, +inf)
    NK_MEDIUM_END = 11,   // restore the outer state; clip [t_A, t_B] to [t_min, best], draw, accept t = t_A + dist / |dir| or nothing
    // 12 = NK_INSTANCE_INLINE (accel items only, below)
    // Cube::hit (cube.rs:64-66) = the list scan of hit.rs:56-67 over its six sides, as ONE node and ONE accel item: the record sits in
    // the rect table (same 48 bytes: (min.x, min.y) (min.z, max.x) (max.y, max.z); rect_mat holds its material) and the sides are
    // tested in Cube::new's order (cube.rs:17-54) with the shrinking closest_so_far.  payload = 8 * record index + side, side = 0
    // in the program and in accel items; a Hit's kp carries the winning side (0 XY z=min, 1 XY z=max, 2 XZ y=min, 3 XZ y=max,
    // 4 YZ x=min, 5 YZ x=max).  The six sides are contiguous in the reference's visit order, so one program index per cube
    // resolves exact ties against other objects exactly as six would; ties among the sides are resolved inside the scan.
    NK_CUBE = 13,
    // D9 (book-2 extension, no reference code): moving_sphere -- Sphere::hit around center0 + (center1 - center0) (ray.time - time0) / (time1 - time0);
    // payload = index into the moving-sphere table (cold part: 10 f64 per record {center0, center1, time0, time1, radius, material})
    NK_MSPHERE = 14
};
static const uint32_t NK_BITS = 4;
static const uint32_t NK_MASK = 15;

// per node: meta[2*n] = kind | payload << 4 ; meta[2*n+1] = skip
struct MediumDev {  // objects/medium.rs:9-13
    double neg_inv_density;  // -1 / d
    int32_t mat;             // phase function (an Isotropic material)
    int32_t pad;
    // where the medium sits in the reference-order program (the accel kernel walks the two copies of the boundary's subtree itself:
    // [n_begin + 1, n_mid) and [n_mid + 1, n_end)); n_end is also the node index of the medium's hit (tie rule)
    uint32_t n_begin, n_mid, n_end;
    // kind | payload << 4 of the boundary when it is ONE world-space sphere (0 otherwise): the accel kernel then answers the two boundary
    // queries with two sphere tests instead of two walks over the medium's part of the program (the same Sphere::hit calls, medium.rs:26-27)
    uint32_t boundary_kp;
};
struct MatDev {   // material.rs:88-212 (+ :213-231, the commented-out Isotropic)
    int32_t type;  // 0 Lambertian, 1 Metal, 2 Dielectric, 3 DiffuseLight, 4 Isotropic
    int32_t tex;   // albedo / emit texture
    double param;  // Metal.fuzz | Dielectric.ir
    // Dielectric only, computed on the host with the reference's operations (the same IEEE results, once instead of per hit):
    double inv_ir;      // 1.0 / ir                                   material.rs:161 (front face)
    double r0_front;    // ((1 - 1/ir) / (1 + 1/ir))^2                reflectance, material.rs:150-153
    double r0_back;     // ((1 - ir) / (1 + ir))^2
};
struct TexDev {   // material.rs:48-84
    int32_t type;  // 0 Constant, 1 Checker, 2 Image, 3 Noise (D9: Perlin marble, book 2)
    int32_t t0, t1;        // Checker: constant-texture ids (.0 when sines < 0, .1 otherwise)
    int32_t w, h;          // Image
    uint32_t texel_off;    // Image: byte offset into texels; Noise: byte offset (8-aligned) of its tables: 768 f64 gradient vectors, 768 permutation bytes
    double color[3];       // Constant; Noise: color[0] = scale
};

// ---------------------------------------------------------------------------
// Accel ("kernel 2"): a BVH2 built on the host over the SAME primitives, used only to
// decide WHICH primitives to test.  Box tests merely cull, so any conservative hierarchy
// yields the reference's closest hit as long as (a) primitives are tested with the
// reference's f64 arithmetic and (b) exact ties are resolved as the reference's visit order
// resolves them: every item carries `order` = its index in the DFS program above, and a
// candidate replaces the current best when t < best or (t == best and order > best.order).
//   Node2 (64 B): both children's boxes in f32, rounded OUTWARD and padded (flatten2.cpp
//   derives the pad from the f32 rounding of ray origins), so the f32 slab test never culls
//   a box the exact test would keep.
//   ref (32 bit): tag = ref >> 30 : 0 inner (Node2 index) | 1 leaf ((count-1) << 26 | first item)
//                 | 2 restore-world marker | 3 done (0xFFFFFFFF)
//   item (8 B): {kind | payload << 4, order}; kind = NodeKind of a primitive, or NK_INSTANCE
//   instance (8 B): {xform index, root ref of its object-space BVH}
// ---------------------------------------------------------------------------
static const uint32_t NK_INSTANCE = 8;
// An instance that has no compact copy (anything but f32-vertex triangles): kernels 5 / 6 enter it in the lane, as kernel 2 enters
// every instance, instead of deferring it; only accel items carry this kind.
static const uint32_t NK_INSTANCE_INLINE = 12;
#ifndef RT_NODE2_PAD
#define RT_NODE2_PAD 2
#endif
struct Node2 {
    float lo_x[2], lo_y[2], lo_z[2], hi_x[2], hi_y[2], hi_z[2];  // [child]
    uint32_t child[2];
    uint32_t pad[RT_NODE2_PAD];
};
static const uint32_t NODE2_F4 = (14 + RT_NODE2_PAD) / 4;  // Node2 stride in 16-byte words
static const uint32_t REF_TAG_SHIFT = 30;
static const uint32_t REF_LEAF = 1u << 30;
static const uint32_t REF_RESTORE = 2u << 30;
static const uint32_t REF_DONE = 0xFFFFFFFFu;
static const uint32_t REF_LEAF_COUNT_SHIFT = 26;
static const uint32_t REF_LEAF_FIRST_MASK = (1u << 26) - 1;
static const int ACCEL_MAX_LEAF = 8;      // limit of the 3-bit count field; the builder's default is ACCEL_DEFAULT_LEAF
static const int ACCEL_DEFAULT_LEAF = 4;
static const int ACCEL_MAX_STACK = 64;

struct FlatView {  // by-value kernel argument
    const char* base;
    uint32_t off_meta, off_boxes, off_spheres, off_sphere_mat, off_rects, off_rect_mat, off_tris, off_xforms;
    uint32_t off_mats, off_texs, off_vpos, off_vnrm, off_texels;
    uint32_t off_media;    // cold part: MediumDev per ConstantMedium, in the reference's visit order
    uint32_t n_media;
    uint32_t off_msph;     // cold part: moving spheres (NK_MSPHERE), 10 f64 each
    uint32_t n_msph;
    uint32_t has_noise;    // 1 if some texture is a noise texture (D9): such scenes take the book-2 kernel variants
    // cold part: eight words {off_meta, off_boxes, off_spheres, off_rects, off_tripre, off_xforms, this record's own offset, n_nodes}:
    // the reference-order program's tables in GLOBAL memory, for the out-of-line walk that settles an exact tie (tie_resolve, kernels.hip)
    uint32_t off_tie_view;
    uint32_t n_nodes;
    uint32_t stage_bytes;  // kernel 1 stages bytes [0, stage_bytes) into LDS: [meta|boxes|spheres|rects|tris|xforms|vpos]
    uint32_t kinds_mask;   // bit k set if some node has kind k
    uint32_t total_bytes;
    // accel (kernel 2)
    uint32_t accel_ok;        // 0: no accel was built (unbounded item, depth overflow, ...): kernel 1 only
    uint32_t off_n2, off_items2, off_inst2;
    uint32_t root2;           // root ref
    uint32_t stage2_begin, stage2_end;  // kernel 2 stages [stage2_begin, stage2_end): [spheres|rects|tris|xforms|vpos|n2|items2|inst2]
    uint32_t stack2;          // stack entries a lane can need
    uint32_t off_tripre;      // per triangle {pa, pb-pa, pc-pa, pad}: 10 f64 (hot part, after tris)
    uint32_t off_lights, n_lights;  // cold part: per light {NK_SPHERE | NK_RECT_XZ, payload index}
    uint32_t off_tripre2;     // accel: triangle records {pa, e0, e1, pad} in ITEM order (leaf-contiguous)
    uint32_t n_nodes2;        // Node2 count; the array is sorted by depth, so a prefix of it = the top of every BVH
    double origin_limit2;     // accel boxes are padded for ray origins with max-abs coordinate <= this (camera checked per render)
    uint32_t n_inst2;            // instances (object-space BVHs under a Transform)
    uint32_t max_inst_nodes2;    // Node2 count of the largest instance BVH
    uint32_t inst_depth2;        // depth of the deepest instance BVH (stack entries a suspended object-space walk can hold)
    uint32_t n_world_items2;     // items2[0 .. n_world_items2) are the world-space BVH's and the INLINE instances' (their leaves are laid out first)
    uint32_t stack2_inline;      // stack entries a lane needs for the world-space walk including the inline instances
    uint32_t n_inline2;          // instances kernels 5 / 6 enter in the lane (NK_INSTANCE_INLINE); 0: their MIXED variants are not needed
    // kernel 5's compact copies of the object-space data (see "Compact instance data" below); coop_data_ok = 0: not available
    uint32_t coop_data_ok;
    uint32_t off_n2q;            // NodeQ per Node2 index (object-space nodes only; world-space entries are unused)
    uint32_t off_tri32;          // Tri32 per item index (object-space leaves only)
    uint32_t off_qgrid;          // QGrid per instance
    uint32_t world_top2;         // world-space BVH nodes all have an index below this (the Node2 array is depth-sorted)
    uint32_t world_depth2;       // depth of the world-space BVH (kernel 5 walks it and the object-space BVHs separately)
};

// ---------------------------------------------------------------------------
// Compact instance data (kernel 5).  The serving waves of kernel 5 are bound by the number of cache lines the L1 has to look
// up, not by arithmetic, so the object-space BVHs and triangles get a second, smaller encoding:
//   NodeQ (32 B, Node2 is 64): both children's boxes on a 16-bit grid laid over the instance's bounds, rounded OUTWARD and
//     padded by P grid units; the serving lane maps its object-space ray into grid coordinates once per request
//     (o_g = (o - mn) * k + shift, d_g = d * k per axis: t is unchanged) and runs the same conservative f32 slab test (box32)
//     on the integers.  box32's proof is coordinate-free: it needs stored bounds that contain the exact box with a margin of
//     4 * 2^-24 * |o|max, here in grid units, hence P = ceil(2^-22 * O_g) + 2 with O_g = the largest grid coordinate a ray
//     origin can have (the +2 covers floor/ceil of the host's f64 products and the f64 rounding of o_g, d_g: < 2^-14 units).
//   Tri32 (48 B, item + hoisted record are 88): the three vertices in f32 -- meshes come out of the OBJ loader as f32 widened
//     to f64 (mesh.rs:160-172), so this is exact; the edges e0 = pb - pa, e1 = pc - pa are formed in f64 by the lane exactly as
//     the host formed them -- plus the item's order and kind|payload, so that a leaf needs no second table.
//   Instances whose vertices are not f32 values, or that hold anything but triangles, have no compact copy: kernel 2 renders.
// ---------------------------------------------------------------------------
struct NodeQ {
    uint32_t lox, loy, loz, hix, hiy, hiz;  // child 0 in the low half, child 1 in the high half
    uint32_t child[2];
};
struct Tri32 {
    float pa[3], pb[3], pc[3];
    uint32_t order, kp;
    float me;  // >= max |component| of the edges pb - pa, pc - pa (rounded up): scales the error bounds of the f32 pre-test (tri_miss32, kernels.hip)
};
struct QGrid {
    double mn[3], k[3], shift, pad;
};
static const double QGRID_MAX = 65535.0;

// Kernel 5 keeps the WORLD-level tables in LDS even though the scene as a whole does not fit: spheres, rects, transforms, the
// instance table and the world-space BVH's items (the mesh data -- triangles, object-space nodes and items -- stays in L2 / HBM).
constexpr uint32_t coop_a16(uint32_t x) { return (x + 15u) & ~15u; }
constexpr uint32_t coop_world_bytes(const FlatView& v) {
    return coop_a16(v.off_rects - v.off_spheres) + coop_a16(v.off_tris - v.off_rects) + coop_a16(v.stage_bytes - v.off_xforms) +
           coop_a16(8u * v.n_inst2) + coop_a16(8u * v.n_world_items2) + coop_a16(64u * v.n_inst2);  // + QGrid per instance
}

}  // namespace rtamd
