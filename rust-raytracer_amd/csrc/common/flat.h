// Flattened ("linear") scene shared by the host flattener and the HIP kernels.
//
// The reference's Arc<dyn Hitable> tree (objects/bvh.rs:29-33, objects/hit.rs:56,
// objects/transform.rs:9-14 ...) becomes ONE program of nodes in depth-first
// pre-order -- exactly the order the reference's recursion visits them (left child
// before right, list items in insertion order) -- so a node's first child is the
// next node and a failed box test jumps to `skip`.  No stack, no pointers.
//
// Everything lives in one blob so it can be staged into LDS with one copy:
//   hot  [meta | boxes | spheres | rects | tris | xforms | vpos]      <- staged into LDS when it fits
//   cold [sphere_mat | rect_mat | mats | texs | vnrm | texels]        <- read once per segment, stays global
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
    NK_XFORM_END = 7     // leave the Transform: restore the world ray
};
static const uint32_t NK_BITS = 4;
static const uint32_t NK_MASK = 15;

// per node: meta[2*n] = kind | payload << 4 ; meta[2*n+1] = skip
struct MatDev {   // material.rs:88-212
    int32_t type;  // 0 Lambertian, 1 Metal, 2 Dielectric, 3 DiffuseLight
    int32_t tex;   // albedo / emit texture
    double param;  // Metal.fuzz | Dielectric.ir
};
struct TexDev {   // material.rs:48-84
    int32_t type;  // 0 Constant, 1 Checker, 2 Image
    int32_t t0, t1;        // Checker: constant-texture ids (.0 when sines < 0, .1 otherwise)
    int32_t w, h;          // Image
    uint32_t texel_off;    // Image: byte offset into texels
    double color[3];       // Constant
};

struct FlatView {  // by-value kernel argument
    const char* base;
    uint32_t off_meta, off_boxes, off_spheres, off_sphere_mat, off_rects, off_rect_mat, off_tris, off_xforms;
    uint32_t off_mats, off_texs, off_vpos, off_vnrm, off_texels;
    uint32_t n_nodes;
    uint32_t stage_bytes;  // hot part: bytes [0, stage_bytes) are staged into LDS by the LDS kernel variant
    uint32_t kinds_mask;   // bit k set if some node has kind k
    uint32_t total_bytes;
    uint32_t pad;
};

}  // namespace rtamd
