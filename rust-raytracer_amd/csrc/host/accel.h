// Host-side builder of the kernel-2 acceleration structure (common/flat.h "Accel").
#pragma once
#include <cstdint>
#include <vector>

#include "scene.h"

namespace rtamd {

struct AccelItem {
    Box box;         // f64 box in the space of its BVH (world, or object space under a Transform)
    uint32_t kp;     // kind | payload << 4
    int32_t order;   // DFS index of the primitive in the reference-order program (tie rule)
};
struct AccelBuild {
    std::vector<Node2> nodes;
    std::vector<Box> cbox;        // per Node2: the two children's exact f64 boxes (unpadded), for the wide collapse
    std::vector<double> npad;     // per Node2: the pad its BVH was built with
    std::vector<uint32_t> items;  // 2 words per item
    std::vector<uint32_t> inst;   // 2 words per instance
    int max_depth = 0;            // deepest root-to-leaf path over all BVHs
    bool ok = true;
};
// Builds one BVH2 (binned SAH) over `items`, boxes padded by `pad` and rounded outward to f32.
// Returns the root ref.  `depth0` is the stack depth already used above this BVH.
uint32_t accel_build_bvh(AccelBuild& out, std::vector<AccelItem>& items, double pad, int depth0);

// Wide accel (common/flat.h "Wide accel"): collapses the BVH2s of `ab` (world BVH root2 + one BVH per instance, roots in
// ab.inst) into one breadth-first Node8 array.  Re-orders ab.items so that the items of a Node8's leaf children are
// contiguous and patches the BVH2 leaf refs (Node2::child, root2, ab.inst) accordingly: both accels share one item array.
struct Accel8Build {
    std::vector<Node8> nodes;
    std::vector<uint32_t> inst;   // 2 words per instance: {xform, root Node8 index}
    uint32_t root = NODE8_NONE;
    int max_depth = 0;            // deepest chain of Node8 over the world BVH plus one instance BVH
    bool ok = false;
};
void accel8_build(AccelBuild& ab, uint32_t& root2, Accel8Build& out);

}  // namespace rtamd
