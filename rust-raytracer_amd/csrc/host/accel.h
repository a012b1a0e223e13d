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
    std::vector<uint32_t> items;  // 2 words per item
    std::vector<uint32_t> inst;   // 2 words per instance
    int max_depth = 0;            // deepest root-to-leaf path over all BVHs
    bool ok = true;
};
// Builds one BVH2 (binned SAH) over `items`, boxes padded by `pad` and rounded outward to f32.
// Returns the root ref.  `depth0` is the stack depth already used above this BVH.
uint32_t accel_build_bvh(AccelBuild& out, std::vector<AccelItem>& items, double pad, int depth0);

}  // namespace rtamd
