// Kernel-2 acceleration structure: binned-SAH BVH2 with conservative f32 child boxes.
//
// This hierarchy is NOT the reference's (objects/bvh.rs:60-83 is a random-axis median split and
// the scene files carry their own trees); it only prunes.  The reference's result is preserved
// because primitives keep their f64 tests and the tie rule is carried by AccelItem::order
// (see common/flat.h).  Host-side, once per rt_scene_commit.
#include "accel.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace rtamd {

namespace {

float round_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
float round_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}
Box merge(const Box& a, const Box& b) {
    Box r;
    for (int i = 0; i < 3; i++) {
        r.mn[i] = std::fmin(a.mn[i], b.mn[i]);
        r.mx[i] = std::fmax(a.mx[i], b.mx[i]);
    }
    return r;
}
double area(const Box& b) {
    double dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return 2.0 * (dx * dy + dy * dz + dz * dx);
}
Box empty_box() {
    const double inf = std::numeric_limits<double>::infinity();
    return Box{{inf, inf, inf}, {-inf, -inf, -inf}};
}

struct Ctx {
    AccelBuild& out;
    std::vector<AccelItem>& items;
    double pad;
    double c_box;  // rt_tuning.sah_box_cost and .max_leaf, snapshotted once per build (read at commit time)
    int max_leaf;
};

const double C_PRIM = 2.0;  // relative costs of a child-box pair test (c_box()) and a primitive test
#ifndef RT_SAH_BINS
#define RT_SAH_BINS 32  // (16: headline -0.4 %, 64: +-0)
#endif
const int BINS = RT_SAH_BINS;
int clamp_max_leaf(int m) {  // items per leaf, 1..ACCEL_DEFAULT_LEAF
    return m < 1 ? ACCEL_DEFAULT_LEAF : (m > ACCEL_DEFAULT_LEAF ? ACCEL_DEFAULT_LEAF : m);
}

uint32_t make_leaf(Ctx& c, int begin, int end) {
    uint32_t first = (uint32_t)(c.out.items.size() / 2);
    for (int i = begin; i < end; i++) {
        c.out.items.push_back(c.items[i].kp);
        c.out.items.push_back((uint32_t)c.items[i].order);
    }
    if (first + (uint32_t)(end - begin) > REF_LEAF_FIRST_MASK) {
        c.out.ok = false;
        return REF_DONE;
    }
    return REF_LEAF | ((uint32_t)(end - begin - 1) << REF_LEAF_COUNT_SHIFT) | first;
}

// force_split: the root of a BVH with >= 2 items is always an inner node with two real children, so that every BVH starts with
// a box test and no node needs an "empty" child
uint32_t build(Ctx& c, int begin, int end, int depth, bool force_split = false) {
    if (depth > c.out.max_depth) c.out.max_depth = depth;
    const int n = end - begin;
    bool has_instance = false;  // an instance must sit alone in its leaf (the traversal enters one instance per leaf)
    for (int i = begin; i < end; i++) has_instance |= (c.items[i].kp & NK_MASK) == NK_INSTANCE;
    if (has_instance && n > 1 && depth >= ACCEL_MAX_STACK - 4) {
        c.out.ok = false;
        return REF_DONE;
    }
    if (n <= 1 || depth >= ACCEL_MAX_STACK - 4) {
        if (n > ACCEL_MAX_LEAF) {  // depth cap hit with too many items: give up on the accel, kernel 1 remains
            c.out.ok = false;
            return REF_DONE;
        }
        return make_leaf(c, begin, end);
    }
    Box bounds = empty_box(), cb = empty_box();
    for (int i = begin; i < end; i++) {
        bounds = merge(bounds, c.items[i].box);
        for (int a = 0; a < 3; a++) {
            double ctr = 0.5 * (c.items[i].box.mn[a] + c.items[i].box.mx[a]);
            cb.mn[a] = std::fmin(cb.mn[a], ctr);
            cb.mx[a] = std::fmax(cb.mx[a], ctr);
        }
    }
    // binned SAH over the three axes
    double best_cost = std::numeric_limits<double>::infinity();
    int best_axis = -1, best_bin = -1;
    const double parent_area = area(bounds);
    for (int a = 0; a < 3; a++) {
        double lo = cb.mn[a], ext = cb.mx[a] - cb.mn[a];
        if (!(ext > 0.) || !std::isfinite(ext)) continue;
        Box bb[BINS];
        int cnt[BINS];
        for (int b = 0; b < BINS; b++) { bb[b] = empty_box(); cnt[b] = 0; }
        for (int i = begin; i < end; i++) {
            double ctr = 0.5 * (c.items[i].box.mn[a] + c.items[i].box.mx[a]);
            int b = (int)((ctr - lo) / ext * BINS);
            if (b < 0) b = 0;
            if (b >= BINS) b = BINS - 1;
            bb[b] = merge(bb[b], c.items[i].box);
            cnt[b]++;
        }
        double right_area[BINS];
        int right_cnt[BINS];
        Box acc = empty_box();
        int k = 0;
        for (int b = BINS - 1; b > 0; b--) {
            acc = merge(acc, bb[b]);
            k += cnt[b];
            right_area[b] = k ? area(acc) : 0.;
            right_cnt[b] = k;
        }
        acc = empty_box();
        k = 0;
        for (int b = 0; b < BINS - 1; b++) {
            acc = merge(acc, bb[b]);
            k += cnt[b];
            if (k == 0 || right_cnt[b + 1] == 0) continue;
            double cost = c.c_box + C_PRIM * (area(acc) * k + right_area[b + 1] * right_cnt[b + 1]) / parent_area;
            if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
        }
    }
    int mid = -1;
    if (best_axis >= 0 && (n > c.max_leaf || has_instance || force_split || best_cost < C_PRIM * n)) {
        double lo = cb.mn[best_axis], ext = cb.mx[best_axis] - cb.mn[best_axis];
        auto it = std::stable_partition(c.items.begin() + begin, c.items.begin() + end, [&](const AccelItem& it2) {
            double ctr = 0.5 * (it2.box.mn[best_axis] + it2.box.mx[best_axis]);
            int b = (int)((ctr - lo) / ext * BINS);
            if (b < 0) b = 0;
            if (b >= BINS) b = BINS - 1;
            return b <= best_bin;
        });
        mid = (int)(it - c.items.begin());
        if (mid == begin || mid == end) mid = -1;
    }
    if (mid < 0) {
        if (n <= c.max_leaf && !has_instance && !force_split) return make_leaf(c, begin, end);
        mid = begin + n / 2;  // identical centroids (e.g. concentric spheres): split by index
    }
    uint32_t idx = (uint32_t)c.out.nodes.size();
    c.out.nodes.push_back(Node2{});
    uint32_t child[2];
    Box cbx[2] = {empty_box(), empty_box()};
    for (int i = begin; i < mid; i++) cbx[0] = merge(cbx[0], c.items[i].box);
    for (int i = mid; i < end; i++) cbx[1] = merge(cbx[1], c.items[i].box);
    child[0] = build(c, begin, mid, depth + 1);
    child[1] = build(c, mid, end, depth + 1);
    Node2& nd = c.out.nodes[idx];
    for (int k = 0; k < 2; k++) {
        nd.lo_x[k] = round_down(cbx[k].mn[0] - c.pad); nd.hi_x[k] = round_up(cbx[k].mx[0] + c.pad);
        nd.lo_y[k] = round_down(cbx[k].mn[1] - c.pad); nd.hi_y[k] = round_up(cbx[k].mx[1] + c.pad);
        nd.lo_z[k] = round_down(cbx[k].mn[2] - c.pad); nd.hi_z[k] = round_up(cbx[k].mx[2] + c.pad);
        nd.child[k] = child[k];
    }
    nd.pad[0] = nd.pad[1] = 0;
    return idx;
}

}  // namespace

// Node2.pad[k] = the smallest reference-order index among the items of child k's subtree: lets a walk that only cares about items
// below some index (the media logic of kernel 2: "what the reference has visited before this medium") skip whole subtrees.
static uint32_t fill_min_order(AccelBuild& out, uint32_t ref) {
    if ((ref >> REF_TAG_SHIFT) == 1u) {
        const uint32_t first = ref & REF_LEAF_FIRST_MASK, cnt = ((ref >> REF_LEAF_COUNT_SHIFT) & 7u) + 1u;
        uint32_t m = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < cnt; i++) m = std::min(m, out.items[2 * (size_t)(first + i) + 1]);
        return m;
    }
    if ((ref >> REF_TAG_SHIFT) != 0u) return 0xFFFFFFFFu;
    Node2& nd = out.nodes[ref];
    const uint32_t c0 = nd.child[0], c1 = nd.child[1];
    const uint32_t m0 = fill_min_order(out, c0), m1 = fill_min_order(out, c1);
    out.nodes[ref].pad[0] = m0;
    out.nodes[ref].pad[1] = m1;
    return std::min(m0, m1);
}

static uint32_t accel_build_bvh_impl(AccelBuild& out, std::vector<AccelItem>& items, double pad, int depth0);
uint32_t accel_build_bvh(AccelBuild& out, std::vector<AccelItem>& items, double pad, int depth0) {
    const uint32_t root = accel_build_bvh_impl(out, items, pad, depth0);
    if (out.ok && root != REF_DONE) fill_min_order(out, root);
    return root;
}
static uint32_t accel_build_bvh_impl(AccelBuild& out, std::vector<AccelItem>& items, double pad, int depth0) {
    if (items.empty() || !std::isfinite(pad)) {
        out.ok = false;
        return REF_DONE;
    }
    for (auto& it : items)
        for (int a = 0; a < 3; a++)
            if (!std::isfinite(it.box.mn[a]) || !std::isfinite(it.box.mx[a])) {
                out.ok = false;
                return REF_DONE;
            }
    const Tuning tun = tuning();
    Ctx c{out, items, pad, tun.c_box > 0. ? tun.c_box : 1.0, clamp_max_leaf(tun.max_leaf)};
    if (items.size() == 1) {
        // A single item still gets one inner node so that the root is box-tested like everything else.  Its second child is a
        // ZERO-SIZE box at the low corner of the first (finite, inside the BVH's bounds, quantisable like any other box): a ray
        // passes the slab test of a point only when it goes through it to within the test's 1e-6 relative slack, and if one
        // ever does, re-testing the same item changes nothing (tie rule by order).  An inverted or infinite "empty" box does
        // NOT work here: the slab test takes min/max per axis, so +-inf bounds pass every ray.
        uint32_t idx = (uint32_t)out.nodes.size();
        out.nodes.push_back(Node2{});
        uint32_t leaf = make_leaf(c, 0, 1);
        Node2& nd = out.nodes[idx];
        const Box& b = items[0].box;
        nd.lo_x[0] = round_down(b.mn[0] - pad); nd.hi_x[0] = round_up(b.mx[0] + pad);
        nd.lo_y[0] = round_down(b.mn[1] - pad); nd.hi_y[0] = round_up(b.mx[1] + pad);
        nd.lo_z[0] = round_down(b.mn[2] - pad); nd.hi_z[0] = round_up(b.mx[2] + pad);
        nd.lo_x[1] = nd.hi_x[1] = nd.lo_x[0];
        nd.lo_y[1] = nd.hi_y[1] = nd.lo_y[0];
        nd.lo_z[1] = nd.hi_z[1] = nd.lo_z[0];
        nd.child[0] = leaf;
        nd.child[1] = leaf;
        nd.pad[0] = nd.pad[1] = 0;
        if (depth0 + 1 > out.max_depth) out.max_depth = depth0 + 1;
        return idx;
    }
    return build(c, 0, (int)items.size(), depth0, true);
}

}  // namespace rtamd
