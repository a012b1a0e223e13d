// Kernel-2 acceleration structure: binned-SAH BVH2 with conservative f32 child boxes.
//
// This hierarchy is NOT the reference's (objects/bvh.rs:60-83 is a random-axis median split and
// the scene files carry their own trees); it only prunes.  The reference's result is preserved
// because primitives keep their f64 tests and the tie rule is carried by AccelItem::order
// (see common/flat.h).  Host-side, once per rt_scene_commit.
#include "accel.h"

#include <algorithm>
#include <cmath>
#include <cstring>
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
};

const double C_PRIM = 2.0;  // relative costs of a child-box pair test (c_box()) and a primitive test
double c_box() { return tuning().c_box > 0. ? tuning().c_box : 1.0; }  // rt_tuning.sah_box_cost, read at commit time
const int BINS = 16;
int max_leaf() {  // items per leaf, 1..ACCEL_DEFAULT_LEAF (rt_tuning.max_leaf, read at commit time)
    const int m = tuning().max_leaf;
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

uint32_t build(Ctx& c, int begin, int end, int depth) {
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
            double cost = c_box() + C_PRIM * (area(acc) * k + right_area[b + 1] * right_cnt[b + 1]) / parent_area;
            if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
        }
    }
    int mid = -1;
    if (best_axis >= 0 && (n > max_leaf() || has_instance || best_cost < C_PRIM * n)) {
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
        if (n <= max_leaf() && !has_instance) return make_leaf(c, begin, end);
        mid = begin + n / 2;  // identical centroids (e.g. concentric spheres): split by index
    }
    uint32_t idx = (uint32_t)c.out.nodes.size();
    c.out.nodes.push_back(Node2{});
    c.out.cbox.resize(2 * c.out.nodes.size());
    c.out.npad.resize(c.out.nodes.size(), c.pad);
    uint32_t child[2];
    Box cbx[2] = {empty_box(), empty_box()};
    for (int i = begin; i < mid; i++) cbx[0] = merge(cbx[0], c.items[i].box);
    for (int i = mid; i < end; i++) cbx[1] = merge(cbx[1], c.items[i].box);
    c.out.cbox[2 * idx] = cbx[0];
    c.out.cbox[2 * idx + 1] = cbx[1];
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

uint32_t accel_build_bvh(AccelBuild& out, std::vector<AccelItem>& items, double pad, int depth0) {
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
    Ctx c{out, items, pad};
    if (items.size() == 1) {
        // a single item still needs one inner node so that the root is box-tested like everything else
        uint32_t idx = (uint32_t)out.nodes.size();
        out.nodes.push_back(Node2{});
        out.cbox.resize(2 * out.nodes.size());
        out.npad.resize(out.nodes.size(), pad);
        out.cbox[2 * idx] = items[0].box;
        out.cbox[2 * idx + 1] = empty_box();
        uint32_t leaf = make_leaf(c, 0, 1);
        Node2& nd = out.nodes[idx];
        const Box& b = items[0].box;
        const float inf = std::numeric_limits<float>::infinity();
        nd.lo_x[0] = round_down(b.mn[0] - pad); nd.hi_x[0] = round_up(b.mx[0] + pad);
        nd.lo_y[0] = round_down(b.mn[1] - pad); nd.hi_y[0] = round_up(b.mx[1] + pad);
        nd.lo_z[0] = round_down(b.mn[2] - pad); nd.hi_z[0] = round_up(b.mx[2] + pad);
        nd.lo_x[1] = nd.lo_y[1] = nd.lo_z[1] = inf;  // empty second child: never hit
        nd.hi_x[1] = nd.hi_y[1] = nd.hi_z[1] = -inf;
        nd.child[0] = leaf;
        nd.child[1] = leaf;  // never reached (empty box); if it ever were, re-testing the same item is harmless
        if (depth0 + 1 > out.max_depth) out.max_depth = depth0 + 1;
        return idx;
    }
    uint32_t root = build(c, 0, (int)items.size(), depth0);
    if (out.ok && (root >> REF_TAG_SHIFT) == 1u) {
        // the SAH kept everything in one leaf: still give the BVH an inner root so that every BVH starts with a box test
        // (and the wide collapse has a node to start from)
        Box all = empty_box();
        for (auto& it : items) all = merge(all, it.box);
        uint32_t idx = (uint32_t)out.nodes.size();
        out.nodes.push_back(Node2{});
        out.cbox.resize(2 * out.nodes.size());
        out.npad.resize(out.nodes.size(), pad);
        out.cbox[2 * idx] = all;
        out.cbox[2 * idx + 1] = empty_box();
        Node2& nd = out.nodes[idx];
        const float inf = std::numeric_limits<float>::infinity();
        nd.lo_x[0] = round_down(all.mn[0] - pad); nd.hi_x[0] = round_up(all.mx[0] + pad);
        nd.lo_y[0] = round_down(all.mn[1] - pad); nd.hi_y[0] = round_up(all.mx[1] + pad);
        nd.lo_z[0] = round_down(all.mn[2] - pad); nd.hi_z[0] = round_up(all.mx[2] + pad);
        nd.lo_x[1] = nd.lo_y[1] = nd.lo_z[1] = inf;
        nd.hi_x[1] = nd.hi_y[1] = nd.hi_z[1] = -inf;
        nd.child[0] = root;
        nd.child[1] = root;
        nd.pad[0] = nd.pad[1] = 0;
        if (depth0 + 1 > out.max_depth) out.max_depth = depth0 + 1;
        return idx;
    }
    return root;
}

// ------------------------------------------------------------------ wide accel (Node8) ----
namespace {

struct WideChild {
    uint32_t ref;  // BVH2 ref: inner Node2 index or leaf
    Box box;       // exact f64 box
};

// quantise one axis of a node: frame origin o (f32), exponent byte e, planes qlo/qhi for every child, all verified with the
// device's own decode  lo = fmaf((float)q, s, o)
bool quantise_axis(const std::vector<WideChild>& ch, const int* slot_of, int axis, double pad, float& o_out, uint8_t& e_out, uint8_t* qlo,
                   uint8_t* qhi) {
    double lo = std::numeric_limits<double>::infinity(), hi = -lo;
    for (auto& c : ch) {
        lo = std::fmin(lo, c.box.mn[axis] - pad);
        hi = std::fmax(hi, c.box.mx[axis] + pad);
    }
    const float o = round_down(lo);
    if (!std::isfinite(o) || !std::isfinite(hi)) return false;
    double ext = hi - (double)o;
    int e0 = -120;
    if (ext > 0.) {
        int ex;
        std::frexp(ext / 255.0, &ex);  // ext/255 = m * 2^ex, m in [0.5, 1)  ->  2^ex >= ext/255
        e0 = std::max(ex - 1, -120);
    }
    for (int e = e0; e <= 100; e++) {
        const float s = std::ldexp(1.0f, e);
        bool ok = true;
        for (size_t i = 0; i < ch.size() && ok; i++) {
            const double clo = ch[i].box.mn[axis] - pad, chi = ch[i].box.mx[axis] + pad;
            double ql = std::floor((clo - (double)o) / (double)s);
            if (ql > 255.) ql = 255.;
            if (ql < 0.) ql = 0.;
            int q = (int)ql;
            while (q > 0 && (double)std::fmaf((float)q, s, o) > clo) q--;
            if ((double)std::fmaf((float)q, s, o) > clo) ok = false;
            int q2 = 0;
            if (ok) {
                double qh = std::ceil((chi - (double)o) / (double)s);
                if (qh < 0.) qh = 0.;
                if (qh > 255.) { ok = false; break; }
                q2 = (int)qh;
                while (q2 <= 255 && (double)std::fmaf((float)q2, s, o) < chi) q2++;
                if (q2 > 255) ok = false;
            }
            if (ok) {
                qlo[slot_of[i]] = (uint8_t)q;
                qhi[slot_of[i]] = (uint8_t)q2;
            }
        }
        if (ok) {
            o_out = o;
            e_out = (uint8_t)(e + 127);
            return e + 127 >= 1 && e + 127 <= 254;
        }
    }
    return false;
}

}  // namespace

void accel8_build(AccelBuild& ab, uint32_t& root2, Accel8Build& out) {
    out = Accel8Build{};
    if (!ab.ok || (root2 >> REF_TAG_SHIFT) != 0u) return;
    const size_t n_inst = ab.inst.size() / 2;
    for (size_t i = 0; i < n_inst; i++)
        if ((ab.inst[2 * i + 1] >> REF_TAG_SHIFT) != 0u) return;  // every BVH has an inner root (accel_build_bvh)
    // breadth-first over all BVHs at once: node index order == depth order
    struct Pending {
        uint32_t node2;  // Node2 the wide node grows from
        int depth;       // 1 = root of its BVH
        bool world;      // part of the world-space BVH (else: of an instance's object-space BVH)
    };
    std::vector<Pending> queue;
    queue.push_back({root2, 1, true});
    out.root = 0;
    out.inst.assign(2 * n_inst, 0u);
    for (size_t i = 0; i < n_inst; i++) {
        out.inst[2 * i] = ab.inst[2 * i];
        out.inst[2 * i + 1] = (uint32_t)queue.size();
        queue.push_back({ab.inst[2 * i + 1], 1, false});
    }
    out.nodes.resize(queue.size());
    std::vector<uint32_t> new_items;            // items in wide-node order
    new_items.reserve(ab.items.size());
    std::vector<uint32_t> new_first(ab.items.size() / 2 + 1, 0xFFFFFFFFu);  // old first item of a BVH2 leaf -> new first
    int depth_world = 0, depth_inst = 0;
    for (size_t qi = 0; qi < queue.size(); qi++) {
        const Pending pn = queue[qi];
        // collapse: open the inner child with the largest surface area until there are 8 children or only leaves
        std::vector<WideChild> ch;
        const double pad = ab.npad[pn.node2];
        for (int k = 0; k < 2; k++) {
            const Box& b = ab.cbox[2 * pn.node2 + k];
            if (b.mn[0] <= b.mx[0]) ch.push_back({ab.nodes[pn.node2].child[k], b});  // skip the empty child of a one-item BVH
        }
        for (;;) {
            if (ch.size() >= 8) break;
            int best = -1;
            double best_area = -1.;
            for (size_t i = 0; i < ch.size(); i++)
                if ((ch[i].ref >> REF_TAG_SHIFT) == 0u) {
                    double a = area(ch[i].box);
                    if (a > best_area) { best_area = a; best = (int)i; }
                }
            if (best < 0) break;
            const uint32_t n2 = ch[best].ref;
            WideChild c0{ab.nodes[n2].child[0], ab.cbox[2 * n2]}, c1{ab.nodes[n2].child[1], ab.cbox[2 * n2 + 1]};
            ch[best] = c0;
            if (c1.box.mn[0] <= c1.box.mx[0]) ch.push_back(c1);
        }
        // slots: bit a of the slot index = the child lies on the + side of the node along axis a (greedy assignment of
        // children to the 8 octant directions by the projection of their centroid offset)
        Box nb = empty_box();
        for (auto& c : ch) nb = merge(nb, c.box);
        double cost[8][8];
        for (size_t i = 0; i < ch.size(); i++)
            for (int s = 0; s < 8; s++) {
                double v = 0.;
                for (int a = 0; a < 3; a++) {
                    double off = 0.5 * (ch[i].box.mn[a] + ch[i].box.mx[a]) - 0.5 * (nb.mn[a] + nb.mx[a]);
                    v += ((s >> a) & 1) ? off : -off;
                }
                cost[i][s] = v;
            }
        int slot_of[8];
        bool child_done[8] = {false}, slot_used[8] = {false};
        for (size_t round = 0; round < ch.size(); round++) {
            int bi = -1, bs = -1;
            for (size_t i = 0; i < ch.size(); i++) {
                if (child_done[i]) continue;
                for (int s = 0; s < 8; s++)
                    if (!slot_used[s] && (bi < 0 || cost[i][s] > cost[bi][bs])) { bi = (int)i; bs = s; }
            }
            slot_of[bi] = bs;
            child_done[bi] = true;
            slot_used[bs] = true;
        }
        Node8 nd;
        std::memset(&nd, 0, sizeof(nd));
        for (int a = 0; a < 3; a++)
            if (!quantise_axis(ch, slot_of, a, pad, nd.o[a], nd.e[a], nd.qlo[a], nd.qhi[a])) return;  // out.ok stays false
        // children in slot order: inner ones get contiguous Node8 indices, leaves get contiguous items
        const uint32_t child_base = (uint32_t)out.nodes.size();
        const uint32_t item_base = (uint32_t)(new_items.size() / 2);
        uint32_t imask = 0, lmask = 0, n_inner = 0, n_item = 0;
        int child_in_slot[8];
        for (int s = 0; s < 8; s++) child_in_slot[s] = -1;
        for (size_t i = 0; i < ch.size(); i++) child_in_slot[slot_of[i]] = (int)i;
        for (int s = 0; s < 8; s++) {
            if (child_in_slot[s] < 0) continue;
            const WideChild& c = ch[child_in_slot[s]];
            if ((c.ref >> REF_TAG_SHIFT) == 0u) {
                imask |= 1u << s;
                nd.meta[s] = 0xFF;
                queue.push_back({c.ref, pn.depth + 1, pn.world});
                n_inner++;
            } else {
                const uint32_t first = c.ref & REF_LEAF_FIRST_MASK, cnt = ((c.ref >> REF_LEAF_COUNT_SHIFT) & 7u) + 1u;
                if (cnt > 4 || n_item + cnt > 32) return;
                lmask |= 1u << s;
                nd.meta[s] = (uint8_t)((cnt << 5) | n_item);
                if (new_first[first] == 0xFFFFFFFFu) new_first[first] = item_base + n_item;
                for (uint32_t j = 0; j < cnt; j++) {
                    new_items.push_back(ab.items[2 * (first + j)]);
                    new_items.push_back(ab.items[2 * (first + j) + 1]);
                }
                n_item += cnt;
            }
        }
        out.nodes.resize(out.nodes.size() + n_inner);
        if (out.nodes.size() >= (1u << 24)) return;
        nd.imask = (uint8_t)imask;
        nd.cb_lm = child_base | (lmask << 24);
        nd.item_base = item_base;
        out.nodes[qi] = nd;
        if (pn.world) depth_world = std::max(depth_world, pn.depth);
        else depth_inst = std::max(depth_inst, pn.depth);
    }
    out.max_depth = depth_world + depth_inst;
    if (new_items.size() != ab.items.size()) return;  // every BVH2 leaf is reached exactly once
    // one item array for both accels: adopt the wide order and re-point the BVH2 leaves
    ab.items.swap(new_items);
    auto patch = [&](uint32_t r) -> uint32_t {
        if ((r >> REF_TAG_SHIFT) != 1u) return r;
        const uint32_t first = r & REF_LEAF_FIRST_MASK;
        return (r & ~REF_LEAF_FIRST_MASK) | new_first[first];
    };
    for (auto& n : ab.nodes) {
        n.child[0] = patch(n.child[0]);
        n.child[1] = patch(n.child[1]);
    }
    out.ok = true;
}

}  // namespace rtamd
