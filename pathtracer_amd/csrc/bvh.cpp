#include "bvh.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>
#include <thread>
#include <queue>
#include <system_error>
#include <utility>
#include <vector>

namespace pt {
namespace {

struct Box {
    float mn[3], mx[3];
};

// reference bounding_box.rs:17-27: min = (pos - rad) - 0.01, max = (pos + rad) + 0.01
Box sphere_box(const Sphere32& s) {
    Box b;
    for (int a = 0; a < 3; a++) {
        b.mn[a] = (s.pos[a] - s.rad) - 0.01f;
        b.mx[a] = (s.pos[a] + s.rad) + 0.01f;
    }
    return b;
}
// reference bounding_box.rs:44-48 (Rust's f32::min/max drop a NaN operand, like fminf/fmaxf)
Box unite(const Box& a, const Box& b) {
    Box r;
    for (int k = 0; k < 3; k++) {
        r.mn[k] = fminf(a.mn[k], b.mn[k]);
        r.mx[k] = fmaxf(a.mx[k], b.mx[k]);
    }
    return r;
}

// ------------------------------------------------------------------ reference tree (bvh.rs)
struct RefNode {
    Box box;
    int32_t left = -1, right = -1;  // arena indices
    uint32_t obj = 0;
    bool leaf = false;
};

struct RefBuilder {
    const Sphere32* sp;
    std::vector<uint32_t> ids;
    std::vector<RefNode> arena;
    int swap_mode;
    uint32_t swap_state;

    int32_t leaf(uint32_t id) {
        RefNode n;
        n.leaf = true;
        n.obj = id;
        n.box = sphere_box(sp[id]);
        arena.push_back(n);
        return (int32_t)arena.size() - 1;
    }
    bool coin() {
        if (swap_mode == 0) return false;
        if (swap_mode == 1) return true;
        uint32_t r = swap_state;
        r ^= r << 13;
        r ^= r >> 17;
        r ^= r << 5;
        swap_state = r;
        return (r >> 16) & 1u;
    }
    // ids[lo,hi) plays the role of the Vec handed to new_interior (bvh.rs:60): sorting a sub-range in
    // place and recursing on its halves is what split_off + two owned Vecs do.
    int32_t interior(size_t lo, size_t hi) {
        size_t n = hi - lo;
        int32_t l, r;
        if (n == 2) {  // bvh.rs:63-70
            l = leaf(ids[lo]);
            r = leaf(ids[lo + 1]);
        } else if (n == 3) {  // bvh.rs:71-80: the last element becomes the left leaf
            l = leaf(ids[hi - 1]);
            r = interior(lo, hi - 1);
        } else {
            // bvh.rs:82-96.  The split axis is chosen from `max - min` of the two sentinel vectors
            // (f32::MIN - f32::MAX = -inf on every axis), never from the box of the spheres, so
            // neither `>` holds and the key is z for every node.
            const float big = std::numeric_limits<float>::max();
            float diff[3] = {-big - big, -big - big, -big - big};
            int axis = (diff[0] > diff[1] && diff[0] > diff[2]) ? 0 : (diff[1] > diff[2] ? 1 : 2);
            const Sphere32* s = sp;
            std::stable_sort(ids.begin() + lo, ids.begin() + hi,
                             [s, axis](uint32_t a, uint32_t b) { return s[a].pos[axis] < s[b].pos[axis]; });
            size_t mid = lo + n / 2;
            l = interior(lo, mid);
            r = interior(mid, hi);
            if (coin()) std::swap(l, r);  // bvh.rs:107-109
        }
        RefNode t;
        t.left = l;
        t.right = r;
        t.box = unite(arena[l].box, arena[r].box);
        arena.push_back(t);
        return (int32_t)arena.size() - 1;
    }
};

void put_box(Node48& n, const Box& b) {
    for (int a = 0; a < 3; a++) n.mn[a] = b.mn[a], n.mx[a] = b.mx[a];
}
// bvh.rs:121-140: reserve the slot, emit left then right, then store the escape index.
void flatten_ref(const std::vector<RefNode>& arena, int32_t i, std::vector<Node48>& out) {
    const RefNode& t = arena[i];
    Node48 n;
    memset(&n, 0, sizeof n);
    put_box(n, t.box);
    if (t.leaf) {
        n.type = 1u;
        n.index = t.obj;
        out.push_back(n);
        return;
    }
    size_t slot = out.size();
    out.push_back(n);
    flatten_ref(arena, t.left, out);
    flatten_ref(arena, t.right, out);
    out[slot].type = 0u;
    out[slot].index = (uint32_t)out.size();
}

// ------------------------------------------------------------------ emit helpers
void set_child_boxes(Node64& n, const float* lo0, const float* hi0, const float* lo1, const float* hi1) {
    n.q0 = f4{lo0[0], lo0[1], lo0[2], hi0[0]};
    n.q1 = f4{hi0[1], hi0[2], lo1[0], lo1[1]};
    n.q2 = f4{lo1[2], hi1[0], hi1[1], hi1[2]};
}
int32_t leaf_ref(uint32_t first, uint32_t count) {
    return (int32_t)~(first | ((count - 1u) << LEAF_COUNT_SHIFT));
}

constexpr uint32_t kMaxDepth = 64;

struct RefConverter {
    const Node48* nodes;
    uint32_t n_nodes, n_spheres;
    Bvh2* out;
    std::vector<uint8_t> seen;
    std::string* err;

    bool fail(const char* m) {
        *err = m;
        return false;
    }
    static uint32_t subtree_end(const Node48& n, uint32_t i) { return n.type == 1u ? i + 1 : n.index; }

    // Converts the subtree occupying [i, end); writes its link to *link.
    bool conv(uint32_t i, uint32_t end, uint32_t depth, int32_t* link) {
        if (i >= n_nodes || end > n_nodes || end <= i) return fail("bvh: subtree range out of bounds");
        const Node48& nd = nodes[i];
        if (nd.type == 1u) {
            if (end != i + 1) return fail("bvh: leaf does not end its range");
            if (nd.index >= n_spheres) return fail("bvh: leaf sphere index out of range");
            if (seen[nd.index]) return fail("bvh: sphere referenced by two leaves");
            seen[nd.index] = 1;
            *link = leaf_ref((uint32_t)out->order.size(), 1);
            out->order.push_back(nd.index);
            return true;
        }
        if (nd.type != 0u) return fail("bvh: unknown node type");
        if (nd.index != end) return fail("bvh: escape index does not match the subtree extent");
        if (depth >= kMaxDepth) return fail("bvh: tree deeper than 64 levels");
        if (depth + 1 > out->max_depth) out->max_depth = depth + 1;
        uint32_t left = i + 1;
        if (left >= end) return fail("bvh: interior node without children");
        uint32_t left_end = subtree_end(nodes[left], left);
        if (left_end <= left || left_end >= end) return fail("bvh: left child overruns its parent");
        uint32_t right = left_end;
        size_t me = out->nodes.size();
        out->nodes.push_back(Node64{});
        // The reference tests a node's box when it visits the node and tests a leaf's sphere without any
        // box of its own (compute.wgsl:244-262): a sphere is tried whenever its parent's box was hit.  A
        // leaf child therefore carries the parent's box here.
        const Node48& l = nodes[left];
        const Node48& r = nodes[right];
        set_child_boxes(out->nodes[me], l.type == 1u ? nd.mn : l.mn, l.type == 1u ? nd.mx : l.mx,
                        r.type == 1u ? nd.mn : r.mn, r.type == 1u ? nd.mx : r.mx);
        int32_t c0, c1;
        if (!conv(left, left_end, depth + 1, &c0)) return false;
        if (!conv(right, end, depth + 1, &c1)) return false;
        out->nodes[me].child0 = c0;
        out->nodes[me].child1 = c1;
        *link = (int32_t)me;
        return true;
    }
};

// ------------------------------------------------------------------ binned SAH
struct Prim {
    Box box;
    float c[3];
    uint32_t id;
};
struct TmpNode {
    Box box;
    int32_t left = -1, right = -1;
    uint32_t first = 0, count = 0;  // leaf when count > 0
};
float half_area(const Box& b) {
    float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return dx * dy + dy * dz + dz * dx;
}
Box empty_box() {
    const float inf = std::numeric_limits<float>::infinity();
    return Box{{inf, inf, inf}, {-inf, -inf, -inf}};
}

constexpr int kBins = 16;
constexpr float kCostNode = 1.0f;  // one interior step: a 64-byte record and two slab tests

// Tunables of the SAH builder; the environment overrides exist for experiments only.
struct SahParams {
    uint32_t max_leaf = MAX_LEAF_SPHERES;  // spheres per leaf, 1..4
    float cost_sphere = 0.6f;              // one sphere test relative to one interior step
    SahParams() {
        if (const char* e = getenv("MI355PT_MAX_LEAF")) {
            int v = atoi(e);
            if (v >= 1 && v <= MAX_LEAF_SPHERES) max_leaf = (uint32_t)v;
        }
        if (const char* e = getenv("MI355PT_COST_SPHERE")) {
            float v = (float)atof(e);
            if (v > 0.0f && v < 100.0f) cost_sphere = v;
        }
    }
};

struct SahBuilder {
    SahParams par;
    std::vector<Prim> prims;
    std::vector<TmpNode> tmp;

    // Builds the subtree over prims[lo, hi) into `tmp` (pre-order) and returns its index there.  The
    // first `fork_levels` levels of large ranges build their right halves on other threads, each into
    // its own vector that is appended afterwards: same tree, same node order as the serial build.
    int32_t build(std::vector<TmpNode>& tmp, uint32_t lo, uint32_t hi, uint32_t depth, bool force_split,
                  int fork_levels) {
        TmpNode node;
        node.box = empty_box();
        Box cb = empty_box();
        for (uint32_t i = lo; i < hi; i++) {
            node.box = unite(node.box, prims[i].box);
            for (int a = 0; a < 3; a++) {
                cb.mn[a] = fminf(cb.mn[a], prims[i].c[a]);
                cb.mx[a] = fmaxf(cb.mx[a], prims[i].c[a]);
            }
        }
        uint32_t n = hi - lo;
        int32_t me = (int32_t)tmp.size();
        tmp.push_back(node);
        auto make_leaf = [&]() {
            tmp[me].first = lo;
            tmp[me].count = n;
            return me;
        };
        if (n == 1) return make_leaf();

        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_plane = -1;
        float parent_area = half_area(node.box);
        bool balanced_only = depth + 24 >= kMaxDepth;  // deep chains: fall back to median splits
        if (!balanced_only && parent_area > 0.0f) {
            for (int a = 0; a < 3; a++) {
                float ext = cb.mx[a] - cb.mn[a];
                if (!(ext > 0.0f)) continue;
                float scale = (float)kBins / ext;
                uint32_t cnt[kBins] = {};
                Box bb[kBins];
                for (int b = 0; b < kBins; b++) bb[b] = empty_box();
                for (uint32_t i = lo; i < hi; i++) {
                    int b = std::min(kBins - 1, (int)((prims[i].c[a] - cb.mn[a]) * scale));
                    cnt[b]++;
                    bb[b] = unite(bb[b], prims[i].box);
                }
                float right_area[kBins];
                uint32_t right_cnt[kBins];
                Box acc = empty_box();
                uint32_t c = 0;
                for (int b = kBins - 1; b > 0; b--) {
                    acc = unite(acc, bb[b]);
                    c += cnt[b];
                    right_area[b] = c ? half_area(acc) : 0.0f;
                    right_cnt[b] = c;
                }
                acc = empty_box();
                c = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    acc = unite(acc, bb[b]);
                    c += cnt[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = kCostNode + par.cost_sphere *
                                                 (half_area(acc) * (float)c +
                                                  right_area[b + 1] * (float)right_cnt[b + 1]) /
                                                 parent_area;
                    if (cost < best_cost) best_cost = cost, best_axis = a, best_plane = b;
                }
            }
        }
        float leaf_cost = par.cost_sphere * (float)n;
        if (!force_split && n <= par.max_leaf && (best_axis < 0 || leaf_cost <= best_cost))
            return make_leaf();

        uint32_t mid;
        if (best_axis >= 0) {
            float ext = cb.mx[best_axis] - cb.mn[best_axis];
            float scale = (float)kBins / ext;
            float lo_c = cb.mn[best_axis];
            int axis = best_axis, plane = best_plane;
            auto it = std::partition(prims.begin() + lo, prims.begin() + hi, [=](const Prim& p) {
                int b = std::min(kBins - 1, (int)((p.c[axis] - lo_c) * scale));
                return b <= plane;
            });
            mid = (uint32_t)(it - prims.begin());
        } else {
            mid = lo;
        }
        if (mid == lo || mid == hi) {
            // coincident centroids (or the depth guard): split the range in the middle along the
            // widest centroid axis
            int axis = 0;
            float e0 = cb.mx[0] - cb.mn[0], e1 = cb.mx[1] - cb.mn[1], e2 = cb.mx[2] - cb.mn[2];
            if (e1 > e0 && e1 >= e2) axis = 1;
            if (e2 > e0 && e2 > e1) axis = 2;
            mid = lo + n / 2;
            std::nth_element(prims.begin() + lo, prims.begin() + mid, prims.begin() + hi,
                             [axis](const Prim& x, const Prim& y) { return x.c[axis] < y.c[axis]; });
        }
        if (fork_levels > 0 && n >= (1u << 16)) {
            std::vector<TmpNode> right_nodes;
            std::future<int32_t> right;
            try {
                right = std::async(std::launch::async, [&, mid, hi, depth, fork_levels]() {
                    return build(right_nodes, mid, hi, depth + 1, false, fork_levels - 1);
                });
            } catch (const std::system_error&) {
                // no thread to be had (a process or thread limit): this subtree is built here; the tree is the same
            }
            int32_t l = build(tmp, lo, mid, depth + 1, false, fork_levels - 1);
            int32_t r = right.valid() ? right.get() : build(right_nodes, mid, hi, depth + 1, false, 0);
            const int32_t shift = (int32_t)tmp.size();
            for (TmpNode& t : right_nodes)
                if (t.count == 0) t.left += shift, t.right += shift;
            tmp.insert(tmp.end(), right_nodes.begin(), right_nodes.end());
            tmp[me].left = l;
            tmp[me].right = r + shift;
            return me;
        }
        int32_t l = build(tmp, lo, mid, depth + 1, false, 0);
        int32_t r = build(tmp, mid, hi, depth + 1, false, 0);
        tmp[me].left = l;
        tmp[me].right = r;
        return me;
    }

    int32_t emit(int32_t t, uint32_t depth, Bvh2& out) {
        const TmpNode& n = tmp[t];
        if (n.count > 0) {
            uint32_t first = (uint32_t)out.order.size();
            for (uint32_t i = 0; i < n.count; i++) out.order.push_back(prims[n.first + i].id);
            return leaf_ref(first, n.count);
        }
        if (depth + 1 > out.max_depth) out.max_depth = depth + 1;
        size_t me = out.nodes.size();
        out.nodes.push_back(Node64{});
        const TmpNode& a = tmp[n.left];
        const TmpNode& b = tmp[n.right];
        set_child_boxes(out.nodes[me], a.box.mn, a.box.mx, b.box.mn, b.box.mx);
        int32_t c0 = emit(n.left, depth + 1, out);
        int32_t c1 = emit(n.right, depth + 1, out);
        out.nodes[me].child0 = c0;
        out.nodes[me].child1 = c1;
        return (int32_t)me;
    }
};

}  // namespace

bool build_reference(const Sphere32* spheres, uint32_t n, int swap_mode, uint32_t swap_seed,
                     std::vector<Node48>& out, std::string& err) {
    out.clear();
    if (n < 2) {
        err = "bvh: a scene needs at least 2 spheres (the reference panics on 0 or 1)";
        return false;
    }
    for (uint32_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++)
            if (spheres[i].pos[a] != spheres[i].pos[a]) {
                err = "bvh: NaN sphere position (the reference panics while sorting)";
                return false;
            }
    RefBuilder b;
    b.sp = spheres;
    b.swap_mode = swap_mode;
    b.swap_state = swap_seed ? swap_seed : 1u;
    b.ids.resize(n);
    for (uint32_t i = 0; i < n; i++) b.ids[i] = i;
    b.arena.reserve(2ull * n);
    int32_t root = b.interior(0, n);
    out.reserve(2ull * n - 1);
    flatten_ref(b.arena, root, out);
    return true;
}

bool from_reference(const Node48* nodes, uint32_t n_nodes, uint32_t n_spheres, Bvh2& out,
                    std::vector<uint32_t>& rank, std::string& err) {
    out = Bvh2{};
    if (n_spheres < 2 || n_nodes != 2ull * n_spheres - 1) {
        err = "bvh: node count must be 2 * spheres - 1";
        return false;
    }
    if (nodes[0].type != 0u) {
        err = "bvh: root must be an interior node";
        return false;
    }
    RefConverter c{nodes, n_nodes, n_spheres, &out, std::vector<uint8_t>(n_spheres, 0), &err};
    out.nodes.reserve(n_spheres - 1);
    out.order.reserve(n_spheres);
    int32_t root;
    if (!c.conv(0, n_nodes, 0, &root)) return false;
    if (out.order.size() != n_spheres) {
        err = "bvh: not every sphere is referenced by a leaf";
        return false;
    }
    rank.assign(n_spheres, 0);
    for (uint32_t i = 0; i < n_spheres; i++) rank[out.order[i]] = i;
    return true;
}

uint32_t pick_hoisted(const Sphere32* spheres, uint32_t n) {
    const char* e = getenv("MI355PT_HOIST");
    if (n < 8 || (e && atoi(e) == 0)) return n;  // a handful of spheres gains nothing (scene2's four: 3 % slower)
    uint32_t big = 0;
    for (uint32_t i = 1; i < n; i++)
        if (spheres[i].rad > spheres[big].rad) big = i;
    Box rest = empty_box();
    for (uint32_t i = 0; i < n; i++)
        if (i != big) rest = unite(rest, sphere_box(spheres[i]));
    const Box bb = sphere_box(spheres[big]);
    for (int a = 0; a < 3; a++)  // it has to touch the rest of the scene: a huge sphere far away is entered by few rays
        if (bb.mn[a] > rest.mx[a] || bb.mx[a] < rest.mn[a]) return n;
    return half_area(bb) >= 4.0f * half_area(rest) ? big : n;
}

bool build_sah(const Sphere32* spheres, uint32_t n, Bvh2& out, std::string& err) {
    out = Bvh2{};
    if (n < 2) {
        err = "bvh: a scene needs at least 2 spheres";
        return false;
    }
    if (n > LEAF_FIRST_MASK) {
        err = "bvh: too many spheres for the leaf encoding";
        return false;
    }
    // A sphere far larger than the rest of the scene together (the ground of every shipped scene: radius 1000
    // under spheres within a dozen units) is kept out of the tree: as a leaf under the root its box would be
    // entered by nearly every ray, each paying a node visit and a leaf step to reach it; the walks test it first
    // instead, which also gives them a closest hit to cull against from the first node on.
    const uint32_t n_all = n;
    const uint32_t hoisted = pick_hoisted(spheres, n);  // n_all: none
    SahBuilder b;
    b.prims.reserve(n);
    for (uint32_t i = 0; i < n; i++) {
        if (i == hoisted) continue;
        Prim p;
        p.box = sphere_box(spheres[i]);
        for (int a = 0; a < 3; a++) p.c[a] = spheres[i].pos[a];
        p.id = i;
        b.prims.push_back(p);
    }
    n = (uint32_t)b.prims.size();
    b.tmp.reserve(2ull * n);
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    int fork_levels = 0;  // 2^levels concurrent subtrees, about as many as there are cores (at most 32)
    while ((1u << fork_levels) < std::min(hw, 32u)) fork_levels++;
    int32_t root = b.build(b.tmp, 0, n, 0, true, fork_levels);
    out.nodes.reserve(n);
    out.order.reserve(n);
    b.emit(root, 0, out);
    if (hoisted != n_all) out.order.push_back(hoisted), out.n_hoisted = 1;
    if (out.max_depth > kMaxDepth) {
        err = "bvh: tree deeper than 64 levels";
        return false;
    }
    return true;
}

void order_top_first(Bvh2& bvh, uint32_t k_top) {
    const uint32_t n = (uint32_t)bvh.nodes.size();
    if (n < 3) return;
    auto area = [&](uint32_t i) {  // surface area of the node's own box = union of its child boxes
        const Node64& nd = bvh.nodes[i];
        const float lo[3] = {std::min(nd.q0.x, nd.q1.z), std::min(nd.q0.y, nd.q1.w), std::min(nd.q0.z, nd.q2.x)};
        const float hi[3] = {std::max(nd.q0.w, nd.q2.y), std::max(nd.q1.x, nd.q2.z), std::max(nd.q1.y, nd.q2.w)};
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    std::vector<uint32_t> old_of_new;
    old_of_new.reserve(n);
    std::priority_queue<std::pair<float, uint32_t>> open;
    open.push({area(0), 0u});
    while (!open.empty() && old_of_new.size() < k_top) {
        const uint32_t i = open.top().second;
        open.pop();
        old_of_new.push_back(i);
        const int32_t c[2] = {bvh.nodes[i].child0, bvh.nodes[i].child1};
        for (int k = 0; k < 2; k++)
            if (c[k] >= 0) open.push({area((uint32_t)c[k]), (uint32_t)c[k]});
    }
    std::vector<uint32_t> todo;
    while (!open.empty()) {  // roots of the subtrees below the top, largest first
        std::vector<uint32_t> stack{open.top().second};
        open.pop();
        while (!stack.empty()) {
            const uint32_t i = stack.back();
            stack.pop_back();
            old_of_new.push_back(i);
            const Node64& nd = bvh.nodes[i];
            if (nd.child1 >= 0) stack.push_back((uint32_t)nd.child1);
            if (nd.child0 >= 0) stack.push_back((uint32_t)nd.child0);
        }
    }
    std::vector<uint32_t> new_of_old(n);
    for (uint32_t k = 0; k < n; k++) new_of_old[old_of_new[k]] = k;
    std::vector<Node64> out(n);
    for (uint32_t k = 0; k < n; k++) {
        Node64 nd = bvh.nodes[old_of_new[k]];
        if (nd.child0 >= 0) nd.child0 = (int32_t)new_of_old[(uint32_t)nd.child0];
        if (nd.child1 >= 0) nd.child1 = (int32_t)new_of_old[(uint32_t)nd.child1];
        out[k] = nd;
    }
    bvh.nodes.swap(out);
}

}  // namespace pt
