#include "scene_pack.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "pt_device.h"

namespace pt {

namespace {

// Directed rounding to binary16 by search over the 31744 non-negative finite halves.
struct HalfTable {
    std::vector<float> v;
    HalfTable() {
        v.reserve(0x7c00);
        for (uint32_t h = 0; h < 0x7c00u; h++) v.push_back(half_bits_to_float(h));
    }
};
const HalfTable& half_table() {
    static const HalfTable t;
    return t;
}
uint16_t half_mag_down(double x) {  // x >= 0: largest half <= x
    const std::vector<float>& v = half_table().v;
    size_t i = (size_t)(std::upper_bound(v.begin(), v.end(), x, [](double a, float b) { return a < (double)b; }) -
                        v.begin());
    return (uint16_t)(i - 1);
}
uint16_t half_mag_up(double x) {  // x >= 0: smallest half >= x, +inf beyond 65504
    const std::vector<float>& v = half_table().v;
    size_t i = (size_t)(std::lower_bound(v.begin(), v.end(), x, [](float a, double b) { return (double)a < b; }) -
                        v.begin());
    return (uint16_t)i;  // v.size() == 0x7c00 is the bit pattern of +inf
}
}  // namespace
uint16_t half_floor(double x) { return x >= 0.0 ? half_mag_down(x) : (uint16_t)(0x8000u | half_mag_up(-x)); }
uint16_t half_ceil(double x) { return x >= 0.0 ? half_mag_up(x) : (uint16_t)(0x8000u | half_mag_down(-x)); }
namespace {

double box_area(const double lo[3], const double hi[3]) {
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// Node64 -> Node32 around `org`.  Returns the mean surface-area ratio half box / exact box over all
// child boxes (infinity when a plane leaves the binary16 range).
double to_half_nodes(const std::vector<Node64>& in, const float org[3], std::vector<Node32>& out) {
    out.resize(in.size());
    double ratio_sum = 0.0;
    for (size_t i = 0; i < in.size(); i++) {
        const Node64& n = in[i];
        const float b[12] = {n.q0.x, n.q0.y, n.q0.z, n.q0.w, n.q1.x, n.q1.y,
                             n.q1.z, n.q1.w, n.q2.x, n.q2.y, n.q2.z, n.q2.w};
        Node32& o = out[i];
        for (int c = 0; c < 2; c++) {
            double lo[3], hi[3], hlo[3], hhi[3];
            for (int k = 0; k < 3; k++) {
                lo[k] = (double)b[c * 6 + k] - (double)org[k];
                hi[k] = (double)b[c * 6 + 3 + k] - (double)org[k];
                const uint16_t l = half_floor(lo[k]), h = half_ceil(hi[k]);
                o.h[c * 6 + k] = l;
                o.h[c * 6 + 3 + k] = h;
                hlo[k] = (double)half_bits_to_float(l);
                hhi[k] = (double)half_bits_to_float(h);
            }
            const double exact = box_area(lo, hi);
            ratio_sum += exact > 0.0 ? box_area(hlo, hhi) / exact : 1.0;
        }
        o.child0 = n.child0;
        o.child1 = n.child1;
    }
    return in.empty() ? 1.0 : ratio_sum / (2.0 * (double)in.size());
}

}  // namespace

bool validate_scene(const SceneArrays& a, std::string& err) {
    if (a.n_spheres < 2) {
        err = "scene: at least 2 spheres are required (the reference's BVH builder panics on 0 or 1)";
        return false;
    }
    if (a.n_spheres > LEAF_FIRST_MASK) {
        err = "scene: too many spheres";
        return false;
    }
    if (!a.spheres || (a.n_lights && !a.lights) || (a.n_lambertians && !a.lambertians) ||
        (a.n_metals && !a.metals) || (a.n_glass && !a.glass)) {
        err = "scene: null array with a non-zero count";
        return false;
    }
    const size_t counts[4] = {a.n_lights, a.n_lambertians, a.n_metals, a.n_glass};
    for (size_t i = 0; i < a.n_spheres; i++) {
        const Sphere32& s = a.spheres[i];
        if (!std::isfinite(s.pos[0]) || !std::isfinite(s.pos[1]) || !std::isfinite(s.pos[2]) ||
            !std::isfinite(s.rad) || s.rad < 0.0f) {
            err = "scene: sphere " + std::to_string(i) + " has a non-finite position or a bad radius";
            return false;
        }
        if (s.mat_type <= 3u && s.mat_index >= counts[s.mat_type]) {
            err = "scene: sphere " + std::to_string(i) + " has a mat_index outside its material array";
            return false;
        }
        if (s.mat_index >= (1u << 29)) {
            err = "scene: mat_index too large";
            return false;
        }
    }
    return true;
}

SceneLayout make_layout(const SceneArrays& a, uint32_t n_nodes, bool half_nodes) {
    auto align16 = [](size_t x) { return (x + 15u) & ~(size_t)15u; };
    const size_t n = a.n_spheres;
    SceneLayout L{};
    size_t off = 0;
    L.off_nodes = (uint32_t)off, L.n_nodes = n_nodes;
    off = align16(off + (size_t)n_nodes * (half_nodes ? sizeof(Node32) : sizeof(Node64)));
    L.off_spheres = (uint32_t)off, L.n_spheres = (uint32_t)n;
    off = align16(off + n * sizeof(f4));
    L.off_info = (uint32_t)off;
    off = align16(off + n * sizeof(u2));
    L.off_lights = (uint32_t)off, L.n_lights = (uint32_t)a.n_lights;
    off = align16(off + a.n_lights * 16);
    L.off_lambertians = (uint32_t)off, L.n_lambertians = (uint32_t)a.n_lambertians;
    off = align16(off + a.n_lambertians * 16);
    L.off_metals = (uint32_t)off, L.n_metals = (uint32_t)a.n_metals;
    off = align16(off + a.n_metals * 16);
    L.off_glass = (uint32_t)off, L.n_glass = (uint32_t)a.n_glass;
    off = align16(off + a.n_glass * 16);
    L.total_bytes = off > 0xfffffff0ull ? 0u : (uint32_t)off;
    L.node_half = half_nodes ? 1u : 0u;
    return L;
}

// Tie ranks without the reference's tree: the spheres' stable order by z (three 11-bit passes of a radix sort over
// keys that order like the floats, -0 with +0) and reference_leaf_rank.
static void tie_ranks_by_z(const Sphere32* spheres, uint32_t n, std::vector<uint32_t>& rank) {
    rank.resize(n);
    if (n < 4u) {  // the reference sorts nothing below four spheres
        for (uint32_t i = 0; i < n; i++) rank[i] = reference_leaf_rank(n, i);
        return;
    }
    std::vector<uint64_t> a(n), b(n);  // key << 32 | sphere index
    for (uint32_t i = 0; i < n; i++) {
        const float z = spheres[i].pos[2] + 0.0f;
        uint32_t bits;
        memcpy(&bits, &z, 4);
        const uint32_t key = (bits & 0x80000000u) ? ~bits : bits | 0x80000000u;
        a[i] = (uint64_t)key << 32 | i;
    }
    for (int shift = 32; shift < 64; shift += 11) {
        size_t count[2049] = {0};
        for (uint32_t i = 0; i < n; i++) count[((a[i] >> shift) & 2047u) + 1u]++;
        for (int k = 0; k < 2048; k++) count[k + 1] += count[k];
        for (uint32_t i = 0; i < n; i++) b[count[(a[i] >> shift) & 2047u]++] = a[i];
        a.swap(b);
    }
    for (uint32_t p = 0; p < n; p++) rank[(uint32_t)a[p]] = reference_leaf_rank(n, p);
}

bool pack_scene(const SceneArrays& a, bool reference_topology, const Node48* ref_nodes, size_t n_ref_nodes,
                std::vector<char>& blob, SceneLayout& L, Bvh2& bvh, std::string& err) {
    if (!validate_scene(a, err)) return false;
    const uint32_t n = (uint32_t)a.n_spheres;
    std::vector<uint32_t> rank;
    std::vector<Node48> own_ref;
    Bvh2 from_ref;
    // Tie ranks -- which of two spheres with bit-identical roots wins -- are the spheres' places among the leaves of
    // the reference's tree: read off that tree when it is walked (a caller's, or the library's own copy of it), else
    // worked out from the spheres' order by z, which is all that tree's shape depends on.
    const bool caller_tree = ref_nodes != nullptr;
    if (!caller_tree && reference_topology) {
        if (!build_reference(a.spheres, n, 0, 1, own_ref, err)) return false;
        ref_nodes = own_ref.data();
        n_ref_nodes = own_ref.size();
    }
    if (ref_nodes) {
        if (n_ref_nodes > 0xffffffffull || !from_reference(ref_nodes, (uint32_t)n_ref_nodes, n, from_ref, rank, err))
            return false;
    } else {
        tie_ranks_by_z(a.spheres, n, rank);
    }
    if (caller_tree || reference_topology) {
        bvh = std::move(from_ref);
    } else if (!build_sah(a.spheres, n, bvh, err)) {
        return false;
    }

    auto align16 = [](size_t x) { return (x + 15u) & ~(size_t)15u; };
    const size_t rest_bytes = align16((size_t)n * sizeof(f4)) + align16((size_t)n * sizeof(u2)) +
                              align16(a.n_lights * 16) + align16(a.n_lambertians * 16) + align16(a.n_metals * 16) +
                              align16(a.n_glass * 16);
    const size_t stack256 = (size_t)std::max(1u, bvh.max_depth) * 256u * 4u;
    // Half-precision boxes for a scene that fits the LDS budget only with them (a walk over LDS
    // sustains several times the record rate of one over the L1/L2 caches), provided rounding the
    // planes outwards inflates the boxes by less than 10 % on average.  A scene that fits with full
    // floats keeps them: the mixed-precision FMA issues slower than the plain one.
    // MI355PT_NODES=float|half overrides the choice (development switch).
    std::vector<Node32> half_nodes;
    L = SceneLayout{};
    const char* nenv = getenv("MI355PT_NODES");
    const bool force_half = nenv && strcmp(nenv, "half") == 0, force_float = nenv && strcmp(nenv, "float") == 0;
    const bool lds_with_float = lds_nodes_bytes((uint32_t)bvh.nodes.size()) + rest_bytes + stack256 <= LDS_SCENE_LIMIT;
    const bool lds_with_half = bvh.nodes.size() * sizeof(Node32) + rest_bytes + stack256 <= LDS_SCENE_LIMIT;
    const bool reference_walk = caller_tree || reference_topology;
    if (!reference_walk && lds_with_half && !force_float && (!lds_with_float || force_half)) {
        std::vector<float> axis(n);
        for (int k = 0; k < 3; k++) {  // origin: median sphere centre, where the detail is
            for (uint32_t i = 0; i < n; i++) axis[i] = a.spheres[i].pos[k];
            std::nth_element(axis.begin(), axis.begin() + n / 2, axis.end());
            L.box_org[k] = axis[n / 2];
        }
        const double inflation = to_half_nodes(bvh.nodes, L.box_org, half_nodes);
        L.node_half = (std::isfinite(inflation) && (inflation <= 1.10 || force_half)) ? 1u : 0u;
        if (!L.node_half) L.box_org[0] = L.box_org[1] = L.box_org[2] = 0.0f;
    }
    // A tree that stays in global memory is renumbered so that any prefix of the node array is the
    // most frequently visited top of the tree; the kernels keep such a prefix in LDS.
    if ((!lds_with_float || reference_walk) && !L.node_half) order_top_first(bvh, 4096);
    {
        const SceneLayout sized = make_layout(a, (uint32_t)bvh.nodes.size(), L.node_half != 0);
        if (sized.total_bytes == 0) {
            err = "scene: device blob exceeds 4 GiB";
            return false;
        }
        const float org[3] = {L.box_org[0], L.box_org[1], L.box_org[2]};
        L = sized;
        L.box_org[0] = org[0], L.box_org[1] = org[1], L.box_org[2] = org[2];
    }
    L.max_depth = bvh.max_depth;
    L.n_hoisted = bvh.n_hoisted;
    L.reference_walk = reference_walk ? 1u : 0u;
    const size_t off = L.total_bytes;
    blob.assign(off, 0);
    if (L.node_half) memcpy(blob.data() + L.off_nodes, half_nodes.data(), half_nodes.size() * sizeof(Node32));
    else memcpy(blob.data() + L.off_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(Node64));
    f4* sp = reinterpret_cast<f4*>(blob.data() + L.off_spheres);
    u2* info = reinterpret_cast<u2*>(blob.data() + L.off_info);
    for (uint32_t i = 0; i < n; i++) {
        const Sphere32& s = a.spheres[bvh.order[i]];
        sp[i] = f4{s.pos[0], s.pos[1], s.pos[2], s.rad * s.rad};
        const uint32_t type = s.mat_type > 3u ? 4u : s.mat_type;
        info[i] = u2{type | (s.mat_index << 3), rank[bvh.order[i]]};
    }
    if (a.n_lights) memcpy(blob.data() + L.off_lights, a.lights, a.n_lights * 16);
    if (a.n_lambertians) memcpy(blob.data() + L.off_lambertians, a.lambertians, a.n_lambertians * 16);
    if (a.n_metals) memcpy(blob.data() + L.off_metals, a.metals, a.n_metals * 16);
    f4* glass = reinterpret_cast<f4*>(blob.data() + L.off_glass);
    for (size_t i = 0; i < a.n_glass; i++) glass[i] = glass_record(a.glass[i]);
    return true;
}

}  // namespace pt
