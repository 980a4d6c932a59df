#include "scene_pack.h"

#include <cmath>
#include <cstring>

namespace pt {

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

bool pack_scene(const SceneArrays& a, bool reference_topology, const Node48* ref_nodes, size_t n_ref_nodes,
                std::vector<char>& blob, SceneLayout& L, Bvh2& bvh, std::string& err) {
    if (!validate_scene(a, err)) return false;
    const uint32_t n = (uint32_t)a.n_spheres;
    std::vector<uint32_t> rank;
    std::vector<Node48> own_ref;
    Bvh2 from_ref;
    // Tie ranks come from the reference-order walk.  Above this size the reference tree is not built
    // just for ranks: two spheres with bit-identical roots then resolve by sphere index.
    const uint32_t kRankLimit = 4u << 20;
    const bool caller_tree = ref_nodes != nullptr;
    if (!caller_tree && (reference_topology || n <= kRankLimit)) {
        if (!build_reference(a.spheres, n, 0, 1, own_ref, err)) return false;
        ref_nodes = own_ref.data();
        n_ref_nodes = own_ref.size();
    }
    if (ref_nodes) {
        if (n_ref_nodes > 0xffffffffull || !from_reference(ref_nodes, (uint32_t)n_ref_nodes, n, from_ref, rank, err))
            return false;
    } else {
        rank.resize(n);
        for (uint32_t i = 0; i < n; i++) rank[i] = i;
    }
    if (caller_tree || reference_topology) {
        bvh = std::move(from_ref);
    } else if (!build_sah(a.spheres, n, bvh, err)) {
        return false;
    }

    auto align16 = [](size_t x) { return (x + 15u) & ~(size_t)15u; };
    size_t off = 0;
    L = SceneLayout{};
    L.off_nodes = (uint32_t)off, L.n_nodes = (uint32_t)bvh.nodes.size();
    off = align16(off + bvh.nodes.size() * sizeof(Node64));
    L.off_spheres = (uint32_t)off, L.n_spheres = n;
    off = align16(off + (size_t)n * sizeof(f4));
    L.off_info = (uint32_t)off;
    off = align16(off + (size_t)n * sizeof(u2));
    L.off_lights = (uint32_t)off, L.n_lights = (uint32_t)a.n_lights;
    off = align16(off + a.n_lights * 16);
    L.off_lambertians = (uint32_t)off, L.n_lambertians = (uint32_t)a.n_lambertians;
    off = align16(off + a.n_lambertians * 16);
    L.off_metals = (uint32_t)off, L.n_metals = (uint32_t)a.n_metals;
    off = align16(off + a.n_metals * 16);
    L.off_glass = (uint32_t)off, L.n_glass = (uint32_t)a.n_glass;
    off = align16(off + a.n_glass * 4);
    if (off > 0xfffffff0ull) {
        err = "scene: device blob exceeds 4 GiB";
        return false;
    }
    L.total_bytes = (uint32_t)off;
    L.max_depth = bvh.max_depth;
    blob.assign(off, 0);
    memcpy(blob.data() + L.off_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(Node64));
    f4* sp = reinterpret_cast<f4*>(blob.data() + L.off_spheres);
    u2* info = reinterpret_cast<u2*>(blob.data() + L.off_info);
    for (uint32_t i = 0; i < n; i++) {
        const Sphere32& s = a.spheres[bvh.order[i]];
        sp[i] = f4{s.pos[0], s.pos[1], s.pos[2], s.rad};
        const uint32_t type = s.mat_type > 3u ? 4u : s.mat_type;
        info[i] = u2{type | (s.mat_index << 3), rank[bvh.order[i]]};
    }
    if (a.n_lights) memcpy(blob.data() + L.off_lights, a.lights, a.n_lights * 16);
    if (a.n_lambertians) memcpy(blob.data() + L.off_lambertians, a.lambertians, a.n_lambertians * 16);
    if (a.n_metals) memcpy(blob.data() + L.off_metals, a.metals, a.n_metals * 16);
    if (a.n_glass) memcpy(blob.data() + L.off_glass, a.glass, a.n_glass * 4);
    return true;
}

}  // namespace pt
