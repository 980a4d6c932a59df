// Turns a scene in wire format into the single device blob the kernels read: picks / converts the
// tree, derives the tie ranks, reorders the spheres into leaf order.  Host only.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "bvh.h"
#include "pt_types.h"

namespace pt {

struct SceneArrays {
    const Sphere32* spheres = nullptr;
    size_t n_spheres = 0;
    const Vec16* lights = nullptr;
    size_t n_lights = 0;
    const Vec16* lambertians = nullptr;
    size_t n_lambertians = 0;
    const Vec16* metals = nullptr;
    size_t n_metals = 0;
    const float* glass = nullptr;
    size_t n_glass = 0;
};

// What the reference would reject by panicking or what its robust buffer access would hide.
bool validate_scene(const SceneArrays& a, std::string& err);

// reference_topology: walk the reference's own tree shape (built here, or ref_nodes when given)
// instead of the SAH tree.  ref_nodes (optional) is a caller-built flattened reference tree; it
// always supplies the tie ranks and, being the caller's explicit choice, the topology.
// binary16 bit patterns of the largest half <= x / the smallest half >= x (-inf / +inf beyond the
// finite range): the directed rounding that keeps half-precision boxes conservative.
uint16_t half_floor(double x);
uint16_t half_ceil(double x);

// Section offsets of the device blob for a tree of n_nodes interior nodes (total_bytes = 0 when the blob
// would exceed 4 GiB); max_depth and box_org are left for the caller.
SceneLayout make_layout(const SceneArrays& a, uint32_t n_nodes, bool half_nodes);

bool pack_scene(const SceneArrays& a, bool reference_topology, const Node48* ref_nodes, size_t n_ref_nodes,
                std::vector<char>& blob, SceneLayout& layout, Bvh2& bvh, std::string& err);

}  // namespace pt
