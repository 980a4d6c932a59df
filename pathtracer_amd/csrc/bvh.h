// Host-side BVH construction.
//   * build_reference(): the reference's own tree (src/renderer/bvh.rs:51-140) in its 48-byte
//     flattened wire format, so a caller can get byte-identical nodes to what the reference uploads.
//   * from_reference(): converts such a flattened tree (ours or a caller's) to the kernels' layout,
//     keeping its topology, and extracts the tie rank (leaf order) of every sphere.
//   * build_sah(): binned-SAH tree with small multi-sphere leaves, the default for rendering.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "pt_types.h"

namespace pt {

struct Bvh2 {
    std::vector<Node64> nodes;    // node 0 is the root; always interior (a scene has >= 2 spheres)
    std::vector<uint32_t> order;  // leaf-ordered sphere ids: leaf (first,count) indexes this array
    uint32_t max_depth = 0;       // interior nodes on the longest root-to-leaf chain
    uint32_t n_hoisted = 0;       // 1: order.back() is in no leaf (SceneLayout::n_hoisted)
};

// The sphere that stays out of the tree (SceneLayout::n_hoisted), or n when there is none: the largest one, if its
// padded box has at least four times the surface of the box around all the others and meets that box (scenes of at
// least eight spheres).  MI355PT_HOIST=0: none.
uint32_t pick_hoisted(const Sphere32* spheres, uint32_t n);

// Renumbers the interior nodes: the first min(k_top, n) indices go to the nodes with the largest
// boxes, parents before children (every prefix is a connected top of the tree); the rest follow in
// depth-first order.  The root stays node 0; leaf links and the sphere order are unchanged.
void order_top_first(Bvh2& bvh, uint32_t k_top);

// swap_mode: 0 never swap children, 1 always, 2 one xorshift32 bit per 4+-sphere node (post-order,
// where the reference calls rand::random(), bvh.rs:107).  Returns false when n < 2 (the reference
// panics, bvh.rs:61-62) or a position is NaN (partial_cmp().unwrap() panics, bvh.rs:91-95).
bool build_reference(const Sphere32* spheres, uint32_t n, int swap_mode, uint32_t swap_seed,
                     std::vector<Node48>& out, std::string& err);

// Validates a flattened reference tree over n_spheres spheres and converts it.  rank[s] = position of
// sphere s among the leaves in flattened order.
bool from_reference(const Node48* nodes, uint32_t n_nodes, uint32_t n_spheres, Bvh2& out,
                    std::vector<uint32_t>& rank, std::string& err);

bool build_sah(const Sphere32* spheres, uint32_t n, Bvh2& out, std::string& err);

}  // namespace pt
