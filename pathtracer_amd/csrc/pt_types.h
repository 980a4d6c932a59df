// Byte layouts at the boundary (exactly what the reference uploads to its storage / uniform
// buffers) and the internal scene layout the kernels consume.
#pragma once
#include <cstddef>
#include <cstdint>

namespace pt {

// ---- reference wire formats (little-endian) ------------------------------------------------
struct Sphere32 {  // reference src/renderer/sphere.rs:26-43
    float pos[3];
    float rad;
    uint32_t mat_type;
    uint32_t mat_index;
    uint32_t pad[2];
};
struct Node48 {  // reference src/renderer/bvh.rs:20-36
    float mn[3];
    uint32_t pad0;
    float mx[3];
    uint32_t pad1;
    uint32_t type;   // 0 = interior, index = escape; 1 = leaf, index = sphere id (bvh.rs:10-11)
    uint32_t index;
    uint32_t pad2[2];
};
struct Vec16 {  // Light{colour}, Lambertian{albedo}, Metal{albedo, roughness}: materials.rs:26-97
    float x, y, z, w;
};
struct Camera64 {  // reference src/renderer/camera.rs:91-113
    float pos[3], pad0;
    float horizontal[3], pad1;
    float vertical[3], pad2;
    float center[3], pad3;
};
struct Globals12 {  // reference src/renderer/globals.rs:3-9
    uint32_t seed;
    int32_t samples;
    int32_t depth;
};
static_assert(sizeof(Sphere32) == 32, "Sphere is 32 bytes on the wire");
static_assert(sizeof(Node48) == 48, "BVHNode is 48 bytes on the wire");
static_assert(sizeof(Vec16) == 16, "materials are 16 bytes on the wire");
static_assert(sizeof(Camera64) == 64, "Camera is 64 bytes on the wire");
static_assert(sizeof(Globals12) == 12, "Globals is 12 bytes on the wire");

enum : uint32_t { MAT_LIGHT = 0, MAT_LAMBERTIAN = 1, MAT_METAL = 2, MAT_GLASS = 3 };  // materials.rs:3-8

// ---- internal device layout ----------------------------------------------------------------
struct alignas(16) f4 {
    float x, y, z, w;
};
struct alignas(8) u2 {
    uint32_t x, y;
};
struct alignas(16) u4 {
    uint32_t x, y, z, w;
};

// One interior node of the binary BVH the kernels walk: both children's boxes and both child
// links in one 64-byte record (four 16-byte loads).
//   q0 = c0.lo.xyz, c0.hi.x   q1 = c0.hi.yz, c1.lo.xy   q2 = c1.lo.z, c1.hi.xyz
//   q3 = child0, child1 (int bits), 0, 0
// child >= 0: interior node index.  child < 0: leaf, ~child = first | (count-1) << 28 over the
// leaf-ordered sphere array.
struct alignas(16) Node64 {
    f4 q0, q1, q2;
    int32_t child0, child1;
    uint32_t pad[2];
};
static_assert(sizeof(Node64) == 64, "Node64");
// The same node with binary16 box planes, relative to SceneLayout::box_org (two 16-byte loads):
//   h[0..11] = c0.lo.xyz, c0.hi.xyz, c1.lo.xyz, c1.hi.xyz   then child0, child1.
// Lower planes are rounded down and upper planes up, so a half box always contains the exact one;
// boxes only cull, which keeps every hit and every pixel value unchanged.  Used for scenes that are
// staged in LDS, where node bytes per visit are what bounds the walk.
struct alignas(16) Node32 {
    uint16_t h[12];
    int32_t child0, child1;
};
static_assert(sizeof(Node32) == 32, "Node32");
constexpr int LEAF_COUNT_SHIFT = 28;
constexpr uint32_t LEAF_FIRST_MASK = (1u << LEAF_COUNT_SHIFT) - 1u;
constexpr int MAX_LEAF_SPHERES = 4;

// Scene blob: one contiguous, 16-byte aligned allocation; offsets in bytes.
struct SceneLayout {
    uint32_t off_nodes, n_nodes;      // Node64[] or, with node_half, Node32[]
    uint32_t node_half;               // 1: nodes are Node32
    uint32_t reference_walk;          // 1: the tree is the reference's (or a caller's) and is walked the reference's
                                      // way: its box arithmetic, no pruning against the closest hit, no box of its
                                      // own for a leaf (a leaf child carries its parent's box)
    float box_org[3];                 // origin the Node32 planes are relative to
    uint32_t off_spheres, n_spheres;  // f4[]  (x, y, z, rad * rad) in leaf order
    uint32_t off_info;                // u2[]  (min(mat_type,4) | mat_index << 3, tie rank) in leaf order
    uint32_t off_lights, n_lights;    // f4[]
    uint32_t off_lambertians, n_lambertians;
    uint32_t off_metals, n_metals;    // f4[]  (albedo, roughness)
    uint32_t off_glass, n_glass;      // f4[]  (ir, 1 / ir, r0(ir), r0(1 / ir)), see glass_record
    uint32_t total_bytes;
    uint32_t max_depth;               // deepest interior-node chain; bounds the traversal stack
    uint32_t n_hoisted;               // 1: the last sphere of the leaf order is in no leaf; every ray that walks the
                                      // tree tests it first (a sphere far larger than everything else -- the ground
                                      // of every shipped scene -- whose box no ray of the scene misses)
};

#if defined(__HIPCC__)
#define PT_TYPES_FN __host__ __device__ inline
#else
#define PT_TYPES_FN inline
#endif
// Position, among the leaves of the reference's tree in the order its walk reaches them (bvh.rs:51-140 without child
// swaps, flattened left first: compute.wgsl:236-268 visits leaves in that order and a later leaf replaces an equal
// t, :129), of the sphere at position p of the n spheres sorted by z.  The reference sorts every range by z
// (bvh.rs:82-96: the axis choice compares sentinel extents, so it is always z) with a stable sort, which leaves a
// sorted range as it is: the tree's shape, and so this position, depends on n and p alone.  Ranges of two keep their
// order, of three put their last element first (bvh.rs:63-80), longer ones split at n / 2.  Scenes of fewer than four
// spheres are never sorted: p is then the sphere's index.
PT_TYPES_FN uint32_t reference_leaf_rank(uint32_t n, uint32_t p) {
    uint32_t lo = 0u, hi = n, base = 0u;
    for (;;) {
        const uint32_t m = hi - lo;
        if (m <= 2u) return base + (p - lo);
        if (m == 3u) return p == hi - 1u ? base : base + 1u + (p - lo);
        const uint32_t mid = lo + m / 2u;
        if (p < mid) hi = mid;
        else base += mid - lo, lo = mid;
    }
}

// A scene whose blob plus the traversal stacks of a 256-lane workgroup stays within this budget is
// staged in LDS by every workgroup (two or more workgroups per CU still fit).
constexpr uint32_t LDS_SCENE_LIMIT = 64u * 1024u;
// In LDS a Node64 is kept as an 80-byte record whose planes are laid out per axis as
// (lo0, lo1, hi0, hi1, lo0, lo1): a ray reads four consecutive floats starting at the first or the
// third, by the sign of its direction on that axis, and gets (near0, near1, far0, far1) without any
// min / max.  The kernels convert while they stage the scene; the blob itself keeps Node64.
// The record is 80 bytes; records lie 88 bytes apart.  With 80 (20 words) a record can start in 16 of the 64
// banks only; 88 (22 words) reaches 32 and the lanes' scattered record fetches collide less: scene1, 256 frames,
// 18.97 ms at a stride of 80, 18.59 at 88, 19.01 at 96, 18.54 at 104, 18.96 at 112 (8-byte alignment is needed
// by the 8-byte fetches, so an odd word count is not available).
#ifndef PT_LDS_NODE_BYTES
#define PT_LDS_NODE_BYTES 88
#endif
constexpr uint32_t LDS_NODE_BYTES = PT_LDS_NODE_BYTES;  // a multiple of 8, at least 80
// Bytes the n records take in LDS: the sections behind them stay 16-byte aligned.
constexpr uint32_t lds_nodes_bytes(uint32_t n) { return (n * LDS_NODE_BYTES + 15u) & ~15u; }
inline uint32_t lds_scene_bytes(const SceneLayout& L) {
    const uint32_t blob = (L.total_bytes + 15u) & ~15u;
    return (L.node_half || L.reference_walk) ? blob : blob + lds_nodes_bytes(L.n_nodes) - L.n_nodes * (uint32_t)sizeof(Node64);
}

}  // namespace pt
