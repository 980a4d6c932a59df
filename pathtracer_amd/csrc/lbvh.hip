// BVH construction on the GPU: Morton codes of the sphere centres, radix sort, the parallel
// hierarchy of Karras ("Maximizing parallelism in the construction of BVHs, octrees and k-d trees",
// HPG 2012), bottom-up box refit, then the device scene blob in the layout the path kernels walk.
// This is the fast path for Renderer::reload_scene (mod.rs:148-155): the reference rebuilds its tree on
// the CPU with a sort per level (bvh.rs:51-113).  The tree is of lower quality than the host's binned
// SAH (median-like splits, leaves of at most two spheres): what is traced is the same, it takes more node visits.
#include <cstring>

#include <hip/hip_runtime.h>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <string>
#include <vector>

#include "gpu.h"
#include "gpu_state.h"
#include "bvh.h"
#include "scene_pack.h"
#include "pt_device.h"

namespace pt {
namespace gpu {

namespace {

constexpr int BLK = 256;

// order-preserving float <-> int, for atomicMin / atomicMax on floats
__device__ __forceinline__ int ordered(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float unordered(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// box[0..2] = min of the centres, box[3..5] = max, as ordered ints
__global__ void lbvh_bounds(const Sphere32* s, uint32_t n, int* box) {
    const uint32_t i = blockIdx.x * BLK + threadIdx.x;
    int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    if (i < n)
        for (int k = 0; k < 3; k++) lo[k] = hi[k] = ordered(s[i].pos[k]);
    for (int k = 0; k < 3; k++) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[k] = min(lo[k], __shfl_down(lo[k], o));
            hi[k] = max(hi[k], __shfl_down(hi[k], o));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&box[k], lo[k]);
            atomicMax(&box[3 + k], hi[k]);
        }
    }
}

__device__ __forceinline__ uint64_t spread21(uint32_t v) {  // bit k of v -> bit 3k
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void lbvh_codes(const Sphere32* s, uint32_t n, const int* box, uint64_t* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * BLK + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
    for (int k = 0; k < 3; k++) {
        const float lo = unordered(box[k]), hi = unordered(box[3 + k]);
        const float ext = hi - lo;
        const float u = ext > 0.0f ? (s[i].pos[k] - lo) / ext : 0.0f;
        q[k] = (uint32_t)fminf(fmaxf(u * 2097152.0f, 0.0f), 2097151.0f);
    }
    keys[i] = spread21(q[0]) << 2 | spread21(q[1]) << 1 | spread21(q[2]);
    vals[i] = i;
}

// length of the common prefix of the keys at sorted positions i and j (ties broken by position)
__device__ __forceinline__ int common_prefix(const uint64_t* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t x = keys[i] ^ keys[j];
    return x ? __clzll((long long)x) : 64 + __clz(i ^ j);
}

// Internal node i of n-1 (node 0 is the root): its children as links (>= 0 internal, ~p leaf at sorted
// position p) and the parent of every child.
__global__ void lbvh_hierarchy(const uint64_t* keys, int n, int32_t* child0, int32_t* child1, int32_t* parent_of_node,
                               int32_t* parent_of_leaf, uint32_t* first_of, uint32_t* count_of, int leaf_max, int* kept_count) {
    const int i = (int)(blockIdx.x * BLK + threadIdx.x);
    if (i >= n - 1) return;
    const int d = common_prefix(keys, n, i, i + 1) - common_prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = common_prefix(keys, n, i, i - d);
    int lmax = 2;
    while (common_prefix(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (common_prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = common_prefix(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (common_prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    if (lo == gamma) {
        child0[i] = ~gamma;
        parent_of_leaf[gamma] = i;
    } else {
        child0[i] = gamma;
        parent_of_node[gamma] = i;
    }
    if (hi == gamma + 1) {
        child1[i] = ~(gamma + 1);
        parent_of_leaf[gamma + 1] = i;
    } else {
        child1[i] = gamma + 1;
        parent_of_node[gamma + 1] = i;
    }
    if (i == 0) parent_of_node[0] = -1;
    // the sorted positions below this node; a node over at most leaf_max spheres becomes one leaf of its
    // parent (the root always stays a node)
    first_of[i] = (uint32_t)lo, count_of[i] = (uint32_t)(hi - lo + 1);
    if (i == 0 || hi - lo + 1 > leaf_max) atomicAdd(kept_count, 1);
}

struct Box {
    float lo[3], hi[3];
};
// BoundingBox::from(&Sphere) (bounding_box.rs:17-27): min = pos - rad - 0.01, max = pos + rad + 0.01
__device__ __forceinline__ Box sphere_box(const Sphere32& s) {
    Box b;
    for (int k = 0; k < 3; k++) {
        b.lo[k] = (s.pos[k] - s.rad) - 0.01f;
        b.hi[k] = (s.pos[k] + s.rad) + 0.01f;
    }
    return b;
}
__device__ __forceinline__ float coherent_load(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One thread per leaf walks towards the root; at every node the second thread to arrive merges the
// boxes of both children and goes on, the first one stops.  node_box: 6 floats per internal node.
__global__ void lbvh_refit(const Sphere32* spheres, const uint32_t* order, int n, const int32_t* child0,
                           const int32_t* child1, const int32_t* parent_of_node, const int32_t* parent_of_leaf,
                           int* arrived, float* node_box, int* max_depth) {
    const int p = (int)(blockIdx.x * BLK + threadIdx.x);
    if (p >= n) return;
    int depth = 0;
    for (int up = parent_of_leaf[p]; up >= 0; up = parent_of_node[up]) depth++;
    atomicMax(max_depth, depth);
    int cur = parent_of_leaf[p];
    while (cur >= 0) {
        __threadfence();
        if (atomicAdd(&arrived[cur], 1) == 0) return;
        Box b;
        for (int k = 0; k < 3; k++) b.lo[k] = INFINITY, b.hi[k] = -INFINITY;
        const int32_t c[2] = {child0[cur], child1[cur]};
        for (int side = 0; side < 2; side++) {
            Box cb;
            if (c[side] < 0) {
                cb = sphere_box(spheres[order[~c[side]]]);
            } else {
                for (int k = 0; k < 3; k++) {
                    cb.lo[k] = coherent_load(node_box + (size_t)c[side] * 6 + k);
                    cb.hi[k] = coherent_load(node_box + (size_t)c[side] * 6 + 3 + k);
                }
            }
            for (int k = 0; k < 3; k++) b.lo[k] = fminf(b.lo[k], cb.lo[k]), b.hi[k] = fmaxf(b.hi[k], cb.hi[k]);
        }
        for (int k = 0; k < 3; k++) node_box[(size_t)cur * 6 + k] = b.lo[k], node_box[(size_t)cur * 6 + 3 + k] = b.hi[k];
        cur = parent_of_node[cur];
    }
}

// Renumbering so that the head of the node array is the top of the tree (what the path kernels keep in
// LDS for scenes in global memory): the `top` nodes with the largest boxes come first, largest first;
// all others keep their relative order, which is spatially coherent (node i sits next to leaf i).
// Sort key of node i: its box's surface area, inverted so that an ascending sort puts the largest first
// (areas are non-negative, so their bit patterns order like the values; the sort is stable and the root
// is node 0, so the root stays first).
__global__ void lbvh_area_keys(const float* node_box, const uint32_t* count_of, int leaf_max, int n_nodes, uint32_t* keys,
                               uint32_t* vals) {
    const int i = (int)(blockIdx.x * BLK + threadIdx.x);
    if (i >= n_nodes) return;
    const float* b = node_box + (size_t)i * 6;
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    const bool kept = i == 0 || (int)count_of[i] > leaf_max;
    // boxes are padded, so a kept node's area is positive and its key below that of the dropped ones
    keys[i] = kept ? ~__float_as_uint(dx * dy + dy * dz + dz * dx) : 0xffffffffu;
    vals[i] = (uint32_t)i;
}
// by_area[r] = node with the r-th largest box.  below[i] = 1 for nodes outside the top.
__global__ void lbvh_mark_top(const uint32_t* by_area, int n_nodes, int top, int kept, uint32_t* below) {
    const int r = (int)(blockIdx.x * BLK + threadIdx.x);
    if (r >= n_nodes) return;
    below[by_area[r]] = (r >= top && r < kept) ? 1u : 0u;  // the first `kept` entries are the kept nodes
}
// new_id of a top node = its rank by area; of any other node = top + number of non-top nodes before it.
__global__ void lbvh_new_ids(const uint32_t* by_area, const uint32_t* below, const uint32_t* below_before, int n_nodes,
                             int top, uint32_t* new_id) {
    const int r = (int)(blockIdx.x * BLK + threadIdx.x);
    if (r >= n_nodes) return;
    if (r < top) new_id[by_area[r]] = (uint32_t)r;
    if (below[r]) new_id[r] = (uint32_t)top + below_before[r];
}

// The 64-byte records the path kernels walk: both child boxes and both links per kept internal node.
__global__ void lbvh_emit(const Sphere32* spheres, const uint32_t* order, int n, const int32_t* child0,
                          const int32_t* child1, const float* node_box, const uint32_t* new_id, const uint32_t* first_of,
                          const uint32_t* count_of, int leaf_max, Node64* out) {
    const int i = (int)(blockIdx.x * BLK + threadIdx.x);
    if (i >= n - 1) return;
    if (i != 0 && (int)count_of[i] <= leaf_max) return;  // folded into a leaf of an ancestor
    Box b[2];
    int32_t link[2];
    const int32_t c[2] = {child0[i], child1[i]};
    for (int side = 0; side < 2; side++) {
        if (c[side] < 0) {
            const uint32_t pos = (uint32_t)~c[side];
            b[side] = sphere_box(spheres[order[pos]]);
            link[side] = (int32_t)~pos;  // leaf of one sphere: first = pos, count - 1 = 0
        } else {
            for (int k = 0; k < 3; k++) {
                b[side].lo[k] = node_box[(size_t)c[side] * 6 + k];
                b[side].hi[k] = node_box[(size_t)c[side] * 6 + 3 + k];
            }
            const uint32_t cnt = count_of[c[side]];
            link[side] = (int)cnt > leaf_max ? (int32_t)new_id[c[side]]
                                             : (int32_t)~(first_of[c[side]] | ((cnt - 1u) << LEAF_COUNT_SHIFT));
        }
    }
    Node64 nd;
    nd.q0 = f4{b[0].lo[0], b[0].lo[1], b[0].lo[2], b[0].hi[0]};
    nd.q1 = f4{b[0].hi[1], b[0].hi[2], b[1].lo[0], b[1].lo[1]};
    nd.q2 = f4{b[1].lo[2], b[1].hi[0], b[1].hi[1], b[1].hi[2]};
    nd.child0 = link[0], nd.child1 = link[1];
    nd.pad[0] = nd.pad[1] = 0u;
    out[new_id[i]] = nd;
}

// Tie ranks: where the reference's own tree would have each sphere among its leaves (reference_leaf_rank), so that
// two spheres with bit-identical roots resolve as in the reference's walk.  That needs the spheres' order by z --
// one stable 32-bit sort -- and nothing of the reference tree itself.
__global__ void lbvh_z_keys(const Sphere32* s, uint32_t n, uint32_t* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * BLK + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = __float_as_uint(s[i].pos[2] + 0.0f);  // -0 sorts with +0, as it compares
    keys[i] = (b & 0x80000000u) ? ~b : b | 0x80000000u;      // unsigned order = float order
    vals[i] = i;
}
__global__ void lbvh_tie_ranks(const uint32_t* by_z, uint32_t n, uint32_t* tie_rank) {
    const uint32_t p = blockIdx.x * BLK + threadIdx.x;
    if (p >= n) return;
    tie_rank[n < 4u ? p : by_z[p]] = reference_leaf_rank(n, p);  // fewer than four spheres: the reference sorts nothing
}

// The spheres the tree is built over: all but the one that stays out of it (SceneLayout::n_hoisted).
__global__ void lbvh_without(const Sphere32* spheres, uint32_t n, uint32_t hoisted, Sphere32* out) {
    const uint32_t i = blockIdx.x * BLK + threadIdx.x;
    if (i + 1u < n) out[i] = spheres[i < hoisted ? i : i + 1u];
}

// Spheres and their material words in leaf (= sorted) order; the sphere that is in no leaf comes last.  `order`
// indexes the array the tree was built over (lbvh_without), `spheres` and `tie_rank` the scene's own.
__global__ void lbvh_gather(const Sphere32* spheres, const uint32_t* order, const uint32_t* tie_rank, uint32_t n, uint32_t hoisted,
                            f4* out_spheres, u2* out_info) {
    const uint32_t p = blockIdx.x * BLK + threadIdx.x;
    if (p >= n) return;
    uint32_t id;
    if (hoisted < n) {
        const uint32_t o = p + 1u < n ? order[p] : 0u;
        id = p + 1u < n ? (o < hoisted ? o : o + 1u) : hoisted;
    } else {
        id = order[p];
    }
    const Sphere32 s = spheres[id];
    out_spheres[p] = f4{s.pos[0], s.pos[1], s.pos[2], s.rad * s.rad};
    const uint32_t type = s.mat_type > 3u ? 4u : s.mat_type;
    out_info[p] = u2{type | (s.mat_index << 3), tie_rank[id]};
}

struct Scratch {  // device allocations of one build, released on every exit path
    std::vector<void*> ptrs;
    ~Scratch() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <class T>
    bool get(T** out, size_t count) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return false;
        ptrs.push_back(p);
        *out = static_cast<T*>(p);
        return true;
    }
};

}  // namespace

bool build_scene_lbvh(Device* d, const SceneArrays& a, SceneLayout& layout, std::string& err) {
    if (!validate_scene(a, err)) return false;
    const uint32_t n_all = (uint32_t)a.n_spheres;
    const uint32_t hoisted = pick_hoisted(a.spheres, n_all);  // n_all: none
    const uint32_t n = hoisted < n_all ? n_all - 1u : n_all;   // spheres the tree is built over
    DeviceGuard guard(d->id);
    PT_HIP(hipStreamSynchronize(d->stream));
    hipStream_t st = d->stream;

    // Subtrees over at most this many spheres become one leaf (MI355PT_LBVH_LEAF overrides; development).
    int leaf_max = 2;
    if (const char* e = getenv("MI355PT_LBVH_LEAF")) leaf_max = std::min(std::max(atoi(e), 1), MAX_LEAF_SPHERES);
    Scratch tmp;
    Sphere32 *all_spheres, *spheres;  // the scene's own array; the one the tree is built over
    uint64_t *keys, *keys_sorted;
    uint32_t *vals, *order;
    int32_t *child0, *child1, *parent_of_node, *parent_of_leaf;
    int *arrived, *box, *max_depth, *kept_count;
    uint32_t *first_of, *count_of;
    float* node_box;
    if (!tmp.get(&all_spheres, n_all) || (hoisted < n_all && !tmp.get(&spheres, n)) || !tmp.get(&keys, n) || !tmp.get(&keys_sorted, n) || !tmp.get(&vals, n) ||
        !tmp.get(&order, n) || !tmp.get(&child0, n) || !tmp.get(&child1, n) || !tmp.get(&parent_of_node, n) ||
        !tmp.get(&parent_of_leaf, n) || !tmp.get(&arrived, n) || !tmp.get(&box, 8) || !tmp.get(&max_depth, 1) ||
        !tmp.get(&kept_count, 1) || !tmp.get(&first_of, n) || !tmp.get(&count_of, n) || !tmp.get(&node_box, (size_t)n * 6)) {
        err = "lbvh: out of device memory";
        return false;
    }
    uint32_t *area_keys, *area_keys_sorted, *area_vals, *by_area, *below, *below_before, *new_id, *tie_rank;
    if (!tmp.get(&tie_rank, n_all) || !tmp.get(&area_keys, n_all) || !tmp.get(&area_keys_sorted, n_all) || !tmp.get(&area_vals, n_all) ||
        !tmp.get(&by_area, n_all) ||
        !tmp.get(&below, n) || !tmp.get(&below_before, n) || !tmp.get(&new_id, n)) {
        err = "lbvh: out of device memory";
        return false;
    }
    const uint32_t n_nodes = n - 1u;
    size_t sort_bytes = 0, sort32_bytes = 0, scan_bytes = 0;
    PT_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, keys, keys_sorted, vals, order, n, 0, 64, st));
    PT_HIP(rocprim::radix_sort_pairs(nullptr, sort32_bytes, area_keys, area_keys_sorted, area_vals, by_area, n_all, 0, 32, st));
    PT_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, below, below_before, 0u, n_nodes, rocprim::plus<uint32_t>(), st));
    sort_bytes = std::max(sort_bytes, std::max(sort32_bytes, scan_bytes));
    char* sort_tmp;
    if (!tmp.get(&sort_tmp, sort_bytes)) {
        err = "lbvh: out of device memory";
        return false;
    }
    d->have_scene = false;

    const int box_init[8] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN, 0, 0};
    PT_HIP(hipMemcpyAsync(all_spheres, a.spheres, (size_t)n_all * sizeof(Sphere32), hipMemcpyHostToDevice, st));
    PT_HIP(hipMemcpyAsync(box, box_init, sizeof box_init, hipMemcpyHostToDevice, st));
    PT_HIP(hipMemsetAsync(arrived, 0, (size_t)n * sizeof(int), st));
    PT_HIP(hipMemsetAsync(max_depth, 0, sizeof(int), st));
    PT_HIP(hipMemsetAsync(kept_count, 0, sizeof(int), st));
    const dim3 grid_all((n_all + BLK - 1) / BLK), grid((n + BLK - 1) / BLK), block(BLK);
    // tie ranks first, over the whole scene: the sort borrows the buffers of the later sort by box area
    hipLaunchKernelGGL(lbvh_z_keys, grid_all, block, 0, st, all_spheres, n_all, area_keys, area_vals);
    PT_HIP(hipGetLastError());
    PT_HIP(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, area_keys, area_keys_sorted, area_vals, by_area, n_all, 0, 32, st));
    hipLaunchKernelGGL(lbvh_tie_ranks, grid_all, block, 0, st, by_area, n_all, tie_rank);
    PT_HIP(hipGetLastError());
    if (hoisted < n_all) {
        hipLaunchKernelGGL(lbvh_without, grid_all, block, 0, st, all_spheres, n_all, hoisted, spheres);
        PT_HIP(hipGetLastError());
    } else {
        spheres = all_spheres;
    }
    hipLaunchKernelGGL(lbvh_bounds, grid, block, 0, st, spheres, n, box);
    hipLaunchKernelGGL(lbvh_codes, grid, block, 0, st, spheres, n, box, keys, vals);
    PT_HIP(hipGetLastError());
    PT_HIP(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys, keys_sorted, vals, order, n, 0, 64, st));
    hipLaunchKernelGGL(lbvh_hierarchy, grid, block, 0, st, keys_sorted, (int)n, child0, child1, parent_of_node,
                       parent_of_leaf, first_of, count_of, leaf_max, kept_count);
    PT_HIP(hipGetLastError());
    // the number of nodes that remain decides the layout of the blob
    int kept = 0;
    PT_HIP(hipMemcpyAsync(&kept, kept_count, sizeof kept, hipMemcpyDeviceToHost, st));
    PT_HIP(hipStreamSynchronize(st));
    if (kept < 1 || (uint32_t)kept > n_nodes) {
        err = "lbvh: inconsistent hierarchy";
        return false;
    }
    SceneLayout L = make_layout(a, (uint32_t)kept, false);
    if (L.total_bytes == 0) {
        err = "scene: device blob exceeds 4 GiB";
        return false;
    }
    const size_t bytes = ((size_t)L.total_bytes + 15u) & ~(size_t)15u;
    if (bytes > d->scene_cap) {
        (void)hipFree(d->scene);
        d->scene = nullptr;
        d->scene_cap = 0;
        PT_HIP(hipMalloc(&d->scene, bytes));
        d->scene_cap = bytes;
    }
    PT_HIP(hipMemsetAsync(d->scene, 0, bytes, st));
    const uint32_t top = std::min((uint32_t)kept, 4096u);
    hipLaunchKernelGGL(lbvh_refit, grid, block, 0, st, spheres, order, (int)n, child0, child1, parent_of_node,
                       parent_of_leaf, arrived, node_box, max_depth);
    hipLaunchKernelGGL(lbvh_area_keys, grid, block, 0, st, node_box, count_of, leaf_max, (int)n_nodes, area_keys, area_vals);
    PT_HIP(hipGetLastError());
    PT_HIP(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, area_keys, area_keys_sorted, area_vals, by_area, n_nodes, 0, 32, st));
    hipLaunchKernelGGL(lbvh_mark_top, grid, block, 0, st, by_area, (int)n_nodes, (int)top, kept, below);
    PT_HIP(hipGetLastError());
    PT_HIP(rocprim::exclusive_scan(sort_tmp, sort_bytes, below, below_before, 0u, n_nodes, rocprim::plus<uint32_t>(), st));
    hipLaunchKernelGGL(lbvh_new_ids, grid, block, 0, st, by_area, below, below_before, (int)n_nodes, (int)top, new_id);
    hipLaunchKernelGGL(lbvh_emit, grid, block, 0, st, spheres, order, (int)n, child0, child1, node_box, new_id,
                       first_of, count_of, leaf_max, reinterpret_cast<Node64*>(d->scene + L.off_nodes));
    hipLaunchKernelGGL(lbvh_gather, grid_all, block, 0, st, all_spheres, order, tie_rank, n_all, hoisted, reinterpret_cast<f4*>(d->scene + L.off_spheres),
                       reinterpret_cast<u2*>(d->scene + L.off_info));
    PT_HIP(hipGetLastError());
    if (a.n_lights) PT_HIP(hipMemcpyAsync(d->scene + L.off_lights, a.lights, a.n_lights * 16, hipMemcpyHostToDevice, st));
    if (a.n_lambertians)
        PT_HIP(hipMemcpyAsync(d->scene + L.off_lambertians, a.lambertians, a.n_lambertians * 16, hipMemcpyHostToDevice, st));
    if (a.n_metals) PT_HIP(hipMemcpyAsync(d->scene + L.off_metals, a.metals, a.n_metals * 16, hipMemcpyHostToDevice, st));
    std::vector<f4> glass(a.n_glass);  // lives until the synchronisation below
    for (size_t i = 0; i < a.n_glass; i++) glass[i] = glass_record(a.glass[i]);
    if (a.n_glass) PT_HIP(hipMemcpyAsync(d->scene + L.off_glass, glass.data(), a.n_glass * 16, hipMemcpyHostToDevice, st));
    int depth = 0;
    PT_HIP(hipMemcpyAsync(&depth, max_depth, sizeof depth, hipMemcpyDeviceToHost, st));
    PT_HIP(hipStreamSynchronize(st));
    L.max_depth = (uint32_t)std::max(1, depth);
    L.n_hoisted = hoisted < n_all ? 1u : 0u;
    layout = L;
    return adopt_layout(d, L, err);
}

}  // namespace gpu
}  // namespace pt
