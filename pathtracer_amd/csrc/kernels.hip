// HIP kernels for gfx950 (MI355X) and their launch code.
//
//   pt_hash_kernel          per-pixel seed hash (compute.wgsl:392-400), once per resize
//   pt_render_sched_kernel  persistent path kernel (default): every lane owns one (pixel, frame) path at a
//                           time and pulls a new one from a work queue the moment its path ends; the wave
//                           runs one kind of step at a time (node / leaf / scatter / restart), chosen by
//                           how many lanes wait for it
//   pt_render_kernel        the same paths in plain lockstep (acquire -> camera -> whole BVH walk ->
//                           scatter per iteration); kept as the simple reference for the scheduled one
//   pt_accumulate_kernel    adds the per-frame results to the accumulator in frame order
//                           (compute.wgsl:418-421), which keeps sums bit-identical to a frame-by-frame run
//   pt_tile_order_kernel    orders the 8x8 tiles by the rays they cost in the last launch
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is opened at run time (see Rccl below)
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#include "gpu.h"
#include "gpu_state.h"
#include "pt_device.h"

namespace pt {
namespace gpu {

// Work queue: waves take chunks of 64-path units from one counter.  One returning atomic per chunk on a
// single address is a device-wide serial resource: with fixed 4-unit chunks it cost 10 % of a scene1
// launch on MI355X (65 M atomics/s) and two thirds of a scene2 launch.  Chunks are therefore up to
// RenderParams::chunk_max units while plenty of work is left and shrink towards the end of the queue,
// where balance matters: chunk = what is left / (2 * waves), clamped to [1, chunk_max]; once tile costs
// are known, a chunk is also limited to about 2 600 rays (rank_chunk), because the expensive tiles come
// first and a 32-unit chunk of them keeps one wave busy for 6 ms.  The host sets chunk_max to 32 (16 for short
// launches of scenes in global memory, see enqueue_in): consecutive units are consecutive frames of one tile, so
// a wave that keeps a tile for many frames finds the part of a large tree it walks in the caches it has just
// filled.
__device__ __forceinline__ uint32_t guided_chunk_units(uint32_t total, uint32_t taken_to, float inv_share,
                                                       uint32_t chunk_max) {
    const uint32_t left = total - min(total, taken_to);
    return max(1u, min(chunk_max, (uint32_t)((float)left * inv_share)));
}

#define PT_LDS(T) __attribute__((address_space(3))) T

// Scene blob staged in LDS: explicit address space so every fetch is a ds_read.
struct LdsMem {
    static constexpr bool ORDERED_NODES = false;
    uint32_t base;
    __device__ __forceinline__ int32_t root() const { return 0; }
    template <class T>
    __device__ __forceinline__ T load(uint32_t off) const {
        // address-space-qualified loads exist for builtin vector types only; bit-cast to the record
        typedef uint32_t w4 __attribute__((ext_vector_type(4)));
        typedef uint32_t w2 __attribute__((ext_vector_type(2)));
        if constexpr (sizeof(T) == 16) {
            w4 v = *reinterpret_cast<const PT_LDS(w4)*>(base + off);
            return __builtin_bit_cast(T, v);
        } else if constexpr (sizeof(T) == 8) {
            w2 v = *reinterpret_cast<const PT_LDS(w2)*>(base + off);
            return __builtin_bit_cast(T, v);
        } else {
            static_assert(sizeof(T) == 4, "LDS loads are 4, 8 or 16 bytes");
            uint32_t v = *reinterpret_cast<const PT_LDS(uint32_t)*>(base + off);
            return __builtin_bit_cast(T, v);
        }
    }
    __device__ __forceinline__ void node64(uint32_t off, f4& q0, f4& q1, f4& q2, u2& ch) const {
        q0 = load<f4>(off), q1 = load<f4>(off + 16), q2 = load<f4>(off + 32), ch = load<u2>(off + 48);
    }
    __device__ __forceinline__ void node32(uint32_t off, u4& a, u4& b) const { a = load<u4>(off), b = load<u4>(off + 16); }
};
// The same with the node section replaced by direction-ordered 80-byte records, LDS_NODE_BYTES = 88 bytes apart, at
// `node_base`; `base` is shifted so that the blob offsets of the other sections still apply.  Links to
// interior nodes are the LDS byte addresses of their records (stage_scene rewrites them), which saves
// the address arithmetic per visit; LDS addresses are small positive numbers, leaf links stay negative.
struct LdsOrderedMem : LdsMem {
    static constexpr bool ORDERED_NODES = true;
    uint32_t node_base;
    __device__ __forceinline__ int32_t root() const { return (int32_t)node_base; }
    __device__ __forceinline__ void node_ordered(uint32_t a, const uint32_t* sgn, f4& ax, f4& ay, f4& az, u2& ch) const {
        typedef uint32_t w2 __attribute__((ext_vector_type(2)));
        auto four = [](uint32_t addr) {  // 8-byte aligned: two halves, merged into one ds_read2_b64
            const w2 lo = *reinterpret_cast<const PT_LDS(w2)*>(addr), hi = *reinterpret_cast<const PT_LDS(w2)*>(addr + 8u);
            return f4{__uint_as_float(lo.x), __uint_as_float(lo.y), __uint_as_float(hi.x), __uint_as_float(hi.y)};
        };
        ax = four(a + sgn[0]), ay = four(a + 24u + sgn[1]), az = four(a + 48u + sgn[2]);
        const w2 c = *reinterpret_cast<const PT_LDS(w2)*>(a + 72u);
        ch = u2{c.x, c.y};
    }
};
// Scene blob in global memory with the first `lds_node_bytes` of the node array (the top of the
// tree, see order_top_first) also staged in LDS.  Random record fetches from LDS sustain more than
// twice the rate of fetches that hit in the vector L1, and the two pipes work side by side.
struct HybridMem {
    static constexpr bool ORDERED_NODES = false;
    const char* base;
    __device__ __forceinline__ int32_t root() const { return 0; }
    uint32_t lds_base;
    uint32_t lds_node_bytes;
    template <class T>
    __device__ __forceinline__ T load(uint32_t off) const {
        return *reinterpret_cast<const T*>(base + off);
    }
    __device__ __forceinline__ void node64(uint32_t off, f4& q0, f4& q1, f4& q2, u2& ch) const {
        if (off < lds_node_bytes) {
            LdsMem l{lds_base};
            q0 = l.load<f4>(off), q1 = l.load<f4>(off + 16), q2 = l.load<f4>(off + 32), ch = l.load<u2>(off + 48);
        } else {
            q0 = load<f4>(off), q1 = load<f4>(off + 16), q2 = load<f4>(off + 32), ch = load<u2>(off + 48);
        }
    }
    __device__ __forceinline__ void node32(uint32_t off, u4& a, u4& b) const { a = load<u4>(off), b = load<u4>(off + 16); }
};
// The same with the nodes as direction-ordered 80-byte records (node_to_ordered) in an array of their own
// (`onodes`, built by pt_order_nodes_kernel; the blob keeps its Node64 section, which this view never reads).
// Links to interior nodes are the byte offsets of their records; the first `lds_node_bytes` bytes of the array
// (whole records) are also in LDS.  A visit costs 12 vector instructions less than over Node64 (no min / max per
// axis), which is what bounds the walk of a large tree once its fetches hit the L1 (see enqueue_in).
constexpr uint32_t GLOBAL_NODE_BYTES = 80;
struct HybridOrderedMem : HybridMem {
    static constexpr bool ORDERED_NODES = true;
    const char* onodes;
    __device__ __forceinline__ void node_ordered(uint32_t a, const uint32_t* sgn, f4& ax, f4& ay, f4& az, u2& ch) const {
        typedef uint32_t w2 __attribute__((ext_vector_type(2)));
        typedef float v4 __attribute__((ext_vector_type(4), aligned(8)));
        if (a < lds_node_bytes) {
            const uint32_t l = lds_base + a;
            auto four = [](uint32_t addr) {
                const w2 lo = *reinterpret_cast<const PT_LDS(w2)*>(addr), hi = *reinterpret_cast<const PT_LDS(w2)*>(addr + 8u);
                return f4{__uint_as_float(lo.x), __uint_as_float(lo.y), __uint_as_float(hi.x), __uint_as_float(hi.y)};
            };
            ax = four(l + sgn[0]), ay = four(l + 24u + sgn[1]), az = four(l + 48u + sgn[2]);
            const w2 c = *reinterpret_cast<const PT_LDS(w2)*>(l + 72u);
            ch = u2{c.x, c.y};
        } else {
            // 32-bit offsets from the uniform base: one address register per fetch
            const uint32_t o0 = a + sgn[0], o1 = a + sgn[1], o2 = a + sgn[2];
            const v4 x = *reinterpret_cast<const v4*>(onodes + o0), y = *reinterpret_cast<const v4*>(onodes + o1 + 24u),
                     z = *reinterpret_cast<const v4*>(onodes + o2 + 48u);
            ax = f4{x.x, x.y, x.z, x.w}, ay = f4{y.x, y.y, y.z, y.w}, az = f4{z.x, z.y, z.z, z.w};
            ch = *reinterpret_cast<const u2*>(onodes + a + 72u);
        }
    }
};
// Per-lane traversal stack in LDS, entry-major so that lane l always hits bank l % 64.
template <int BLK>
struct LdsStack {
    uint32_t base;  // byte address of this lane's entry 0
    __device__ __forceinline__ void push(int sp, int32_t v) const {
        *reinterpret_cast<PT_LDS(int32_t)*>(base + (uint32_t)sp * (BLK * 4u)) = v;
    }
    __device__ __forceinline__ int32_t pop(int sp) const {
        return *reinterpret_cast<const PT_LDS(int32_t)*>(base + (uint32_t)sp * (BLK * 4u));
    }
};

// The same stack with only its first `lds_entries` entries in LDS; deeper entries (rare: one per
// pending far child beyond that depth) go to a per-lane area in global memory.  Keeps the LDS cost of
// a deep tree's stacks bounded, which is what leaves room for the node prefix.
template <int BLK>
struct SpillStack {
    uint32_t base;        // byte address of this lane's entry 0 in LDS
    int32_t* spill;       // this lane's entry `lds_entries`, stride BLK
    int32_t lds_entries;
    __device__ __forceinline__ void push(int sp, int32_t v) const {
        if (sp < lds_entries) *reinterpret_cast<PT_LDS(int32_t)*>(base + (uint32_t)sp * (BLK * 4u)) = v;
        else spill[(size_t)(sp - lds_entries) * BLK] = v;
    }
    __device__ __forceinline__ int32_t pop(int sp) const {
        if (sp < lds_entries) return *reinterpret_cast<const PT_LDS(int32_t)*>(base + (uint32_t)sp * (BLK * 4u));
        return spill[(size_t)(sp - lds_entries) * BLK];
    }
};

struct RenderParams {
    const char* scene;
    SceneLayout L;
    Camera64 cam;
    uint32_t width, height, tiles_x, tiles_per_frame, n_pix;
    int32_t samples, depth;
    uint32_t total_units;  // frames in this launch * tiles_per_frame
    // Work order: unit u = (tile rank u / n_frames, frame u % n_frames); tile = tile_perm[rank] (or the
    // rank itself without a permutation).  The permutation lists tiles by decreasing measured cost, so
    // the long paths start early and the queue ends with the cheapest tiles.
    uint32_t n_frames;
    float inv_n_frames;
    const uint32_t* tile_perm;
    const uint16_t* rank_chunk;  // with tile_perm: chunk size by the rank the chunk starts in (cost-aware)
    const uint32_t* seeds;       // one per frame of this launch
    const uint32_t* pixel_hash;  // [n_pix]
    f4* frame_out;               // [frame][tile rank][slot]: rgb = final_colour / samples, w = rays traced;
                                 // with `compact`: three floats per result, no ray count
    uint32_t compact;            // 1: 12-byte results; the rays traced are summed per wave into ray_slots instead
                                 // (launches that measure the tile costs need the count per path: 16-byte results)
    unsigned long long* ray_slots;
    uint32_t* work_counter;
    unsigned long long* counters;  // nodes, tests, accepted, bounce[5]
    uint32_t scene_lds_bytes;
    int32_t glass_always_draws;
    uint64_t watchdog_ticks;  // 100 MHz ticks without a finished path after which a wave gives up (scheduling-bug guard)
    uint32_t thr[6];  // pt_render_sched_kernel: lanes that must wait for a step before it runs: node, leaf,
                      // (unused: terminal lanes are served by the restart step), opaque, glass, restart
    float inv_tiles_x;
    uint32_t small_ids;  // 1: unit and tile ids < 2^24 (divmod24 is exact)
    float inv_share;  // 1 / (2 * waves in the grid), for the chunk size
    uint32_t chunk_max;
    // SCENE_GLOBAL only: scene_lds_bytes is the node prefix kept in LDS, the stack keeps
    // stack_lds_entries in LDS and the rest in stack_spill[block][entry][lane]
    int32_t* stack_spill;
    uint32_t stack_lds_entries, stack_spill_entries;
    // Exchange variant of pt_render_sched_kernel: LDS offset of the workgroup's two path pools (0: not in use)
    uint32_t pool_off;
    const char* onodes;  // SCENE_GLOBAL_ORDERED: the tree as direction-ordered records
    const uint32_t* tile_lists;  // pt_render_sched_kernel: per tile TILE_LIST_WORDS words, the candidates of its camera rays (null: none)
    float rcp_w, rcp_h;          // 1 / width, 1 / height
    uint32_t short_div;          // 1: the short division by width and height was verified for this image size (PixelDiv)
    // The buffered results of the PREVIOUS launch, still to be added to the accumulator (null: none): the first
    // `resolve_wgs` workgroups of this launch add them before they take paths (see the kernel's prologue).
    f4* accum;
    const f4* prev_results;
    const uint32_t* prev_perm;
    uint32_t prev_frames, prev_active_tiles, resolve_wgs;
};

// Paths waiting for a scatter step, parked in LDS by the exchange variant: per class (lambertian / metal, glass)
// a stack of POOL_CAP records of 16 words, stored as four planes of 16-byte quarters so that consecutive slots
// hit consecutive banks.  Header: per class a lock word and the entry count.
constexpr uint32_t POOL_CAP = 64, POOL_HEADER = 64, POOL_BYTES = POOL_HEADER + 2u * POOL_CAP * 64u;
// A restarting wave takes a batch out of a pool that holds at least POOL_MIN_* paths; a wave parks the lanes it
// has waiting for a class when they are at least POOL_DEP_MIN_* (and fewer than the class's threshold).
// Compile-time on purpose: as kernel arguments they would sit in scalar registers through the whole loop.
#ifndef PT_POOL_TUNING
#define PT_POOL_TUNING 48, 32, 4, 2
#endif
constexpr uint32_t POOL_TUNING[4] = {PT_POOL_TUNING};
// Chunks of the work queue a workgroup of the exchange variant takes at once (0: every wave takes its own).  The
// waves of a compute unit then work on the same few tiles -- the same part of a large tree in its L1 -- and the
// queue sees a sixteenth of the atomics: 1 M spheres at 3840x2160, 64 frames, 44.1 ms with 0, 43.7 with 4, 43.1 with
// 16, 43.6 with 32; scene1 1920x1080 x 256 frames 17.32 -> 17.17 ms.
#ifndef PT_WG_QUEUE
#define PT_WG_QUEUE 16
#endif
constexpr uint32_t POOL_MIN_OPAQUE = POOL_TUNING[0], POOL_MIN_GLASS = POOL_TUNING[1];
constexpr uint32_t POOL_DEP_MIN_OPAQUE = POOL_TUNING[2], POOL_DEP_MIN_GLASS = POOL_TUNING[3];
// Scenes in LDS park twice as many at the least (scene1 1920x1080 x 256 frames: 13.95 -> 13.83 ms); trees in global
// memory are better off with the smaller minima (1 M spheres at 3840x2160 x 64: 31.6 against 32.0 ms).
constexpr uint32_t POOL_DEP_LDS_FACTOR = 2;

// The direction-ordered records of a tree that stays in global memory (adopt_layout).
__global__ void pt_order_nodes_kernel(const uint32_t* nodes64, uint32_t n, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t q[16], rec[20];
#pragma unroll
    for (int k = 0; k < 16; k++) q[k] = nodes64[(size_t)i * 16u + k];
    const uint32_t l0 = (int32_t)q[12] >= 0 ? q[12] * GLOBAL_NODE_BYTES : q[12];
    const uint32_t l1 = (int32_t)q[13] >= 0 ? q[13] * GLOBAL_NODE_BYTES : q[13];
    node_to_ordered(q, l0, l1, rec);
#pragma unroll
    for (int k = 0; k < 20; k++) out[(size_t)i * 20u + k] = rec[k];
}

__global__ void pt_hash_kernel(uint32_t* out, uint32_t w, uint32_t h) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    out[i] = pixel_hash(i % w, i / w, w, h);
}

// Candidate lists of the camera rays, one thread per 8x8 tile (tile_list_build): once per camera, scene and image
// size.  Reads the blob in global memory whatever the path kernels stage in LDS.
__global__ void pt_tile_lists_kernel(const char* scene, SceneLayout L, Camera64 cam, uint32_t width, uint32_t height,
                                     uint32_t tiles_x, uint32_t tiles, uint32_t max_list, uint32_t* lists, uint32_t* count) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles) return;
    PtrMem mem{scene};
    uint32_t rec[TILE_LIST_WORDS];
    for (uint32_t k = 1; k < TILE_LIST_WORDS; k++) rec[k] = 0u;
    tile_list_build(mem, L, cam, t % tiles_x, t / tiles_x, width, height, max_list, rec);
    typedef uint32_t w4 __attribute__((ext_vector_type(4)));
    w4* out = reinterpret_cast<w4*>(lists + (size_t)t * TILE_LIST_WORDS);
    for (uint32_t k = 0; k < TILE_LIST_WORDS / 4u; k++) out[k] = w4{rec[4 * k], rec[4 * k + 1], rec[4 * k + 2], rec[4 * k + 3]};
    if (rec[0] != TILE_LIST_WALK) atomicAdd(count, 1u), atomicAdd(count + 1, rec[0]);  // statistics
    if (rec[0] == 0u) atomicAdd(count + 2, 1u);  // tiles that show nothing but sky (sky_block of pt_accumulate_kernel)
}

__device__ __forceinline__ uint32_t lane_rank(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

// SCENE: where the scene blob is read from and which node format it holds.
// SCENE_REFERENCE: memory as SCENE_GLOBAL, nodes walked the reference's way (SceneLayout::reference_walk).
// SCENE_GLOBAL_ORDERED: memory as SCENE_GLOBAL, nodes read as direction-ordered records from RenderParams::onodes.
enum { SCENE_GLOBAL = 0, SCENE_LDS = 1, SCENE_LDS_HALF = 2, SCENE_REFERENCE = 3, SCENE_GLOBAL_ORDERED = 4 };
template <int SCENE> struct SceneMem { typedef HybridMem type; };
template <> struct SceneMem<SCENE_GLOBAL_ORDERED> { typedef HybridOrderedMem type; };
template <> struct SceneMem<SCENE_LDS> { typedef LdsOrderedMem type; };
template <> struct SceneMem<SCENE_LDS_HALF> { typedef LdsMem type; };

// Copies what the workgroup keeps in LDS: the whole blob (LDS scenes; float nodes become
// direction-ordered records, LDS_NODE_BYTES apart, on the way) or the node prefix (global scenes).
template <int SCENE, int BLK>
__device__ __forceinline__ void stage_scene(const RenderParams& P, uint32_t lds0) {
    typedef uint32_t w4 __attribute__((ext_vector_type(4)));
    typedef uint32_t w2 __attribute__((ext_vector_type(2)));
    const w4* src = reinterpret_cast<const w4*>(SCENE == SCENE_GLOBAL_ORDERED ? P.onodes : P.scene);
    if constexpr (SCENE == SCENE_LDS) {
        const uint32_t n = P.L.n_nodes;  // the node section starts the blob
        for (uint32_t i = threadIdx.x; i < n; i += BLK) {
            const w4 q0 = src[i * 4u], q1 = src[i * 4u + 1u], q2 = src[i * 4u + 2u], q3 = src[i * 4u + 3u];
            const uint32_t q[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
            // links to interior nodes become the LDS addresses of their records
            const uint32_t l0 = (int32_t)q3.x >= 0 ? lds0 + q3.x * LDS_NODE_BYTES : q3.x;
            const uint32_t l1 = (int32_t)q3.y >= 0 ? lds0 + q3.y * LDS_NODE_BYTES : q3.y;
            uint32_t rec[20];
            node_to_ordered(q, l0, l1, rec);
            const uint32_t a = lds0 + i * LDS_NODE_BYTES;
#pragma unroll
            for (uint32_t k = 0; k < 10; k++) *reinterpret_cast<PT_LDS(w2)*>(a + 8u * k) = w2{rec[2 * k], rec[2 * k + 1]};
        }
        const uint32_t rest = (P.scene_lds_bytes - lds_nodes_bytes(n)) / 16u;
        for (uint32_t i = threadIdx.x; i < rest; i += BLK)
            *reinterpret_cast<PT_LDS(w4)*>(lds0 + lds_nodes_bytes(n) + i * 16u) = src[n * 4u + i];
    } else {
        for (uint32_t i = threadIdx.x; i < P.scene_lds_bytes / 16u; i += BLK)
            *reinterpret_cast<PT_LDS(w4)*>(lds0 + i * 16u) = src[i];
    }
    __syncthreads();
}

// One finished pixel-frame.
__device__ __forceinline__ void store_result(const RenderParams& P, uint32_t index, f4 rgb, uint32_t rays, uint32_t& lane_rays) {
    if (P.compact) {
        float* o = reinterpret_cast<float*>(P.frame_out) + (size_t)index * 3u;
        o[0] = rgb.x, o[1] = rgb.y, o[2] = rgb.z;
        lane_rays += rays;
    } else {
        rgb.w = (float)rays;  // summed per pixel by the accumulation kernel: ray total and tile costs
        P.frame_out[index] = rgb;
    }
}
// At the end of a path kernel: the rays of a compact launch, one atomic per wave on a striped counter.
__device__ __forceinline__ void flush_rays(const RenderParams& P, uint32_t lane_rays) {
    if (!P.compact) return;
    const unsigned long long total = wave_sum((unsigned long long)lane_rays);
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63u) == 0 && total) atomicAdd(P.ray_slots + (size_t)(wave % RAY_SLOTS) * RAY_SLOT_STRIDE, total);
}

template <int SCENE, bool COUNT, int BLK>
__global__ void __launch_bounds__(BLK) pt_render_kernel(const RenderParams P) {
    constexpr bool LDS_SCENE = SCENE == SCENE_LDS || SCENE == SCENE_LDS_HALF, HALF = SCENE == SCENE_LDS_HALF;
    extern __shared__ __align__(16) char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(PT_LDS(char)*)smem;
    stage_scene<SCENE, BLK>(P, lds0);
    using Mem = typename SceneMem<SCENE>::type;
    using Stack = typename std::conditional<LDS_SCENE, LdsStack<BLK>, SpillStack<BLK>>::type;
    Mem mem;
    Stack stack;
    stack.base = lds0 + P.scene_lds_bytes + threadIdx.x * 4u;
    if constexpr (SCENE == SCENE_LDS) {
        mem.node_base = lds0;
        mem.base = lds0 + lds_nodes_bytes(P.L.n_nodes) - P.L.n_nodes * (uint32_t)sizeof(Node64);
    } else if constexpr (LDS_SCENE) {
        mem.base = lds0;
    } else {
        mem.base = P.scene, mem.lds_base = lds0, mem.lds_node_bytes = P.scene_lds_bytes;
        if constexpr (SCENE == SCENE_GLOBAL_ORDERED) mem.onodes = P.onodes;
        stack.spill = P.stack_spill + (size_t)blockIdx.x * P.stack_spill_entries * BLK + threadIdx.x;
        stack.lds_entries = (int32_t)P.stack_lds_entries;
    }

    const CameraRegs cam = load_camera(P.cam);
    const float fw = (float)P.width, fh = (float)P.height;
    const uint32_t lane = threadIdx.x & 63u;

    // lane state
    bool alive = false, need_ray = false;
    PathState ps;
    ps.o = ps.d = ps.colour = mk(0.0f, 0.0f, 0.0f);
    ps.rng = 1u;
    ps.bounce = 0;
    V3 final_colour = mk(0.0f, 0.0f, 0.0f);
    uint32_t out_index = 0, pxy = 0, nrays = 0, lane_rays = 0;
    int32_t sample_i = 0;
    Counters cnt = {0, 0, 0};
    uint32_t bounce_cnt[5] = {0, 0, 0, 0, 0};

    // wave state (uniform)
    uint32_t next = 0, end = 0;
    uint32_t chunk = guided_chunk_units(P.total_units, 0u, P.inv_share, P.chunk_max);
    bool exhausted = false;

    for (;;) {
        // ---- acquire: idle lanes take consecutive path ids from the wave's chunk of the queue
        unsigned long long m = __ballot(!alive);
        while (m) {
            if (next == end) {
                if (exhausted) break;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(P.work_counter, chunk);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);  // lane 0 is active here: the step is entered by the whole wave
                if (base >= P.total_units) {
                    exhausted = true;
                    break;
                }
                next = base * 64u;
                end = min(base + chunk, P.total_units) * 64u;
                chunk = guided_chunk_units(P.total_units, base + chunk, P.inv_share, P.chunk_max);
            }
            const uint32_t avail = end - next;
            const uint32_t rank = lane_rank(m);
            if (!alive && rank < avail) {
                const uint32_t pid = next + rank;
                const uint32_t unit = pid >> 6, slot = pid & 63u;
                const uint32_t tile_rank = unit / P.n_frames;
                const uint32_t frame = unit - tile_rank * P.n_frames;
                const uint32_t tile = P.tile_perm ? P.tile_perm[tile_rank] : tile_rank;
                const uint32_t result_index = (frame * P.tiles_per_frame + tile_rank) * 64u + slot;
                uint32_t px, py;
                tile_pixel(tile, slot, P.tiles_x, px, py);
                if (px < P.width && py < P.height) {
                    const uint32_t pix = py * P.width + px;
                    ps.rng = pixel_seed(P.seeds[frame], P.pixel_hash[pix]);
                    out_index = result_index;
                    pxy = px | (py << 16);
                    final_colour = mk(0.0f, 0.0f, 0.0f);
                    sample_i = 0;
                    nrays = 0;
                    alive = true;
                    need_ray = true;
                }
            }
            next += min((uint32_t)__popcll(m), avail);
            m = __ballot(!alive);
        }
        if (__ballot(alive) == 0ull) break;

        // ---- camera: compute.wgsl:405-413
        if (alive && need_ray) {
            ps.d = camera_dir(cam, pxy & 0xffffu, pxy >> 16, fw, fh, ps.rng);
            ps.o = cam.pos;
            ps.colour = mk(1.0f, 1.0f, 1.0f);
            ps.bounce = 0;
            need_ray = false;
        }

        // ---- extend + scatter
        if (alive) {
            Hit hit;
            closest_hit<Mem, Stack, COUNT, HALF, SCENE == SCENE_REFERENCE>(mem, P.L, stack, ps.o, ps.d, hit, cnt);
            nrays++;
            V3 radiance;
            uint32_t mat;
            const bool done = scatter(mem, P.L, hit, ps, P.depth, P.glass_always_draws != 0, radiance, mat);
            if (COUNT) {
#pragma unroll
                for (uint32_t k = 0; k < 5; k++) bounce_cnt[k] += (mat == k) ? 1u : 0u;
            }
            if (done) {
                final_colour = final_colour + radiance;
                sample_i++;
                if (sample_i < P.samples) {
                    need_ray = true;
                } else {
                    f4 o4;
                    o4.x = final_colour.x, o4.y = final_colour.y, o4.z = final_colour.z;
                    if (P.samples != 1) {  // x / 1.0f == x: the common case needs no division
                        const float fs = (float)P.samples;
                        o4.x = final_colour.x / fs;
                        o4.y = final_colour.y / fs;
                        o4.z = final_colour.z / fs;
                    }
                    o4.w = 0.0f;
                    store_result(P, out_index, o4, nrays, lane_rays);
                    alive = false;
                }
            }
        }
    }

    flush_rays(P, lane_rays);
    if (COUNT) {
        unsigned long long v[8] = {cnt.nodes,     cnt.tests,     cnt.accepted,  bounce_cnt[0],
                                   bounce_cnt[1], bounce_cnt[2], bounce_cnt[3], bounce_cnt[4]};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            unsigned long long s = wave_sum(v[k]);
            if (lane == 0) atomicAdd(&P.counters[k], s);
        }
    }
}

// accum[p] += vec4(result[f][p].rgb, 0) for f = 0..n_frames-1 in order; ray counts summed on the side
// (total, and per tile into `tile_cost` when given: the input of pt_tile_order_kernel).  Results lie in
// [frame][tile rank][slot] order, ranks being those of the launch that produced them: one thread per
// (tile rank, slot), i.e. one wave per tile; at any moment all waves stream through the same frame.
// (A [tile rank][frame][slot] layout makes the waves' simultaneous reads a power-of-two stride apart:
// measured 6x slower at 3840x2160.)
// `block` / `thread`: which 256 texels (in rank order) and which of them; the callers' workgroups may be larger.
__device__ __forceinline__ void resolve_block(uint32_t block, uint32_t thread, f4* accum, const f4* results, uint32_t width, uint32_t height,
                                              uint32_t tiles_x, uint32_t tiles, uint32_t active_tiles, uint32_t n_frames,
                                              const uint32_t* tile_perm, unsigned long long* ray_total, float* tile_cost, uint32_t compact,
                                              const unsigned long long* gave_up, unsigned long long* launches_added) {
    const uint32_t gid = block * 256u + thread;
    if (gid == 0) atomicAdd(launches_added, 1ull);  // what finish() credits: launches whose results reached the accumulator
    const uint32_t rank = gid >> 6, slot = gid & 63u;
    unsigned long long rays = 0;
    uint32_t tile = 0;
    if (rank < active_tiles) {  // ranks beyond are sky tiles (sky_block); `tiles` is the stride of the layout
        tile = tile_perm ? tile_perm[rank] : rank;
        const uint32_t px = (tile % tiles_x) * 8u + (slot & 7u), py = (tile / tiles_x) * 8u + (slot >> 3);
        if (px < width && py < height) {
            const uint32_t i = py * width + px;
            f4 a = accum[i];
            const f4* src = results + (size_t)rank * 64u + slot;
            const size_t frame_stride = (size_t)tiles * 64u;
            if (compact) {  // 12-byte results; the path kernel has counted the rays
                const float* c = reinterpret_cast<const float*>(results) + ((size_t)rank * 64u + slot) * 3u;
                const size_t stride = frame_stride * 3u;
                // the sums are sequential by definition, the fetches are not: eight frames' results are requested
                // before the first is added (one fetch at a time leaves a wave idle for a memory latency per frame)
                uint32_t f = 0;
                for (; f + 8u <= n_frames; f += 8u) {
                    float v[8][3];
#pragma unroll
                    for (uint32_t k = 0; k < 8u; k++) {
                        const float* s = c + (size_t)(f + k) * stride;
                        v[k][0] = __builtin_nontemporal_load(s), v[k][1] = __builtin_nontemporal_load(s + 1), v[k][2] = __builtin_nontemporal_load(s + 2);
                    }
#pragma unroll
                    for (uint32_t k = 0; k < 8u; k++) a.x += v[k][0], a.y += v[k][1], a.z += v[k][2], a.w += 0.0f;
                }
                for (; f < n_frames; f++) {
                    const float* s = c + (size_t)f * stride;
                    a.x += s[0];
                    a.y += s[1];
                    a.z += s[2];
                    a.w += 0.0f;
                }
            } else {
                for (uint32_t f = 0; f < n_frames; f++) {
                    f4 s = src[(size_t)f * frame_stride];
                    a.x += s.x;
                    a.y += s.y;
                    a.z += s.z;
                    a.w += 0.0f;
                    rays += (unsigned long long)s.w;
                }
            }
            accum[i] = a;
        }
    }
    rays = wave_sum(rays);
    if ((thread & 63u) == 0) {
        // one atomic per wave: spread over RAY_SLOTS addresses in different cache lines (atomics on one
        // address run one at a time, ~12 ns each: 130 k of them cost 1.5 ms at 3840x2160)
        if (rays) atomicAdd(ray_total + (size_t)(block % RAY_SLOTS) * RAY_SLOT_STRIDE, rays);
        if (tile_cost && rank < active_tiles) tile_cost[tile] = (float)rays;
    }
}

// Wave-scheduled variant.  Every lane is a small state machine kept in Walk::cur -- interior node
// pending, leaf pending, walk finished, one of three scatter classes pending, idle (needs a path or a
// camera ray), exited -- and the wave runs one kind of step at a time, only when enough lanes wait
// for it (ballot + popcount against a threshold).  Lanes in a thinly populated state wait until their
// state fills up, so the long scatter bodies run with many lanes active instead of being serialised
// over sparsely populated material branches, and the node loop stops when its population drops below
// the threshold instead of dragging a few long walks along with most lanes idle.  When no state
// reaches its threshold every non-empty state runs once, which guarantees progress.  Scheduling only
// reorders work between independent paths; every path performs exactly the arithmetic of the
// lockstep kernel.
// (All per-lane state lives in vector registers: a bool per lane would be a lane mask in a scalar register
// pair, updated with three scalar instructions per nested branch, and the scalar unit is the scarcer one.)
constexpr int32_t LANE_IDLE = INT32_MIN + 1;      // needs a path
constexpr int32_t LANE_NEXT = INT32_MIN + 2;      // has a path, needs the camera ray of its next sample
constexpr int32_t LANE_TERMINAL = INT32_MIN + 3;  // path ends at this hit / miss; radiance still to add
constexpr int32_t LANE_EXIT = INT32_MIN + 4;      // queue empty, nothing left to do
constexpr int32_t LANE_OPAQUE = INT32_MIN + 5;
constexpr int32_t LANE_GLASS = INT32_MIN + 6;
__device__ __forceinline__ bool lane_wants_restart(int32_t cur) { return (uint32_t)(cur - LANE_IDLE) <= 2u; }
// Node / leaf visits per pass of their loops before the population is counted again.  Lanes only leave
// these loops, so counting every second node visit costs at most one thin visit and saves the count's
// scalar instructions (measured on scene1, 256 frames: 21.46 ms with 1, 21.30 with 2, 21.44 with 3,
// 21.60 with 4; two leaf visits: no gain).
#ifndef PT_NODE_UNROLL
#define PT_NODE_UNROLL 2
#endif
#ifndef PT_LEAF_UNROLL
#define PT_LEAF_UNROLL 1
#endif
enum { PH_NODE = 0, PH_LEAF, PH_TERMINAL, PH_OPAQUE, PH_GLASS, PH_IDLE, PH_COUNT };

// The kernel's arguments as they lie in the kernarg segment, for the parts of a kernel that run rarely (the
// restart step: once per path).  Read through this view at the point of use, their values occupy scalar
// registers only there; as plain kernel arguments the compiler loads them once and keeps all of them for the
// whole persistent loop, and at eight waves per SIMD (80 scalar registers) that means spills, each of which is
// a vector instruction (v_writelane / v_readlane).  The empty asm keeps the compiler from hoisting the loads.
typedef const __attribute__((address_space(4))) RenderParams* KernargView;
__device__ __forceinline__ KernargView kernarg_view() {
    KernargView q = (KernargView)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return q;
}

// FAST: one sample per path and 12-byte results (what every launch but the occasional cost-measuring one is): the
// per-path sample sum, sample counter, pixel coordinates and ray count need no registers of their own.
template <int SCENE, bool COUNT, int BLK, bool XCHG = false, bool FAST = false>
__global__ void __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(BLK == 1024 ? 8 : 1, 8))) pt_render_sched_kernel(const RenderParams P) {
    constexpr bool LDS_SCENE = SCENE == SCENE_LDS || SCENE == SCENE_LDS_HALF, HALF = SCENE == SCENE_LDS_HALF;
    extern __shared__ __align__(16) char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(PT_LDS(char)*)smem;
    if constexpr (XCHG) {  // empty pools, locks open (stage_scene ends with the barrier)
        if (threadIdx.x < POOL_HEADER / 4u) *reinterpret_cast<PT_LDS(uint32_t)*>(lds0 + P.pool_off + threadIdx.x * 4u) = 0u;
    }
    stage_scene<SCENE, BLK>(P, lds0);
    using Mem = typename SceneMem<SCENE>::type;
    using Stack = typename std::conditional<LDS_SCENE, LdsStack<BLK>, SpillStack<BLK>>::type;
    Mem mem;
    Stack stack;
    stack.base = lds0 + P.scene_lds_bytes + threadIdx.x * 4u;
    if constexpr (SCENE == SCENE_LDS) {
        mem.node_base = lds0;
        mem.base = lds0 + lds_nodes_bytes(P.L.n_nodes) - P.L.n_nodes * (uint32_t)sizeof(Node64);
    } else if constexpr (LDS_SCENE) {
        mem.base = lds0;
    } else {
        mem.base = P.scene, mem.lds_base = lds0, mem.lds_node_bytes = P.scene_lds_bytes;
        if constexpr (SCENE == SCENE_GLOBAL_ORDERED) mem.onodes = P.onodes;
        stack.spill = P.stack_spill + (size_t)blockIdx.x * P.stack_spill_entries * BLK + threadIdx.x;
        stack.lds_entries = (int32_t)P.stack_lds_entries;
    }

    const uint32_t lane = threadIdx.x & 63u;

    PathState ps;
    ps.o = ps.d = ps.colour = mk(0.0f, 0.0f, 0.0f);
    ps.rng = 1u;
    ps.bounce = 0;
    Walk w;
    walk_begin<HALF>(w, P.L, ps.o, mk(1.0f, 1.0f, 1.0f));
    w.cur = LANE_IDLE;
    uint32_t mbits = 0;  // material word of the hit, valid in the scatter states
    V3 final_colour = mk(0.0f, 0.0f, 0.0f);
    uint32_t out_index = 0, pxy = 0, rays = 0;  // rays of the lane's current path
    uint32_t lane_rays = 0;                     // rays of all its finished paths (compact launches)
    int32_t sample_i = 0;
    Counters cnt = {0, 0, 0};
    uint32_t bounce_cnt[5] = {0, 0, 0, 0, 0};
    uint32_t ph_exec[PH_COUNT] = {0}, ph_lanes[PH_COUNT] = {0};  // COUNT: wave-level step executions

    uint32_t next = 0, end = 0;  // wave-uniform: the wave's chunk of the path queue
    uint32_t chunk;
    {
        KernargView Q = kernarg_view();
        chunk = guided_chunk_units(Q->total_units, 0u, Q->inv_share, Q->chunk_max);
        if (Q->rank_chunk) chunk = min(chunk, (uint32_t)Q->rank_chunk[0]);
    }
    bool exhausted = false;
    bool force = false;
#define PT_THR(ph) (force ? 1u : P.thr[ph])

    // Prologue: the buffered results of the previous launch are added to the accumulator by the first few workgroups
    // of this one, before they take paths.  That work is bound by HBM reads and barely issues instructions, so it
    // runs beside the other waves' path work for free, where a kernel of its own between two path kernels costs its
    // full duration (and a kernel on a second queue beside the path kernel costs the path kernel a third of its rate).
    if constexpr (XCHG && FAST) {
        KernargView Q = kernarg_view();
        const uint32_t wgs = min(Q->resolve_wgs, gridDim.x);  // a small launch has fewer workgroups than that
        if (Q->prev_results != nullptr && blockIdx.x < wgs && P.counters[10] == 0ull) {
            const uint32_t blocks = (Q->prev_active_tiles * 64u + 255u) / 256u;
            for (uint32_t b = blockIdx.x * (BLK / 256u) + (threadIdx.x >> 8); b < blocks; b += wgs * (BLK / 256u))
                resolve_block(b, threadIdx.x & 255u, reinterpret_cast<f4*>(Q->accum), Q->prev_results, Q->width, Q->height, Q->tiles_x,
                              Q->tiles_per_frame, Q->prev_active_tiles, Q->prev_frames, Q->prev_perm, Q->ray_slots, nullptr, 1u,
                              P.counters + 10, P.counters + 9);
        }
    }

    // One sample of a path has ended with `radiance`.
    auto finish_sample = [&](V3 radiance) {
        if constexpr (FAST) {
            // the same additions as below with the sum still at zero (0 + x keeps x's bits except for -0)
            float* o = reinterpret_cast<float*>(P.frame_out) + (size_t)out_index * 3u;
            // written once, read once by the accumulation kernel: streaming stores (accumulation 1.10 -> 1.04 ms per
            // 256 frames at 1920x1080)
            __builtin_nontemporal_store(0.0f + radiance.x, o), __builtin_nontemporal_store(0.0f + radiance.y, o + 1),
                __builtin_nontemporal_store(0.0f + radiance.z, o + 2);
            w.cur = LANE_IDLE;
            return;
        }
        final_colour = final_colour + radiance;
        sample_i++;
        w.cur = LANE_NEXT;
        if (sample_i >= P.samples) {
            f4 o4;
            o4.x = final_colour.x, o4.y = final_colour.y, o4.z = final_colour.z;
            if (P.samples != 1) {  // x / 1.0f == x: the common case needs no division
                const float fs = (float)P.samples;
                o4.x = final_colour.x / fs;
                o4.y = final_colour.y / fs;
                o4.z = final_colour.z / fs;
            }
            o4.w = 0.0f;
            store_result(P, out_index, o4, rays, lane_rays);
            w.cur = LANE_IDLE;
        }
    };
    // A lane whose walk has ended fetches the material word of its hit and moves to its scatter class.
    auto classify = [&]() {
        if (w.cur == WALK_FINISHED) {
            Hit hit;
            walk_result(w, hit);
            mbits = hit_material(mem, P.L, hit);
            const uint32_t cls = scatter_class(mbits, hit.back);
            w.cur = cls == SC_OPAQUE ? LANE_OPAQUE : (cls == SC_GLASS ? LANE_GLASS : LANE_TERMINAL);
            if (COUNT) {
                const uint32_t type = mbits & 7u;
                const uint32_t slot = type == MAT_SKY ? 4u : type;  // unknown (4) is not counted
#pragma unroll
                for (uint32_t k = 0; k < 5; k++) bounce_cnt[k] += (slot == k && type != MAT_UNKNOWN) ? 1u : 0u;
            }
        }
    };
    uint32_t waiting = 0;  // lanes counted by the last call of wants()
    auto wants = [&](bool mine, int ph) -> bool {
        const uint32_t n = (uint32_t)__popcll(__ballot(mine));
        waiting = n;
        const bool go = n >= PT_THR(ph);  // thresholds are >= 1
        if (COUNT && go) ph_exec[ph]++, ph_lanes[ph] += n;
        return go;
    };

    // ---- exchange of waiting paths between the waves of the workgroup (XCHG).  A lane whose walk has ended
    // holds nothing but its path: ray, throughput, generator, bounce count, the hit, where its result goes.
    // A wave that has too few lanes waiting for a scatter class to run it parks those paths in the class's pool
    // and frees the lanes; a wave that restarts takes a batch out of a well-filled pool into its idle lanes and
    // runs that scatter with many lanes at once.  One lock per pool, held by lane 0 across the copy.
    uint32_t x_dep = 0, x_dep_lanes = 0, x_wd = 0, x_wd_lanes = 0;
    [[maybe_unused]] const uint32_t pool = lds0 + P.pool_off;
    auto pool_word = [&](uint32_t off) { return reinterpret_cast<PT_LDS(uint32_t)*>(pool + off); };
    // Lane 0 only; the holder never waits for anything, so this ends.  False when the lock could not be had (a bug
    // must end the launch, not hang the GPU): flagged like a watchdog expiry -- the launch's results are discarded --
    // and the caller stays out of the critical section and makes the wave leave the scheduler loop (`abandon`).
    bool abandon = false;  // wave-uniform
    auto pool_lock = [&](uint32_t cls) -> bool {
        uint32_t expected = 0u, spins = 0u;
        while (!__hip_atomic_compare_exchange_strong(pool_word(cls * 8u), &expected, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP)) {
            expected = 0u;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) {
                atomicAdd(&P.counters[10], 1ull);
                return false;
            }
        }
        return true;
    };
    constexpr uint32_t NO_ROOM = 0xffffffffu, NO_LOCK = 0xfffffffeu;
    auto pool_unlock = [&](uint32_t cls) {
        __hip_atomic_store(pool_word(cls * 8u), 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto pool_count = [&](uint32_t cls) -> uint32_t {
        const uint32_t c = __hip_atomic_load(pool_word(cls * 8u + 4u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    };
    // Parks the paths of the lanes with `mine` (all in scatter class `cls`); false when the pool has no room.
    auto deposit = [&](uint32_t cls, bool mine) -> bool {
        const unsigned long long m = __ballot(mine);
        const uint32_t n = (uint32_t)__popcll(m);
        uint32_t base = NO_ROOM;
        if (lane == 0) {
            if (pool_lock(cls)) {
                const uint32_t c = __hip_atomic_load(pool_word(cls * 8u + 4u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (c + n <= POOL_CAP) base = c;
                else pool_unlock(cls);
            } else {
                base = NO_LOCK;
            }
        }
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= NO_LOCK) {
            if (base == NO_LOCK) abandon = true;
            return false;
        }
        if (mine) {
            // one word per store, straight from the registers the values live in (16-byte stores would need them
            // copied into aligned register quads first: vector instructions, the scarce kind here)
            const uint32_t a = pool + POOL_HEADER + cls * (POOL_CAP * 64u) + (base + lane_rank(m)) * 4u;
            auto put = [&](uint32_t k, uint32_t v) { *reinterpret_cast<PT_LDS(uint32_t)*>(a + k * (POOL_CAP * 4u)) = v; };
            put(0, __float_as_uint(ps.o.x)), put(1, __float_as_uint(ps.o.y)), put(2, __float_as_uint(ps.o.z));
            put(3, __float_as_uint(ps.d.x)), put(4, __float_as_uint(ps.d.y)), put(5, __float_as_uint(ps.d.z));
            put(6, __float_as_uint(ps.colour.x)), put(7, __float_as_uint(ps.colour.y)), put(8, __float_as_uint(ps.colour.z));
            put(9, ps.rng), put(10, (uint32_t)ps.bounce), put(11, __float_as_uint(w.best_t));
            put(12, (uint32_t)w.best_code), put(13, mbits), put(14, out_index);
            if constexpr (!FAST) put(15, rays);
            w.cur = LANE_IDLE;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // every lane's record is written before the count says so
        if (lane == 0) {
            __hip_atomic_store(pool_word(cls * 8u + 4u), base + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pool_unlock(cls);
        }
        if (COUNT) x_dep++, x_dep_lanes += n;
        return true;
    };
    // Takes up to one path per lane with `free_lane` out of pool `cls` when it holds at least `wmin`; the lanes
    // that got one are in that class's scatter state afterwards.  Returns whether anything was taken.
    auto withdraw = [&](uint32_t cls, bool free_lane, uint32_t wmin) -> bool {
        if (pool_count(cls) < wmin) return false;  // a look without the lock
        const unsigned long long m = __ballot(free_lane);
        const uint32_t k = (uint32_t)__popcll(m);
        if (k == 0u) return false;
        uint32_t base = 0u, take = 0u;
        if (lane == 0) {
            if (pool_lock(cls)) {
                const uint32_t c = __hip_atomic_load(pool_word(cls * 8u + 4u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                take = c >= wmin ? min(c, k) : 0u;
                base = c - take;
                if (take) __hip_atomic_store(pool_word(cls * 8u + 4u), base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else pool_unlock(cls);
            } else {
                base = NO_LOCK;
            }
        }
        take = (uint32_t)__builtin_amdgcn_readfirstlane((int)take);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (take == 0u) {
            if (base == NO_LOCK) abandon = true;
            return false;
        }
        const uint32_t rank = lane_rank(m);
        if (free_lane && rank < take) {
            const uint32_t a = pool + POOL_HEADER + cls * (POOL_CAP * 64u) + (base + rank) * 4u;
            auto get = [&](uint32_t k) { return *reinterpret_cast<const PT_LDS(uint32_t)*>(a + k * (POOL_CAP * 4u)); };
            ps.o = mk(__uint_as_float(get(0)), __uint_as_float(get(1)), __uint_as_float(get(2)));
            ps.d = mk(__uint_as_float(get(3)), __uint_as_float(get(4)), __uint_as_float(get(5)));
            ps.colour = mk(__uint_as_float(get(6)), __uint_as_float(get(7)), __uint_as_float(get(8)));
            ps.rng = get(9), ps.bounce = (int32_t)get(10);
            w.best_t = __uint_as_float(get(11)), w.best_code = (int32_t)get(12);
            mbits = get(13), out_index = get(14);
            if constexpr (!FAST) rays = get(15), final_colour = mk(0.0f, 0.0f, 0.0f), sample_i = 0;  // one sample per path in this variant
            w.cur = cls ? LANE_GLASS : LANE_OPAQUE;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // every lane has its record before the slots are free again
        if (lane == 0) pool_unlock(cls);
        if (COUNT) x_wd++, x_wd_lanes += take;
        return true;
    };

    uint64_t t_progress = __builtin_amdgcn_s_memrealtime();  // 100 MHz; when a path of this wave last ended
    for (uint32_t round = 0;; round++) {
        // Watchdog: a scheduling bug must end the launch, not hang the GPU.  It measures the time since a path
        // of this wave last ended, so a long launch that makes progress never trips it.
        if ((round & 1023u) == 1023u && __builtin_amdgcn_s_memrealtime() - t_progress > P.watchdog_ticks) {
            if (lane == 0) atomicAdd(&P.counters[10], 1ull);
            break;
        }
        if constexpr (XCHG) {
            if (abandon) break;  // a pool lock could not be had: the pools are not to be trusted, the launch is void
        }
        bool ran = false;
        // ---- interior nodes: keep stepping while enough lanes have one pending
        if (wants(walk_is_node(w.cur), PH_NODE)) {
            ran = true;
            for (;;) {
#pragma unroll
                for (int twice = 0; twice < PT_NODE_UNROLL; twice++) {
                    if (walk_is_node(w.cur)) {
                        if constexpr (SCENE == SCENE_REFERENCE) walk_node_reference<Mem, Stack, COUNT>(mem, P.L, stack, w, ps.o, ps.d, cnt);
                        else walk_node<Mem, Stack, COUNT, HALF>(mem, P.L, stack, w, cnt);
                    }
                }
                const uint32_t n = (uint32_t)__popcll(__ballot(walk_is_node(w.cur)));
                if (n < PT_THR(PH_NODE)) break;  // a forced round drains the state
                if (COUNT) ph_exec[PH_NODE]++, ph_lanes[PH_NODE] += n;
            }
        }
        // ---- leaves
        if (wants(walk_is_leaf(w.cur), PH_LEAF)) {
            ran = true;
            // One sphere per lane and step: lanes never wait for the longest leaf among them
            // (whole-leaf steps measured slower).  A forced round drains the state, as above:
            // 1-frame 1080p launch 1.05 -> 0.71 ms.
            for (;;) {
#pragma unroll
                for (int twice = 0; twice < PT_LEAF_UNROLL; twice++)
                    if (walk_is_leaf(w.cur)) walk_leaf_one<Mem, Stack, COUNT>(mem, P.L, stack, w, ps.o, ps.d, cnt);
                const uint32_t n = (uint32_t)__popcll(__ballot(walk_is_leaf(w.cur)));
                if (n < PT_THR(PH_LEAF)) break;
                if (COUNT) ph_exec[PH_LEAF]++, ph_lanes[PH_LEAF] += n;
            }
        }
        // Enough lanes are back at interior nodes (they came out of leaves): walk on before looking at the
        // scatter states, whose tests cost a dozen scalar instructions per round (scene1, 256 frames:
        // 20.95 -> 20.70 ms; asking for 8 or 16 lanes more than the node threshold: no gain).
        if (!force && (uint32_t)__popcll(__ballot(walk_is_node(w.cur))) >= P.thr[PH_NODE]) continue;
        // ---- finished walks: fetch the material word and sort the lane into its scatter class
        classify();
        // ---- restart, first half (exchange variant): close the paths that end here, then fill idle lanes from a
        // pool that holds enough paths of one scatter class; the scatter steps below then run them (a batch is at
        // least as large as the class's threshold: the host sees to pool_min >= thr)
        bool do_restart = false;
        [[maybe_unused]] const bool may_park = !force && !exhausted;  // a wave that has seen the queue empty only takes
        if constexpr (XCHG) {
            do_restart = wants(lane_wants_restart(w.cur), PH_IDLE);
            if (do_restart || (exhausted && __ballot(w.cur == LANE_IDLE || w.cur == LANE_EXIT))) {
                if (__ballot(w.cur == LANE_TERMINAL)) t_progress = __builtin_amdgcn_s_memrealtime();
                if (w.cur == LANE_TERMINAL) finish_sample(scatter_terminal(mem, P.L, mbits, ps));
                // glass first (the class that runs thinnest on its own); with the queue empty whatever is left
                if (withdraw(1u, w.cur == LANE_IDLE || w.cur == LANE_EXIT, exhausted ? 1u : POOL_MIN_GLASS)) ran = true;
                if (withdraw(0u, w.cur == LANE_IDLE || w.cur == LANE_EXIT, exhausted ? 1u : POOL_MIN_OPAQUE)) ran = true;
            }
        }
        // ---- lambertian / metal
        bool run_opaque = wants(w.cur == LANE_OPAQUE, PH_OPAQUE);
        if constexpr (XCHG) {
            if (!run_opaque && may_park && waiting >= POOL_DEP_MIN_OPAQUE * (LDS_SCENE ? POOL_DEP_LDS_FACTOR : 1u)) {
                if (deposit(0u, w.cur == LANE_OPAQUE)) ran = true;
                else run_opaque = true;  // the pool is full: the waves that empty it are behind
            }
        }
        if (run_opaque) {
            ran = true;
            if (w.cur == LANE_OPAQUE) {
                Hit hit;
                walk_result(w, hit);
                V3 radiance;
                if (scatter_opaque(mem, P.L, hit, mbits, ps, P.depth, radiance)) {
                    finish_sample(radiance);
                } else {
                    walk_begin<HALF>(w, P.L, ps.o, ps.d, mem.root());
                    if constexpr (SCENE != SCENE_REFERENCE) walk_test_hoisted<Mem, COUNT>(mem, P.L, w, ps.o, ps.d, cnt);
                    if constexpr (FAST) lane_rays++; else rays++;
                }
            }
        }
        // ---- glass
        bool run_glass = wants(w.cur == LANE_GLASS, PH_GLASS);
        if constexpr (XCHG) {
            if (!run_glass && may_park && waiting >= POOL_DEP_MIN_GLASS * (LDS_SCENE ? POOL_DEP_LDS_FACTOR : 1u)) {
                if (deposit(1u, w.cur == LANE_GLASS)) ran = true;
                else run_glass = true;
            }
        }
        if (run_glass) {
            ran = true;
            if (w.cur == LANE_GLASS) {
                Hit hit;
                walk_result(w, hit);
                V3 radiance;
                if (scatter_glass(mem, P.L, hit, mbits, ps, P.depth, P.glass_always_draws != 0, radiance)) {
                    finish_sample(radiance);
                } else {
                    walk_begin<HALF>(w, P.L, ps.o, ps.d, mem.root());
                    if constexpr (SCENE != SCENE_REFERENCE) walk_test_hoisted<Mem, COUNT>(mem, P.L, w, ps.o, ps.d, cnt);
                    if constexpr (FAST) lane_rays++; else rays++;
                }
            }
        }
        // ---- restart: close the paths that end at this hit / miss, hand consecutive path ids from the
        // wave's chunk to lanes without a path, start the next sample.  One step on purpose: every
        // path needs both halves exactly once, and a separate state would only thin out the others.
        if constexpr (!XCHG) do_restart = wants(lane_wants_restart(w.cur), PH_IDLE);
        if (do_restart) {
            ran = true;
            KernargView Q = kernarg_view();  // what only this step needs is read here (see kernarg_view)
            if constexpr (!XCHG) {
                if (__ballot(w.cur == LANE_TERMINAL)) t_progress = __builtin_amdgcn_s_memrealtime();
                if (w.cur == LANE_TERMINAL) finish_sample(scatter_terminal(mem, P.L, mbits, ps));
            }
            uint32_t pxy_step = 0;
            uint32_t& pxy_now = FAST ? pxy_step : pxy;  // with one sample per path a lane gets its camera ray in this very step
            unsigned long long m = __ballot(w.cur == LANE_IDLE);
            while (m) {
                if (next == end) {
                    if (exhausted) break;
                    uint32_t base = 0, taken_to;
                    if constexpr (XCHG && PT_WG_QUEUE > 0) {
                        // The workgroup takes PT_WG_QUEUE chunks from the queue at once and its waves share them out:
                        // the waves of a compute unit then work on the same few tiles.  State in the pool header:
                        // lock, next unit, end (both ~0 once the queue has been seen empty).
                        uint32_t got = 0;
                        if (lane == 0 && !pool_lock(4u)) got = NO_LOCK;
                        if (lane == 0 && got != NO_LOCK) {
                            uint32_t n = __hip_atomic_load(pool_word(36u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            uint32_t e = __hip_atomic_load(pool_word(40u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (n >= e && n != 0xffffffffu) {
                                const uint32_t big = chunk * (uint32_t)PT_WG_QUEUE;
                                n = atomicAdd(Q->work_counter, big);
                                e = min(n + big, Q->total_units);
                                if (n >= Q->total_units) n = e = 0xffffffffu;
                            }
                            if (n != 0xffffffffu) got = min(chunk, e - n), base = n, n += got;
                            __hip_atomic_store(pool_word(36u), n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_store(pool_word(40u), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            pool_unlock(4u);
                        }
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                        if (got == 0u || got == NO_LOCK) {
                            exhausted = true;
                            abandon = got == NO_LOCK;
                            break;
                        }
                        next = base * 64u;
                        end = (base + got) * 64u;
                        taken_to = base + got;
                    } else {
                        if (lane == 0) base = atomicAdd(Q->work_counter, chunk);
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);  // lane 0 is active here: the step is entered by the whole wave
                        if (base >= Q->total_units) {
                            exhausted = true;
                            break;
                        }
                        next = base * 64u;
                        end = min(base + chunk, Q->total_units) * 64u;
                        taken_to = base + chunk;
                    }
                    {
                        chunk = guided_chunk_units(Q->total_units, taken_to, Q->inv_share, Q->chunk_max);
                        if (Q->rank_chunk && taken_to < Q->total_units) {
                            // about where the queue will stand at this wave's next visit
                            const uint32_t r = min(Q->tiles_per_frame - 1u, (uint32_t)((float)taken_to * Q->inv_n_frames));
                            chunk = min(chunk, (uint32_t)Q->rank_chunk[r]);
                        }
                    }
                }
                const uint32_t avail = end - next;
                const uint32_t rank = lane_rank(m);
                if (w.cur == LANE_IDLE && rank < avail) {
                    const uint32_t pid = next + rank;
                    const uint32_t unit = pid >> 6, slot = pid & 63u;
                    uint32_t frame, tile, tx, ty;
                    if (Q->small_ids) {  // uniform: every id below 2^24, float-reciprocal division
                        divmod24(unit, Q->n_frames, Q->inv_n_frames, tile, frame);
                    } else {
                        tile = unit / Q->n_frames, frame = unit - tile * Q->n_frames;
                    }
                    const uint32_t result_index = (frame * Q->tiles_per_frame + tile) * 64u + slot;
                    if (Q->tile_perm) tile = Q->tile_perm[tile];
                    if (Q->small_ids) {
                        divmod24(tile, Q->tiles_x, Q->inv_tiles_x, ty, tx);
                    } else {
                        ty = tile / Q->tiles_x, tx = tile - ty * Q->tiles_x;
                    }
                    const uint32_t px = tx * 8u + (slot & 7u), py = ty * 8u + (slot >> 3);
                    if (px < Q->width && py < Q->height) {
                        const uint32_t pix = py * Q->width + px;
                        ps.rng = pixel_seed(Q->seeds[frame], Q->pixel_hash[pix]);
                        out_index = result_index;
                        pxy_now = px | (py << 16);
                        if constexpr (!FAST) final_colour = mk(0.0f, 0.0f, 0.0f), sample_i = 0, rays = 0;
                        w.cur = LANE_NEXT;
                    }
                }
                next += min((uint32_t)__popcll(m), avail);
                m = __ballot(w.cur == LANE_IDLE);
            }
            if (w.cur == LANE_IDLE) w.cur = LANE_EXIT;  // the queue is empty: this lane is finished
            if (w.cur == LANE_NEXT) {
                const CameraRegs cam = {mk(Q->cam.pos[0], Q->cam.pos[1], Q->cam.pos[2]), mk(Q->cam.horizontal[0], Q->cam.horizontal[1], Q->cam.horizontal[2]),
                                        mk(Q->cam.vertical[0], Q->cam.vertical[1], Q->cam.vertical[2]), mk(Q->cam.center[0], Q->cam.center[1], Q->cam.center[2])};
                const bool short_div = Q->short_div != 0u;
                ps.d = camera_dir(cam, pxy_now & 0xffffu, pxy_now >> 16, PixelDiv((float)Q->width, Q->rcp_w, short_div),
                                  PixelDiv((float)Q->height, Q->rcp_h, short_div), ps.rng);
                ps.o = cam.pos;
                ps.colour = mk(1.0f, 1.0f, 1.0f);
                ps.bounce = 0;
                if constexpr (FAST) lane_rays++; else rays++;
                bool walks = true;
                if constexpr (SCENE != SCENE_REFERENCE) {
                    // Camera rays test the candidates of their tile instead of walking the tree (tile_list_build);
                    // the lanes of a restart step share a tile or two, so the loop below runs nearly uniform.  Such
                    // a ray needs none of the walk's per-ray set-up (reciprocal direction, signs, stack).
                    if (Q->tile_lists) {
                        const uint32_t tile_id = (pxy_now >> 19) * Q->tiles_x + ((pxy_now & 0xffffu) >> 3);
                        const uint32_t* rec = Q->tile_lists + (size_t)tile_id * TILE_LIST_WORDS;
                        // count and the first three candidates in one fetch: all that nine tiles in ten hold
                        typedef uint32_t w4 __attribute__((ext_vector_type(4)));
                        const w4 head = *reinterpret_cast<const w4*>(rec);
                        const uint32_t n = head.x;
                        if (n != TILE_LIST_WALK) {
                            walks = false;
                            w.best_t = INFINITY, w.best_code = -1, w.sp = 0;
                            if (n > 0u) walk_test_sphere<Mem, COUNT>(mem, P.L, w, ps.o, ps.d, head.y, cnt);
                            if (n > 1u) walk_test_sphere<Mem, COUNT>(mem, P.L, w, ps.o, ps.d, head.z, cnt);
                            if (n > 2u) walk_test_sphere<Mem, COUNT>(mem, P.L, w, ps.o, ps.d, head.w, cnt);
                            for (uint32_t k = 3; k < n; k++) walk_test_sphere<Mem, COUNT>(mem, P.L, w, ps.o, ps.d, rec[1u + k], cnt);
                            w.cur = WALK_FINISHED;
                        }
                    }
                }
                if (walks) {
                    walk_begin<HALF>(w, P.L, ps.o, ps.d, mem.root());
                    if constexpr (SCENE != SCENE_REFERENCE) walk_test_hoisted<Mem, COUNT>(mem, P.L, w, ps.o, ps.d, cnt);
                }
            }
        }
        if (!ran) {
            if (__ballot(w.cur != LANE_EXIT) == 0ull) {
                // Nothing left in this wave.  Parked paths are taken by waves that have seen the queue empty (this
                // one has); a wave that parks one later has not, and will see it empty -- and take them -- itself.
                if constexpr (XCHG) {
                    if (pool_count(0) + pool_count(1) == 0u) break;
                } else {
                    break;
                }
            }
            force = true;
        } else {
            force = false;
        }
    }

    flush_rays(P, lane_rays);
#ifdef PT_WAVE_END_HIST
    // development build: when the waves end, as a histogram over the device's 100 MHz clock (buckets of
    // 2^PT_WAVE_END_HIST ticks, modulo 64): the shape of a launch's tail
    if (lane == 0) atomicAdd(&P.counters[32 + ((__builtin_amdgcn_s_memrealtime() >> PT_WAVE_END_HIST) & 63u)], 1ull);
#endif
    if (COUNT) {
        unsigned long long v[8] = {cnt.nodes,     cnt.tests,     cnt.accepted,  bounce_cnt[0],
                                   bounce_cnt[1], bounce_cnt[2], bounce_cnt[3], bounce_cnt[4]};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            unsigned long long s = wave_sum(v[k]);
            if (lane == 0) atomicAdd(&P.counters[k], s);
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < PH_COUNT; k++) {
                atomicAdd(&P.counters[16 + 2 * k], (unsigned long long)ph_exec[k]);
                atomicAdd(&P.counters[17 + 2 * k], (unsigned long long)ph_lanes[k]);
            }
            if constexpr (XCHG) {
                atomicAdd(&P.counters[11], (unsigned long long)x_dep), atomicAdd(&P.counters[12], (unsigned long long)x_dep_lanes);
                atomicAdd(&P.counters[13], (unsigned long long)x_wd), atomicAdd(&P.counters[14], (unsigned long long)x_wd_lanes);
            }
        }
    }
}


// The tiles at ranks [first_rank, tiles) of the order show nothing but sky (their candidate lists are empty): every
// path there is one camera ray and the sky colour.  No queue and no result buffer: a workgroup owns a tile, its
// four waves work out the colours of four consecutive frames of the tile's 64 pixels side by side (the colours
// do not depend on one another, only the sums do), and the first wave adds them to its running sums in frame
// order.  Same arithmetic as the path kernels followed by the accumulation kernel (sky_pixel_frame), so the
// texel's bits are the same.
constexpr uint32_t SKY_FRAMES = 4;  // frames in flight per tile = waves per workgroup
__device__ __forceinline__ void sky_block(uint32_t block, f4* accum, Camera64 cam64, uint32_t width, uint32_t height,
                                                                      uint32_t tiles_x, uint32_t first_rank, uint32_t n_frames,
                                                                      const uint32_t* tile_perm, const uint32_t* seeds,
                                                                      const uint32_t* pixel_hash, unsigned long long* ray_total,
                                                                      const unsigned long long* gave_up, float rcp_w, float rcp_h,
                                                                      uint32_t short_div) {
    __shared__ float colour[SKY_FRAMES][3][64];
    const uint32_t slot = threadIdx.x & 63u, sub = threadIdx.x >> 6;
    const uint32_t tile = tile_perm[first_rank + block];
    const uint32_t px = (tile % tiles_x) * 8u + (slot & 7u), py = (tile / tiles_x) * 8u + (slot >> 3);
    const bool inside = px < width && py < height;  // the barriers below are reached by every lane
    const CameraRegs cam = load_camera(cam64);
    const PixelDiv fw((float)width, rcp_w, short_div != 0u), fh((float)height, rcp_h, short_div != 0u);
    const uint32_t i = inside ? py * width + px : 0u, hash = pixel_hash[i];
    f4 a = accum[i];
    for (uint32_t f0 = 0; f0 < n_frames; f0 += SKY_FRAMES) {
        const uint32_t f = f0 + sub;
        if (f < n_frames) {
            const V3 c = sky_pixel_frame(cam, px, py, fw, fh, seeds[f], hash);
            colour[sub][0][slot] = c.x, colour[sub][1][slot] = c.y, colour[sub][2][slot] = c.z;
        }
        __syncthreads();
        if (sub == 0) {
            const uint32_t n = min(SKY_FRAMES, n_frames - f0);
            for (uint32_t k = 0; k < n; k++) a.x += colour[k][0][slot], a.y += colour[k][1][slot], a.z += colour[k][2][slot], a.w += 0.0f;
        }
        __syncthreads();
    }
    if (sub == 0) {
        if (inside) accum[i] = a;
        const unsigned long long rays = wave_sum(inside ? (unsigned long long)n_frames : 0ull);  // one closest-hit query per path
        if (slot == 0 && rays) atomicAdd(ray_total + (size_t)(block % RAY_SLOTS) * RAY_SLOT_STRIDE, rays);
    }
}


// The accumulation of one launch as ONE kernel: `sky_blocks` of its workgroups each take a sky tile (sky_block:
// bound by vector instructions), the others 256 texels of the buffered results (resolve_block: bound by HBM
// reads).  The two kinds are interleaved in proportion to their numbers, so that -- workgroups being dispatched in
// index order -- both are in flight on every compute unit from the first to the last moment of the grid.
__global__ void __launch_bounds__(256) pt_accumulate_kernel(f4* accum, const f4* results, Camera64 cam64, uint32_t width, uint32_t height,
                                                           uint32_t tiles_x, uint32_t tiles, uint32_t active_tiles, uint32_t n_frames,
                                                           const uint32_t* tile_perm, const uint32_t* seeds, const uint32_t* pixel_hash,
                                                           unsigned long long* ray_total, float* tile_cost, uint32_t compact,
                                                           const unsigned long long* gave_up, unsigned long long* launches_added,
                                                           uint32_t sky_blocks, float rcp_w, float rcp_h, uint32_t short_div) {
    // The path kernel's watchdog fired (now or in an earlier launch of the queue): its results are incomplete,
    // so nothing of this launch is added.
    if (*gave_up) return;
    // sky tiles among the first i workgroups: floor(i * sky_blocks / gridDim.x)
    const uint32_t before = (uint32_t)((uint64_t)blockIdx.x * sky_blocks / gridDim.x);
    const uint32_t upto = (uint32_t)(((uint64_t)blockIdx.x + 1u) * sky_blocks / gridDim.x);
    if (upto > before)
        sky_block(before, accum, cam64, width, height, tiles_x, active_tiles, n_frames, tile_perm, seeds, pixel_hash, ray_total,
                  gave_up, rcp_w, rcp_h, short_div);
    else
        resolve_block(blockIdx.x - before, threadIdx.x, accum, results, width, height, tiles_x, tiles, active_tiles, n_frames, tile_perm,
                      ray_total, tile_cost, compact, gave_up, launches_added);
}

// Tiles (8x8 pixels) ordered by decreasing cost (rays traced in them during the last launch): one
// workgroup, a 256-bucket counting sort.  Ties land in arbitrary order, which nothing depends on.
constexpr int ORDER_BLK = 1024;
// rank_chunk[r]: units a chunk starting in the r-th tile of that order should hold so that it costs about
// `chunk_rays` rays (the cost of one unit of a tile = its measured rays / frames_measured).
// With `lists` (the camera rays' candidate lists) the tiles whose list is empty -- sky and nothing else -- come
// last, whatever they cost: launches that leave them to the sky blocks of pt_accumulate_kernel end their queue before them.
__global__ void __launch_bounds__(ORDER_BLK) pt_tile_order_kernel(const float* tile_cost, uint32_t tiles, uint32_t* perm,
                                                                 uint16_t* rank_chunk, float frames_measured,
                                                                 float chunk_rays, uint32_t chunk_max, const uint32_t* lists) {
    __shared__ uint32_t hist[257];
    __shared__ float red[ORDER_BLK];
    float mx = 0.0f;
    for (uint32_t t = threadIdx.x; t < tiles; t += ORDER_BLK) mx = fmaxf(mx, tile_cost[t]);
    red[threadIdx.x] = mx;
    if (threadIdx.x < 257) hist[threadIdx.x] = 0u;
    __syncthreads();
    for (int o = ORDER_BLK / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    const float scale = 255.0f / fmaxf(red[0], 1.0f);
    auto bucket = [&](uint32_t t) {  // 0 = most expensive, 256 = sky only
        if (lists && lists[(size_t)t * TILE_LIST_WORDS] == 0u) return 256u;
        return 255u - min(255u, (uint32_t)(tile_cost[t] * scale));
    };
    for (uint32_t t = threadIdx.x; t < tiles; t += ORDER_BLK) atomicAdd(&hist[bucket(t)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {  // exclusive prefix over the buckets
        uint32_t run = 0;
        for (int b = 0; b < 257; b++) {
            const uint32_t n = hist[b];
            hist[b] = run;
            run += n;
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < tiles; t += ORDER_BLK) {
        const float c = tile_cost[t];
        const uint32_t r = atomicAdd(&hist[bucket(t)], 1u);
        perm[r] = t;
        const float unit_rays = fmaxf(c / frames_measured, 1.0f);
        rank_chunk[r] = (uint16_t)max(1u, min(chunk_max, (uint32_t)(chunk_rays / unit_rays)));
    }
}

// Is PixelDiv's short division by `d` exact?  Compares it with the general division for every binary32 numerator
// with bits in [lo, hi] and for zero; any difference sets *bad.  Run once per image size and divisor (resize).
__global__ void pt_verify_div_kernel(float d, float rcp, uint32_t lo, uint32_t hi, uint32_t* bad) {
    const PixelDiv div(d, rcp, true);
    bool differs = false;
    for (uint64_t b = (uint64_t)lo + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b <= hi; b += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)b);
        differs |= __float_as_uint(div.short_quotient(x)) != __float_as_uint(x / d);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) differs |= __float_as_uint(div.short_quotient(0.0f)) != __float_as_uint(0.0f / d);
    if (__ballot(differs) && (threadIdx.x & 63u) == 0) atomicOr(bad, 1u);
}

__global__ void pt_fill_kernel(f4* p, uint32_t n, f4 v) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------ host
// Development switches (environment variables; not part of the ABI), read once per context.
static long env_long(const char* name, long dflt) {
    const char* e = getenv(name);
    return e && *e ? atol(e) : dflt;
}
static void read_switches(Switches& s) {
    s.max_batch = (uint32_t)std::max(0l, env_long("MI355PT_MAX_BATCH", 0));
    s.frame_buf_gb = (uint32_t)std::max(0l, env_long("MI355PT_FRAME_BUF_GB", 0));
    s.block = (uint32_t)env_long("MI355PT_BLOCK", 0);
    s.stack_lds = env_long("MI355PT_STACK_LDS", 5);
    s.node_cache_kb = env_long("MI355PT_NODE_CACHE_KB", -1);
    s.tile_order = env_long("MI355PT_TILE_ORDER", 1) != 0;
    s.chunk = (uint32_t)std::min(1024l, std::max(0l, env_long("MI355PT_CHUNK", 0)));
    if (const char* e = getenv("MI355PT_CHUNK_SHARE")) s.chunk_share = atof(e) > 0.0 ? (float)atof(e) : 2.0f;
    if (const char* e = getenv("MI355PT_CHUNK_RAYS")) s.chunk_rays = std::max(64.0f, (float)atof(e));
    if (const char* e = getenv("MI355PT_KERNEL")) s.lockstep = strcmp(e, "lockstep") == 0;
    s.watchdog_ms = (uint64_t)std::max(1l, env_long("MI355PT_WATCHDOG_MS", 20000));
    s.exchange = env_long("MI355PT_XCHG", 1) != 0;
    s.exchange_global = env_long("MI355PT_XCHG_GLOBAL", 1) != 0;
    {
        const char* nodes = getenv("MI355PT_NODES");
        s.ordered_global = !(nodes && strcmp(nodes, "plain") == 0);
    }
    s.measure_every = (uint32_t)std::max(1l, env_long("MI355PT_MEASURE_EVERY", 64));
    s.defer_resolve = env_long("MI355PT_DEFER_RESOLVE", 1) != 0;
    s.resolve_wgs = (uint32_t)std::min(4096l, std::max(1l, env_long("MI355PT_RESOLVE_WGS", 16)));
    s.short_div = env_long("MI355PT_SHORT_DIV", 1) != 0;
    s.sky_tiles = env_long("MI355PT_SKY_TILES", 1) != 0;
    s.tile_lists = env_long("MI355PT_TILE_LISTS", 1) != 0;
    s.tile_list_max = (uint32_t)std::min<long>(TILE_LIST_CAP, std::max(0l, env_long("MI355PT_TILE_LIST_MAX", TILE_LIST_CAP)));
    if (const char* senv = getenv("MI355PT_XCHG_SCHED")) {
        unsigned t[6];
        if (sscanf(senv, "%u,%u,%u,%u,%u,%u", &t[0], &t[1], &t[2], &t[3], &t[4], &t[5]) == 6)
            for (int k = 0; k < 6; k++) s.xthr[k] = t[k] < 1 ? 1 : (t[k] > 65 ? 65 : t[k]);
    }
    if (const char* senv = getenv("MI355PT_SCHED")) {
        unsigned t[6];
        if (sscanf(senv, "%u,%u,%u,%u,%u,%u", &t[0], &t[1], &t[2], &t[3], &t[4], &t[5]) == 6)
            for (int k = 0; k < 6; k++) s.thr[k] = t[k] < 1 ? 1 : (t[k] > 64 ? 64 : t[k]);
    }
}

// Static workspace of render_static(): callers that cannot allocate (tight serving loops, harnesses that forbid
// hipMalloc) render with the frame buffer, the seeds, the work-queue counter and
// the stack spill area in this array instead of in buffers the context grows on demand.  Every device the
// library runs on holds its own copy at its own address until the process ends.  In host code the array's
// name is only a handle for hipGetSymbolAddress; kernels and memsets get the looked-up address.
constexpr size_t STATIC_WS_BYTES = (size_t)160 << 20;
constexpr size_t STATIC_WS_COUNTER = 0;        // work-queue counter
constexpr size_t STATIC_WS_SEEDS = 256;        // STATIC_MAX_FRAMES seeds
constexpr size_t STATIC_WS_DATA = 256 + (size_t)STATIC_MAX_FRAMES * 4;  // stack spill area, then results
static_assert(STATIC_WS_DATA % 256 == 0, "sections of the static workspace are 256-byte aligned");
static __device__ __align__(256) unsigned char scratch[STATIC_WS_BYTES];

static std::mutex g_static_ws_mutex;
static unsigned char* g_static_ws[64] = {nullptr};  // by device ordinal
// Address of `scratch` on device `id` (which must be current), looked up once per device.
static bool static_ws_address(int id, unsigned char** out, std::string& err) {
    std::lock_guard<std::mutex> lock(g_static_ws_mutex);
    if (id < 0 || id >= 64) {
        *out = nullptr;  // render_static reports it
        return true;
    }
    if (!g_static_ws[id]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            err = std::string("hipGetSymbolAddress(static workspace): ") + hipGetErrorString(e);
            return false;
        }
        g_static_ws[id] = static_cast<unsigned char*>(p);
    }
    *out = g_static_ws[id];
    return true;
}

bool create(int device_id, void* stream, Device** out, std::string& err) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        err = "no HIP device available (this library has no CPU path)";
        return false;
    }
    if (device_id < 0 || device_id >= n) {
        err = "device ordinal out of range";
        return false;
    }
    DeviceGuard guard(device_id);
    Device* d = new Device();
    d->id = device_id;
    read_switches(d->sw);
    if (hipGetDeviceProperties(&d->prop, device_id) != hipSuccess) {
        delete d;
        err = "hipGetDeviceProperties failed";
        return false;
    }
    if (stream) {
        d->stream = (hipStream_t)stream;
    } else {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&d->stream, hipStreamNonBlocking, greatest) != hipSuccess) {
            delete d;
            err = "hipStreamCreate failed";
            return false;
        }
        d->own_stream = true;
    }
    if (hipMalloc(&d->work_counter, 256) != hipSuccess || hipMalloc(&d->counters, COUNTER_BYTES) != hipSuccess ||
        hipMalloc(&d->ray_slots, RAY_SLOTS * RAY_SLOT_STRIDE * 8) != hipSuccess || hipMalloc(&d->list_count, 256) != hipSuccess) {
        destroy(d);
        err = "hipMalloc failed";
        return false;
    }
    (void)hipMemset(d->counters, 0, COUNTER_BYTES);
    (void)hipMemset(d->ray_slots, 0, RAY_SLOTS * RAY_SLOT_STRIDE * 8);
    if (!static_ws_address(device_id, &d->static_ws, err)) {
        destroy(d);
        return false;
    }
    *out = d;
    return true;
}

void destroy(Device* d) {
    if (!d) return;
    DeviceGuard guard(d->id);
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    for (auto& t : d->timed) d->event_pool.push_back(t.begin), d->event_pool.push_back(t.end);
    for (hipEvent_t e : d->event_pool) (void)hipEventDestroy(e);
    (void)hipFree(d->scene);
    (void)hipFree(d->onodes);
    (void)hipFree(d->accum);
    (void)hipFree(d->pixel_hash);
    (void)hipFree(d->frame_buf);
    (void)hipFree(d->tile_cost);
    (void)hipFree(d->tile_perm);
    (void)hipFree(d->rank_chunk);
    (void)hipFree(d->tile_lists);
    (void)hipFree(d->list_count);
    (void)hipFree(d->seeds);
    (void)hipFree(d->stack_spill);
    (void)hipFree(d->work_counter);
    (void)hipFree(d->counters);
    (void)hipFree(d->ray_slots);
    if (d->own_stream && d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
}

int device_id(Device* d) { return d->id; }
bool scene_in_lds(Device* d) { return d->have_scene && d->scene_in_lds; }
void* accum_device_ptr(Device* d) { return d->accum; }
void* stream_handle(Device* d) { return d->stream; }

bool device_info(Device* d, int* cus, int* clock_khz, size_t* lds_per_cu, char name[256]) {
    if (cus) *cus = d->prop.multiProcessorCount;
    if (clock_khz) *clock_khz = d->prop.clockRate;
    if (lds_per_cu) *lds_per_cu = d->prop.maxSharedMemoryPerMultiProcessor;
    if (name) {
        strncpy(name, d->prop.name, 255);
        name[255] = 0;
    }
    return true;
}

bool synchronize(Device* d, std::string& err) {
    DeviceGuard guard(d->id);
    PT_HIP(hipStreamSynchronize(d->stream));
    return true;
}
#define PT_SYNC_ALL(d) PT_HIP(hipStreamSynchronize((d)->stream))

static uint32_t stack_bytes(const SceneLayout& L, uint32_t block) { return std::max(1u, L.max_depth) * block * 4u; }

bool upload_scene(Device* d, const void* blob, const SceneLayout& layout, std::string& err) {
    DeviceGuard guard(d->id);
    PT_SYNC_ALL(d);
    size_t bytes = ((size_t)layout.total_bytes + 15u) & ~(size_t)15u;
    if (bytes > d->scene_cap) {
        (void)hipFree(d->scene);
        d->scene = nullptr;
        d->scene_cap = 0;
        PT_HIP(hipMalloc(&d->scene, bytes));
        d->scene_cap = bytes;
    }
    PT_HIP(hipMemset(d->scene, 0, bytes));
    PT_HIP(hipMemcpy(d->scene, blob, layout.total_bytes, hipMemcpyHostToDevice));
    return adopt_layout(d, layout, err);
}

bool adopt_layout(Device* d, const SceneLayout& layout, std::string& err) {
    d->layout = layout;
    d->have_scene = true;
    d->perm_valid = false;
    d->lists_valid = false;
    d->scene_in_lds = !layout.reference_walk && lds_scene_bytes(layout) + stack_bytes(layout, 256) <= LDS_SCENE_LIMIT;
    if (layout.node_half && !d->scene_in_lds) {
        err = "scene: half-precision nodes are only walked from LDS";
        d->have_scene = false;
        return false;
    }
    // A tree that stays in global memory is walked through direction-ordered records, built here from the blob's
    // Node64 section (MI355PT_NODES=plain keeps the Node64 walk); links are byte offsets, hence the size limit.
    d->onodes_valid = false;
    if (!d->scene_in_lds && !layout.reference_walk && layout.off_nodes == 0 && d->sw.ordered_global &&
        (uint64_t)layout.n_nodes * GLOBAL_NODE_BYTES < (1ull << 31)) {
        DeviceGuard guard(d->id);
        const size_t need = (size_t)layout.n_nodes * GLOBAL_NODE_BYTES;
        if (need > d->onodes_cap) {
            (void)hipFree(d->onodes);
            d->onodes = nullptr;
            d->onodes_cap = 0;
            if (hipMalloc(&d->onodes, need) != hipSuccess) {  // no room for the second copy: the Node64 walk needs none
                (void)hipGetLastError();
                d->onodes = nullptr;
                return true;
            }
            d->onodes_cap = need;
        }
        hipLaunchKernelGGL(pt_order_nodes_kernel, dim3((layout.n_nodes + 255u) / 256u), dim3(256), 0, d->stream,
                           reinterpret_cast<const uint32_t*>(d->scene + layout.off_nodes), layout.n_nodes,
                           reinterpret_cast<uint32_t*>(d->onodes));
        PT_HIP(hipGetLastError());
        PT_HIP(hipStreamSynchronize(d->stream));
        d->onodes_valid = true;
    }
    return true;
}

bool resize(Device* d, uint32_t width, uint32_t height, std::string& err) {
    if (width == 0 || height == 0 || width > 65535u || height > 65535u) {
        err = "image size must be in [1, 65535] per side";
        return false;
    }
    DeviceGuard guard(d->id);
    PT_SYNC_ALL(d);
    uint32_t n_pix = width * height;
    (void)hipFree(d->accum);
    (void)hipFree(d->pixel_hash);
    (void)hipFree(d->tile_cost);
    (void)hipFree(d->tile_perm);
    (void)hipFree(d->rank_chunk);
    (void)hipFree(d->tile_lists);
    d->rank_chunk = nullptr;
    d->tile_lists = nullptr;
    d->lists_valid = false;
    d->accum = nullptr;
    d->pixel_hash = nullptr;
    d->tile_cost = nullptr, d->tile_perm = nullptr;
    d->perm_valid = false;
    const size_t n_tiles = (size_t)((width + 7) / 8) * ((height + 7) / 8);
    PT_HIP(hipMalloc(&d->accum, (size_t)n_pix * sizeof(f4)));
    PT_HIP(hipMalloc(&d->pixel_hash, (size_t)n_pix * sizeof(uint32_t)));
    PT_HIP(hipMalloc(&d->tile_cost, n_tiles * sizeof(float)));
    PT_HIP(hipMalloc(&d->tile_perm, n_tiles * sizeof(uint32_t)));
    PT_HIP(hipMalloc(&d->rank_chunk, n_tiles * sizeof(uint16_t)));
    PT_HIP(hipMalloc(&d->tile_lists, n_tiles * TILE_LIST_WORDS * sizeof(uint32_t)));
    d->width = width, d->height = height, d->n_pix = n_pix;
    hipLaunchKernelGGL(pt_hash_kernel, dim3((n_pix + 255) / 256), dim3(256), 0, d->stream, d->pixel_hash, width,
                       height);
    PT_HIP(hipGetLastError());
    // The short division of the camera rays' pixel coordinates (PixelDiv) is used only if it gives the general
    // division's bits for every numerator there can be -- px + r with px < 65536 and r in [2^-32, 1] or r = 0 --
    // which the device checks here, once per size, for both divisors (0.8 G quotients: under a millisecond).
    d->rcp_w = 1.0f / (float)width, d->rcp_h = 1.0f / (float)height;
    d->short_div = false;
    if (d->sw.short_div) {
        PT_HIP(hipMemsetAsync(d->list_count + 8, 0, 4, d->stream));
        hipLaunchKernelGGL(pt_verify_div_kernel, dim3(4096), dim3(256), 0, d->stream, (float)width, d->rcp_w, 0x2F800000u, 0x47800000u, d->list_count + 8);
        hipLaunchKernelGGL(pt_verify_div_kernel, dim3(4096), dim3(256), 0, d->stream, (float)height, d->rcp_h, 0x2F800000u, 0x47800000u, d->list_count + 8);
        PT_HIP(hipGetLastError());
        uint32_t bad = 1;
        PT_HIP(hipStreamSynchronize(d->stream));
        PT_HIP(hipMemcpy(&bad, d->list_count + 8, 4, hipMemcpyDeviceToHost));
        d->short_div = bad == 0u;
    }
    return clear(d, err);
}

bool clear(Device* d, std::string& err) {
    if (!d->accum) {
        err = "no accumulator (resize first)";
        return false;
    }
    DeviceGuard guard(d->id);
    f4 v{0.0f, 0.0f, 0.0f, 1.0f};  // LoadOp::Clear(0,0,0,1), reference mod.rs:182-187
    hipLaunchKernelGGL(pt_fill_kernel, dim3((d->n_pix + 255) / 256), dim3(256), 0, d->stream, d->accum, d->n_pix,
                       v);
    PT_HIP(hipGetLastError());
    PT_HIP(hipMemsetAsync(d->counters, 0, COUNTER_BYTES, d->stream));
    d->added_seen = 0;
    PT_HIP(hipMemsetAsync(d->ray_slots, 0, RAY_SLOTS * RAY_SLOT_STRIDE * 8, d->stream));
    PT_SYNC_ALL(d);
    return true;
}

bool read_accum(Device* d, float* rgba, size_t n_floats, std::string& err) {
    if (n_floats != (size_t)d->n_pix * 4) {
        err = "read_accum: n_floats must be 4 * width * height";
        return false;
    }
    DeviceGuard guard(d->id);
    PT_SYNC_ALL(d);
    PT_HIP(hipMemcpy(rgba, d->accum, n_floats * sizeof(float), hipMemcpyDeviceToHost));
    return true;
}

bool read_scene_blob(Device* d, void* out, size_t cap, size_t& need, SceneLayout& layout, std::string& err) {
    if (!d->have_scene) {
        err = "scene_blob: no scene";
        return false;
    }
    layout = d->layout;
    need = d->layout.total_bytes;
    if (!out || cap < need) return true;  // the caller asked for the size
    DeviceGuard guard(d->id);
    PT_SYNC_ALL(d);
    PT_HIP(hipMemcpy(out, d->scene, need, hipMemcpyDeviceToHost));
    return true;
}

bool write_accum(Device* d, const float* rgba, size_t n_floats, std::string& err) {
    if (n_floats != (size_t)d->n_pix * 4) {
        err = "write_accum: n_floats must be 4 * width * height";
        return false;
    }
    DeviceGuard guard(d->id);
    PT_SYNC_ALL(d);
    PT_HIP(hipMemcpy(d->accum, rgba, n_floats * sizeof(float), hipMemcpyHostToDevice));
    return true;
}

template <int SCENE, bool COUNT, int BLK>
static bool launch_render(Device* d, const RenderParams& P, uint32_t lds_bytes, bool lockstep, std::string& err) {
    auto kern = lockstep ? pt_render_kernel<SCENE, COUNT, BLK> : pt_render_sched_kernel<SCENE, COUNT, BLK>;
    const bool fast = !lockstep && !COUNT && P.samples == 1 && P.compact;
    if constexpr (!COUNT) {
        if (fast) kern = pt_render_sched_kernel<SCENE, false, BLK, false, true>;
    }
    if constexpr ((SCENE == SCENE_LDS || SCENE == SCENE_GLOBAL || SCENE == SCENE_GLOBAL_ORDERED) && BLK == 1024) {  // exchange variant
        if (P.pool_off && !lockstep) kern = pt_render_sched_kernel<SCENE, COUNT, BLK, true>;
        if constexpr (!COUNT) {
            if (P.pool_off && fast) kern = pt_render_sched_kernel<SCENE, false, BLK, true, true>;
        }
    }
    if (lds_bytes > 48u * 1024u)
        PT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds_bytes));
    int per_cu = 0;
    PT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, BLK, lds_bytes));
    if (per_cu < 1) per_cu = 1;
    uint32_t grid = (uint32_t)d->prop.multiProcessorCount * (uint32_t)per_cu;
    // never launch more waves than there are 64-path units
    uint32_t max_blocks = (P.total_units + (BLK / 64) - 1u) / (BLK / 64);
    grid = std::max(1u, std::min(grid, max_blocks));
    RenderParams Q = P;
    const float share = d->sw.chunk_share;  // chunks are at most 1/share of an even split of what is left
    Q.inv_share = 1.0f / (share * (float)grid * (float)(BLK / 64));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLK), lds_bytes, d->stream, Q);
    PT_HIP(hipGetLastError());
    return true;
}

template <int BLK>
static bool launch_render_blk(Device* d, const RenderParams& P, uint32_t lds_bytes, int scene_mode, bool count,
                              bool lockstep, std::string& err) {
    if (scene_mode == SCENE_LDS_HALF)
        return count ? launch_render<SCENE_LDS_HALF, true, BLK>(d, P, lds_bytes, lockstep, err)
                     : launch_render<SCENE_LDS_HALF, false, BLK>(d, P, lds_bytes, lockstep, err);
    if (scene_mode == SCENE_LDS)
        return count ? launch_render<SCENE_LDS, true, BLK>(d, P, lds_bytes, lockstep, err)
                     : launch_render<SCENE_LDS, false, BLK>(d, P, lds_bytes, lockstep, err);
    if (scene_mode == SCENE_REFERENCE)
        return count ? launch_render<SCENE_REFERENCE, true, BLK>(d, P, lds_bytes, lockstep, err)
                     : launch_render<SCENE_REFERENCE, false, BLK>(d, P, lds_bytes, lockstep, err);
    if constexpr (BLK != 256) {  // the 256-lane instances (MI355PT_BLOCK=256 only) are not built for it
        if (scene_mode == SCENE_GLOBAL_ORDERED)
            return count ? launch_render<SCENE_GLOBAL_ORDERED, true, BLK>(d, P, lds_bytes, lockstep, err)
                         : launch_render<SCENE_GLOBAL_ORDERED, false, BLK>(d, P, lds_bytes, lockstep, err);
    }
    return count ? launch_render<SCENE_GLOBAL, true, BLK>(d, P, lds_bytes, lockstep, err)
                 : launch_render<SCENE_GLOBAL, false, BLK>(d, P, lds_bytes, lockstep, err);
}

static hipEvent_t take_event(Device* d) {
    if (!d->event_pool.empty()) {
        hipEvent_t e = d->event_pool.back();
        d->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

bool render(Device* d, const Camera64& cam, const RenderConfig& cfg, const uint32_t* seeds, uint32_t n_frames,
            RenderStats& stats, std::string& err) {
    return enqueue(d, cam, cfg, seeds, n_frames, stats, err) && finish(d, stats, err);
}

// Adds the buffered results that no path kernel's prologue has taken (the last launch of a queue, or a launch
// followed by one that has no prologue) with a kernel of their own.
static bool launch_pending_resolve(Device* d, std::string& err) {
    Device::PendingResolve& p = d->pending_resolve;
    if (!p.active) return true;
    const uint32_t tiles_x = (d->width + 7) / 8;
    hipLaunchKernelGGL(pt_accumulate_kernel, dim3((p.active_tiles * 64u + 255u) / 256u), dim3(256), 0, d->stream, d->accum, p.results,
                       Camera64{}, d->width, d->height, tiles_x, p.tiles, p.active_tiles, p.frames, p.perm, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr, d->ray_slots, (float*)nullptr, 1u, d->counters + 10, d->counters + 9, 0u, 0.0f, 0.0f, 0u);
    PT_HIP(hipGetLastError());
    p.active = false;
    return true;
}

bool finish(Device* d, RenderStats& stats, std::string& err) {
    if (!d->in_flight) return true;
    DeviceGuard guard(d->id);
    // what still waits for a prologue is added now, before the wait below (timed like every accumulation; a failure
    // is reported after the clean-up)
    bool flushed = true;
    if (d->pending_resolve.active) {
        hipEvent_t b = take_event(d), e = take_event(d);
        const bool timed = b && e && hipEventRecord(b, d->stream) == hipSuccess;
        flushed = launch_pending_resolve(d, err);
        if (timed && flushed && hipEventRecord(e, d->stream) == hipSuccess) {
            d->timed.push_back(Device::Timed{b, e, true});
        } else {
            if (b) d->event_pool.push_back(b);
            if (e) d->event_pool.push_back(e);
        }
    }
    // whatever happens below, the queue is empty afterwards: a failed call must not poison the next one
    struct Reset {
        Device* d;
        ~Reset() {
            for (auto& t : d->timed) d->event_pool.push_back(t.begin), d->event_pool.push_back(t.end);
            d->timed.clear();
            d->seed_staging.clear();
            d->seeds_head = 0;
            d->in_flight = false;
            d->pending.clear();
        }
    } reset{d};
    PT_SYNC_ALL(d);
    if (!flushed) return false;
    for (auto& t : d->timed) {
        float ms = 0.0f;
        PT_HIP(hipEventElapsedTime(&ms, t.begin, t.end));
        (t.resolve ? stats.resolve_ms : stats.render_ms) += ms;
    }
    unsigned long long host_counters[28];
    PT_HIP(hipMemcpy(host_counters, d->counters, sizeof host_counters, hipMemcpyDeviceToHost));
    for (int k = 0; k < 12; k++) stats.phase[k] = host_counters[16 + k];
    stats.nodes = host_counters[0];
    stats.tests = host_counters[1];
    stats.accepted = host_counters[2];
    for (int k = 0; k < 5; k++) stats.bounces[k] = host_counters[3 + k];
    for (int k = 0; k < 4; k++) stats.exchange[k] = host_counters[11 + k];
    {
        unsigned long long slots[RAY_SLOTS * RAY_SLOT_STRIDE];
        PT_HIP(hipMemcpy(slots, d->ray_slots, sizeof slots, hipMemcpyDeviceToHost));
        stats.rays = 0;
        for (uint32_t k = 0; k < RAY_SLOTS; k++) stats.rays += slots[(size_t)k * RAY_SLOT_STRIDE];
    }
    stats.list_tiles = stats.list_candidates = 0;
    if (d->lists_in_use) stats.list_tiles = d->list_counts[0], stats.list_candidates = d->list_counts[1];
    stats.sky_tiles = d->sky_tiles_last;
    stats.short_div = d->short_div;
    // The accumulation kernel counts the launches it has added (counters[9]); it skips a launch whose path kernel
    // gave up and every later one that was already queued, so the accumulator holds exactly the first `added` of
    // the queued launches, and only those are credited.
    const size_t added = std::min<size_t>(d->pending.size(), (size_t)(host_counters[9] - d->added_seen));
    d->added_seen = host_counters[9];
    stats.dropped_frames = 0;
    for (size_t k = 0; k < d->pending.size(); k++) {
        if (k < added) stats.paths += d->pending[k].paths, stats.launches++;
        else stats.dropped_frames += d->pending[k].frames;
    }
    if (host_counters[10]) {
        // Reported once.
        PT_HIP(hipMemset(d->counters + 10, 0, 8));
        err = "render: kernel watchdog expired (" + std::to_string(host_counters[10]) +
              " waves gave up); the frames of that launch and of the launches queued behind it were not added";
        return false;
    }
    return true;
}

// `fixed` == nullptr: the context's own buffers, grown to what the call needs.  Otherwise the call works in
// the given area, allocates nothing (no device memory, no events: its launches are not timed) and fails
// with a message when the area is too small.
static bool enqueue_in(Device* d, const Workspace* fixed, const Camera64& cam, const RenderConfig& cfg,
                       const uint32_t* seeds, uint32_t n_frames, RenderStats& stats, std::string& err) {
    if (!d->have_scene) {
        err = "render: no scene";
        return false;
    }
    if (!d->accum) {
        err = "render: no accumulator";
        return false;
    }
    if (cfg.samples < 1) {
        err = "render: settings.samples must be >= 1";
        return false;
    }
    if (n_frames == 0) return true;
    if (cfg.depth < 0) {
        // trace_path's loop `while i <= depth` never runs (compute.wgsl:289): every path returns
        // colour * light = 0, no ray is traced and no random number beyond the jitter is drawn, so
        // the frames add exactly zero to every texel.
        stats.paths += (uint64_t)d->n_pix * (uint64_t)cfg.samples * n_frames;
        return true;
    }
    DeviceGuard guard(d->id);
    const Switches& sw = d->sw;

    const uint32_t tiles_x = (d->width + 7) / 8, tiles_y = (d->height + 7) / 8;
    const uint32_t tiles = tiles_x * tiles_y;
    // Frames per launch.  The persistent kernel ends with a tail in which most waves have run out of
    // work while a few finish their longest paths (about 1 ms at 1080p, whatever the launch size), so
    // launches should be long: one launch covers as many frames as the frame buffer may hold --
    // FRAME_BUF_DEFAULT_BYTES by default, at most a quarter of the free device memory, MI355PT_FRAME_BUF_GB
    // overrides -- and 32-bit path ids allow.  Measured on MI355X, scene1 1080p, 256 frames: 4 launches
    // 32.4 ms, 2 launches 30.3 ms, 1 launch 29.0 ms.
    uint32_t batch = cfg.max_batch ? cfg.max_batch : sw.max_batch;
    if (fixed) {
        // What the area holds if no spill space is needed; the launch is shaped (workgroup size, queue chunks,
        // exchange variant) for this length, and clamped further below once the spill area is known.
        batch = (uint32_t)std::max<size_t>(1, std::min<size_t>(fixed->frame_bytes / ((size_t)tiles * 64u * 12u), 4096));
    } else if (batch == 0) {
        size_t budget = FRAME_BUF_DEFAULT_BYTES;
        if (sw.frame_buf_gb) {
            budget = (size_t)sw.frame_buf_gb << 30;
        } else {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const size_t held = d->frame_cap * sizeof(f4);  // ours already
                budget = std::min(budget, (free_b + held) / 8);  // two halves: together at most a quarter
            }
        }
        size_t by_mem = budget / ((size_t)tiles * 64u * sizeof(f4));
        batch = (uint32_t)std::max<size_t>(1, std::min<size_t>(by_mem, 4096));
    }
    batch = std::min<uint32_t>(batch, std::max<uint32_t>(1u, (0xffffffffu / 64u) / tiles));
    batch = std::min(batch, n_frames);
    // equal launches rather than full ones and a short rest (256 frames at 84 per launch: 4 x 64, not 3 x 84 + 4)
    auto balanced = [](uint32_t n, uint32_t per_launch) {
        const uint32_t launches = (n + per_launch - 1u) / per_launch;
        return (n + launches - 1u) / launches;
    };
    if (batch) batch = balanced(n_frames, batch);
    size_t need = (size_t)batch * tiles * 64u;  // whole tiles: [frame][tile rank][64 slots]
    if (!fixed && 2u * need > d->frame_cap) {
        if (!finish(d, stats, err)) return false;  // nothing may still read or write the old buffer
        (void)hipFree(d->frame_buf);
        d->frame_buf = nullptr;
        d->frame_cap = 0;
        PT_HIP(hipMalloc(&d->frame_buf, 2u * need * sizeof(f4)));
        d->frame_cap = 2u * need;
        d->frame_half = 0;
    }
    if (!fixed && n_frames > d->seeds_cap) {
        if (!finish(d, stats, err)) return false;
        (void)hipFree(d->seeds);
        d->seeds = nullptr;
        d->seeds_cap = 0;
        const size_t cap = std::max<size_t>(4096, 2 * (size_t)n_frames);
        PT_HIP(hipMalloc(&d->seeds, cap * 4));
        d->seeds_cap = cap;
        d->seeds_head = 0;
    }
    if (!fixed && d->seeds_head + n_frames > d->seeds_cap && !finish(d, stats, err)) return false;  // ring is full: drain

    RenderParams P;
    memset(&P, 0, sizeof P);
    P.scene = d->scene;
    P.L = d->layout;
    P.cam = cam;
    P.width = d->width, P.height = d->height, P.tiles_x = tiles_x, P.tiles_per_frame = tiles, P.n_pix = d->n_pix;
    P.samples = cfg.samples, P.depth = cfg.depth;
    P.pixel_hash = d->pixel_hash;
    P.work_counter = fixed ? fixed->work_counter : d->work_counter;
    P.counters = d->counters;
    P.glass_always_draws = cfg.glass_always_draws ? 1 : 0;
    P.rcp_w = d->rcp_w, P.rcp_h = d->rcp_h, P.short_div = d->short_div ? 1u : 0u;
    const bool lockstep = sw.lockstep;
    P.watchdog_ticks = sw.watchdog_ms * 100000ull;  // 100 MHz
    P.inv_tiles_x = 1.0f / (float)tiles_x;
    for (int k = 0; k < 6; k++) P.thr[k] = sw.thr[k];
    // Units a wave takes from the queue at once (consecutive units are consecutive frames of one tile: the same
    // pixels, so the same part of a large tree -- 1 M spheres at 3840x2160, 32 frames: 26.8 ms at 4 units,
    // 24.6 at 8, 24.0 at 16 and 32; 10 M spheres 30.9 / 27.3 / 27.1 at 4 / 16 / 32; 8-frame launches are best at 16).
    P.chunk_max = sw.chunk ? sw.chunk : ((d->scene_in_lds || std::min(n_frames, batch) >= 16u) ? 32u : 16u);
    // Short launches (fewer than about 56 64-path units per resident wave: below 15 frames at 1920x1080) are a
    // few per cent faster with the smaller workgroups; longer ones with 1024 lanes at 8 waves per SIMD (measured
    // at 1920x1080: 8 frames 1.15 ms with 512 lanes against 1.23 with 1024; 16 frames 1.75 / 1.74; 24 frames
    // 2.39 / 2.26; 32 frames 3.03 / 2.78; 256 frames 7 % apart).
    const bool long_launch = (uint64_t)std::min(n_frames, batch) * tiles >= 56ull * 32ull * (uint64_t)d->prop.multiProcessorCount;
    const bool in_lds = d->scene_in_lds;
    int scene_mode = d->layout.reference_walk ? SCENE_REFERENCE
                     : in_lds ? (d->layout.node_half ? SCENE_LDS_HALF : SCENE_LDS) : SCENE_GLOBAL;
    const size_t lds_cu = 160u * 1024u;
    uint32_t block, lds_bytes;
    size_t spill_in_use = 0;  // entries of the spill area this call's kernels address
    if (in_lds) {
        P.scene_lds_bytes = lds_scene_bytes(d->layout);
        // Workgroup size: the scene copy in LDS is shared by the workgroup, the traversal stacks are per
        // lane.  Larger workgroups amortise the copy (more waves fit next to it); deep trees make the
        // stacks dominate and favour smaller ones.  Pick what keeps more waves resident per CU.
        // Registers: the kernels need 70 vector registers (7 waves per SIMD: 28 per CU, 24 in whole 512-lane
        // workgroups); the 1024-lane instance of the scheduled kernel is compiled for 64 (8 per SIMD, two
        // workgroups per CU, one spilled register) -- more waves raise the instruction-issue ceiling the
        // kernel runs at: scene1, 256 frames, 20.65 ms with 3 x 512 lanes per CU, 19.20 ms with 2 x 1024.
        auto waves_per_cu = [&](uint32_t blk) {
            size_t per_block = (size_t)P.scene_lds_bytes + stack_bytes(d->layout, blk);
            size_t blocks = per_block ? lds_cu / per_block : 8;
            const size_t cap = blk == 1024 ? 32 : (blk == 512 ? 24 : 28);
            return std::min<size_t>(blocks * (blk / 64), cap);
        };
        block = waves_per_cu(512) > waves_per_cu(256) ? 512u : 256u;
        if (!lockstep && long_launch && waves_per_cu(1024) > waves_per_cu(block)) block = 1024u;
        if (sw.block == 256 || sw.block == 512 || sw.block == 1024) block = sw.block;
        lds_bytes = P.scene_lds_bytes + stack_bytes(d->layout, block);
        // Exchange variant (paths waiting for a scatter step move between the waves of a workgroup through two
        // pools in LDS): one sample per path, float nodes, 1024 lanes, and the pools must not cost a workgroup per CU.
        // Scenes of a handful of spheres gain nothing from it (scene2 / scene3: 1 % slower), scene1's 176 nodes 2.4 %.
        if (sw.exchange && !lockstep && block == 1024u && cfg.samples == 1 && !d->layout.node_half && d->layout.n_nodes >= 32u &&
            lds_cu / (lds_bytes + POOL_BYTES) >= lds_cu / lds_bytes) {
            P.pool_off = (lds_bytes + 15u) & ~15u;
            lds_bytes = P.pool_off + POOL_BYTES;
            for (int k = 0; k < 6; k++) P.thr[k] = sw.xthr[k];
            P.thr[PH_OPAQUE] = std::min(P.thr[PH_OPAQUE], POOL_MIN_OPAQUE);  // a batch out of a pool runs at once
            P.thr[PH_GLASS] = std::min(P.thr[PH_GLASS], POOL_MIN_GLASS);
        }
    } else {
        // Scene in global memory.  LDS holds, per workgroup, the first stack entries of every lane and
        // as much of the top of the tree as still lets 24 waves stay resident on a CU.  Measured on
        // MI355X with 1 M spheres at 3840x2160: 6 entries + 40 KB of nodes in 512-lane workgroups
        // 18.3 ms per 16 frames; everything in global memory 21.7 ms; a 64 KB prefix at 8 waves per CU
        // 33 ms (occupancy matters more than the size of the prefix).  With today's kernel 5 entries are slightly
        // better than 6 (1 M spheres, 64 frames: 43.0 against 43.3 ms; 10 M spheres, 32 frames: 25.3 against 25.9).
        // MI355PT_STACK_LDS (entries) and MI355PT_NODE_CACHE_KB override the split (development).
        block = (lockstep || !long_launch) ? 512u : 1024u;  // the scheduled kernel's 1024-lane instance runs 8 waves per SIMD
        if (sw.block == 256 || sw.block == 512 || sw.block == 1024) block = sw.block;
        const uint32_t depth = std::max(1u, d->layout.max_depth);
        P.stack_lds_entries = (uint32_t)std::min<long>(depth, std::max<long>(1, sw.stack_lds));
        P.stack_spill_entries = std::max(1u, depth - P.stack_lds_entries);
        const uint32_t stack_lds = P.stack_lds_entries * block * 4u;
        const uint32_t blocks_per_cu = std::max(1u, (block == 1024u ? 32u : 24u) / (block / 64u));
        const long budget = (long)(lds_cu / blocks_per_cu) - (long)stack_lds - 2048;  // LDS is granted in granules
        long cache = sw.node_cache_kb >= 0 ? sw.node_cache_kb * 1024 : budget;
        cache = std::min<long>(cache, (long)lds_cu - (long)stack_lds);
        if (scene_mode == SCENE_GLOBAL && d->onodes_valid && block != 256u) scene_mode = SCENE_GLOBAL_ORDERED, P.onodes = d->onodes;
        const long node_bytes = scene_mode == SCENE_GLOBAL_ORDERED ? (long)GLOBAL_NODE_BYTES : (long)sizeof(Node64);
        cache = std::min<long>(cache, 4096l * node_bytes);  // the renumbered prefix (order_top_first)
        cache = std::min<long>(cache, (long)d->layout.n_nodes * node_bytes);
        cache = std::max<long>(cache, 0) / node_bytes * node_bytes;  // whole records
        P.scene_lds_bytes = (d->layout.off_nodes == 0) ? (uint32_t)cache : 0u;
        lds_bytes = P.scene_lds_bytes + stack_lds;
        // Exchange variant, as for scenes in LDS (1 M spheres at 3840x2160, 32 frames: 28.5 -> 27.0 ms; 100 k spheres
        // 24.9 -> 23.6); the pools come out of the node prefix.
        if (sw.exchange_global && sw.exchange && !lockstep && block == 1024u && cfg.samples == 1 &&
            (scene_mode == SCENE_GLOBAL || scene_mode == SCENE_GLOBAL_ORDERED)) {
            if (P.scene_lds_bytes >= POOL_BYTES + 2u * (uint32_t)node_bytes)
                P.scene_lds_bytes -= (POOL_BYTES + (uint32_t)node_bytes - 1u) / (uint32_t)node_bytes * (uint32_t)node_bytes;
            lds_bytes = P.scene_lds_bytes + stack_lds;
            P.pool_off = (lds_bytes + 15u) & ~15u;
            lds_bytes = P.pool_off + POOL_BYTES;
            for (int k = 0; k < 6; k++) P.thr[k] = sw.xthr[k];
            P.thr[PH_OPAQUE] = std::min(P.thr[PH_OPAQUE], POOL_MIN_OPAQUE);
            P.thr[PH_GLASS] = std::min(P.thr[PH_GLASS], POOL_MIN_GLASS);
        }
        const size_t need_spill = (size_t)d->prop.multiProcessorCount * (2048u / block) * P.stack_spill_entries * block;
        if (fixed) {
            if (need_spill > fixed->spill_entries) {
                err = "render_static: the traversal stacks of this scene need " + std::to_string(need_spill * 4) +
                      " bytes of spill space, the static workspace has " + std::to_string(fixed->spill_entries * 4);
                return false;
            }
            P.stack_spill = fixed->stack_spill;
            spill_in_use = need_spill;
        } else if (need_spill > d->spill_cap) {
            if (!finish(d, stats, err)) return false;
            (void)hipFree(d->stack_spill);
            d->stack_spill = nullptr;
            d->spill_cap = 0;
            PT_HIP(hipMalloc(&d->stack_spill, need_spill * sizeof(int32_t)));
            d->spill_cap = need_spill;
        }
        if (!fixed) P.stack_spill = d->stack_spill;
    }
    if (lds_bytes > lds_cu) {
        err = "render: traversal stacks do not fit in LDS (tree too deep)";
        return false;
    }

    char* frame_buf = reinterpret_cast<char*>(d->frame_buf);
    if (fixed) {
        // what the area holds after the spill area: whole frames of 12-byte results (these launches never
        // measure tile costs, so no result carries a ray count)
        const size_t spill_bytes = (spill_in_use * 4u + 255u) & ~(size_t)255u;
        const size_t per_frame = (size_t)tiles * 64u * 12u;
        const size_t room = fixed->frame_bytes > spill_bytes ? fixed->frame_bytes - spill_bytes : 0;
        if (room < per_frame) {
            err = "render_static: one " + std::to_string(d->width) + "x" + std::to_string(d->height) + " frame needs " +
                  std::to_string(per_frame) + " bytes of results, the static workspace has " + std::to_string(room);
            return false;
        }
        frame_buf = static_cast<char*>(fixed->frame_buf) + spill_bytes;
        batch = (uint32_t)std::min<size_t>({(size_t)batch, room / per_frame, (size_t)n_frames});
        batch = balanced(n_frames, batch);
    }

    // From here on work is queued: whatever fails, finish() has to run before the buffers are touched again.
    d->in_flight = true;
    // Candidate lists of the camera rays: rebuilt on the stream, ahead of the launches that read them, when the
    // camera, the scene or the image size changed (no allocation: the buffer came with the accumulator).  The
    // reference-walk mode tries exactly the spheres the reference's walk tries, so it keeps the walk.
    d->lists_in_use = false;
    if (sw.tile_lists && !lockstep && !d->layout.reference_walk && d->tile_lists) {
        d->lists_in_use = true;
        if (!d->lists_valid || memcmp(&cam, &d->lists_cam, sizeof cam) != 0) {
            PT_HIP(hipMemsetAsync(d->list_count, 0, 16, d->stream));
            hipLaunchKernelGGL(pt_tile_lists_kernel, dim3((tiles + 63u) / 64u), dim3(64), 0, d->stream, d->scene, d->layout, cam,
                               d->width, d->height, tiles_x, tiles, sw.tile_list_max, d->tile_lists, d->list_count);
            PT_HIP(hipGetLastError());
            // the counts decide how later launches are split, so they are read back now: once per camera
            PT_HIP(hipStreamSynchronize(d->stream));
            PT_HIP(hipMemcpy(d->list_counts, d->list_count, sizeof d->list_counts, hipMemcpyDeviceToHost));
            d->lists_valid = true;
            d->lists_cam = cam;
            d->perm_sky_last = false;  // an order made before these lists does not know their sky tiles
        }
        P.tile_lists = d->tile_lists;
    }
    uint32_t* dev_seeds;
    if (fixed) {
        // the caller's array is read before this call returns (render_static waits), so it needs no host copy
        dev_seeds = fixed->seeds;
        PT_HIP(hipMemcpyAsync(dev_seeds, seeds, (size_t)n_frames * 4, hipMemcpyHostToDevice, d->stream));
    } else {
        dev_seeds = d->seeds + d->seeds_head;
        d->seeds_head += n_frames;
        d->seed_staging.emplace_back(seeds, seeds + n_frames);
        PT_HIP(hipMemcpyAsync(dev_seeds, d->seed_staging.back().data(), (size_t)n_frames * 4, hipMemcpyHostToDevice,
                              d->stream));
    }

    // Rays a chunk should cost: a wave traces about 6.5 M rays/s, so 2 600 rays are 400 us per chunk and
    // 6 144 waves issue 15 M queue atomics/s.  MI355PT_CHUNK_RAYS overrides (development switch).
    const float chunk_rays = sw.chunk_rays;
    const bool order_tiles = sw.tile_order;  // MI355PT_TILE_ORDER=0 keeps row-major tile order
    bool use_perm = order_tiles && d->perm_valid && memcmp(&cam, &d->perm_cam, sizeof cam) == 0 &&
                    cfg.depth == d->perm_depth && cfg.samples == d->perm_samples;
    // one launch (path kernel or accumulation kernel) between two events; nothing is left half-recorded
    auto timed_launch = [&](bool resolve, auto&& launch) -> bool {
        if (fixed) return launch();
        hipEvent_t b = take_event(d), e = take_event(d);
        bool ok = b && e && hipEventRecord(b, d->stream) == hipSuccess;
        if (!ok) err = "render: could not record a timing event";
        if (ok) ok = launch();
        if (ok && hipEventRecord(e, d->stream) != hipSuccess) {
            ok = false;
            err = "render: could not record a timing event";
        }
        if (!ok) {
            if (b) d->event_pool.push_back(b);
            if (e) d->event_pool.push_back(e);
            return false;
        }
        d->timed.push_back(Device::Timed{b, e, resolve});
        return true;
    };
    // A launch that measures the tile costs runs the kernel's general instance with 16-byte results, about twice the
    // time per frame, and the costs are a property of scene and camera, not of the frames: when one is due it gets
    // MEASURE_FRAMES frames of its own and the rest of the call is spread over launches of the plain kind (a
    // 1000-frame call at 1280x720 used to measure on its first 500 frames).
    constexpr uint32_t MEASURE_FRAMES = 16;
    for (uint32_t f0 = 0, nb = 0; f0 < n_frames; f0 += nb) {
        nb = std::min(batch, n_frames - f0);
        const bool measure_due = !fixed && order_tiles && nb >= 8u && (!use_perm || d->since_measure + 1u >= sw.measure_every);
        if (measure_due && nb > MEASURE_FRAMES) {
            nb = MEASURE_FRAMES;
            const uint32_t rest = n_frames - f0 - nb;
            if (rest) batch = balanced(rest, batch);  // the rest in equal launches, none longer than the buffers were sized for
        }
        P.frame_out = reinterpret_cast<f4*>(frame_buf);
        P.total_units = nb * tiles;
        P.n_frames = nb;
        P.inv_n_frames = 1.0f / (float)nb;
        P.tile_perm = use_perm ? d->tile_perm : nullptr;
        P.rank_chunk = use_perm ? d->rank_chunk : nullptr;
        P.small_ids = (P.total_units < (1u << 24) && tiles < (1u << 24)) ? 1u : 0u;
        P.seeds = dev_seeds + f0;
        // The tile costs are measured on the first launch that is long enough to be representative after the
        // camera or the settings changed, and again every 64th launch (they are a property of scene and camera;
        // the refresh only guards against a first measurement on an unusually short launch); measuring needs
        // the ray count per path (16-byte results, the kernel's general instance: 10-20 % slower), all other
        // launches write and re-read 12 bytes per path.
        const bool measure = measure_due;
        P.compact = measure ? 0u : 1u;
        // Tiles that show nothing but sky (empty candidate lists; last in an order made with the lists) are left to
        // the sky blocks of pt_accumulate_kernel on launches of the plain kind: one sample per path, no counting, no cost measurement.
        uint32_t sky_tiles = 0;
        if (sw.sky_tiles && use_perm && d->perm_sky_last && d->lists_in_use && !measure && !cfg.count && cfg.samples == 1 &&
            d->list_counts[2] < tiles)
            sky_tiles = d->list_counts[2];
        const uint32_t active_tiles = tiles - sky_tiles;
        P.total_units = nb * active_tiles;
        P.small_ids = (P.total_units < (1u << 24) && tiles < (1u << 24)) ? 1u : 0u;
        d->sky_tiles_last = sky_tiles;
        P.ray_slots = d->ray_slots;
        P.accum = d->accum;
        // The buffered results of the previous launch: added by this launch's path kernel in its prologue when the
        // instance that runs has one (exchange variant, one sample per path, 12-byte results), else by a kernel of
        // their own now.  A measuring launch rewrites the tile order those results are laid out by: flush first.
        const bool has_prologue = !lockstep && !cfg.count && cfg.samples == 1 && P.compact && P.pool_off != 0u && block == 1024u &&
                                  (scene_mode == SCENE_LDS || scene_mode == SCENE_GLOBAL || scene_mode == SCENE_GLOBAL_ORDERED);
        P.prev_results = nullptr;
        if (d->pending_resolve.active) {
            if (has_prologue && !measure) {
                const Device::PendingResolve& pr = d->pending_resolve;
                P.prev_results = pr.results, P.prev_perm = pr.perm, P.prev_frames = pr.frames, P.prev_active_tiles = pr.active_tiles;
                P.resolve_wgs = sw.resolve_wgs;
            } else if (!timed_launch(true, [&]() { return launch_pending_resolve(d, err); })) {
                return false;
            }
        }
        if (!fixed) P.frame_out = d->frame_buf + (size_t)d->frame_half * (d->frame_cap / 2u);
        PT_HIP(hipMemsetAsync(P.work_counter, 0, 4, d->stream));
        if (!timed_launch(false, [&]() {
                if (block == 512) return launch_render_blk<512>(d, P, lds_bytes, scene_mode, cfg.count, lockstep, err);
                if (block == 1024) return launch_render_blk<1024>(d, P, lds_bytes, scene_mode, cfg.count, lockstep, err);
                return launch_render_blk<256>(d, P, lds_bytes, scene_mode, cfg.count, lockstep, err);
            }))
            return false;
        d->pending_resolve.active = false;  // taken by the prologue of the kernel just queued (or flushed above)
        // Accumulate in frame order (same stream: launches of one context never overlap).  The buffered results wait
        // for the next launch's prologue (or finish()) unless this launch measures tile costs; the sky tiles, which
        // need no buffer, are added now.
        // (only behind a kernel that has the prologue itself: the next launch then has it too; without one the
        // buffered results are better added beside the sky tiles, in the one grid below -- with a prologue in every
        // one-sample instance scene3 gained 1.3 % per queued 256-frame step and scene2 nothing at 16 to 512 workgroups)
        // (nor behind the last launch of a call whose caller waits at once: no launch follows that could take them)
        const bool defer = sw.defer_resolve && !fixed && !measure && has_prologue && !(cfg.wait_follows && f0 + nb >= n_frames);
        const uint32_t resolve_blocks = defer ? 0u : (active_tiles * 64u + 255u) / 256u;
        if (sky_tiles + resolve_blocks > 0u &&
            !timed_launch(true, [&]() {
                static_assert(64 * SKY_FRAMES == 256, "sky tiles and buffered results share one workgroup size");
                hipLaunchKernelGGL(pt_accumulate_kernel, dim3(sky_tiles + resolve_blocks), dim3(256), 0, d->stream,
                                   d->accum, P.frame_out, cam, d->width, d->height, tiles_x, tiles, active_tiles, nb, P.tile_perm, P.seeds,
                                   d->pixel_hash, d->ray_slots, measure ? d->tile_cost : nullptr, P.compact, d->counters + 10,
                                   d->counters + 9, sky_tiles, P.rcp_w, P.rcp_h, P.short_div);
                PT_HIP(hipGetLastError());
                return true;
            }))
            return false;
        if (defer) {
            d->pending_resolve.active = true;
            d->pending_resolve.results = P.frame_out, d->pending_resolve.perm = P.tile_perm;
            d->pending_resolve.frames = nb, d->pending_resolve.tiles = tiles, d->pending_resolve.active_tiles = active_tiles;
            d->frame_half ^= 1u;
        }
        if (measure) {
            hipLaunchKernelGGL(pt_tile_order_kernel, dim3(1), dim3(ORDER_BLK), 0, d->stream, d->tile_cost, tiles,
                               d->tile_perm, d->rank_chunk, (float)nb, chunk_rays, P.chunk_max,
                               d->lists_in_use ? d->tile_lists : nullptr);
            PT_HIP(hipGetLastError());
            d->perm_sky_last = d->lists_in_use;
            d->perm_valid = true;
            d->perm_cam = cam;
            d->perm_depth = cfg.depth, d->perm_samples = cfg.samples;
            d->since_measure = 0;
            use_perm = true;   // later launches of this call are ordered after the kernel above
        } else {
            d->since_measure++;
        }
        stats.exchange_on = P.pool_off != 0u;
        d->pending.push_back(Device::Pending{nb, (uint64_t)d->n_pix * (uint64_t)cfg.samples * nb});
    }
    return true;
}

bool enqueue(Device* d, const Camera64& cam, const RenderConfig& cfg, const uint32_t* seeds, uint32_t n_frames,
             RenderStats& stats, std::string& err) {
    return enqueue_in(d, nullptr, cam, cfg, seeds, n_frames, stats, err);
}

bool render_static(Device* d, const Camera64& cam, const RenderConfig& cfg, const uint32_t* seeds, uint32_t n_frames,
                   RenderStats& stats, std::string& err) {
    if (!finish(d, stats, err)) return false;  // queued work first: it may still use the context's buffers
    if (!d->static_ws) {
        err = "render_static: no static workspace on this device (device ordinal above 63)";
        return false;
    }
    if (n_frames > STATIC_MAX_FRAMES) {
        err = "render_static: at most " + std::to_string(STATIC_MAX_FRAMES) + " frames per call";
        return false;
    }
    Workspace ws;
    ws.fixed = true;
    ws.work_counter = reinterpret_cast<uint32_t*>(d->static_ws + STATIC_WS_COUNTER);
    ws.seeds = reinterpret_cast<uint32_t*>(d->static_ws + STATIC_WS_SEEDS);
    ws.stack_spill = reinterpret_cast<int32_t*>(d->static_ws + STATIC_WS_DATA);
    ws.spill_entries = (STATIC_WS_BYTES - STATIC_WS_DATA) / 4u;
    ws.frame_buf = d->static_ws + STATIC_WS_DATA;  // enqueue_in places the results behind the spill area in use
    ws.frame_bytes = STATIC_WS_BYTES - STATIC_WS_DATA;
    const bool queued = enqueue_in(d, &ws, cam, cfg, seeds, n_frames, stats, err);
    std::string err2;
    const bool done = finish(d, stats, queued ? err : err2);  // waits: the area and `seeds` are free again on return
    return queued && done;
}

// ------------------------------------------------------------------------------ accumulators of several devices
__global__ void pt_add_rgb_kernel(f4* dst, const f4* src, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        f4 a = dst[i];
        const f4 b = src[i];
        a.x += b.x, a.y += b.y, a.z += b.z;
        dst[i] = a;
    }
}
__global__ void pt_set_alpha_kernel(f4* p, uint32_t n, float alpha) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i].w = alpha;
}

// RCCL, opened when first needed: the library works without it on one device.
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::map<std::vector<int>, std::vector<ncclComm_t>> comms;  // by device list; kept until the process ends
};
static std::mutex g_rccl_mutex;
static Rccl g_rccl;
static bool rccl_open(std::string& err) {
    if (g_rccl.handle) return true;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        err = std::string("reduce: cannot open librccl.so: ") + dlerror();
        return false;
    }
    Rccl& r = g_rccl;
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    r.Reduce = reinterpret_cast<decltype(r.Reduce)>(dlsym(h, "ncclReduce"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!r.CommInitAll || !r.Reduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
        err = "reduce: librccl.so lacks an expected entry point";
        dlclose(h);
        return false;
    }
    r.handle = h;
    return true;
}
#define PT_NCCL(call)                                                                   \
    do {                                                                                \
        ncclResult_t r_ = (call);                                                       \
        if (r_ != ncclSuccess) {                                                        \
            err = std::string(#call) + ": " + g_rccl.GetErrorString(r_);                \
            return false;                                                               \
        }                                                                               \
    } while (0)

static bool same_image(Device* const* devs, size_t n, std::string& err) {
    for (size_t i = 0; i < n; i++) {
        if (!devs[i] || !devs[i]->accum) {
            err = "reduce: a context has no accumulator";
            return false;
        }
        if (devs[i]->width != devs[0]->width || devs[i]->height != devs[0]->height) {
            err = "reduce: the contexts differ in image size";
            return false;
        }
    }
    return true;
}

bool reduce_accum(Device* const* devs, size_t n, size_t root, std::string& err) {
    if (n == 0 || root >= n) {
        err = "reduce: root out of range";
        return false;
    }
    if (!same_image(devs, n, err)) return false;
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < i; j++)
            if (devs[i] == devs[j]) {
                err = "reduce: a context is listed twice";
                return false;
            }
    const uint32_t n_pix = devs[0]->n_pix;
    const dim3 grid((n_pix + 255u) / 256u), block(256);
    // Nothing is in flight on any of the contexts (the caller has joined them).  One context per device
    // represents it -- the root on its device, the first listed elsewhere -- and first takes in the others that
    // share its device with a plain kernel; RCCL then sums one accumulator per device into the root's.
    std::map<int, Device*> rep;
    rep[devs[root]->id] = devs[root];
    for (size_t i = 0; i < n; i++) rep.emplace(devs[i]->id, devs[i]);
    for (size_t i = 0; i < n; i++) {
        Device* r = rep[devs[i]->id];
        if (r == devs[i]) continue;
        DeviceGuard guard(r->id);
        hipLaunchKernelGGL(pt_add_rgb_kernel, grid, block, 0, r->stream, r->accum, devs[i]->accum, n_pix);
        PT_HIP(hipGetLastError());
    }
    if (rep.size() > 1) {
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        if (!rccl_open(err)) return false;
        std::vector<int> ids;
        for (auto& kv : rep) ids.push_back(kv.first);  // ascending: the rank of a device is its place in this list
        auto it = g_rccl.comms.find(ids);
        if (it == g_rccl.comms.end()) {
            std::vector<ncclComm_t> comms(ids.size());
            PT_NCCL(g_rccl.CommInitAll(comms.data(), (int)ids.size(), ids.data()));
            it = g_rccl.comms.emplace(ids, std::move(comms)).first;
        }
        int root_rank = 0;
        for (size_t k = 0; k < ids.size(); k++)
            if (ids[k] == devs[root]->id) root_rank = (int)k;
        // one reduce of W*H*4 floats, every device's part queued on that device's render stream
        PT_NCCL(g_rccl.GroupStart());
        bool ok = true;
        ncclResult_t bad = ncclSuccess;
        for (size_t k = 0; k < ids.size() && ok; k++) {
            Device* r = rep[ids[k]];
            DeviceGuard guard(r->id);
            bad = g_rccl.Reduce(r->accum, r->accum, (size_t)n_pix * 4u, ncclFloat, ncclSum, root_rank, it->second[k], r->stream);
            ok = bad == ncclSuccess;
        }
        const ncclResult_t end = g_rccl.GroupEnd();
        if (!ok || end != ncclSuccess) {
            err = std::string("ncclReduce: ") + g_rccl.GetErrorString(ok ? end : bad);
            return false;
        }
        // the alpha channel is the constant 1 of the clear colour in every accumulator, not a sum
        DeviceGuard guard(devs[root]->id);
        hipLaunchKernelGGL(pt_set_alpha_kernel, grid, block, 0, devs[root]->stream, devs[root]->accum, n_pix, 1.0f);
        PT_HIP(hipGetLastError());
    }
    for (auto& kv : rep) {
        DeviceGuard guard(kv.first);
        PT_HIP(hipStreamSynchronize(kv.second->stream));
    }
    return true;
}

bool debug_counters(Device* d, unsigned long long* out, size_t n, std::string& err) {
    DeviceGuard guard(d->id);
    PT_HIP(hipStreamSynchronize(d->stream));
    PT_HIP(hipMemcpy(out, d->counters, std::min(n * 8, COUNTER_BYTES), hipMemcpyDeviceToHost));
    return true;
}

bool set_alpha(Device* d, float alpha, std::string& err) {
    if (!d->accum) {
        err = "no accumulator";
        return false;
    }
    DeviceGuard guard(d->id);
    hipLaunchKernelGGL(pt_set_alpha_kernel, dim3((d->n_pix + 255u) / 256u), dim3(256), 0, d->stream, d->accum, d->n_pix, alpha);
    PT_HIP(hipGetLastError());
    PT_HIP(hipStreamSynchronize(d->stream));
    return true;
}

bool reduce_accum_comm(Device* d, void* comm, int root, std::string& err) {
    if (!d->accum || !comm) {
        err = "reduce: no accumulator or no communicator";
        return false;
    }
    {
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        if (!rccl_open(err)) return false;
    }
    DeviceGuard guard(d->id);
    PT_NCCL(g_rccl.Reduce(d->accum, d->accum, (size_t)d->n_pix * 4u, ncclFloat, ncclSum, root, static_cast<ncclComm_t>(comm), d->stream));
    PT_HIP(hipStreamSynchronize(d->stream));
    return true;
}

}  // namespace gpu
}  // namespace pt
