// HIP kernels for gfx950 (MI355X) and their launch code.
//
//   pt_hash_kernel          per-pixel seed hash (compute.wgsl:392-400), once per resize
//   pt_render_sched_kernel  persistent path kernel: a wave owns one 8x8-pixel tile at a time for ALL frames of
//                           the launch; every lane traces one (pixel, frame) path at a time and takes the next
//                           one of the wave's tile the moment its path ends; the wave runs one kind of step at a
//                           time (node / leaf / scatter / restart), chosen by how many lanes wait for it; finished
//                           paths park their colour in a small per-wave ring (L2-resident) and the wave adds
//                           completed frames to the accumulator in frame order (compute.wgsl:418-421), which
//                           keeps sums bit-identical to a frame-by-frame run
//   pt_tile_order_kernel    orders the 8x8 tiles by the rays they cost in the last launch
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <type_traits>
#include <vector>

#include "gpu.h"
#include "gpu_state.h"
#include "pt_device.h"

namespace pt {
namespace gpu {

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
// The same with the node section replaced by direction-ordered 80-byte records (LDS_NODE_BYTES) at
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


// Work and results.  A job is one 8x8-pixel tile for a range of job_frames consecutive frames of the launch; the
// 64 paths of one (tile, frame) form a row.  Jobs are numbered range-major -- job j = (range j / n_tiles, tile
// tile_perm[j % n_tiles]) -- and taken from one atomic counter, one at a time.  A wave hands the paths of
// consecutive rows to its lanes as they fall idle, so at any moment its lanes hold paths of a few consecutive
// frames of one tile (and of the next job while the last paths of the previous one finish).
//
// Frames must reach a texel in frame order (compute.wgsl:418-421 runs once per frame; float addition does not
// commute bit for bit), but paths finish out of order.  A finished path therefore parks its colour in the slot of
// its row in the wave's ring -- MAX_RING_SLOTS slots of 64 x 3 floats in global memory, private to the wave, taken
// lowest-free-first so that the few slots in use stay in the L2 -- and counts the row's pending paths down in
// LDS.  At each restart step the wave adds the rows that are complete, oldest first, to the tile's sums: lane p
// adds pixel p.  The sums live in the accumulator before and after a job and in a private row of the wave (one more
// ring slot) in between, so a job reads and writes each of its texels in the accumulator once.
//
// Ranges of one tile must be added in order too, and they are different jobs, usually of different waves on
// different XCDs.  The queue's order makes the wait vanish: job (range k, tile t) comes n_tiles jobs after
// (range k - 1, tile t), by which time the latter is normally long done.  The hand-off itself is explicit: the wave
// that finishes range k - 1 stores the texels with sc1 (write-through) stores, waits for them (s_waitcnt vmcnt(0))
// and then raises tile_done[t] to k with an agent-scope store; the wave that owns range k polls tile_done[t] with
// an agent-scope load before its first commit -- it keeps its finished rows in the ring until then -- and reads
// the texels with sc1 loads (every texel load in the kernel is one: no stale L1 line is ever used).  A job only
// ever waits for a job with a smaller number, so the smallest unfinished job can always finish.
constexpr uint32_t MAX_RING_SLOTS = 32;                   // rows in flight per wave (a 51-ray path spans ~25)
constexpr uint32_t ROW_PIXELS = 64;                       // 8 x 8
constexpr uint32_t RING_SLOT_FLOATS = ROW_PIXELS * 3u;    // rgb per pixel
constexpr uint32_t RING_WAVE_FLOATS = MAX_RING_SLOTS * RING_SLOT_FLOATS + ROW_PIXELS * 4u;  // + the private sums row
// per wave and row in flight: {tile x | tile y << 13 | slot << 26 | last row of its job << 31, pending paths,
// rays traced, range index of the job}
constexpr uint32_t ROWS_LDS_BYTES = MAX_RING_SLOTS * 16u;

struct RenderParams {
    const char* scene;
    SceneLayout L;
    Camera64 cam;
    uint32_t width, height, tiles_x, n_tiles;
    int32_t samples, depth;
    uint32_t n_frames;             // frames of this launch
    uint32_t job_frames;           // frames per job
    uint32_t n_ranges;             // ceil(n_frames / job_frames); jobs = n_ranges * n_tiles
    float inv_n_tiles;
    // Job j renders tile tile_perm[j % n_tiles] (or j % n_tiles without a permutation).  The permutation lists
    // tiles by decreasing measured cost, so within every range the expensive tiles start first and the queue ends
    // with the cheapest ones.
    const uint32_t* tile_perm;
    uint32_t* tile_done;           // [n_tiles] ranges of the tile added to the accumulator so far (zero at launch)
    const uint32_t* seeds;         // one per frame of this launch
    const uint32_t* pixel_hash;    // [width * height]
    f4* accum;                     // [width * height]
    float* ring;                   // [waves of the grid][RING_WAVE_FLOATS]
    uint32_t ring_slots;           // slots in use, <= MAX_RING_SLOTS
    float* tile_cost;              // rays traced per tile, added up over the jobs (may be null; zero at launch)
    unsigned long long* ray_slots;
    uint32_t* work_counter;
    unsigned long long* counters;  // nodes, tests, accepted, bounce[5], -, -, watchdog, ...; from 16: step statistics
    uint32_t scene_lds_bytes;
    uint32_t rows_lds_off;         // byte offset from the workgroup's LDS base to the waves' row tables
    int32_t glass_always_draws;
    uint64_t watchdog_ticks;  // 100 MHz ticks without a finished path after which a wave gives up (scheduling-bug guard)
    uint32_t thr[6];  // lanes that must wait for a step before it runs: node, leaf, (unused: terminal lanes are
                      // served by the restart step), opaque, glass, restart
    float inv_tiles_x;
    uint32_t small_ids;  // 1: job and tile ids < 2^24 (divmod24 is exact)
    // SCENE_GLOBAL only: scene_lds_bytes is the node prefix kept in LDS, the stack keeps
    // stack_lds_entries in LDS and the rest in stack_spill[block][entry][lane]
    int32_t* stack_spill;
    uint32_t stack_lds_entries, stack_spill_entries;
};

__global__ void pt_hash_kernel(uint32_t* out, uint32_t w, uint32_t h) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    out[i] = pixel_hash(i % w, i / w, w, h);
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
enum { SCENE_GLOBAL = 0, SCENE_LDS = 1, SCENE_LDS_HALF = 2, SCENE_REFERENCE = 3 };
template <int SCENE> struct SceneMem { typedef HybridMem type; };
template <> struct SceneMem<SCENE_LDS> { typedef LdsOrderedMem type; };
template <> struct SceneMem<SCENE_LDS_HALF> { typedef LdsMem type; };

// Copies what the workgroup keeps in LDS: the whole blob (LDS scenes; float nodes become
// direction-ordered 80-byte records on the way) or the node prefix (global scenes).
template <int SCENE, int BLK>
__device__ __forceinline__ void stage_scene(const RenderParams& P, uint32_t lds0) {
    typedef uint32_t w4 __attribute__((ext_vector_type(4)));
    typedef uint32_t w2 __attribute__((ext_vector_type(2)));
    const w4* src = reinterpret_cast<const w4*>(P.scene);
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
        const uint32_t rest = (P.scene_lds_bytes - n * LDS_NODE_BYTES) / 16u;
        for (uint32_t i = threadIdx.x; i < rest; i += BLK)
            *reinterpret_cast<PT_LDS(w4)*>(lds0 + n * LDS_NODE_BYTES + i * 16u) = src[n * 4u + i];
    } else {
        for (uint32_t i = threadIdx.x; i < P.scene_lds_bytes / 16u; i += BLK)
            *reinterpret_cast<PT_LDS(w4)*>(lds0 + i * 16u) = src[i];
    }
    __syncthreads();
}

// At the end of a path kernel: the rays the wave traced, one atomic per wave on a striped counter.
__device__ __forceinline__ void flush_rays(const RenderParams& P, uint32_t lane_rays) {
    const unsigned long long total = wave_sum((unsigned long long)lane_rays);
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63u) == 0 && total) atomicAdd(P.ray_slots + (size_t)(wave % RAY_SLOTS) * RAY_SLOT_STRIDE, total);
}

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// Every lane is a small state machine kept in Walk::cur -- interior node pending, leaf pending, walk
// finished, one of three scatter classes pending, idle (needs a path or a camera ray), exited -- and the
// wave runs one kind of step at a time, only when enough lanes wait for it (ballot + popcount against a
// threshold).  Lanes in a thinly populated state wait until their state fills up, so the long scatter
// bodies run with many lanes active instead of being serialised over sparsely populated material
// branches, and the node loop stops when its population drops below the threshold instead of dragging a
// few long walks along with most lanes idle.  When no state reaches its threshold every non-empty state
// runs once, which guarantees progress.  Scheduling only reorders work between independent paths; every
// path performs exactly the arithmetic of compute.wgsl's trace_path.
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

template <int SCENE, bool COUNT, int BLK>
__global__ void __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(BLK == 1024 ? 8 : 1, 8))) pt_render_sched_kernel(const RenderParams P) {
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
        mem.base = lds0 + P.L.n_nodes * (LDS_NODE_BYTES - (uint32_t)sizeof(Node64));
    } else if constexpr (LDS_SCENE) {
        mem.base = lds0;
    } else {
        mem.base = P.scene, mem.lds_base = lds0, mem.lds_node_bytes = P.scene_lds_bytes;
        stack.spill = P.stack_spill + (size_t)blockIdx.x * P.stack_spill_entries * BLK + threadIdx.x;
        stack.lds_entries = (int32_t)P.stack_lds_entries;
    }

    const CameraRegs cam = load_camera(P.cam);
    const float fw = (float)P.width, fh = (float)P.height;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_in_block = uniform(threadIdx.x >> 6);
    // the wave's row table in LDS and its ring in global memory
    const uint32_t rows_lds = lds0 + P.rows_lds_off + wave_in_block * ROWS_LDS_BYTES;
    float* const ring = P.ring + (size_t)(blockIdx.x * (BLK / 64) + wave_in_block) * RING_WAVE_FLOATS;
    f4* const sums_row = reinterpret_cast<f4*>(ring + MAX_RING_SLOTS * RING_SLOT_FLOATS) + lane;  // this lane's private sums

    PathState ps;
    ps.o = ps.d = ps.colour = mk(0.0f, 0.0f, 0.0f);
    ps.rng = 1u;
    ps.bounce = 0;
    Walk w;
    walk_begin<HALF>(w, P.L, ps.o, mk(1.0f, 1.0f, 1.0f));
    w.cur = LANE_IDLE;
    uint32_t mbits = 0;  // material word of the hit, valid in the scatter states
    V3 final_colour = mk(0.0f, 0.0f, 0.0f);
    // path tag: pixel slot of the tile (6 bits) | ring slot << 6 (5 bits) | row-table index << 11 (5 bits)
    uint32_t tag = 0, pxy = 0, rays = 0;  // rays of the lane's current path
    uint32_t lane_rays = 0;               // rays of all its finished paths
    int32_t sample_i = 0;
    Counters cnt = {0, 0, 0};
    uint32_t bounce_cnt[5] = {0, 0, 0, 0, 0};
    uint32_t ph_exec[PH_COUNT] = {0}, ph_lanes[PH_COUNT] = {0};  // COUNT: wave-level step executions

    // wave-uniform: rows issued / committed so far, free ring slots, the row and job being issued
    uint32_t i_seq = 0, c_seq = 0;
    uint32_t free_mask = P.ring_slots >= 32u ? 0xffffffffu : (1u << P.ring_slots) - 1u;
    uint32_t row_next = ROW_PIXELS;   // pixel slots of the open row already handed out (ROW_PIXELS: no open row)
    uint32_t row_tag = 0;             // ring slot << 6 | row-table index << 11 of the open row
    uint32_t tile_xy = 0;             // tile x | tile y << 13 of the job being issued
    uint32_t f_issue = 0, f_end = 0;  // its next frame and the end of its range (equal: a new job is needed)
    uint32_t k_issue = 0;             // its range index
    uint32_t seed_cur = 0;
    bool exhausted = false, stalled = false;
    bool c_first = true;              // the next row to commit is the first of its job
    uint32_t job_rays = 0;            // rays of the committed rows of the job being committed
    bool force = false;
    uint64_t t_progress = __builtin_amdgcn_s_memrealtime();  // 100 MHz; when a path of this wave last finished

    // Adds the rows that are complete, oldest first, to the sums of their tile (lane p: pixel p).
    auto try_commit = [&]() {
        typedef uint32_t w4 __attribute__((ext_vector_type(4)));
        bool have = false, a_ok = false;   // have: `a` holds the running sums of the job being committed
        uint32_t a_idx = 0;
        f4 a = {0.0f, 0.0f, 0.0f, 0.0f};
        while (c_seq != i_seq) {
            const w4 rec = *reinterpret_cast<const PT_LDS(w4)*>(rows_lds + (c_seq & (MAX_RING_SLOTS - 1u)) * 16u);
            if (uniform(rec.y)) break;  // paths pending
            const uint32_t info = uniform(rec.x), row_rays = uniform(rec.z), k = uniform(rec.w);
            const uint32_t tx = info & 0x1fffu, ty = (info >> 13) & 0x1fffu, slot = (info >> 26) & 31u;
            const bool last = (info >> 31) != 0u;
            const uint32_t tile = ty * P.tiles_x + tx;
            if (!have) {
                if (c_first && k) {  // ranges 0 .. k-1 of this tile must be in the accumulator: one poll, no spinning
                    if (__hip_atomic_load(P.tile_done + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k) break;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drops this CU's L1 lines; the texel loads below are sc1 besides
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the wave's own result stores are in L2
                const uint32_t px = tx * 8u + (lane & 7u), py = ty * 8u + (lane >> 3);
                a_ok = px < P.width && py < P.height;
                a_idx = py * P.width + px;
                if (a_ok) {
                    // texels: sc1 loads only (written by another wave's sc1 stores, or before the launch);
                    // the private sums: this wave's own earlier stores, read past the L1 like the ring
                    const float* src = c_first ? reinterpret_cast<const float*>(P.accum + a_idx) : reinterpret_cast<const float*>(sums_row);
                    a.x = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    a.y = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    a.z = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                have = true;
            }
            if (a_ok) {
                const float* r = ring + (slot * ROW_PIXELS + lane) * 3u;
                a.x += __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a.y += __hip_atomic_load(r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a.z += __hip_atomic_load(r + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            job_rays += row_rays;
            free_mask |= 1u << slot;
            c_seq++;
            c_first = false;
            if (last) {  // the job is complete: its sums go to the accumulator, the next range of the tile may follow
                if (a_ok) {
                    float* dst = reinterpret_cast<float*>(P.accum + a_idx);  // alpha stays as it is
                    __hip_atomic_store(dst, a.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: write-through
                    __hip_atomic_store(dst + 1, a.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 2, a.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (k + 1u < P.n_ranges) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every lane's texel stores have left the L2
                    if (lane == 0) __hip_atomic_store(P.tile_done + tile, k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (P.tile_cost && lane == 0) atomicAdd(P.tile_cost + tile, (float)job_rays);
                have = false;
                c_first = true;
                job_rays = 0;
            }
        }
        if (have && a_ok) *sums_row = a;
    };
    // One sample of a path has ended with `radiance`.
    auto finish_sample = [&](V3 radiance) {
        final_colour = final_colour + radiance;
        sample_i++;
        w.cur = LANE_NEXT;
        if (sample_i >= P.samples) {
            V3 o = final_colour;
            if (P.samples != 1) {  // x / 1.0f == x: the common case needs no division
                const float fs = (float)P.samples;
                o.x = final_colour.x / fs;
                o.y = final_colour.y / fs;
                o.z = final_colour.z / fs;
            }
            float* dst = ring + (tag & 0x7ffu) * 3u;
            dst[0] = o.x, dst[1] = o.y, dst[2] = o.z;
            // one pending path fewer in the row, its rays added
            {
                PT_LDS(uint32_t)* rec = reinterpret_cast<PT_LDS(uint32_t)*>(rows_lds + (tag >> 11) * 16u);
                __hip_atomic_fetch_add(rec + 2, rays, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(rec + 1, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            lane_rays += rays;
            w.cur = LANE_IDLE;
        }
    };
    auto wants = [&](bool mine, int ph) -> bool {
        const uint32_t n = (uint32_t)__popcll(__ballot(mine));
        const bool go = n >= (force ? 1u : P.thr[ph]);  // thresholds are >= 1
        if (COUNT && go) ph_exec[ph]++, ph_lanes[ph] += n;
        return go;
    };

    bool gave_up = false;
    for (uint32_t round = 0;; round++) {
        // Watchdog: a scheduling bug must end the launch, not hang the GPU.
        if ((round & 1023u) == 1023u && __builtin_amdgcn_s_memrealtime() - t_progress > P.watchdog_ticks) {
            unsigned long long nth = 0;
            if (lane == 0) nth = atomicAdd(&P.counters[10], 1ull);
#ifdef PT_DEBUG_WATCHDOG
            nth = __shfl(nth, 0);
            if (nth < 6 && lane == 0) {
                typedef uint32_t w2 __attribute__((ext_vector_type(2)));
                const w2 rec = *reinterpret_cast<const PT_LDS(w2)*>(rows_lds + (c_seq & (MAX_RING_SLOTS - 1u)) * 16u);
                printf("gaveup blk %u wave %u lane %u: i %u c %u free %x row_next %u f_issue %u exh %d stalled %d force %d cur %x tag %x rec %x %x\n",
                       blockIdx.x, wave_in_block, lane, i_seq, c_seq, free_mask, row_next, f_issue, (int)exhausted, (int)stalled,
                       (int)force, (uint32_t)w.cur, tag, rec.x, rec.y);
            }
#endif
            gave_up = true;
            break;
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
                if (n < (force ? 1u : P.thr[PH_NODE])) break;  // a forced round drains the state
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
                if (n < (force ? 1u : P.thr[PH_LEAF])) break;
                if (COUNT) ph_exec[PH_LEAF]++, ph_lanes[PH_LEAF] += n;
            }
        }
        // Enough lanes are back at interior nodes (they came out of leaves): walk on before looking at the
        // scatter states, whose tests cost a dozen scalar instructions per round (scene1, 256 frames:
        // 20.95 -> 20.70 ms; asking for 8 or 16 lanes more than the node threshold: no gain).
        if (!force && (uint32_t)__popcll(__ballot(walk_is_node(w.cur))) >= P.thr[PH_NODE]) continue;
        // ---- finished walks: fetch the material word and sort the lane into its scatter class
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
        // ---- lambertian / metal
        if (wants(w.cur == LANE_OPAQUE, PH_OPAQUE)) {
            ran = true;
            if (w.cur == LANE_OPAQUE) {
                Hit hit;
                walk_result(w, hit);
                V3 radiance;
                if (scatter_opaque(mem, P.L, hit, mbits, ps, P.depth, radiance)) {
                    finish_sample(radiance);
                } else {
                    walk_begin<HALF>(w, P.L, ps.o, ps.d, mem.root());
                    rays++;
                }
            }
        }
        // ---- glass
        if (wants(w.cur == LANE_GLASS, PH_GLASS)) {
            ran = true;
            if (w.cur == LANE_GLASS) {
                Hit hit;
                walk_result(w, hit);
                V3 radiance;
                if (scatter_glass(mem, P.L, hit, mbits, ps, P.depth, P.glass_always_draws != 0, radiance)) {
                    finish_sample(radiance);
                } else {
                    walk_begin<HALF>(w, P.L, ps.o, ps.d, mem.root());
                    rays++;
                }
            }
        }
        // ---- restart: add the completed rows to the accumulator, close the paths that end at this hit / miss,
        // hand the next pixel slots of the wave's tile to lanes without a path, start the next sample.  One step
        // on purpose: every path needs these halves exactly once, and a separate state would only thin out the
        // others.  Idle lanes of a wave whose ring is full do not ask for it (they wait for a row to complete),
        // except in a forced round: the path that completed the oldest row may have ended in a scatter step
        // (depth used up), after which nobody else would look at the ring again.
        if (wants(lane_wants_restart(w.cur) && !(stalled && !force && w.cur == LANE_IDLE), PH_IDLE)) {
            ran = true;
            stalled = false;
            try_commit();
            if (__ballot(w.cur == LANE_TERMINAL)) t_progress = __builtin_amdgcn_s_memrealtime();
            if (w.cur == LANE_TERMINAL) finish_sample(scatter_terminal(mem, P.L, mbits, ps));
            unsigned long long m = __ballot(w.cur == LANE_IDLE);
            while (m) {
                if (row_next == ROW_PIXELS) {  // open the next row
                    if (f_issue == f_end && exhausted) break;
                    if (i_seq - c_seq == P.ring_slots) {
                        try_commit();  // rows may have completed in this very step
                        if (i_seq - c_seq == P.ring_slots) {
                            stalled = true;
                            break;
                        }
                    }
                    if (f_issue == f_end) {  // the next job of the queue
                        uint32_t job = 0;
                        if (lane == 0) job = atomicAdd(P.work_counter, 1u);
                        job = uniform(job);  // lane 0 is active here: the step is entered by the whole wave
                        if (job >= P.n_tiles * P.n_ranges) {
                            exhausted = true;
                            break;
                        }
                        uint32_t k, rank, tx, ty;
                        if (P.small_ids) divmod24(job, P.n_tiles, P.inv_n_tiles, k, rank);
                        else k = job / P.n_tiles, rank = job - k * P.n_tiles;
                        const uint32_t tile = uniform(P.tile_perm ? P.tile_perm[rank] : rank);
                        if (P.small_ids) divmod24(tile, P.tiles_x, P.inv_tiles_x, ty, tx);
                        else ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
                        tile_xy = uniform(tx | (ty << 13));
                        k_issue = uniform(k);
                        f_issue = k_issue * P.job_frames;
                        f_end = min(f_issue + P.job_frames, P.n_frames);
                    }
                    const uint32_t slot = (uint32_t)__builtin_ctz(free_mask);  // not empty: fewer rows live than slots
                    free_mask &= free_mask - 1u;
                    const uint32_t ri = i_seq & (MAX_RING_SLOTS - 1u);
                    const uint32_t vx = min(8u, P.width - (tile_xy & 0x1fffu) * 8u), vy = min(8u, P.height - (tile_xy >> 13) * 8u);
                    const uint32_t last = f_issue + 1u == f_end ? 0x80000000u : 0u;
                    if (lane == 0) {
                        typedef uint32_t w4 __attribute__((ext_vector_type(4)));
                        *reinterpret_cast<PT_LDS(w4)*>(rows_lds + ri * 16u) = w4{tile_xy | (slot << 26) | last, vx * vy, 0u, k_issue};
                    }
                    row_tag = (slot << 6) | (ri << 11);
                    seed_cur = uniform(P.seeds[f_issue]);
                    f_issue++;
                    i_seq++;
                    row_next = 0;
                }
                const uint32_t avail = ROW_PIXELS - row_next;
                const uint32_t rank = lane_rank(m);
                if (w.cur == LANE_IDLE && rank < avail) {
                    const uint32_t slot = row_next + rank;
                    const uint32_t px = (tile_xy & 0x1fffu) * 8u + (slot & 7u), py = (tile_xy >> 13) * 8u + (slot >> 3);
                    if (px < P.width && py < P.height) {
                        ps.rng = pixel_seed(seed_cur, P.pixel_hash[py * P.width + px]);
                        tag = row_tag | slot;
                        pxy = px | (py << 16);
                        final_colour = mk(0.0f, 0.0f, 0.0f);
                        sample_i = 0;
                        rays = 0;
                        w.cur = LANE_NEXT;
                    }
                }
                row_next += min((uint32_t)__popcll(m), avail);
                m = __ballot(w.cur == LANE_IDLE);
            }
            if (w.cur == LANE_IDLE && exhausted) w.cur = LANE_EXIT;  // the queue is empty: this lane is finished
            if (w.cur == LANE_NEXT) {
                ps.d = camera_dir(cam, pxy & 0xffffu, pxy >> 16, fw, fh, ps.rng);
                ps.o = cam.pos;
                ps.colour = mk(1.0f, 1.0f, 1.0f);
                ps.bounce = 0;
                walk_begin<HALF>(w, P.L, ps.o, ps.d, mem.root());
                rays++;
            }
        }
        if (!ran) {
            if (__ballot(w.cur != LANE_EXIT) == 0ull) break;
            force = true;
        } else {
            force = false;
        }
    }

    // The rows that completed in the last restart step; rows of a range whose predecessor is still being added by
    // another wave wait for it (that wave never waits for this one: it owns a job with a smaller number).
    while (!gave_up) {
        try_commit();
        if (c_seq == i_seq) break;
        __builtin_amdgcn_s_sleep(32);
        if (__builtin_amdgcn_s_memrealtime() - t_progress > P.watchdog_ticks) {
            if (lane == 0) atomicAdd(&P.counters[10], 1ull);
            gave_up = true;
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
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < PH_COUNT; k++) {
                atomicAdd(&P.counters[16 + 2 * k], (unsigned long long)ph_exec[k]);
                atomicAdd(&P.counters[17 + 2 * k], (unsigned long long)ph_lanes[k]);
            }
        }
    }
}

// Tiles (8x8 pixels) ordered by decreasing cost (rays traced in them during the last launch): one
// workgroup, a 256-bucket counting sort.  Ties land in arbitrary order, which nothing depends on.
constexpr int ORDER_BLK = 1024;
__global__ void __launch_bounds__(ORDER_BLK) pt_tile_order_kernel(const float* tile_cost, uint32_t tiles, uint32_t* perm) {
    __shared__ uint32_t hist[256];
    __shared__ float red[ORDER_BLK];
    float mx = 0.0f;
    for (uint32_t t = threadIdx.x; t < tiles; t += ORDER_BLK) mx = fmaxf(mx, tile_cost[t]);
    red[threadIdx.x] = mx;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
    __syncthreads();
    for (int o = ORDER_BLK / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    const float scale = 255.0f / fmaxf(red[0], 1.0f);
    auto bucket = [&](float c) { return 255u - min(255u, (uint32_t)(c * scale)); };  // 0 = most expensive
    for (uint32_t t = threadIdx.x; t < tiles; t += ORDER_BLK) atomicAdd(&hist[bucket(tile_cost[t])], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {  // exclusive prefix over 256 buckets
        uint32_t run = 0;
        for (int b = 0; b < 256; b++) {
            const uint32_t n = hist[b];
            hist[b] = run;
            run += n;
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < tiles; t += ORDER_BLK) perm[atomicAdd(&hist[bucket(tile_cost[t])], 1u)] = t;
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
    s.block = (uint32_t)env_long("MI355PT_BLOCK", 0);
    s.stack_lds = env_long("MI355PT_STACK_LDS", 6);
    s.node_cache_kb = env_long("MI355PT_NODE_CACHE_KB", -1);
    s.tile_order = env_long("MI355PT_TILE_ORDER", 1) != 0;
    s.ring_slots = (uint32_t)std::min<long>(MAX_RING_SLOTS, std::max(1l, env_long("MI355PT_RING_SLOTS", MAX_RING_SLOTS)));
    s.watchdog_ms = (uint64_t)std::max(1l, env_long("MI355PT_WATCHDOG_MS", 20000));
    s.job_frames_min = (uint32_t)std::max(1l, env_long("MI355PT_JOB_FRAMES_MIN", 4));
    s.job_frames_max = (uint32_t)std::max<long>(s.job_frames_min, env_long("MI355PT_JOB_FRAMES_MAX", 32));
    const uint32_t default_thr[6] = {12, 8, 24, 16, 6, 32};  // node, leaf, (unused), opaque, glass, restart
    for (int k = 0; k < 6; k++) s.thr[k] = default_thr[k];
    if (const char* senv = getenv("MI355PT_SCHED")) {
        unsigned t[6];
        if (sscanf(senv, "%u,%u,%u,%u,%u,%u", &t[0], &t[1], &t[2], &t[3], &t[4], &t[5]) == 6)
            for (int k = 0; k < 6; k++) s.thr[k] = t[k] < 1 ? 1 : (t[k] > 64 ? 64 : t[k]);
    }
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
    if (hipMalloc(&d->work_counter, 256) != hipSuccess || hipMalloc(&d->counters, 256) != hipSuccess ||
        hipMalloc(&d->ray_slots, RAY_SLOTS * RAY_SLOT_STRIDE * 8) != hipSuccess) {
        destroy(d);
        err = "hipMalloc failed";
        return false;
    }
    (void)hipMemset(d->counters, 0, 256);
    (void)hipMemset(d->ray_slots, 0, RAY_SLOTS * RAY_SLOT_STRIDE * 8);
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
    (void)hipFree(d->accum);
    (void)hipFree(d->pixel_hash);
    (void)hipFree(d->ring);
    (void)hipFree(d->tile_cost);
    (void)hipFree(d->tile_perm);
    (void)hipFree(d->tile_done);
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
static uint32_t rows_bytes(uint32_t block) { return (block / 64u) * ROWS_LDS_BYTES; }

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
    d->scene_in_lds = !layout.reference_walk &&
                      lds_scene_bytes(layout) + stack_bytes(layout, 256) + rows_bytes(256) <= LDS_SCENE_LIMIT;
    if (layout.node_half && !d->scene_in_lds) {
        err = "scene: half-precision nodes are only walked from LDS";
        d->have_scene = false;
        return false;
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
    (void)hipFree(d->tile_done);
    d->tile_done = nullptr;
    d->accum = nullptr;
    d->pixel_hash = nullptr;
    d->tile_cost = nullptr, d->tile_perm = nullptr;
    d->perm_valid = false;
    const size_t n_tiles = (size_t)((width + 7) / 8) * ((height + 7) / 8);
    PT_HIP(hipMalloc(&d->accum, (size_t)n_pix * sizeof(f4)));
    PT_HIP(hipMalloc(&d->pixel_hash, (size_t)n_pix * sizeof(uint32_t)));
    PT_HIP(hipMalloc(&d->tile_cost, ((n_tiles * sizeof(float)) + 15u) & ~(size_t)15u));
    PT_HIP(hipMalloc(&d->tile_perm, n_tiles * sizeof(uint32_t)));
    PT_HIP(hipMalloc(&d->tile_done, ((n_tiles * sizeof(uint32_t)) + 15u) & ~(size_t)15u));
    d->width = width, d->height = height, d->n_pix = n_pix;
    hipLaunchKernelGGL(pt_hash_kernel, dim3((n_pix + 255) / 256), dim3(256), 0, d->stream, d->pixel_hash, width,
                       height);
    PT_HIP(hipGetLastError());
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
    PT_HIP(hipMemsetAsync(d->counters, 0, 256, d->stream));
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
static bool launch_render(Device* d, RenderParams& P, uint32_t lds_bytes, float* ring, size_t ring_cap_bytes,
                          std::string& err) {
    auto kern = pt_render_sched_kernel<SCENE, COUNT, BLK>;
    if (lds_bytes > 48u * 1024u)
        PT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds_bytes));
    int per_cu = 0;
    PT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, BLK, lds_bytes));
    if (per_cu < 1) per_cu = 1;
    uint32_t grid = (uint32_t)d->prop.multiProcessorCount * (uint32_t)per_cu;
    // never launch more waves than there are jobs
    const uint32_t max_blocks = (uint32_t)std::min<uint64_t>(0x7fffffffull, ((uint64_t)P.n_tiles * P.n_ranges + (BLK / 64) - 1u) / (BLK / 64));
    grid = std::max(1u, std::min(grid, max_blocks));
    // and never more than the ring holds
    const size_t per_block = (size_t)(BLK / 64) * RING_WAVE_FLOATS * sizeof(float);
    grid = (uint32_t)std::min<size_t>(grid, ring_cap_bytes / per_block);
    if (grid == 0) {
        err = "render: the result ring is too small for one workgroup";
        return false;
    }
    P.ring = ring;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLK), lds_bytes, d->stream, P);
    PT_HIP(hipGetLastError());
    return true;
}

template <int BLK>
static bool launch_render_blk(Device* d, RenderParams& P, uint32_t lds_bytes, int scene_mode, bool count, float* ring,
                              size_t ring_cap_bytes, std::string& err) {
    if (scene_mode == SCENE_LDS_HALF)
        return count ? launch_render<SCENE_LDS_HALF, true, BLK>(d, P, lds_bytes, ring, ring_cap_bytes, err)
                     : launch_render<SCENE_LDS_HALF, false, BLK>(d, P, lds_bytes, ring, ring_cap_bytes, err);
    if (scene_mode == SCENE_LDS)
        return count ? launch_render<SCENE_LDS, true, BLK>(d, P, lds_bytes, ring, ring_cap_bytes, err)
                     : launch_render<SCENE_LDS, false, BLK>(d, P, lds_bytes, ring, ring_cap_bytes, err);
    if (scene_mode == SCENE_REFERENCE)
        return count ? launch_render<SCENE_REFERENCE, true, BLK>(d, P, lds_bytes, ring, ring_cap_bytes, err)
                     : launch_render<SCENE_REFERENCE, false, BLK>(d, P, lds_bytes, ring, ring_cap_bytes, err);
    return count ? launch_render<SCENE_GLOBAL, true, BLK>(d, P, lds_bytes, ring, ring_cap_bytes, err)
                 : launch_render<SCENE_GLOBAL, false, BLK>(d, P, lds_bytes, ring, ring_cap_bytes, err);
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

bool finish(Device* d, RenderStats& stats, std::string& err) {
    if (!d->in_flight) return true;
    DeviceGuard guard(d->id);
    // whatever happens below, the queue is empty afterwards: a failed call must not poison the next one
    struct Reset {
        Device* d;
        ~Reset() {
            for (auto& t : d->timed) d->event_pool.push_back(t.begin), d->event_pool.push_back(t.end);
            d->timed.clear();
            d->seed_staging.clear();
            d->seeds_head = 0;
            d->in_flight = false;
            d->pending_paths = 0;
            d->pending_launches = 0;
        }
    } reset{d};
    PT_SYNC_ALL(d);
    for (auto& t : d->timed) {
        float ms = 0.0f;
        PT_HIP(hipEventElapsedTime(&ms, t.begin, t.end));
        stats.render_ms += ms;
    }
    unsigned long long host_counters[28];
    PT_HIP(hipMemcpy(host_counters, d->counters, sizeof host_counters, hipMemcpyDeviceToHost));
    for (int k = 0; k < 12; k++) stats.phase[k] = host_counters[16 + k];
    stats.nodes = host_counters[0];
    stats.tests = host_counters[1];
    stats.accepted = host_counters[2];
    for (int k = 0; k < 5; k++) stats.bounces[k] = host_counters[3 + k];
    {
        std::vector<unsigned long long> slots(RAY_SLOTS * RAY_SLOT_STRIDE);
        PT_HIP(hipMemcpy(slots.data(), d->ray_slots, slots.size() * 8, hipMemcpyDeviceToHost));
        stats.rays = 0;
        for (uint32_t k = 0; k < RAY_SLOTS; k++) stats.rays += slots[(size_t)k * RAY_SLOT_STRIDE];
    }
    stats.paths += d->pending_paths;
    stats.launches += d->pending_launches;
    if (host_counters[10]) {
        // reported once: the rows the waves had completed are in the accumulator, the rest of the launch is missing
        PT_HIP(hipMemset(d->counters + 10, 0, 8));
        err = "render: kernel watchdog expired (" + std::to_string(host_counters[10]) +
              " waves gave up); the accumulator lacks part of the frames of that call";
        return false;
    }
    return true;
}

bool enqueue(Device* d, const Camera64& cam, const RenderConfig& cfg, const uint32_t* seeds, uint32_t n_frames,
             RenderStats& stats, std::string& err) {
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

    const uint32_t tiles_x = (d->width + 7) / 8, tiles_y = (d->height + 7) / 8;
    const uint32_t tiles = tiles_x * tiles_y;
    // Frames per launch.  A launch ends with a tail in which the waves run out of tiles one after the other, so
    // launches should be long; nothing but the seed buffer grows with the number of frames.
    uint32_t batch = cfg.max_batch ? cfg.max_batch : d->sw.max_batch;
    if (batch == 0) batch = 4096;
    batch = std::min(batch, n_frames);

    // ---- launch geometry
    const Switches& sw = d->sw;
    const bool in_lds = d->scene_in_lds;
    const int scene_mode = d->layout.reference_walk ? SCENE_REFERENCE
                           : in_lds ? (d->layout.node_half ? SCENE_LDS_HALF : SCENE_LDS) : SCENE_GLOBAL;
    const size_t lds_cu = 160u * 1024u;
    const uint32_t cus = (uint32_t)d->prop.multiProcessorCount;
    // Few jobs (small images, few frames): small workgroups, so that the waves spread over all compute units.
    const bool few_tiles = (uint64_t)tiles * std::max(1u, batch / d->sw.job_frames_min) < cus * 24ull;
    // Short launches (below 32 frames at 1920x1080) are a few per cent faster with the smaller workgroups; long
    // ones 7 % faster with 1024 lanes at 8 waves per SIMD.
    const bool long_launch = (uint64_t)batch * tiles >= 96ull * 32ull * (uint64_t)cus;
    RenderParams P;
    memset(&P, 0, sizeof P);
    uint32_t block, lds_bytes;
    if (in_lds) {
        P.scene_lds_bytes = lds_scene_bytes(d->layout);
        // Workgroup size: the scene copy in LDS is shared by the workgroup, the traversal stacks and the row
        // tables are per lane / wave.  Larger workgroups amortise the copy (more waves fit next to it); deep trees
        // make the stacks dominate and favour smaller ones.  Pick what keeps more waves resident per CU.
        // Registers: the kernel needs about 70 vector registers (7 waves per SIMD: 28 per CU, 24 in whole
        // 512-lane workgroups); its 1024-lane instance is compiled for 64 (8 per SIMD, two workgroups per CU) --
        // more waves raise the instruction-issue ceiling the kernel runs at: scene1, 256 frames, 20.65 ms with
        // 3 x 512 lanes per CU, 19.20 ms with 2 x 1024.
        auto waves_per_cu = [&](uint32_t blk) {
            size_t per_block = (size_t)P.scene_lds_bytes + stack_bytes(d->layout, blk) + rows_bytes(blk);
            size_t blocks = per_block ? lds_cu / per_block : 8;
            const size_t cap = blk == 1024 ? 32 : (blk == 512 ? 24 : 28);
            return std::min<size_t>(blocks * (blk / 64), cap);
        };
        block = waves_per_cu(512) > waves_per_cu(256) ? 512u : 256u;
        if (long_launch && waves_per_cu(1024) > waves_per_cu(block)) block = 1024u;
        if (few_tiles) block = 256u;
        if (sw.block == 256 || sw.block == 512 || sw.block == 1024) block = sw.block;
        P.rows_lds_off = P.scene_lds_bytes + stack_bytes(d->layout, block);
        lds_bytes = P.rows_lds_off + rows_bytes(block);
    } else {
        // Scene in global memory.  LDS holds, per workgroup, the first stack entries of every lane and
        // as much of the top of the tree as still lets 24 waves stay resident on a CU.  Measured on
        // MI355X with 1 M spheres at 3840x2160: 6 entries + 40 KB of nodes in 512-lane workgroups
        // 18.3 ms per 16 frames; everything in global memory 21.7 ms; a 64 KB prefix at 8 waves per CU
        // 33 ms (occupancy matters more than the size of the prefix).
        // MI355PT_STACK_LDS (entries) and MI355PT_NODE_CACHE_KB override the split (development).
        block = !long_launch ? 512u : 1024u;  // the 1024-lane instance runs 8 waves per SIMD
        if (few_tiles) block = 256u;
        if (sw.block == 256 || sw.block == 512 || sw.block == 1024) block = sw.block;
        const uint32_t depth = std::max(1u, d->layout.max_depth);
        P.stack_lds_entries = (uint32_t)std::min<long>(depth, std::max<long>(1, sw.stack_lds));
        P.stack_spill_entries = std::max(1u, depth - P.stack_lds_entries);
        const uint32_t stack_lds = P.stack_lds_entries * block * 4u;
        const uint32_t blocks_per_cu = std::max(1u, (block == 1024u ? 32u : 24u) / (block / 64u));
        const long budget = (long)(lds_cu / blocks_per_cu) - (long)stack_lds - (long)rows_bytes(block) - 2048;  // LDS is granted in granules
        long cache = sw.node_cache_kb >= 0 ? sw.node_cache_kb * 1024 : budget;
        cache = std::min<long>(cache, (long)lds_cu - (long)stack_lds - (long)rows_bytes(block));
        cache = std::min<long>(cache, 4096l * (long)sizeof(Node64));  // the renumbered prefix (order_top_first)
        cache = std::min<long>(cache, (long)d->layout.n_nodes * (long)sizeof(Node64));
        cache = std::max<long>(cache, 0) / 64 * 64;
        P.scene_lds_bytes = (d->layout.off_nodes == 0) ? (uint32_t)cache : 0u;
        P.rows_lds_off = P.scene_lds_bytes + stack_lds;
        lds_bytes = P.rows_lds_off + rows_bytes(block);
        const size_t need_spill = (size_t)cus * (2048u / block) * P.stack_spill_entries * block;
        if (need_spill > d->spill_cap) {
            if (!finish(d, stats, err)) return false;
            (void)hipFree(d->stack_spill);
            d->stack_spill = nullptr;
            d->spill_cap = 0;
            PT_HIP(hipMalloc(&d->stack_spill, need_spill * sizeof(int32_t)));
            d->spill_cap = need_spill;
        }
        P.stack_spill = d->stack_spill;
    }
    if (lds_bytes > lds_cu) {
        err = "render: traversal stacks do not fit in LDS (tree too deep)";
        return false;
    }
    // The result ring: one per wave that can be resident.
    P.ring_slots = sw.ring_slots;
    {
        const size_t need = (size_t)cus * 32u * RING_WAVE_FLOATS * sizeof(float);
        if (need > d->ring_cap) {
            if (!finish(d, stats, err)) return false;
            (void)hipFree(d->ring);
            d->ring = nullptr;
            d->ring_cap = 0;
            PT_HIP(hipMalloc(&d->ring, need));
            d->ring_cap = need;
        }
    }
    if (n_frames > d->seeds_cap) {
        if (!finish(d, stats, err)) return false;
        (void)hipFree(d->seeds);
        d->seeds = nullptr;
        d->seeds_cap = 0;
        const size_t cap = std::max<size_t>(4096, 2 * (size_t)n_frames);
        PT_HIP(hipMalloc(&d->seeds, cap * 4));
        d->seeds_cap = cap;
        d->seeds_head = 0;
    }
    if (d->seeds_head + n_frames > d->seeds_cap && !finish(d, stats, err)) return false;  // ring is full: drain

    // From here on work is queued: whatever fails, finish() has to run before the buffers are touched again.
    d->in_flight = true;
    uint32_t* const dev_seeds = d->seeds + d->seeds_head;
    d->seeds_head += n_frames;
    d->seed_staging.emplace_back(seeds, seeds + n_frames);
    PT_HIP(hipMemcpyAsync(dev_seeds, d->seed_staging.back().data(), (size_t)n_frames * 4, hipMemcpyHostToDevice,
                          d->stream));

    P.scene = d->scene;
    P.L = d->layout;
    P.cam = cam;
    P.width = d->width, P.height = d->height, P.tiles_x = tiles_x, P.n_tiles = tiles;
    P.samples = cfg.samples, P.depth = cfg.depth;
    P.pixel_hash = d->pixel_hash;
    P.accum = d->accum;
    P.work_counter = d->work_counter;
    P.counters = d->counters;
    P.ray_slots = d->ray_slots;
    P.glass_always_draws = cfg.glass_always_draws ? 1 : 0;
    P.watchdog_ticks = sw.watchdog_ms * 100000ull;  // 100 MHz
    P.inv_tiles_x = 1.0f / (float)tiles_x;
    P.inv_n_tiles = 1.0f / (float)tiles;
    P.tile_done = d->tile_done;
    for (int k = 0; k < 6; k++) P.thr[k] = sw.thr[k];

    bool use_perm = sw.tile_order && d->perm_valid && memcmp(&cam, &d->perm_cam, sizeof cam) == 0 &&
                    cfg.depth == d->perm_depth && cfg.samples == d->perm_samples;
    for (uint32_t f0 = 0; f0 < n_frames; f0 += batch) {
        const uint32_t nb = std::min(batch, n_frames - f0);
        P.n_frames = nb;
        // Frames per job: small enough that the queue holds several jobs per resident wave and ends with short
        // ones, large enough that a job's texel traffic (one read, one write per job) stays small.
        {
            const uint64_t target_jobs = 8ull * cus * 32ull;
            uint64_t jf = (uint64_t)nb * tiles / target_jobs;
            jf = std::max<uint64_t>(sw.job_frames_min, std::min<uint64_t>(jf, sw.job_frames_max));
            P.job_frames = (uint32_t)std::min<uint64_t>(jf, nb);
            P.n_ranges = (nb + P.job_frames - 1u) / P.job_frames;
            if ((uint64_t)P.n_ranges * tiles > 0xffffffffull) {
                err = "render: too many jobs for one launch";
                return false;
            }
            P.small_ids = (uint64_t)P.n_ranges * tiles < (1ull << 24) ? 1u : 0u;
        }
        if (P.n_ranges > 1) PT_HIP(hipMemsetAsync(d->tile_done, 0, (((size_t)tiles * 4u) + 15u) & ~(size_t)15u, d->stream));
        P.tile_perm = use_perm ? d->tile_perm : nullptr;
        P.seeds = dev_seeds + f0;
        // The tile costs are measured on the first launch that is long enough to be representative after the
        // camera or the settings changed, and again every 16th launch.
        const bool measure = sw.tile_order && nb >= 8u && (!use_perm || d->since_measure >= 15u);
        P.tile_cost = measure ? d->tile_cost : nullptr;
        if (measure) PT_HIP(hipMemsetAsync(d->tile_cost, 0, (((size_t)tiles * 4u) + 15u) & ~(size_t)15u, d->stream));
        PT_HIP(hipMemsetAsync(d->work_counter, 0, 4, d->stream));
        hipEvent_t ev_begin = take_event(d), ev_end = take_event(d);
        bool ok = ev_begin && ev_end && hipEventRecord(ev_begin, d->stream) == hipSuccess;
        if (ok) {
            if (block == 512) ok = launch_render_blk<512>(d, P, lds_bytes, scene_mode, cfg.count, d->ring, d->ring_cap, err);
            else if (block == 1024) ok = launch_render_blk<1024>(d, P, lds_bytes, scene_mode, cfg.count, d->ring, d->ring_cap, err);
            else ok = launch_render_blk<256>(d, P, lds_bytes, scene_mode, cfg.count, d->ring, d->ring_cap, err);
        } else {
            err = "render: could not record a timing event";
        }
        if (ok && hipEventRecord(ev_end, d->stream) != hipSuccess) {
            ok = false;
            err = "render: could not record a timing event";
        }
        if (!ok) {  // nothing is left half-recorded: the events go back to the pool
            if (ev_begin) d->event_pool.push_back(ev_begin);
            if (ev_end) d->event_pool.push_back(ev_end);
            return false;
        }
        d->timed.push_back(Device::Timed{ev_begin, ev_end});
        if (measure) {
            hipLaunchKernelGGL(pt_tile_order_kernel, dim3(1), dim3(ORDER_BLK), 0, d->stream, d->tile_cost, tiles,
                               d->tile_perm);
            PT_HIP(hipGetLastError());
            d->perm_valid = true;
            d->perm_cam = cam;
            d->perm_depth = cfg.depth, d->perm_samples = cfg.samples;
            d->since_measure = 0;
            use_perm = true;   // later launches of this call are ordered after the kernel above
        } else {
            d->since_measure++;
        }
        d->pending_launches++;
        d->pending_paths += (uint64_t)d->n_pix * (uint64_t)cfg.samples * nb;
    }
    return true;
}

}  // namespace gpu
}  // namespace pt
