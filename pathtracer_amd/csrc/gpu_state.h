// State of one GPU context and the error-check macro, shared by the .hip files of the library.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <deque>
#include <string>
#include <vector>

#include "gpu.h"
#include "pt_types.h"
#include "pt_device.h"

namespace pt {
namespace gpu {

constexpr size_t COUNTER_BYTES = 1024;  // 128 kernel counters: 0..31 statistics and watchdog, 32..95 development histograms
constexpr uint32_t RAY_SLOTS = 64, RAY_SLOT_STRIDE = 16;  // 64 ray counters, 128 bytes apart

// Makes `id` the calling thread's current device for the lifetime of the guard and puts the caller's
// device back afterwards: a host that also drives HIP (or torch) on another device keeps its own.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int id) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != id) (void)hipSetDevice(id);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// Development switches (environment variables, not part of the ABI), read once when a context is created.
struct Switches {
    uint32_t max_batch = 0;     // MI355PT_MAX_BATCH: frames per launch (0: what the frame buffer holds)
    uint32_t frame_buf_gb = 0;  // MI355PT_FRAME_BUF_GB: frame buffer budget (0: FRAME_BUF_DEFAULT_BYTES)
    uint32_t block = 0;         // MI355PT_BLOCK: workgroup size 256 | 512 | 1024 (0: chosen per launch)
    long stack_lds = 5;         // MI355PT_STACK_LDS: stack entries kept in LDS for scenes walked from global memory
    long node_cache_kb = -1;    // MI355PT_NODE_CACHE_KB: node prefix kept in LDS for those scenes (-1: what fits)
    bool tile_order = true;     // MI355PT_TILE_ORDER=0: tiles in row-major order
    uint32_t chunk = 0;         // MI355PT_CHUNK: largest work-queue chunk in units (0: 32; 16 for launches shorter than 16 frames of a scene in global memory)
    float chunk_share = 2.0f;   // MI355PT_CHUNK_SHARE
    float chunk_rays = 2600.0f; // MI355PT_CHUNK_RAYS
    bool lockstep = false;      // MI355PT_KERNEL=lockstep: the simple kernel
    uint64_t watchdog_ms = 20000;  // MI355PT_WATCHDOG_MS
    uint32_t thr[6] = {12, 8, 24, 16, 6, 32};  // MI355PT_SCHED: step thresholds node, leaf, -, opaque, glass, restart
    bool ordered_global = true;                 // MI355PT_NODES=plain: trees in global memory are walked as Node64 records
    bool exchange_global = true;                // MI355PT_XCHG_GLOBAL=0: no exchange variant for trees in global memory
    bool exchange = true;                       // MI355PT_XCHG=0: never the exchange variant of the scheduled kernel
    uint32_t xthr[6] = {10, 6, 24, 24, 16, 32}; // MI355PT_XCHG_SCHED: step thresholds of the exchange variant (65: never locally)
    uint32_t measure_every = 64;                // MI355PT_MEASURE_EVERY: the tile costs are measured again on every n-th launch
    bool defer_resolve = true;                  // MI355PT_DEFER_RESOLVE=0: every launch's results are added by a kernel right behind it
    uint32_t resolve_wgs = 16;                  // MI355PT_RESOLVE_WGS: workgroups of a path kernel that add the previous launch's results
    bool short_div = true;                      // MI355PT_SHORT_DIV=0: pixel coordinates are always divided the general way
    bool sky_tiles = true;                      // MI355PT_SKY_TILES=0: tiles that show only sky go through the path kernel like the rest
    bool tile_lists = true;                     // MI355PT_TILE_LISTS=0: camera rays walk the tree like every other ray
    uint32_t tile_list_max = TILE_LIST_CAP;     // MI355PT_TILE_LIST_MAX: tiles with more candidates keep the walk
};
// Results of one launch wait in a frame buffer for the accumulation kernel; it is at most this large (and at
// most a quarter of the free device memory): 334 frames at 1920x1080, 84 at 3840x2160.
constexpr size_t FRAME_BUF_DEFAULT_BYTES = (size_t)8 << 30;

struct Device {
    int id = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipDeviceProp_t prop;
    Switches sw;

    char* scene = nullptr;
    size_t scene_cap = 0;
    char* onodes = nullptr;  // direction-ordered records of a tree that stays in global memory (adopt_layout)
    size_t onodes_cap = 0;
    bool onodes_valid = false;
    SceneLayout layout{};
    bool have_scene = false;
    bool scene_in_lds = false;

    uint32_t width = 0, height = 0, n_pix = 0;
    f4* accum = nullptr;
    uint32_t* pixel_hash = nullptr;
    // Results of one launch, frame-major; added to the accumulator in frame order by pt_accumulate_kernel on
    // the same stream.  (Running that kernel, or the next launch's path kernel, on a second stream under
    // the tail of the current one was measured 5-10 % slower on MI355X: concurrent kernels get in the
    // way of the persistent kernel's workgroup placement.)
    f4* frame_buf = nullptr;
    size_t frame_cap = 0;  // in f4 elements; two halves, written by alternate launches
    uint32_t frame_half = 0;  // the half the next launch writes
    // Buffered results of the last launch that are not in the accumulator yet: the next launch's path kernel adds
    // them in its prologue, finish() (or a launch that cannot) with a kernel of its own.
    struct PendingResolve {
        bool active = false;
        const f4* results = nullptr;
        const uint32_t* perm = nullptr;
        uint32_t frames = 0, tiles = 0, active_tiles = 0;
    } pending_resolve;
    // Cost-ordered work: rays per tile of the last measuring launch, tiles sorted by it, and what it was measured for
    float* tile_cost = nullptr;    // [tiles]
    uint32_t* tile_perm = nullptr; // [tiles]
    uint16_t* rank_chunk = nullptr; // [tiles] chunk size for a chunk that starts in the tile of that rank
    bool perm_valid = false;
    uint32_t since_measure = 0;  // launches since the tile costs were last measured
    Camera64 perm_cam{};
    int32_t perm_depth = 0, perm_samples = 0;
    // Candidate lists of the camera rays (pt_tile_lists_kernel): [tiles][TILE_LIST_WORDS], allocated with the
    // accumulator, rebuilt on the render stream when camera, scene or image size changed
    uint32_t* tile_lists = nullptr;
    uint32_t* list_count = nullptr;  // two words: tiles with a list, entries of all lists (written by the list kernel)
    bool lists_in_use = false;       // the last launch read the lists
    uint32_t list_counts[4] = {0, 0, 0, 0};  // host copy of list_count: tiles with a list, entries, tiles with an empty list
    bool perm_sky_last = false;      // tile_perm was made with these lists: the tiles with an empty list are its last ranks
    uint32_t sky_tiles_last = 0;     // tiles the last launch left to pt_accumulate_kernel
    float rcp_w = 0.0f, rcp_h = 0.0f;  // 1 / width, 1 / height
    bool short_div = false;          // PixelDiv's short sequence verified for this width and height (resize)
    bool lists_valid = false;
    Camera64 lists_cam{};
    uint32_t* seeds = nullptr;  // ring: every enqueued call owns a contiguous range until finish()
    size_t seeds_cap = 0, seeds_head = 0;
    std::deque<std::vector<uint32_t>> seed_staging;  // host copies kept alive until finish()
    struct Timed {
        hipEvent_t begin, end;
        bool resolve;
    };
    std::vector<Timed> timed;          // launches enqueued since the last finish(), both events recorded
    std::vector<hipEvent_t> event_pool;
    bool in_flight = false;
    // queued launches in order: frames and paths of each, credited by finish() once their results have been added
    struct Pending {
        uint32_t frames;
        uint64_t paths;
    };
    std::vector<Pending> pending;
    unsigned long long added_seen = 0;  // launches the accumulation kernel had added at the last finish() (counters[9])
    int32_t* stack_spill = nullptr;  // deep traversal-stack entries of scenes walked from global memory
    size_t spill_cap = 0;            // in entries
    uint32_t* work_counter = nullptr;  // [1]
    unsigned long long* counters = nullptr;  // [32] kernel counters, watchdog, step statistics
    unsigned long long* ray_slots = nullptr; // [RAY_SLOTS * RAY_SLOT_STRIDE] rays traced, striped
    // This device's copy of the library's static workspace (render_static): its address on this device,
    // looked up once when the first context on the device is created.
    unsigned char* static_ws = nullptr;
};

// Device memory one render call works in beside the context's own buffers.  enqueue() grows the context's
// buffers to what a call needs; render_static() describes a fixed area and nothing is ever allocated.
struct Workspace {
    void* frame_buf = nullptr;       // results of one launch
    size_t frame_bytes = 0;
    uint32_t* seeds = nullptr;       // one per frame of the call
    int32_t* stack_spill = nullptr;
    size_t spill_entries = 0;
    uint32_t* work_counter = nullptr;
    bool fixed = false;              // true: capacities are final, no timing events are created
};

#define PT_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
            return false;                                                                     \
        }                                                                                     \
    } while (0)


}  // namespace gpu
}  // namespace pt
