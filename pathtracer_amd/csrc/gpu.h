// Host-visible interface of the HIP side (implemented in kernels.hip).  Plain C++ types only.
#pragma once
#include <cstdint>
#include <string>

#include "pt_types.h"

namespace pt {
struct SceneArrays;  // scene_pack.h
namespace gpu {

struct Device;  // one GPU: stream, scene blob, accumulator, sample buffer, work counter

struct RenderStats {
    uint64_t paths = 0;      // primary rays started (pixel-frames x samples)
    uint64_t rays = 0;       // closest-hit queries
    uint64_t nodes = 0;      // interior nodes fetched (two box tests each)   [counting launches only]
    uint64_t tests = 0;      // sphere tests                                  [counting launches only]
    uint64_t accepted = 0;   // sphere tests that replaced the closest record [counting launches only]
    uint64_t bounces[5] = {0, 0, 0, 0, 0};  // light, lambertian, metal, glass, sky  [counting only]
    uint64_t phase[12] = {0};  // per step kind (node, leaf, terminal, opaque, glass, idle): wave-level
                                                  // executions, lanes served  [counting only]
    double render_ms = 0.0;  // device time of the path kernels (HIP events on the render stream)
    double resolve_ms = 0.0; // device time of the frame-order accumulation kernels
    uint32_t launches = 0;
    uint64_t exchange[4] = {0, 0, 0, 0};  // exchange variant: deposits, lanes deposited, withdrawals, lanes withdrawn [counting only]
    bool exchange_on = false;             // the last launch ran the exchange variant
    uint32_t list_tiles = 0;              // 8x8 tiles whose camera rays test a candidate list in the last launch (0: all walk)
    uint32_t list_candidates = 0;         // entries of those lists
    uint32_t sky_tiles = 0;               // tiles the last launch left to pt_accumulate_kernel (empty lists: sky only)
    bool short_div = false;               // pixel coordinates are divided by the verified short sequence (PixelDiv)
    uint32_t dropped_frames = 0;          // set by finish() when it fails: queued frames that were never added to the accumulator
};

struct RenderConfig {
    int32_t samples = 1;
    int32_t depth = 50;
    bool glass_always_draws = true;
    bool count = false;       // run the instrumented kernel variant (slower)
    uint32_t max_batch = 0;   // frames per launch; 0 = choose from the image size
    bool wait_follows = false;  // the caller waits right after queueing (a blocking render): nothing can overlap the call's last launch
};

bool create(int device_id, void* stream, Device** out, std::string& err);
void destroy(Device* d);
bool upload_scene(Device* d, const void* blob, const SceneLayout& layout, std::string& err);
// Builds the tree on the device (lbvh.hip) from the scene arrays and leaves the finished blob there.
bool build_scene_lbvh(Device* d, const pt::SceneArrays& arrays, SceneLayout& layout, std::string& err);
// Records `layout` as the scene now in d's blob and decides where the kernels read it from.
bool adopt_layout(Device* d, const SceneLayout& layout, std::string& err);
bool resize(Device* d, uint32_t width, uint32_t height, std::string& err);
bool clear(Device* d, std::string& err);
// Accumulates n_frames frames (seeds[f] = Globals.seed of frame f) in frame order.
bool render(Device* d, const Camera64& cam, const RenderConfig& cfg, const uint32_t* seeds, uint32_t n_frames,
            RenderStats& stats, std::string& err);
// The same in two halves: enqueue() returns once the work is queued on the device's stream; finish() waits for everything queued so far and folds its timings and counters into `stats`.
// Every other entry point below expects finish() to have been called.
bool enqueue(Device* d, const Camera64& cam, const RenderConfig& cfg, const uint32_t* seeds, uint32_t n_frames,
             RenderStats& stats, std::string& err);
bool finish(Device* d, RenderStats& stats, std::string& err);
// render() without any allocation: results, seeds, work-queue counter and stack spill area live in the
// library's static workspace (one per device, shared by every context on it: calls that overlap on one
// device race).  At most STATIC_MAX_FRAMES frames per call; fails when one frame of results does not fit.
constexpr uint32_t STATIC_MAX_FRAMES = 1024;
bool render_static(Device* d, const Camera64& cam, const RenderConfig& cfg, const uint32_t* seeds, uint32_t n_frames,
                   RenderStats& stats, std::string& err);
// Sums the accumulators of n contexts (same image size, nothing in flight) into devs[root]'s: contexts that
// share a device are added with a kernel, one accumulator per device is then summed by a single RCCL reduce
// (one process, ncclCommInitAll over the devices involved; librccl.so is opened on first use).  The other
// accumulators are left in an unspecified state.
bool reduce_accum(Device* const* devs, size_t n, size_t root, std::string& err);
// The same step for hosts that run one process per device and own the communicator: this context's part of an
// ncclReduce(sum) to rank `root`, queued on its stream and waited for.  On ranks other than root the
// accumulator is unchanged; on root the alpha channel holds the rank count afterwards (set_alpha puts 1 back).
bool reduce_accum_comm(Device* d, void* nccl_comm, int root, std::string& err);
bool set_alpha(Device* d, float alpha, std::string& err);
// Raw kernel counters (development builds read their histograms here).
bool debug_counters(Device* d, unsigned long long* out, size_t n, std::string& err);
bool read_accum(Device* d, float* rgba, size_t n_floats, std::string& err);
bool write_accum(Device* d, const float* rgba, size_t n_floats, std::string& err);
// The packed scene as the device holds it (tree included, whoever built it) and its layout; `need` = its size, copied
// only when `out` holds that much.  For the tests: a tree built on the device is checked and walked on the CPU.
bool read_scene_blob(Device* d, void* out, size_t cap, size_t& need, SceneLayout& layout, std::string& err);
void* accum_device_ptr(Device* d);
void* stream_handle(Device* d);
bool synchronize(Device* d, std::string& err);
int device_id(Device* d);
bool scene_in_lds(Device* d);
// Chip facts used by callers to size work and report rooflines.
bool device_info(Device* d, int* compute_units, int* clock_khz, size_t* lds_per_cu, char name[256]);

}  // namespace gpu
}  // namespace pt
