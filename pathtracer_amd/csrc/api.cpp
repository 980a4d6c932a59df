// C ABI of libmi355pt.so (see include/mi355pt.h).  Host-side state machine of the reference's
// `Renderer` (src/renderer/mod.rs:91-320) over the HIP device code in kernels.hip.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/mi355pt.h"
#include "bvh.h"
#include "gpu.h"
#include "scene_io.h"
#include "scene_pack.h"

using namespace pt;

struct mi355pt_scene {
    Scene scene;
};

struct mi355pt_ctx {
    mi355pt_desc desc{};
    gpu::Device* dev = nullptr;
    std::string error;

    // host copy of the scene in wire format (SceneBuffers, mod.rs:32-42)
    std::vector<Sphere32> spheres;
    std::vector<Vec16> lights, lambertians, metals;
    std::vector<float> glass;
    bool have_scene = false;
    Bvh2 bvh;
    SceneLayout layout{};
    bool scene_in_lds = false;

    // camera: CameraConfig (mod.rs:96) or a caller-supplied uniform
    bool have_camera = false;
    bool camera_from_config = false;
    CameraSettings cam_settings;
    Camera64 camera{};

    Settings settings;
    uint32_t width = 0, height = 0;
    int32_t frames = -1;  // Renderer::samples (mod.rs:102,132)
    bool dirty = true;    // mod.rs:103,133
    uint32_t seed_index = 0;
    gpu::RenderStats stats;
};

namespace {

thread_local std::string g_create_error;  // failures without a context: per calling thread

int fail(mi355pt_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->error = msg;
    else g_create_error = msg;
    return code;
}

// Waits for work queued by mi355pt_render_async and folds its timings / counters into the context.
int settle(mi355pt_ctx* c) {
    std::string err;
    if (!gpu::finish(c->dev, c->stats, err)) {
        // frames that were queued but never added (a launch gave up) do not count: samples(), the divisor of
        // to_rgb8 / save_png and the seed schedule continue from what the accumulator holds
        const uint32_t dropped = c->stats.dropped_frames;
        c->stats.dropped_frames = 0;
        c->frames -= (int32_t)std::min<uint32_t>(dropped, (uint32_t)std::max(c->frames, 0));
        c->seed_index -= std::min(dropped, c->seed_index);
        return fail(c, MI355PT_ERR_DEVICE, err);
    }
    return MI355PT_OK;
}
#define SETTLE_OR_RETURN(c)              \
    do {                                 \
        int settle_rc_ = settle(c);      \
        if (settle_rc_ != MI355PT_OK) return settle_rc_; \
    } while (0)

// Builds (or converts) the tree, packs the device blob and uploads it.  ref_nodes may be null.
int install_tree(mi355pt_ctx* c, const Node48* ref_nodes, size_t n_ref_nodes) {
    SceneArrays a;
    a.spheres = c->spheres.data(), a.n_spheres = c->spheres.size();
    a.lights = c->lights.data(), a.n_lights = c->lights.size();
    a.lambertians = c->lambertians.data(), a.n_lambertians = c->lambertians.size();
    a.metals = c->metals.data(), a.n_metals = c->metals.size();
    a.glass = c->glass.data(), a.n_glass = c->glass.size();
    std::string err;
    std::vector<char> blob;
    c->have_scene = false;
    if (c->desc.bvh_mode == MI355PT_BVH_LBVH && !ref_nodes) {
        c->bvh = Bvh2{};
        if (!gpu::build_scene_lbvh(c->dev, a, c->layout, err)) return fail(c, MI355PT_ERR_DEVICE, err);
        c->have_scene = true;
        c->dirty = true;
        return MI355PT_OK;
    }
    if (!pack_scene(a, c->desc.bvh_mode == MI355PT_BVH_REFERENCE, ref_nodes, n_ref_nodes, blob, c->layout, c->bvh, err))
        return fail(c, MI355PT_ERR_ARG, err);
    if (!gpu::upload_scene(c->dev, blob.data(), c->layout, err)) return fail(c, MI355PT_ERR_DEVICE, err);
    c->have_scene = true;
    c->dirty = true;
    return MI355PT_OK;
}

void rebuild_camera(mi355pt_ctx* c) {
    if (c->camera_from_config) {
        // aspect = width as f32 / height as f32 (mod.rs:115,164)
        float aspect = (float)c->width / (float)c->height;
        camera_build(c->cam_settings.pos, c->cam_settings.yaw, c->cam_settings.pitch, c->cam_settings.vfov, aspect,
                     c->camera);
    }
}

// Nothing may unwind through the C ABI: allocation failures of the host-side copies and builders (million-sphere
// scenes) and anything unexpected become error codes.
template <class F>
int guarded(mi355pt_ctx* ctx, F&& body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(ctx, MI355PT_ERR_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(ctx, MI355PT_ERR_DEVICE, std::string("unexpected failure: ") + e.what());
    } catch (...) {
        return fail(ctx, MI355PT_ERR_DEVICE, "unexpected failure");
    }
}

void copy_err(const std::string& e, char* buf, size_t len) {
    if (!buf || !len) return;
    size_t n = e.size() < len - 1 ? e.size() : len - 1;
    memcpy(buf, e.data(), n);
    buf[n] = 0;
}

}  // namespace

extern "C" {

int mi355pt_create(const mi355pt_desc* desc, mi355pt_ctx** out) {
    return guarded(nullptr, [&]() -> int {
        if (!desc || !out) return fail(nullptr, MI355PT_ERR_ARG, "create: null argument");
        *out = nullptr;
        if (desc->width == 0 || desc->height == 0) return fail(nullptr, MI355PT_ERR_ARG, "create: zero image size");
        if (desc->bvh_mode > MI355PT_BVH_LBVH) return fail(nullptr, MI355PT_ERR_ARG, "create: unknown bvh_mode");
        mi355pt_ctx* c = new (std::nothrow) mi355pt_ctx();
        if (!c) return fail(nullptr, MI355PT_ERR_NOMEM, "create: out of memory");
        c->desc = *desc;
        std::string err;
        if (!gpu::create(desc->device, desc->stream, &c->dev, err)) {
            delete c;
            return fail(nullptr, MI355PT_ERR_DEVICE, err);
        }
        c->width = desc->width, c->height = desc->height;
        if (!gpu::resize(c->dev, c->width, c->height, err)) {
            gpu::destroy(c->dev);
            delete c;
            return fail(nullptr, MI355PT_ERR_DEVICE, err);
        }
        *out = c;
        return MI355PT_OK;
    });
}

void mi355pt_destroy(mi355pt_ctx* ctx) {
    if (!ctx) return;
    (void)settle(ctx);
    gpu::destroy(ctx->dev);
    delete ctx;
}

const char* mi355pt_last_error(const mi355pt_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int mi355pt_set_scene(mi355pt_ctx* c, const void* spheres32, size_t n_spheres, const void* lights16, size_t n_lights,
                      const void* lambertians16, size_t n_lambertians, const void* metals16, size_t n_metals,
                      const void* glass4, size_t n_glass) {
    return guarded(c, [&]() -> int {
        if (!c) return MI355PT_ERR_ARG;
        SETTLE_OR_RETURN(c);
        {
            SceneArrays a;
            a.spheres = static_cast<const Sphere32*>(spheres32), a.n_spheres = n_spheres;
            a.lights = static_cast<const Vec16*>(lights16), a.n_lights = n_lights;
            a.lambertians = static_cast<const Vec16*>(lambertians16), a.n_lambertians = n_lambertians;
            a.metals = static_cast<const Vec16*>(metals16), a.n_metals = n_metals;
            a.glass = static_cast<const float*>(glass4), a.n_glass = n_glass;
            std::string err;
            if (!validate_scene(a, err)) return fail(c, MI355PT_ERR_ARG, err);
        }
        const Sphere32* sp = static_cast<const Sphere32*>(spheres32);
        c->have_scene = false;
        c->spheres.assign(sp, sp + n_spheres);
        auto vec16 = [](const void* p, size_t n) {
            const Vec16* v = static_cast<const Vec16*>(p);
            return n ? std::vector<Vec16>(v, v + n) : std::vector<Vec16>();
        };
        c->lights = vec16(lights16, n_lights);
        c->lambertians = vec16(lambertians16, n_lambertians);
        c->metals = vec16(metals16, n_metals);
        const float* g = static_cast<const float*>(glass4);
        c->glass = n_glass ? std::vector<float>(g, g + n_glass) : std::vector<float>();
        return install_tree(c, nullptr, 0);
    });
}

int mi355pt_set_bvh(mi355pt_ctx* c, const void* nodes48, size_t n_nodes) {
    return guarded(c, [&]() -> int {
        if (!c) return MI355PT_ERR_ARG;
        if (c->spheres.empty()) return fail(c, MI355PT_ERR_STATE, "set_bvh: call set_scene first");
        if (!nodes48) return fail(c, MI355PT_ERR_ARG, "set_bvh: null nodes");
        SETTLE_OR_RETURN(c);
        return install_tree(c, static_cast<const Node48*>(nodes48), n_nodes);
    });
}

int mi355pt_set_camera(mi355pt_ctx* c, const void* camera64) {
    if (!c || !camera64) return MI355PT_ERR_ARG;
    memcpy(&c->camera, camera64, 64);
    c->camera_from_config = false;
    c->have_camera = true;
    c->dirty = true;
    return MI355PT_OK;
}

int mi355pt_set_camera_config(mi355pt_ctx* c, const float pos[3], float yaw_deg, float pitch_deg, float vfov_deg) {
    if (!c || !pos) return MI355PT_ERR_ARG;
    memcpy(c->cam_settings.pos, pos, 12);
    c->cam_settings.yaw = yaw_deg, c->cam_settings.pitch = pitch_deg, c->cam_settings.vfov = vfov_deg;
    c->camera_from_config = true;
    c->have_camera = true;
    rebuild_camera(c);
    c->dirty = true;
    return MI355PT_OK;
}

int mi355pt_get_camera(const mi355pt_ctx* c, void* out) {
    if (!c || !out || !c->have_camera) return MI355PT_ERR_ARG;
    memcpy(out, &c->camera, 64);
    return MI355PT_OK;
}

int mi355pt_set_settings(mi355pt_ctx* c, int32_t samples, int32_t depth) {
    if (!c) return MI355PT_ERR_ARG;
    if (samples < 1) return fail(c, MI355PT_ERR_ARG, "set_settings: samples must be >= 1");
    c->settings.samples = samples;
    c->settings.depth = depth;
    return MI355PT_OK;
}

int mi355pt_resize(mi355pt_ctx* c, uint32_t width, uint32_t height) {
    return guarded(c, [&]() -> int {
        if (!c) return MI355PT_ERR_ARG;
        SETTLE_OR_RETURN(c);
        std::string err;
        if (!gpu::resize(c->dev, width, height, err)) return fail(c, MI355PT_ERR_DEVICE, err);
        c->width = width, c->height = height;
        c->dirty = true;
        return MI355PT_OK;
    });
}

int mi355pt_clear(mi355pt_ctx* c) {
    return guarded(c, [&]() -> int {
        if (!c) return MI355PT_ERR_ARG;
        SETTLE_OR_RETURN(c);
        std::string err;
        rebuild_camera(c);
        if (!gpu::clear(c->dev, err)) return fail(c, MI355PT_ERR_DEVICE, err);
        c->dirty = false;
        c->frames = 0;
        c->seed_index = 0;
        c->stats = gpu::RenderStats{};
        return MI355PT_OK;
    });
}

static int render_queued(mi355pt_ctx* c, uint32_t n_frames, const uint32_t* seeds, bool wait_follows) {
    return guarded(c, [&]() -> int {
        if (!c) return MI355PT_ERR_ARG;
        if (!c->have_scene) return fail(c, MI355PT_ERR_STATE, "render: no scene (call set_scene)");
        if (!c->have_camera) return fail(c, MI355PT_ERR_STATE, "render: no camera (call set_camera or set_camera_config)");
        if (c->dirty) {  // mod.rs:169-195
            int r = mi355pt_clear(c);
            if (r != MI355PT_OK) return r;
        }
        if (n_frames == 0) return MI355PT_OK;
        std::vector<uint32_t> own;
        if (!seeds) {
            own.resize(n_frames);
            for (uint32_t f = 0; f < n_frames; f++) own[f] = frame_seed(c->desc.base_seed, c->seed_index + f);
            seeds = own.data();
        }
        gpu::RenderConfig cfg;
        cfg.samples = c->settings.samples;
        cfg.depth = c->settings.depth;
        cfg.glass_always_draws = !(c->desc.flags & MI355PT_FLAG_GLASS_SHORT_CIRCUIT);
        cfg.count = (c->desc.flags & MI355PT_FLAG_COUNT) != 0;
        cfg.wait_follows = wait_follows;
        std::string err;
        if (!gpu::enqueue(c->dev, c->camera, cfg, seeds, n_frames, c->stats, err)) return fail(c, MI355PT_ERR_DEVICE, err);
        c->frames += (int32_t)n_frames;
        c->seed_index += n_frames;
        return MI355PT_OK;
    });
}

int mi355pt_render_async(mi355pt_ctx* c, uint32_t n_frames, const uint32_t* seeds) {
    return render_queued(c, n_frames, seeds, false);
}

int mi355pt_render_static(mi355pt_ctx* c, uint32_t n_frames, const uint32_t* seeds) {
    return guarded(c, [&]() -> int {
        if (!c) return MI355PT_ERR_ARG;
        if (!c->have_scene) return fail(c, MI355PT_ERR_STATE, "render: no scene (call set_scene)");
        if (!c->have_camera) return fail(c, MI355PT_ERR_STATE, "render: no camera (call set_camera or set_camera_config)");
        if (n_frames > gpu::STATIC_MAX_FRAMES)
            return fail(c, MI355PT_ERR_ARG, "render_static: at most " + std::to_string(gpu::STATIC_MAX_FRAMES) + " frames per call");
        SETTLE_OR_RETURN(c);
        if (c->dirty) {  // mod.rs:169-195; the clear is a fill kernel and two memsets, no allocation
            int r = mi355pt_clear(c);
            if (r != MI355PT_OK) return r;
        }
        if (n_frames == 0) return MI355PT_OK;
        uint32_t own[gpu::STATIC_MAX_FRAMES];
        if (!seeds) {
            for (uint32_t f = 0; f < n_frames; f++) own[f] = frame_seed(c->desc.base_seed, c->seed_index + f);
            seeds = own;
        }
        gpu::RenderConfig cfg;
        cfg.samples = c->settings.samples;
        cfg.depth = c->settings.depth;
        cfg.glass_always_draws = !(c->desc.flags & MI355PT_FLAG_GLASS_SHORT_CIRCUIT);
        cfg.count = (c->desc.flags & MI355PT_FLAG_COUNT) != 0;
        std::string err;
        if (!gpu::render_static(c->dev, c->camera, cfg, seeds, n_frames, c->stats, err)) {
            const bool arg = err.rfind("render_static:", 0) == 0;  // a size the workspace cannot hold
            if (!arg && c->stats.dropped_frames && c->stats.dropped_frames < n_frames) {
                // a launch gave up: the launches before it were added and count
                c->frames += (int32_t)(n_frames - c->stats.dropped_frames);
                c->seed_index += n_frames - c->stats.dropped_frames;
            }
            c->stats.dropped_frames = 0;
            return fail(c, arg ? MI355PT_ERR_ARG : MI355PT_ERR_DEVICE, err);
        }
        c->frames += (int32_t)n_frames;
        c->seed_index += n_frames;
        return MI355PT_OK;
    });
}

void mi355pt_partition_frames(uint32_t total, uint32_t parts, uint32_t part, uint32_t* first, uint32_t* count) {
    uint32_t f = 0, n = 0;
    if (parts && part < parts) {  // contiguous blocks; the first total % parts blocks hold one frame more
        const uint32_t q = total / parts, r = total % parts;
        f = part * q + (part < r ? part : r);
        n = q + (part < r ? 1u : 0u);
    }
    if (first) *first = f;
    if (count) *count = n;
}

int mi355pt_reduce(mi355pt_ctx* const* ctxs, size_t n, size_t root) {
    return guarded(nullptr, [&]() -> int {
        if (!ctxs || n == 0 || root >= n) return fail(nullptr, MI355PT_ERR_ARG, "reduce: bad arguments");
        for (size_t i = 0; i < n; i++)
            if (!ctxs[i]) {
                // readable from the root context when there is one, and from mi355pt_last_error(NULL) in any case
                if (ctxs[root]) ctxs[root]->error = "reduce: null context";
                return fail(nullptr, MI355PT_ERR_ARG, "reduce: null context");
            }
        mi355pt_ctx* r = ctxs[root];
        std::vector<gpu::Device*> devs(n);
        int64_t frames = 0;
        for (size_t i = 0; i < n; i++) {
            int rc = settle(ctxs[i]);
            if (rc != MI355PT_OK) return i == root ? rc : fail(r, rc, "reduce: context " + std::to_string(i) + ": " + ctxs[i]->error);
            if (ctxs[i]->dirty || ctxs[i]->frames < 0)
                return fail(r, MI355PT_ERR_STATE, "reduce: context " + std::to_string(i) + " holds no rendered frames (its accumulator is due to be cleared)");
            devs[i] = ctxs[i]->dev;
            frames += ctxs[i]->frames;
        }
        std::string err;
        if (!gpu::reduce_accum(devs.data(), n, root, err)) return fail(r, MI355PT_ERR_DEVICE, err);
        for (size_t i = 0; i < n; i++) {
            if (i == root) continue;
            r->stats.paths += ctxs[i]->stats.paths, r->stats.rays += ctxs[i]->stats.rays;
            ctxs[i]->dirty = true;  // its accumulator is spent: the next render starts a new accumulation
        }
        r->frames = (int32_t)frames;
        return MI355PT_OK;
    });
}

int mi355pt_reduce_comm(mi355pt_ctx* c, void* nccl_comm, int root_rank, int my_rank, int32_t total_frames) {
    return guarded(c, [&]() -> int {
        if (!c || !nccl_comm) return MI355PT_ERR_ARG;
        SETTLE_OR_RETURN(c);
        if (c->dirty || c->frames < 0) return fail(c, MI355PT_ERR_STATE, "reduce: no rendered frames (the accumulator is due to be cleared)");
        std::string err;
        if (!gpu::reduce_accum_comm(c->dev, nccl_comm, root_rank, err)) return fail(c, MI355PT_ERR_DEVICE, err);
        if (my_rank == root_rank) {
            if (!gpu::set_alpha(c->dev, 1.0f, err)) return fail(c, MI355PT_ERR_DEVICE, err);
            c->frames = total_frames;
        } else {
            c->dirty = true;
        }
        return MI355PT_OK;
    });
}

int mi355pt_wait(mi355pt_ctx* c) {
    return guarded(c, [&]() -> int {
        if (!c) return MI355PT_ERR_ARG;
        return settle(c);
    });
}

int mi355pt_render(mi355pt_ctx* c, uint32_t n_frames, const uint32_t* seeds) {
    int r = render_queued(c, n_frames, seeds, true);
    return r == MI355PT_OK ? mi355pt_wait(c) : r;
}

int32_t mi355pt_frames(const mi355pt_ctx* c) { return c ? c->frames : -1; }

int mi355pt_read_accum(mi355pt_ctx* c, float* rgba, size_t n_floats) {
    return guarded(c, [&]() -> int {
        if (!c || !rgba) return MI355PT_ERR_ARG;
        SETTLE_OR_RETURN(c);
        std::string err;
        if (!gpu::read_accum(c->dev, rgba, n_floats, err)) return fail(c, MI355PT_ERR_ARG, err);
        return MI355PT_OK;
    });
}

int mi355pt_write_accum(mi355pt_ctx* c, const float* rgba, size_t n_floats, int32_t frames) {
    return guarded(c, [&]() -> int {
        if (!c || !rgba) return MI355PT_ERR_ARG;
        SETTLE_OR_RETURN(c);
        std::string err;
        if (!gpu::write_accum(c->dev, rgba, n_floats, err)) return fail(c, MI355PT_ERR_ARG, err);
        c->frames = frames;
        c->dirty = false;
        return MI355PT_OK;
    });
}

// Not part of the interface (no declaration in mi355pt.h): the tests read the packed scene back, so that a tree
// built on the device can be checked node by node and walked by the CPU harness of the kernels' per-lane code.
// `layout` receives the pt::SceneLayout (layout_bytes must be its size); *size the blob's; the blob is copied when
// cap >= *size.
extern "C" int mi355pt_debug_scene_blob(mi355pt_ctx* c, void* buf, size_t cap, size_t* size, void* layout, size_t layout_bytes) {
    return guarded(c, [&]() -> int {
        if (!c || !size || !layout || layout_bytes != sizeof(pt::SceneLayout)) return MI355PT_ERR_ARG;
        if (!c->have_scene) return fail(c, MI355PT_ERR_STATE, "scene_blob: no scene (call set_scene)");
        SETTLE_OR_RETURN(c);
        std::string err;
        pt::SceneLayout L;
        size_t need = 0;
        if (!gpu::read_scene_blob(c->dev, buf, cap, need, L, err)) return fail(c, MI355PT_ERR_DEVICE, err);
        *size = need;
        memcpy(layout, &L, sizeof L);
        return MI355PT_OK;
    });
}

int mi355pt_to_rgb8(mi355pt_ctx* c, int32_t frames, uint8_t* rgb, size_t n_bytes) {
    return guarded(c, [&]() -> int {
        if (!c || !rgb) return MI355PT_ERR_ARG;
        if (n_bytes != (size_t)c->width * c->height * 3) return fail(c, MI355PT_ERR_ARG, "to_rgb8: n_bytes must be 3 * width * height");
        if (frames < 1) return fail(c, MI355PT_ERR_ARG, "to_rgb8: frames must be >= 1");
        SETTLE_OR_RETURN(c);
        std::vector<float> acc((size_t)c->width * c->height * 4);
        std::string err;
        if (!gpu::read_accum(c->dev, acc.data(), acc.size(), err)) return fail(c, MI355PT_ERR_DEVICE, err);
        postprocess_rgb8(acc.data(), c->width, c->height, frames, rgb);
        return MI355PT_OK;
    });
}

int mi355pt_save_png(mi355pt_ctx* c, int32_t frames, const char* path) {
    return guarded(c, [&]() -> int {
        if (!c || !path) return MI355PT_ERR_ARG;
        std::vector<uint8_t> rgb((size_t)c->width * c->height * 3);
        int r = mi355pt_to_rgb8(c, frames, rgb.data(), rgb.size());
        if (r != MI355PT_OK) return r;
        std::string err;
        if (!write_png_rgb8(path, rgb.data(), c->width, c->height, err)) return fail(c, MI355PT_ERR_IO, err);
        return MI355PT_OK;
    });
}

int mi355pt_stats_get(mi355pt_ctx* c, mi355pt_stats* out) {
    return guarded(c, [&]() -> int {
        if (!c || !out) return MI355PT_ERR_ARG;
        SETTLE_OR_RETURN(c);
        memset(out, 0, sizeof *out);
        out->paths = c->stats.paths, out->rays = c->stats.rays;
        out->nodes = c->stats.nodes, out->tests = c->stats.tests, out->accepted = c->stats.accepted;
        for (int k = 0; k < 5; k++) out->bounces[k] = c->stats.bounces[k];
        for (int k = 0; k < 12; k++) out->phase[k] = c->stats.phase[k];
        out->render_ms = c->stats.render_ms, out->resolve_ms = c->stats.resolve_ms;
        out->launches = c->stats.launches;
        out->bvh_nodes = c->have_scene ? c->layout.n_nodes : 0u;
        out->bvh_depth = c->have_scene ? c->layout.max_depth : 0u;
        out->scene_bytes = c->layout.total_bytes;
        out->scene_in_lds = c->have_scene && gpu::scene_in_lds(c->dev) ? 1u : 0u;
        out->node_format = c->have_scene ? c->layout.node_half : 0u;
        out->exchange_on = c->stats.exchange_on ? 1u : 0u;
        for (int k = 0; k < 4; k++) out->exchange[k] = c->stats.exchange[k];
        out->list_tiles = c->stats.list_tiles, out->list_candidates = c->stats.list_candidates;
        out->sky_tiles = c->stats.sky_tiles;
        out->short_div = c->stats.short_div ? 1u : 0u;
        return MI355PT_OK;
    });
}

int mi355pt_accum_device_ptr(mi355pt_ctx* c, void** ptr, void** stream) {
    if (!c) return MI355PT_ERR_ARG;
    SETTLE_OR_RETURN(c);
    if (ptr) *ptr = gpu::accum_device_ptr(c->dev);
    if (stream) *stream = gpu::stream_handle(c->dev);
    return MI355PT_OK;
}

#ifdef PT_WAVE_END_HIST
// development builds only: the raw kernel counters
int mi355pt_debug_counters(mi355pt_ctx* c, uint64_t* out, size_t n) {
    std::string err;
    return c && out && gpu::debug_counters(c->dev, reinterpret_cast<unsigned long long*>(out), n, err) ? MI355PT_OK : MI355PT_ERR_DEVICE;
}
#endif

// ---- host-side pieces ---------------------------------------------------------------------------
int mi355pt_camera_build(const float pos[3], float yaw_deg, float pitch_deg, float vfov_deg, float aspect, void* out) {
    if (!pos || !out) return MI355PT_ERR_ARG;
    Camera64 c;
    camera_build(pos, yaw_deg, pitch_deg, vfov_deg, aspect, c);
    memcpy(out, &c, 64);
    return MI355PT_OK;
}

int64_t mi355pt_bvh_build_reference(const void* spheres32, size_t n_spheres, int swap_mode, uint32_t swap_seed,
                                    void* nodes48_out, size_t cap_nodes) {
    if (!spheres32 || !nodes48_out || n_spheres > 0x7fffffffull) return MI355PT_ERR_ARG;
    if (swap_mode < 0 || swap_mode > 2) return MI355PT_ERR_ARG;
    std::vector<Node48> nodes;
    std::string err;
    if (!build_reference(static_cast<const Sphere32*>(spheres32), (uint32_t)n_spheres, swap_mode, swap_seed, nodes, err)) {
        g_create_error = err;
        return MI355PT_ERR_ARG;
    }
    if (nodes.size() > cap_nodes) return MI355PT_ERR_ARG;
    memcpy(nodes48_out, nodes.data(), nodes.size() * sizeof(Node48));
    return (int64_t)nodes.size();
}

uint32_t mi355pt_frame_seed(uint32_t base_seed, uint32_t frame) { return frame_seed(base_seed, frame); }

int mi355pt_postprocess_rgb8(const float* accum_rgba, uint32_t width, uint32_t height, int32_t frames, uint8_t* rgb) {
    return guarded(nullptr, [&]() -> int {
        if (!accum_rgba || !rgb || frames < 1) return MI355PT_ERR_ARG;
        postprocess_rgb8(accum_rgba, width, height, frames, rgb);
        return MI355PT_OK;
    });
}

int mi355pt_write_png_rgb8(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height) {
    return guarded(nullptr, [&]() -> int {
        if (!path || !rgb || !width || !height) return MI355PT_ERR_ARG;
        std::string err;
        if (!write_png_rgb8(path, rgb, width, height, err)) {
            g_create_error = err;
            return MI355PT_ERR_IO;
        }
        return MI355PT_OK;
    });
}

// ---- RON ------------------------------------------------------------------------------------------
int mi355pt_scene_parse_ron(const char* text, size_t len, mi355pt_scene** out, char* errbuf, size_t errbuf_len) {
    return guarded(nullptr, [&]() -> int {
        if (!text || !out) return MI355PT_ERR_ARG;
        *out = nullptr;
        std::unique_ptr<mi355pt_scene> s(new (std::nothrow) mi355pt_scene());
        if (!s) return MI355PT_ERR_NOMEM;
        std::string err;
        if (!parse_scene_ron(std::string(text, len), s->scene, err)) {
            copy_err(err, errbuf, errbuf_len);
            return MI355PT_ERR_IO;
        }
        *out = s.release();
        return MI355PT_OK;
    });
}

int mi355pt_scene_load_ron(const char* path, mi355pt_scene** out, char* errbuf, size_t errbuf_len) {
    return guarded(nullptr, [&]() -> int {
        if (!path || !out) return MI355PT_ERR_ARG;
        *out = nullptr;
        std::string text, err;
        if (!read_file(path, text, err)) {
            copy_err(err, errbuf, errbuf_len);
            return MI355PT_ERR_IO;
        }
        return mi355pt_scene_parse_ron(text.data(), text.size(), out, errbuf, errbuf_len);
    });
}

mi355pt_scene* mi355pt_scene_default(void) {
    mi355pt_scene* s = new (std::nothrow) mi355pt_scene();
    if (s) s->scene = default_scene();
    return s;
}

void mi355pt_scene_free(mi355pt_scene* scene) { delete scene; }

size_t mi355pt_scene_array(const mi355pt_scene* s, int which, const void** data) {
    if (!s) return 0;
    const void* p = nullptr;
    size_t n = 0;
    switch (which) {
        case 0: p = s->scene.spheres.data(), n = s->scene.spheres.size(); break;
        case 1: p = s->scene.lights.data(), n = s->scene.lights.size(); break;
        case 2: p = s->scene.lambertians.data(), n = s->scene.lambertians.size(); break;
        case 3: p = s->scene.metals.data(), n = s->scene.metals.size(); break;
        case 4: p = s->scene.glass.data(), n = s->scene.glass.size(); break;
        default: break;
    }
    if (data) *data = p;
    return n;
}

void mi355pt_scene_camera(const mi355pt_scene* s, float out6[6]) {
    if (!s || !out6) return;
    memcpy(out6, s->scene.camera.pos, 12);
    out6[3] = s->scene.camera.yaw, out6[4] = s->scene.camera.pitch, out6[5] = s->scene.camera.vfov;
}

int mi355pt_set_scene_handle(mi355pt_ctx* c, const mi355pt_scene* s) {
    return guarded(c, [&]() -> int {
        if (!c || !s) return MI355PT_ERR_ARG;
        const Scene& sc = s->scene;
        int r = mi355pt_set_scene(c, sc.spheres.data(), sc.spheres.size(), sc.lights.data(), sc.lights.size(),
                                  sc.lambertians.data(), sc.lambertians.size(), sc.metals.data(), sc.metals.size(),
                                  sc.glass.data(), sc.glass.size());
        if (r != MI355PT_OK) return r;
        return mi355pt_set_camera_config(c, sc.camera.pos, sc.camera.yaw, sc.camera.pitch, sc.camera.vfov);
    });
}

int mi355pt_settings_load_ron(const char* path, int32_t* samples, int32_t* depth, char* errbuf, size_t errbuf_len) {
    return guarded(nullptr, [&]() -> int {
        if (!path || !samples || !depth) return MI355PT_ERR_ARG;
        std::string text, err;
        Settings s;
        if (!read_file(path, text, err) || !parse_settings_ron(text, s, err)) {
            copy_err(err, errbuf, errbuf_len);
            return MI355PT_ERR_IO;
        }
        *samples = s.samples, *depth = s.depth;
        return MI355PT_OK;
    });
}

}  // extern "C"
