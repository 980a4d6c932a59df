/* mi355pt.h -- C ABI of libmi355pt.so: the reference path tracer's compute path on AMD MI355X.
 *
 * The library replaces what the reference's `Renderer` (src/renderer/mod.rs:91-320) does through
 * wgpu: it owns the scene buffers and the RGBA32F accumulator, and one call to mi355pt_render()
 * stands for N iterations of `Renderer::render` + submit + poll (src/bin/offline.rs:44-52).  All
 * inputs are the byte blobs the reference uploads (its `Bytes` trait, src/renderer/bytes.rs):
 *
 *   Sphere   32 B  pos.xyz f32 | rad f32 | mat_type u32 | mat_index u32 | 8 B pad   sphere.rs:26-43
 *   BVHNode  48 B  min.xyz, pad | max.xyz, pad | type u32 | index u32 | 8 B pad    bvh.rs:20-36
 *   Camera   64 B  pos, pad | horizontal, pad | vertical, pad | center, pad        camera.rs:91-113
 *   Light / Lambertian 16 B  rgb f32, pad;  Metal 16 B  rgb, roughness;  Glass 4 B  ir
 *                                                                                   materials.rs:26-117
 *   mat_type: 0 light, 1 lambertian, 2 metal, 3 glass (materials.rs:3-8); anything else shades black
 *   (compute.wgsl:356-358).
 *
 * Conventions: every function returns MI355PT_OK (0) or a negative error code and never throws or
 * aborts; mi355pt_last_error() gives the message of the last failure on that context.  Pointers are
 * borrowed for the duration of the call; the library copies.  A context is owned by one thread at a
 * time.  There is no CPU fallback: without a usable HIP device mi355pt_create() fails.
 */
#ifndef MI355PT_H
#define MI355PT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi355pt_ctx mi355pt_ctx;

enum {
    MI355PT_OK = 0,
    MI355PT_ERR_ARG = -1,     /* bad argument / failed validation */
    MI355PT_ERR_STATE = -2,   /* call order (e.g. render before set_scene) */
    MI355PT_ERR_DEVICE = -3,  /* HIP error, no device */
    MI355PT_ERR_IO = -4,      /* file could not be read / written / parsed */
    MI355PT_ERR_NOMEM = -5
};

enum {
    MI355PT_BVH_SAH = 0,       /* library builds a binned-SAH tree (default).  Results equal MI355PT_BVH_REFERENCE's bit for
                                  bit except where the sphere test's cancellation lets a ray "hit" a sphere whose
                                  padded box it misses (far, tiny spheres; then the result depends on which boxes a
                                  walk tests, in the reference as much as here; also paths that travel ~1000 units,
                                  e.g. inside a glass ground sphere): no pixel in the reference's scenes or at
                                  100 k spheres, 1 texel in 900 renders of random scenes a quarter of which have a glass
                                  ground sphere of radius 1000, <= 0.1 % of pixels at 1 M spheres (measured 0.04 %, horizon
                                  rows), <= 0.3 % at 10 M spheres (measured 0.23 %) */
    MI355PT_BVH_REFERENCE = 1, /* library builds the reference's own tree (bvh.rs:51-140, no child swap) and walks
                                  it the reference's way (compute.wgsl:85-96, 236-268: its box arithmetic, no
                                  pruning, no box test for a leaf): exactly the spheres the reference tries are
                                  tried, so results equal the reference's even for the rare rays whose computed
                                  root lies outside its sphere's box; several times slower than the default */
    MI355PT_BVH_LBVH = 2       /* tree built on the GPU (Morton order, leaves of one or two spheres): fastest set_scene,
                                  more node visits per ray; same results as MI355PT_BVH_SAH (equal roots of two
                                  spheres resolve by the reference's leaf order here too: it follows from the
                                  spheres' order by z, one more sort on the device) */
};

enum {
    MI355PT_FLAG_GLASS_SHORT_CIRCUIT = 1u << 0, /* `cannot_refract || ...` skips the random draw
                                                   (compute.wgsl:344); default: always draws */
    MI355PT_FLAG_COUNT = 1u << 1                /* run the instrumented kernels (visit counters) */
};

typedef struct mi355pt_desc {
    uint32_t width, height; /* Renderer::new(.., width, height)               mod.rs:106-112 */
    int32_t device;         /* HIP device ordinal */
    uint32_t base_seed;     /* seed schedule when mi355pt_render() gets seeds == NULL */
    uint32_t bvh_mode;      /* MI355PT_BVH_* used when no tree is supplied with mi355pt_set_bvh() */
    uint32_t flags;         /* MI355PT_FLAG_* */
    void* stream;           /* hipStream_t to run on, or NULL for a private stream */
} mi355pt_desc;

typedef struct mi355pt_stats {
    uint64_t paths;      /* primary rays since the last clear (pixels x frames x samples) */
    uint64_t rays;       /* closest-hit queries since the last clear */
    uint64_t nodes;      /* interior BVH nodes fetched        (MI355PT_FLAG_COUNT only) */
    uint64_t tests;      /* ray-sphere tests                  (MI355PT_FLAG_COUNT only) */
    uint64_t accepted;   /* tests that replaced the closest   (MI355PT_FLAG_COUNT only) */
    uint64_t bounces[5]; /* light, lambertian, metal, glass, sky (MI355PT_FLAG_COUNT only) */
    uint64_t phase[12];  /* per step kind (node, leaf, terminal, opaque, glass, idle): wave-level executions, lanes served
                            (MI355PT_FLAG_COUNT only) */
    double render_ms;    /* device time of the path kernels, HIP events on the render stream */
    double resolve_ms;   /* device time of the accumulation kernels */
    uint32_t launches;   /* path-kernel launches */
    uint32_t bvh_nodes;  /* interior nodes of the tree in use */
    uint32_t bvh_depth;  /* its depth */
    uint32_t scene_bytes;/* size of the device scene blob */
    uint32_t scene_in_lds; /* 1 when the blob is staged in LDS by every workgroup (small scenes; 0 under
                              MI355PT_BVH_REFERENCE / mi355pt_set_bvh, whose walk reads global memory) */
    uint32_t node_format;  /* 0: 64-byte nodes, float boxes; 1: 32-byte nodes, binary16 boxes (LDS scenes only) */
    uint32_t exchange_on;  /* 1 when the last launch ran the kernel variant that exchanges waiting paths between waves */
    uint64_t exchange[4];  /* that variant: pool deposits, lanes deposited, withdrawals, lanes withdrawn
                              (MI355PT_FLAG_COUNT only) */
    uint32_t list_tiles;      /* 8x8 pixel tiles whose camera rays tested a per-tile candidate list instead of walking
                                 the tree in the last launch (0: every camera ray walked) */
    uint32_t list_candidates; /* spheres on those lists, all tiles together */
    uint32_t sky_tiles;       /* tiles with an empty list (nothing but sky) that the last launch accumulated pixel by pixel
                                 in registers, bypassing the path kernel's queue and the per-frame result buffer */
    uint32_t short_div;       /* 1 when the camera rays' divisions by width and height run as a three-instruction sequence that the
                                 device verified, for this image size, to give the correctly rounded quotient for every
                                 possible pixel coordinate */
} mi355pt_stats;

/* ---- context: Renderer::new / drop (mod.rs:106-146) ---------------------------------------- */
int mi355pt_create(const mi355pt_desc* desc, mi355pt_ctx** out);
void mi355pt_destroy(mi355pt_ctx* ctx);
const char* mi355pt_last_error(const mi355pt_ctx* ctx); /* ctx may be NULL: last create() failure */

/* ---- scene: SceneBuffers::new (mod.rs:44-88), Renderer::reload_scene (mod.rs:148-155) --------
 * Replaces all scene buffers, rebuilds the tree and marks the accumulator dirty (the next render
 * clears it).  Rejected (MI355PT_ERR_ARG): fewer than 2 spheres (bvh.rs:61-62 panics), non-finite
 * position or radius, negative radius, a mat_index outside its material array. */
int mi355pt_set_scene(mi355pt_ctx* ctx, const void* spheres32, size_t n_spheres, const void* lights16,
                      size_t n_lights, const void* lambertians16, size_t n_lambertians, const void* metals16,
                      size_t n_metals, const void* glass4, size_t n_glass);
/* Optional: use a caller-built tree in the reference's flattened format (the bytes bound at
 * @group(1) @binding(0), compute.wgsl:184-185).  Call after mi355pt_set_scene().  The library keeps
 * the topology, re-encodes the nodes for its kernels and walks them as under MI355PT_BVH_REFERENCE. */
int mi355pt_set_bvh(mi355pt_ctx* ctx, const void* nodes48, size_t n_nodes);

/* ---- camera: the uniform at @group(0) @binding(1) (compute.wgsl:178-179) ---------------------
 * Either the 64-byte blob itself, or the reference's CameraSettings (degrees; camera.rs:3-12), from
 * which the blob is rebuilt on every resize as Renderer::render does (mod.rs:169-174).  Both mark
 * the accumulator dirty like camera_config_mut (mod.rs:254-257). */
int mi355pt_set_camera(mi355pt_ctx* ctx, const void* camera64);
int mi355pt_set_camera_config(mi355pt_ctx* ctx, const float pos[3], float yaw_deg, float pitch_deg,
                              float vfov_deg);
int mi355pt_get_camera(const mi355pt_ctx* ctx, void* camera64_out);

/* Renderer::reload_settings (mod.rs:157-160): does NOT reset the accumulator. */
int mi355pt_set_settings(mi355pt_ctx* ctx, int32_t samples, int32_t depth);
/* Renderer::resize (mod.rs:162-166): new accumulator, aspect = w / h, dirty. */
int mi355pt_resize(mi355pt_ctx* ctx, uint32_t width, uint32_t height);
/* The LoadOp::Clear of mod.rs:176-195 done now: accumulator = (0,0,0,1), frames = 0. */
int mi355pt_clear(mi355pt_ctx* ctx);

/* ---- the hot path: n_frames x { Renderer::render (mod.rs:168-244) + compute.wgsl main } --------
 * Frame f runs the shader with Globals{seed = seeds[f], samples, depth}; with seeds == NULL the
 * seed of the k-th frame since the last clear is mi355pt_frame_seed(base_seed, k).  Frames are added
 * to each texel in order.  Synchronous: the accumulator is complete on return. */
int mi355pt_render(mi355pt_ctx* ctx, uint32_t n_frames, const uint32_t* seeds);
/* The same without waiting: returns once the frames are queued on the device, so consecutive calls run
 * back to back without the host in between.  The seeds are copied before returning.  mi355pt_wait() -- and every call that reads or replaces device state
 * (read_accum, to_rgb8, save_png, stats_get, accum_device_ptr, clear, resize, set_scene, set_bvh,
 * destroy) -- first waits for everything queued; a device error of queued work is reported there.
 * No counterpart in the reference, which blocks on device.poll(Wait) after every frame
 * (offline.rs:50-52, main.rs:160-161); mi355pt_render() = mi355pt_render_async() + mi355pt_wait(). */
int mi355pt_render_async(mi355pt_ctx* ctx, uint32_t n_frames, const uint32_t* seeds);
int mi355pt_wait(mi355pt_ctx* ctx);
/* mi355pt_render() for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc): same frames, same bits, but the per-launch results, the seeds, the work-queue counter and the stack
 * spill area live in a static 160 MiB device array of the library instead of buffers the context grows on
 * demand, and the call makes no hipMalloc / hipFree / event creation.  It is an ordinary blocking call (it
 * copies the seeds up, waits for its launches and reads counters back), so it cannot be recorded into a HIP
 * graph.  The array is part of the library's code object: every device the process uses holds its 160 MiB from
 * the moment the library is loaded there, whether or not this entry point is ever called.  No counterpart in
 * the reference.  Limits (MI355PT_ERR_ARG with a message otherwise): at most 1024 frames per
 * call; one frame of results (12 bytes per pixel of the image rounded up to 8x8 tiles) plus the scene's stack
 * spill area must fit in the array -- 3840x2160 does, with any scene that fits beside it.  Launches are as long
 * as the array allows (6 frames at 1920x1080), so this is slower than mi355pt_render(); it is not timed
 * (render_ms / resolve_ms do not advance) and never re-measures the tile order.  Every device holds one copy of
 * the array, shared by all contexts on that device, and its contents are overwritten by each call: two such
 * calls that overlap on one device (from two contexts / threads) race.  Synchronous. */
int mi355pt_render_static(mi355pt_ctx* ctx, uint32_t n_frames, const uint32_t* seeds);
/* Renderer::samples (mod.rs:246-248): frames since the last clear, -1 before the first render
 * (queued frames count at once). */
int32_t mi355pt_frames(const mi355pt_ctx* ctx);

/* ---- output: Renderer::start_save + SaveInfo::finish (mod.rs:263-315, 396-435) ---------------
 * read_accum: the RGBA32F texels, tight rows (no 256-byte row padding), n_floats = 4*w*h.
 * write_accum: the inverse (restore a saved accumulator; no reference counterpart).
 * to_rgb8: rgb * (1/frames), powf(1/2.2), round(clamp(x,0,1)*255); n_bytes = 3*w*h.
 * save_png: to_rgb8 + an 8-bit RGB PNG at `path` (the reference always writes ./img.png). */
int mi355pt_read_accum(mi355pt_ctx* ctx, float* rgba, size_t n_floats);
int mi355pt_write_accum(mi355pt_ctx* ctx, const float* rgba, size_t n_floats, int32_t frames);
int mi355pt_to_rgb8(mi355pt_ctx* ctx, int32_t frames, uint8_t* rgb, size_t n_bytes);
int mi355pt_save_png(mi355pt_ctx* ctx, int32_t frames, const char* path);

/* Counters and device timings since the last clear (no reference counterpart; the reference only
 * prints wall seconds, offline.rs:83). */
int mi355pt_stats_get(mi355pt_ctx* ctx, mi355pt_stats* out);
/* Device pointer of the accumulator (float4[w*h]) and the stream it is written on, for callers
 * that combine accumulators across GPUs with a collective (no reference counterpart). */
int mi355pt_accum_device_ptr(mi355pt_ctx* ctx, void** ptr, void** stream);

/* ---- several GPUs: frames are independent (one seed each, compute.wgsl:400), so a render splits by frame index
 * over any number of contexts, one per device, each with the same scene, camera, settings and size; their
 * accumulators differ from a one-device render of all frames only in the order of the float additions.  No
 * counterpart in the reference (one device, offline.rs:44-52).
 * partition_frames: the one split used everywhere (CLI, bench, tests): contiguous blocks, the first
 *   total % parts of them one frame longer.
 * reduce: one process driving n contexts.  Joins them, then sums every accumulator into ctxs[root]'s: contexts
 *   sharing a device are added by a kernel, one accumulator per device goes through ONE ncclReduce(sum) of
 *   4*w*h floats over xGMI (RCCL, opened at run time; communicators are made with ncclCommInitAll and kept).
 *   Afterwards ctxs[root] counts the frames of all (mi355pt_frames, to_rgb8, save_png see the whole image);
 *   the other contexts are spent: their next render starts a new accumulation.
 * reduce_comm: the same step for a host that runs one process per GPU and owns an ncclComm_t (or a communicator
 *   of a framework built on RCCL): this context's part of the reduce to rank root_rank.  total_frames = frames
 *   of all ranks, recorded on the root. */
void mi355pt_partition_frames(uint32_t total, uint32_t parts, uint32_t part, uint32_t* first, uint32_t* count);
int mi355pt_reduce(mi355pt_ctx* const* ctxs, size_t n, size_t root);
int mi355pt_reduce_comm(mi355pt_ctx* ctx, void* nccl_comm, int root_rank, int my_rank, int32_t total_frames);

/* ---- host-side pieces of the path, usable without a context or a GPU --------------------------- */
/* CameraConfig::new(settings, aspect).build().bytes()                     camera.rs:36-57,71-113 */
int mi355pt_camera_build(const float pos[3], float yaw_deg, float pitch_deg, float vfov_deg, float aspect,
                         void* camera64_out);
/* flatten(BVHTree::new(spheres)).bytes()  (bvh.rs:51-140).  swap_mode 0: never swap children,
 * 1: always, 2: one xorshift32 bit per node with >= 4 spheres (where the reference calls
 * rand::random()).  Writes 2n-1 nodes; returns the node count or a negative error. */
int64_t mi355pt_bvh_build_reference(const void* spheres32, size_t n_spheres, int swap_mode, uint32_t swap_seed,
                                    void* nodes48_out, size_t cap_nodes);
uint32_t mi355pt_frame_seed(uint32_t base_seed, uint32_t frame);
/* SaveInfo::finish's pixel math on a host buffer, and the PNG encoder. */
int mi355pt_postprocess_rgb8(const float* accum_rgba, uint32_t width, uint32_t height, int32_t frames,
                             uint8_t* rgb_out);
int mi355pt_write_png_rgb8(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height);

/* ---- RON files: load_ron (lib.rs:6-14) for Scene (scene.rs:5-14) and Settings (mod.rs:26-30) -----
 * A scene handle owns the parsed arrays in wire format.  mi355pt_scene_default() is Scene::default()
 * (scene.rs:15-31), what offline.rs:37 falls back to when the file cannot be loaded. */
typedef struct mi355pt_scene mi355pt_scene;
int mi355pt_scene_load_ron(const char* path, mi355pt_scene** out, char* errbuf, size_t errbuf_len);
int mi355pt_scene_parse_ron(const char* text, size_t len, mi355pt_scene** out, char* errbuf, size_t errbuf_len);
mi355pt_scene* mi355pt_scene_default(void);
void mi355pt_scene_free(mi355pt_scene* scene);
/* which: 0 spheres(32 B) 1 lights 2 lambertians 3 metals (16 B) 4 glass (4 B).  Returns the element
 * count and points *data at the library-owned array (valid until mi355pt_scene_free). */
size_t mi355pt_scene_array(const mi355pt_scene* scene, int which, const void** data);
/* camera: pos[3], yaw, pitch, vfov in degrees */
void mi355pt_scene_camera(const mi355pt_scene* scene, float out6[6]);
/* set_scene + set_camera_config from a handle */
int mi355pt_set_scene_handle(mi355pt_ctx* ctx, const mi355pt_scene* scene);
int mi355pt_settings_load_ron(const char* path, int32_t* samples, int32_t* depth, char* errbuf,
                              size_t errbuf_len);

#ifdef __cplusplus
}
#endif
#endif /* MI355PT_H */
