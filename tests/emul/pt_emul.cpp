// Test harness only: runs the kernels' per-lane code (pathtracer_amd/csrc/pt_device.h) and the
// library's host-side scene packing on the CPU, one path after another, so that tree building,
// traversal and scatter logic can be compared with the oracle where no GPU exists.  It is built by
// tests/conftest.py into tests/emul/ and is not part of libmi355pt.so.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../pathtracer_amd/csrc/pt_device.h"
#include "../../pathtracer_amd/csrc/scene_pack.h"

using namespace pt;

namespace {
struct HostStack {
    int32_t a[80];
    void push(int sp, int32_t v) { a[sp] = v; }
    int32_t pop(int sp) const { return a[sp]; }
};
std::string g_err;
int g_brute_force = 0;
int g_tile_lists = 1;         // camera rays test their tile's candidate list (the device's default) instead of walking
uint32_t g_tile_list_max = TILE_LIST_CAP;
uint64_t g_list_rays = 0, g_list_tests = 0, g_walk_tiles = 0, g_list_tiles = 0;
bool g_log = true;
const bool g_trace = getenv("PT_EMUL_TRACE") != nullptr;  // every ray of every path to stderr (debugging aid)
uint32_t g_last_hoisted = 0;  // SceneLayout::n_hoisted of the last emul_render
std::vector<char> g_blob;   // a packed scene read back from the device (emul_set_blob): rendered instead of the harness's own packing
SceneLayout g_blob_layout;
}  // namespace

extern "C" {

const char* emul_error() { return g_err.c_str(); }
// A packed scene as the device holds it (tree included) and its layout: emul_render then walks that one, whatever
// arrays it is given (n = 0: back to the harness's own packing).  Lets a tree built on the GPU be walked on the CPU.
int emul_set_blob(const void* blob, size_t n, const void* layout, size_t layout_bytes) {
    g_blob.clear();
    if (n == 0) return 0;
    if (layout_bytes != sizeof(SceneLayout)) return -1;
    memcpy(&g_blob_layout, layout, sizeof g_blob_layout);
    if (g_blob_layout.total_bytes != n) return -1;
    g_blob.assign((const char*)blob, (const char*)blob + n);
    return 0;
}
// Structure of the tree in the blob set by emul_set_blob: every interior node reached exactly once from node 0, every
// sphere in exactly one leaf, every child box inside its parent's box for it, every sphere's padded box
// (bounding_box.rs:17-27; the radius is the root of the stored square, hence the small tolerance) inside its leaf's
// box, no chain of interior nodes longer than the layout says (it sizes the traversal stack).  0 = sound; a positive code names the failed check.
int emul_check_blob_tree() {
    if (g_blob.empty()) return -1;
    const SceneLayout& L = g_blob_layout;
    if (L.node_half) return -2;
    const Node64* nodes = reinterpret_cast<const Node64*>(g_blob.data() + L.off_nodes);
    const f4* spheres = reinterpret_cast<const f4*>(g_blob.data() + L.off_spheres);
    std::vector<uint8_t> seen_node(L.n_nodes, 0), seen_sphere(L.n_spheres, 0);
    struct Todo { int32_t link; uint32_t depth; float lo[3], hi[3]; };
    std::vector<Todo> todo;
    auto child_box = [&](const Node64& nd, int k, float* lo, float* hi) {
        const float* q = reinterpret_cast<const float*>(&nd);
        for (int a = 0; a < 3; a++) lo[a] = q[k * 6 + a], hi[a] = q[k * 6 + 3 + a];
    };
    Todo root{0, 1u, {-INFINITY, -INFINITY, -INFINITY}, {INFINITY, INFINITY, INFINITY}};
    todo.push_back(root);
    uint32_t deepest = 0;
    while (!todo.empty()) {
        const Todo t = todo.back();
        todo.pop_back();
        if (t.link >= 0) {
            if ((uint32_t)t.link >= L.n_nodes || seen_node[t.link]++) return 3;
            deepest = std::max(deepest, t.depth);
            const Node64& nd = nodes[t.link];
            const int32_t c[2] = {nd.child0, nd.child1};
            for (int k = 0; k < 2; k++) {
                Todo n{c[k], t.depth + 1u, {0, 0, 0}, {0, 0, 0}};
                child_box(nd, k, n.lo, n.hi);
                for (int a = 0; a < 3; a++)
                    if (!(n.lo[a] >= t.lo[a] && n.hi[a] <= t.hi[a] && n.lo[a] <= n.hi[a])) return 4;
                todo.push_back(n);
            }
        } else {
            const uint32_t ref = ~(uint32_t)t.link, first = ref & LEAF_FIRST_MASK, count = (ref >> LEAF_COUNT_SHIFT) + 1u;
            for (uint32_t k = 0; k < count; k++) {
                if (first + k >= L.n_spheres || seen_sphere[first + k]++) return 6;
                const f4 s = spheres[first + k];
                const float r = sqrtf(s.w), c[3] = {s.x, s.y, s.z};
                for (int a = 0; a < 3; a++) {
                    const float tol = 4e-7f * (fabsf(c[a]) + r + 0.01f);
                    if (!((c[a] - r) - 0.01f >= t.lo[a] - tol && (c[a] + r) + 0.01f <= t.hi[a] + tol)) return 5;
                }
            }
        }
    }
    if (L.n_hoisted) seen_sphere[L.n_spheres - 1u]++;  // in no leaf on purpose
    for (uint8_t v : seen_node) if (v != 1) return 7;
    for (uint8_t v : seen_sphere) if (v != 1) return 8;
    if (deepest > L.max_depth) return 9;
    return 0;
}
// 1: closest hits are found by testing every sphere (no tree, no boxes); 2: the same with each sphere's own
// padded box as the only cull, logging every root that box would hide
void emul_set_brute_force(int mode) { g_brute_force = mode; }
// on = 0: camera rays walk the tree like every other ray; max_list: longest list a tile may have before its rays walk
void emul_set_tile_lists(int on, uint32_t max_list) { g_tile_lists = on, g_tile_list_max = max_list < TILE_LIST_CAP ? max_list : TILE_LIST_CAP; }
// camera rays served by a list, sphere tests they made, tiles that keep the walk, tiles with a list (last render)
void emul_tile_list_stats(uint64_t* out) { out[0] = g_list_rays, out[1] = g_list_tests, out[2] = g_walk_tiles, out[3] = g_list_tiles; }

// counters: nodes, tests, accepted, bounce[5], rays, paths
int emul_render(const void* spheres32, size_t n_spheres, const void* lights, size_t n_lights, const void* lamb,
                size_t n_lamb, const void* metals, size_t n_metals, const void* glass, size_t n_glass,
                int reference_topology, const void* ref_nodes, size_t n_ref_nodes, const void* cam64, uint32_t w,
                uint32_t h, int32_t samples, int32_t depth, int glass_always_draws, uint32_t n_frames,
                const uint32_t* seeds, float* accum, uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1,
                uint64_t* counters, uint32_t* bvh_info) {
    SceneArrays a;
    a.spheres = (const Sphere32*)spheres32, a.n_spheres = n_spheres;
    a.lights = (const Vec16*)lights, a.n_lights = n_lights;
    a.lambertians = (const Vec16*)lamb, a.n_lambertians = n_lamb;
    a.metals = (const Vec16*)metals, a.n_metals = n_metals;
    a.glass = (const float*)glass, a.n_glass = n_glass;
    std::vector<char> blob;
    SceneLayout L;
    Bvh2 bvh;
    if (!g_blob.empty()) {
        blob = g_blob, L = g_blob_layout;
    } else if (!pack_scene(a, reference_topology != 0, (const Node48*)ref_nodes, n_ref_nodes, blob, L, bvh, g_err)) {
        return -1;
    }
    g_last_hoisted = L.n_hoisted;
    if (bvh_info) bvh_info[0] = (uint32_t)bvh.nodes.size(), bvh_info[1] = bvh.max_depth, bvh_info[2] = L.total_bytes, bvh_info[3] = L.node_half;
    PtrMem mem{blob.data()};
    // A float-node scene that the device would stage in LDS is walked there through direction-ordered
    // records; the harness converts the same way and walks the same code.
    const bool ordered = !L.node_half && !L.reference_walk && lds_scene_bytes(L) + (size_t)std::max(1u, L.max_depth) * 256u * 4u <= LDS_SCENE_LIMIT;
    std::vector<char> ordered_nodes;
    PtrOrderedMem omem;
    omem.base = blob.data();
    if (ordered) {
        ordered_nodes.resize((size_t)L.n_nodes * LDS_NODE_BYTES);
        for (uint32_t i = 0; i < L.n_nodes; i++) {
            uint32_t q[16], rec[LDS_NODE_BYTES / 4u] = {0};
            memcpy(q, blob.data() + L.off_nodes + (size_t)i * 64u, 64);
            const uint32_t l0 = (int32_t)q[12] >= 0 ? q[12] * LDS_NODE_BYTES : q[12];
            const uint32_t l1 = (int32_t)q[13] >= 0 ? q[13] * LDS_NODE_BYTES : q[13];
            node_to_ordered(q, l0, l1, rec);
            memcpy(ordered_nodes.data() + (size_t)i * LDS_NODE_BYTES, rec, LDS_NODE_BYTES);
        }
        omem.nodes = ordered_nodes.data();
    }
    // A float-node tree that stays in global memory is walked through the same records, 80 bytes apart, links as
    // byte offsets (the device builds them when it adopts the scene; MI355PT_NODES=plain walks the Node64 records).
    const char* nodes_env = getenv("MI355PT_NODES");
    const bool ordered_global = !ordered && !L.node_half && !L.reference_walk && !(nodes_env && strcmp(nodes_env, "plain") == 0) &&
                                (uint64_t)L.n_nodes * 80u < (1ull << 31);
    if (ordered_global) {
        ordered_nodes.resize((size_t)L.n_nodes * 80u);
        for (uint32_t i = 0; i < L.n_nodes; i++) {
            uint32_t q[16], rec[20];
            memcpy(q, blob.data() + L.off_nodes + (size_t)i * 64u, 64);
            const uint32_t l0 = (int32_t)q[12] >= 0 ? q[12] * 80u : q[12];
            const uint32_t l1 = (int32_t)q[13] >= 0 ? q[13] * 80u : q[13];
            node_to_ordered(q, l0, l1, rec);
            memcpy(ordered_nodes.data() + (size_t)i * 80u, rec, 80);
        }
        omem.nodes = ordered_nodes.data();
    }
    if (bvh_info && ordered) bvh_info[3] = 2;
    if (bvh_info && L.reference_walk) bvh_info[3] = 3;
    if (bvh_info && ordered_global) bvh_info[3] = 4;
    Camera64 cam64v;
    memcpy(&cam64v, cam64, 64);
    CameraRegs cam = load_camera(cam64v);
    Counters cnt = {0, 0, 0};
    uint64_t big[10] = {0};
    const float fw = (float)w, fh = (float)h;
    // per-tile candidate lists for camera rays, built on first use with the code the device's list kernel runs
    const bool use_lists = g_tile_lists && !g_brute_force && !L.reference_walk;
    const uint32_t tiles_x = (w + 7u) / 8u, tiles_y = (h + 7u) / 8u;
    std::vector<uint32_t> lists;
    std::vector<uint8_t> have_list;
    if (use_lists) lists.resize((size_t)tiles_x * tiles_y * TILE_LIST_WORDS), have_list.assign((size_t)tiles_x * tiles_y, 0);
    g_list_rays = g_list_tests = g_walk_tiles = g_list_tiles = 0;
    for (uint32_t y = y0; y < y1; y++)
        for (uint32_t x = x0; x < x1; x++) {
            float* texel = accum + 4ull * ((uint64_t)y * w + x);
            const uint32_t* rec = nullptr;
            if (use_lists) {
                const size_t tile = (size_t)(y >> 3) * tiles_x + (x >> 3);
                uint32_t* r = lists.data() + tile * TILE_LIST_WORDS;
                if (!have_list[tile]) {
                    tile_list_build(mem, L, cam64v, x >> 3, y >> 3, w, h, g_tile_list_max, r);
                    have_list[tile] = 1;
                    (r[0] == TILE_LIST_WALK ? g_walk_tiles : g_list_tiles)++;
                }
                if (r[0] != TILE_LIST_WALK) rec = r;
            }
            const uint32_t hash = pixel_hash(x, y, w, h);
            for (uint32_t f = 0; f < n_frames; f++) {
                PathState ps;
                ps.rng = pixel_seed(seeds[f], hash);
                V3 final_colour = mk(0.0f, 0.0f, 0.0f);
                for (int32_t s = 0; s < samples; s++) {
                    ps.d = camera_dir(cam, x, y, fw, fh, ps.rng);
                    ps.o = cam.pos;
                    ps.colour = mk(1.0f, 1.0f, 1.0f);
                    ps.bounce = 0;
                    big[9]++;
                    for (;;) {
                        Hit hit;
                        HostStack st;
                        cnt = Counters{0, 0, 0};
                        if (rec && ps.bounce == 0) {  // camera ray: its tile's candidates, no tree
                            Walk w;
                            walk_begin<false>(w, L, ps.o, ps.d);
                            walk_tile_list<PtrMem, true>(mem, L, w, ps.o, ps.d, rec, cnt);
                            walk_result(w, hit);
                            g_list_rays++, g_list_tests += cnt.tests;
                        } else if (g_brute_force) {  // every sphere, no boxes: what no tree can change
                            Walk w;
                            walk_begin<false>(w, L, ps.o, ps.d);
                            for (uint32_t si = 0; si < L.n_spheres; si++) {
                                if (g_brute_force == 2 && g_blob.empty()) {  // with every sphere's own padded box as the only cull
                                    f4 sp = mem.load<f4>(L.off_spheres + si * 16u);
                                    sp.w = a.spheres[bvh.order[si]].rad;  // the blob holds rad * rad
                                    float tn;
                                    const bool in = slab((sp.x - sp.w) - 0.01f, (sp.y - sp.w) - 0.01f, (sp.z - sp.w) - 0.01f,
                                                         (sp.x + sp.w) + 0.01f, (sp.y + sp.w) + 0.01f, (sp.z + sp.w) + 0.01f, w.inv,
                                                         w.ood, w.best_t, tn);
                                    if (!in) {
                                        Walk probe = w;
                                        probe.best_t = INFINITY, probe.best_code = -1;
                                        walk_test_sphere<PtrMem, false>(mem, L, probe, ps.o, ps.d, si, cnt);
                                        if (probe.best_code >= 0 && probe.best_t < w.best_t && g_log)
                                            fprintf(stderr, "pixel %u,%u frame %u bounce %d: sphere %u (r %g) root t=%.9g culled by its own box; o=(%.9g %.9g %.9g) d=(%.9g %.9g %.9g) best_t=%.9g\n",
                                                    x, y, f, ps.bounce, si, sp.w, probe.best_t, ps.o.x, ps.o.y, ps.o.z, ps.d.x, ps.d.y, ps.d.z, w.best_t);
                                        continue;
                                    }
                                }
                                walk_test_sphere<PtrMem, true>(mem, L, w, ps.o, ps.d, si, cnt);
                            }
                            walk_result(w, hit);
                        } else if (L.reference_walk) closest_hit<PtrMem, HostStack, true, false, true>(mem, L, st, ps.o, ps.d, hit, cnt);
                        else if (ordered || ordered_global) closest_hit<PtrOrderedMem, HostStack, true, false>(omem, L, st, ps.o, ps.d, hit, cnt);
                        else if (L.node_half) closest_hit<PtrMem, HostStack, true, true>(mem, L, st, ps.o, ps.d, hit, cnt);
                        else closest_hit<PtrMem, HostStack, true, false>(mem, L, st, ps.o, ps.d, hit, cnt);
                        big[0] += cnt.nodes, big[1] += cnt.tests, big[2] += cnt.accepted, big[8]++;
                        if (g_trace)
                            fprintf(stderr, "pixel %u,%u frame %u sample %d bounce %d: o=(%.9g %.9g %.9g) d=(%.9g %.9g %.9g) t=%.9g sphere(leaf order)=%d back=%d\n", x, y,
                                    f, s, ps.bounce, ps.o.x, ps.o.y, ps.o.z, ps.d.x, ps.d.y, ps.d.z, hit.t, hit.idx, (int)hit.back);
                        V3 radiance;
                        uint32_t mat;
                        bool done = scatter(mem, L, hit, ps, depth, glass_always_draws != 0, radiance, mat);
                        if (mat < 5) big[3 + mat]++;
                        if (done) {
                            final_colour = final_colour + radiance;
                            break;
                        }
                    }
                }
                const float fs = (float)samples;
                texel[0] += final_colour.x / fs;
                texel[1] += final_colour.y / fs;
                texel[2] += final_colour.z / fs;
                texel[3] += 0.0f;
            }
        }
    if (counters) memcpy(counters, big, sizeof big);
    return 0;
}

// Directed binary16 rounding of scene_pack: lo[i] <= x[i] <= hi[i] as floats (inf allowed).
void emul_half_bounds(const double* x, size_t n, float* lo, float* hi) {
    for (size_t i = 0; i < n; i++) {
        lo[i] = half_bits_to_float(half_floor(x[i]));
        hi[i] = half_bits_to_float(half_ceil(x[i]));
    }
}

// SAH tree over the spheres, renumbered with order_top_first(k_top); returns 0 when the result is
// the same tree (every sphere in exactly one leaf, every interior node reachable exactly once from
// node 0, same depth), parents precede children inside the top prefix, and areas do not grow along
// the prefix faster than the priority order allows; a positive code names the failed check.
int emul_check_top_first(const void* spheres32, size_t n, uint32_t k_top) {
    Bvh2 bvh;
    if (!build_sah((const Sphere32*)spheres32, (uint32_t)n, bvh, g_err)) return 1;
    const uint32_t depth_before = bvh.max_depth;
    const size_t nodes_before = bvh.nodes.size();
    order_top_first(bvh, k_top);
    if (bvh.nodes.size() != nodes_before || bvh.max_depth != depth_before) return 2;
    std::vector<uint8_t> seen_node(bvh.nodes.size(), 0), seen_sphere(n, 0);
    std::vector<std::pair<int32_t, uint32_t>> todo{{0, 1u}};
    uint32_t deepest = 0;
    while (!todo.empty()) {
        auto [link, depth] = todo.back();
        todo.pop_back();
        if (link >= 0) {
            if ((size_t)link >= bvh.nodes.size() || seen_node[link]++) return 3;
            deepest = std::max(deepest, depth);
            const Node64& nd = bvh.nodes[link];
            const int32_t c[2] = {nd.child0, nd.child1};
            for (int k = 0; k < 2; k++) {
                if (c[k] >= 0 && (uint32_t)link < k_top && (uint32_t)c[k] < k_top && c[k] <= link) return 4;
                if (c[k] >= 0 && (uint32_t)link >= k_top && (uint32_t)c[k] < k_top) return 5;  // prefix is closed upwards
                todo.push_back({c[k], depth + 1});
            }
        } else {
            const uint32_t ref = ~(uint32_t)link, first = ref & LEAF_FIRST_MASK, count = (ref >> LEAF_COUNT_SHIFT) + 1u;
            for (uint32_t k = 0; k < count; k++) {
                if (first + k >= n || seen_sphere[bvh.order[first + k]]++) return 6;
            }
        }
    }
    if (bvh.n_hoisted) seen_sphere[bvh.order.back()]++;  // in no leaf on purpose
    for (uint8_t v : seen_node) if (v != 1) return 7;
    for (uint8_t v : seen_sphere) if (v != 1) return 8;
    if (deepest != depth_before) return 9;
    return 0;
}

// Candidate lists of every 8x8 tile (tile_list_build, the code of the device's list kernel) over the SAH tree of
// the scene: out[tiles][TILE_LIST_WORDS]; order_out (n_spheres, optional) receives the caller's sphere index of
// every leaf-order index the lists use.
int emul_tile_lists(const void* spheres32, size_t n_spheres, const void* lights, size_t n_lights, const void* lamb,
                    size_t n_lamb, const void* metals, size_t n_metals, const void* glass, size_t n_glass, const void* cam64,
                    uint32_t w, uint32_t h, uint32_t max_list, uint32_t* out, uint32_t* order_out) {
    SceneArrays a;
    a.spheres = (const Sphere32*)spheres32, a.n_spheres = n_spheres;
    a.lights = (const Vec16*)lights, a.n_lights = n_lights;
    a.lambertians = (const Vec16*)lamb, a.n_lambertians = n_lamb;
    a.metals = (const Vec16*)metals, a.n_metals = n_metals;
    a.glass = (const float*)glass, a.n_glass = n_glass;
    std::vector<char> blob;
    SceneLayout L;
    Bvh2 bvh;
    if (!pack_scene(a, false, nullptr, 0, blob, L, bvh, g_err)) return -1;
    PtrMem mem{blob.data()};
    Camera64 cam;
    memcpy(&cam, cam64, 64);
    const uint32_t tiles_x = (w + 7u) / 8u, tiles_y = (h + 7u) / 8u;
    for (uint32_t ty = 0; ty < tiles_y; ty++)
        for (uint32_t tx = 0; tx < tiles_x; tx++) {
            uint32_t* rec = out + ((size_t)ty * tiles_x + tx) * TILE_LIST_WORDS;
            for (uint32_t k = 0; k < TILE_LIST_WORDS; k++) rec[k] = 0u;
            tile_list_build(mem, L, cam, tx, ty, w, h, max_list, rec);
        }
    if (order_out)
        for (size_t i = 0; i < n_spheres; i++) order_out[i] = bvh.order[i];
    return 0;
}

uint32_t emul_last_hoisted() { return g_last_hoisted; }
uint32_t emul_reference_leaf_rank(uint32_t n, uint32_t pos) { return reference_leaf_rank(n, pos); }
uint32_t emul_pixel_hash(uint32_t px, uint32_t py, uint32_t w, uint32_t h) { return pixel_hash(px, py, w, h); }
float emul_det_sin(float x) { return det_sin(x); }

}  // extern "C"
