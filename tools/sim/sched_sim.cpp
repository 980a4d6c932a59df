// Wave-scheduling model: replays the real step sequences of paths (traced with the kernels' own
// per-lane code on the CPU) through a 64-lane wave under different scheduling policies and counts
// wave-level step executions, weighted by the instruction cost of each step kind.  Development tool
// for choosing the kernel's scheduling; not part of the library.
//   g++ -std=c++17 -O2 -ffp-contract=off -I../../pathtracer_amd/csrc -o sched_sim sched_sim.cpp \
//       ../../pathtracer_amd/csrc/scene_pack.cpp ../../pathtracer_amd/csrc/bvh.cpp ../../pathtracer_amd/csrc/scene_io.cpp -lz
//   ./sched_sim scene.ron W H
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "pt_device.h"
#include "scene_io.h"
#include "scene_pack.h"

using namespace pt;

struct HostStack {
    int32_t a[96];
    void push(int sp, int32_t v) { a[sp] = v; }
    int32_t pop(int sp) const { return a[sp]; }
};

struct Segment {
    std::string steps;  // 'N' interior node step, 'L' one-sphere leaf step
    char cls;           // 'T' terminal, 'O' opaque, 'G' glass
    bool ends;          // the path ends in this segment's scatter
};
typedef std::vector<Segment> Path;

enum Kind { K_NODE, K_LEAF, K_OPAQUE, K_GLASS, K_RESTART, K_RETIRE, K_REFILL, K_COUNT };
static const char* kind_name[K_COUNT] = {"node", "leaf", "opaque", "glass", "restart", "retire", "refill"};
// vector instructions per wave-level execution of a step (from the kernel's ISA)
static double cost[K_COUNT] = {39, 75, 330, 300, 130, 25, 30};

struct Tally {
    double exec[K_COUNT] = {0}, lanes[K_COUNT] = {0};
    void add(int k, int n) { exec[k] += 1, lanes[k] += n; }
    double total() const {
        double t = 0;
        for (int k = 0; k < K_COUNT; k++) t += exec[k] * cost[k];
        return t;
    }
};

struct Lane {
    int path = -1, seg = 0, pos = 0;
    char st = 'I';  // N, L, F (walk finished), O, G, T, I (idle), X (exit), E (empty, pool policy)
};

static char next_state(const Path& p, int seg, int pos) {
    const Segment& s = p[seg];
    if (pos < (int)s.steps.size()) return s.steps[pos];
    return 'F';
}

// ---- policy A: the kernel's current scheduling (every lane owns one path through all its stages)
static void run_lane_owner(const std::vector<Path>& paths, size_t first, size_t last, const int thr[5], Tally& tl) {
    Lane ln[64];
    size_t next = first;
    bool force = false;
    auto count = [&](char c) { int n = 0; for (auto& l : ln) n += l.st == c; return n; };
    auto advance_walk = [&](Lane& l) {
        l.pos++;
        l.st = next_state(paths[l.path], l.seg, l.pos);
        if (l.st == 'F') l.st = paths[l.path][l.seg].cls;  // classification is folded into the loop
    };
    for (;;) {
        bool ran = false;
        int n = count('N');
        if (n >= thr[0] || (force && n)) {
            ran = true;
            for (;;) {
                tl.add(K_NODE, n);
                for (auto& l : ln) if (l.st == 'N') advance_walk(l);
                n = count('N');
                if (n < thr[0]) break;
            }
        }
        n = count('L');
        if (n >= thr[1] || (force && n)) {
            ran = true;
            for (;;) {
                tl.add(K_LEAF, n);
                for (auto& l : ln) if (l.st == 'L') advance_walk(l);
                n = count('L');
                if (n < thr[1]) break;
            }
        }
        auto scatter = [&](char c, int kind, int th) {
            int m = count(c);
            if (m >= th || (force && m)) {
                ran = true;
                tl.add(kind, m);
                for (auto& l : ln) if (l.st == c) {
                    const Segment& s = paths[l.path][l.seg];
                    if (s.ends) l.st = 'I', l.path = -1;
                    else l.seg++, l.pos = 0, l.st = next_state(paths[l.path], l.seg, 0);
                }
            }
        };
        scatter('O', K_OPAQUE, thr[2]);
        scatter('G', K_GLASS, thr[3]);
        n = count('I') + count('T');
        if (n >= thr[4] || (force && n)) {
            ran = true;
            tl.add(K_RESTART, n);
            for (auto& l : ln) {
                if (l.st == 'T') l.st = 'I', l.path = -1;
                if (l.st == 'I') {
                    if (next < last) l.path = (int)next++, l.seg = 0, l.pos = 0, l.st = next_state(paths[l.path], 0, 0);
                    else l.st = 'X';
                }
            }
        }
        if (!ran) {
            if (count('X') == 64) break;
            force = true;
        } else force = false;
    }
}

// ---- policy C: one step at a time, the kind chosen by score = population * weight[kind]; a kind
// needs at least `floor_` lanes unless nothing else can run.  weight > 1 favours cheap steps.
static void run_greedy(const std::vector<Path>& paths, size_t first, size_t last, const double weight[5], int floor_,
                       Tally& tl) {
    Lane ln[64];
    size_t next = first;
    auto count = [&](char c) { int n = 0; for (auto& l : ln) n += l.st == c; return n; };
    auto advance_walk = [&](Lane& l) {
        l.pos++;
        l.st = next_state(paths[l.path], l.seg, l.pos);
        if (l.st == 'F') l.st = paths[l.path][l.seg].cls;
    };
    for (;;) {
        int pop[5] = {count('N'), count('L'), count('O'), count('G'), count('I') + count('T')};
        if (count('X') == 64) break;
        int best = -1;
        double best_score = 0;
        for (int pass = 0; pass < 2 && best < 0; pass++)
            for (int k = 0; k < 5; k++) {
                if (!pop[k] || (pass == 0 && pop[k] < floor_)) continue;
                const double sc = pop[k] * weight[k];
                if (sc > best_score) best_score = sc, best = k;
            }
        if (best < 0) break;
        if (best == 0) { tl.add(K_NODE, pop[0]); for (auto& l : ln) if (l.st == 'N') advance_walk(l); }
        else if (best == 1) { tl.add(K_LEAF, pop[1]); for (auto& l : ln) if (l.st == 'L') advance_walk(l); }
        else if (best == 2 || best == 3) {
            const char c = best == 2 ? 'O' : 'G';
            tl.add(best == 2 ? K_OPAQUE : K_GLASS, pop[best]);
            for (auto& l : ln) if (l.st == c) {
                const Segment& s = paths[l.path][l.seg];
                if (s.ends) l.st = 'I', l.path = -1;
                else l.seg++, l.pos = 0, l.st = next_state(paths[l.path], l.seg, 0);
            }
        } else {
            tl.add(K_RESTART, pop[4]);
            for (auto& l : ln) {
                if (l.st == 'T') l.st = 'I', l.path = -1;
                if (l.st == 'I') {
                    if (next < last) l.path = (int)next++, l.seg = 0, l.pos = 0, l.st = next_state(paths[l.path], 0, 0);
                    else l.st = 'X';
                }
            }
        }
    }
}

// ---- policy B: lanes only walk; paths waiting for a scatter step are parked in a per-wave pool
struct Parked { int path, seg; };
static void run_pool(const std::vector<Path>& paths, size_t first, size_t last, const int thr[8], int cap, Tally& tl,
                     double* max_pool) {
    // thr: node, leaf, opaque queue, glass queue, empties before a restart, finished before a retire, empties before a refill
    Lane ln[64];
    for (auto& l : ln) l.st = 'E';
    std::deque<Parked> ray, qo, qg;
    size_t next = first;
    bool force = false;
    auto count = [&](char c) { int n = 0; for (auto& l : ln) n += l.st == c; return n; };
    auto pooled = [&]() { return (int)(ray.size() + qo.size() + qg.size()); };
    auto advance_walk = [&](Lane& l) {
        l.pos++;
        l.st = next_state(paths[l.path], l.seg, l.pos);
    };
    for (;;) {
        bool ran = false;
        // refill empty lanes from the ray queue
        int e = count('E');
        if (e && !ray.empty() && (e >= thr[6] || force || count('N') + count('L') < thr[0])) {
            int m = 0;
            for (auto& l : ln) if (l.st == 'E' && !ray.empty()) {
                Parked p = ray.front(); ray.pop_front();
                l.path = p.path, l.seg = p.seg, l.pos = 0, l.st = next_state(paths[l.path], l.seg, 0), m++;
            }
            tl.add(K_REFILL, m);
            ran = true;
        }
        int n = count('N');
        if (n >= thr[0] || (force && n)) {
            ran = true;
            for (;;) {
                tl.add(K_NODE, n);
                for (auto& l : ln) if (l.st == 'N') advance_walk(l);
                n = count('N');
                if (n < thr[0]) break;
                if (count('E') + count('F') >= thr[6] && (!ray.empty() || count('F'))) break;  // go and refill
            }
        }
        n = count('L');
        if (n >= thr[1] || (force && n)) {
            ran = true;
            for (;;) {
                tl.add(K_LEAF, n);
                for (auto& l : ln) if (l.st == 'L') advance_walk(l);
                n = count('L');
                if (n < thr[1]) break;
            }
        }
        // retire finished walks: terminal paths end here, the others are parked by class
        n = count('F');
        if (n >= thr[5] || (force && n)) {
            int m = 0, swaps = 0;
            for (auto& l : ln) if (l.st == 'F') {
                const Segment& s = paths[l.path][l.seg];
                if (s.cls == 'T') { l.st = 'E', l.path = -1, m++; continue; }
                if (pooled() >= cap) {
                    // pool full: exchange with a parked ray (needs no free slot), else keep the finished walk
                    if (ray.empty()) continue;
                    Parked p = ray.front(); ray.pop_front();
                    (s.cls == 'O' ? qo : qg).push_back({l.path, l.seg});
                    l.path = p.path, l.seg = p.seg, l.pos = 0, l.st = next_state(paths[l.path], l.seg, 0), m++, swaps++;
                    continue;
                }
                (s.cls == 'O' ? qo : qg).push_back({l.path, l.seg});
                l.st = 'E', l.path = -1, m++;
            }
            if (m) tl.add(K_RETIRE, m), ran = true;
            if (swaps) tl.add(K_REFILL, swaps);
        }
        auto scatter = [&](std::deque<Parked>& q, int kind, int th) {
            int m = (int)std::min<size_t>(q.size(), 64);
            const bool starving = count('N') + count('L') < thr[0] && ray.empty();
            if (m >= th || ((force || starving) && m)) {
                ran = true;
                tl.add(kind, m);
                for (int i = 0; i < m; i++) {
                    Parked p = q.front(); q.pop_front();
                    if (!paths[p.path][p.seg].ends) ray.push_back({p.path, p.seg + 1});
                }
            }
        };
        scatter(qo, K_OPAQUE, thr[2]);
        scatter(qg, K_GLASS, thr[3]);
        // restart: new paths straight into empty lanes, only when no parked ray is waiting
        e = count('E');
        if (ray.empty() && next < last && (e >= thr[4] || (force && e))) {
            int m = 0;
            for (auto& l : ln) if (l.st == 'E' && next < last)
                l.path = (int)next++, l.seg = 0, l.pos = 0, l.st = next_state(paths[l.path], 0, 0), m++;
            tl.add(K_RESTART, m);
            ran = true;
        }
        *max_pool = std::max(*max_pool, (double)pooled());
        if (!ran) {
            if (next >= last && count('E') == 64 && pooled() == 0) break;
            if (force) {
                fprintf(stderr, "pool policy stuck: N %d L %d F %d E %d ray %zu qo %zu qg %zu next %zu last %zu\n", count('N'),
                        count('L'), count('F'), count('E'), ray.size(), qo.size(), qg.size(), next, last);
                exit(2);
            }
            force = true;
        } else force = false;
    }
}

// ---- policy D: a workgroup of `G` waves that exchange paths waiting for a scatter step through shared
// pools (one per scatter class).  A wave whose population of a class is below its run threshold deposits
// those lanes (they become idle at once and are refilled by the next restart); a restart step first
// withdraws from a pool that holds at least `wmin` paths and runs that scatter with the lanes it got.
// Waves advance on their own clocks (instruction slots); the next wave to act is the one furthest behind.
struct GroupCfg {
    int waves = 16;
    int thr[5] = {12, 8, 16, 6, 32};  // node, leaf, opaque run, glass run, restart
    int dep_max[2] = {64, 64};        // deposit only populations below this (and below the run threshold)
    int wmin[2] = {32, 16};           // pool size from which a restart withdraws
    int cap[2] = {128, 128};
};
static void run_group(const std::vector<Path>& paths, size_t first, size_t last, const GroupCfg& cfg, Tally& tl,
                      double* max_pool) {
    struct Wave {
        Lane ln[64];
        double clock = 0;
        bool force = false, done = false;
    };
    std::vector<Wave> wv(cfg.waves);
    std::deque<Parked> pool[2];
    size_t next = first;
    auto step = [&](Wave& w, int kind, int n) { tl.add(kind, n), w.clock += cost[kind]; };
    for (;;) {
        Wave* wp = nullptr;
        for (auto& w : wv) if (!w.done && (!wp || w.clock < wp->clock)) wp = &w;
        if (!wp) break;
        Wave& w = *wp;
        auto count = [&](char c) { int n = 0; for (auto& l : w.ln) n += l.st == c; return n; };
        auto advance_walk = [&](Lane& l) {
            l.pos++;
            l.st = next_state(paths[l.path], l.seg, l.pos);
            if (l.st == 'F') l.st = paths[l.path][l.seg].cls;
        };
        auto scatter_lane = [&](Lane& l) {
            const Segment& sg = paths[l.path][l.seg];
            if (sg.ends) l.st = 'I', l.path = -1;
            else l.seg++, l.pos = 0, l.st = next_state(paths[l.path], l.seg, 0);
        };
        bool ran = false;
        int n = count('N');
        if (n >= cfg.thr[0] || (w.force && n)) {
            ran = true;
            for (;;) {
                step(w, K_NODE, n);
                for (auto& l : w.ln) if (l.st == 'N') advance_walk(l);
                n = count('N');
                if (n < cfg.thr[0]) break;
            }
        }
        n = count('L');
        if (n >= cfg.thr[1] || (w.force && n)) {
            ran = true;
            for (;;) {
                step(w, K_LEAF, n);
                for (auto& l : w.ln) if (l.st == 'L') advance_walk(l);
                n = count('L');
                if (n < cfg.thr[1]) break;
            }
        }
        for (int c = 0; c < 2; c++) {
            const char ch = c == 0 ? 'O' : 'G';
            const int kind = c == 0 ? K_OPAQUE : K_GLASS;
            int m = count(ch);
            if (!m) continue;
            if (m >= cfg.thr[2 + c] || (w.force && (int)pool[c].size() + m > cfg.cap[c])) {
                ran = true;
                step(w, kind, m);
                for (auto& l : w.ln) if (l.st == ch) scatter_lane(l);
            } else if (m < cfg.dep_max[c] && (int)pool[c].size() + m <= cfg.cap[c]) {
                ran = true;
                step(w, K_RETIRE, m);
                for (auto& l : w.ln) if (l.st == ch) pool[c].push_back({l.path, l.seg}), l.st = 'I', l.path = -1;
            } else if (w.force) {
                ran = true;
                step(w, kind, m);
                for (auto& l : w.ln) if (l.st == ch) scatter_lane(l);
            }
        }
        n = count('I') + count('T');
        const bool queue_empty = next >= last;
        if (n >= cfg.thr[4] || (w.force && n) || (queue_empty && n && (pool[0].size() || pool[1].size()))) {
            bool did = false;
            int terminal = count('T');
            for (auto& l : w.ln) if (l.st == 'T') l.st = 'I', l.path = -1;
            // withdraw: the fuller pool first
            for (int pass = 0; pass < 2; pass++) {
                const int c = (pool[0].size() >= pool[1].size()) == (pass == 0) ? 0 : 1;
                const int idle = count('I');
                const int have = (int)pool[c].size();
                if (!idle || !have) continue;
                if (have < cfg.wmin[c] && !queue_empty && !w.force) continue;
                const int take = std::min(idle, have);
                const char ch = c == 0 ? 'O' : 'G';
                int got = 0;
                for (auto& l : w.ln) if (l.st == 'I' && got < take) {
                    Parked p = pool[c].front(); pool[c].pop_front();
                    l.path = p.path, l.seg = p.seg, l.st = ch, got++;
                }
                step(w, K_REFILL, got);
                const int m = count(ch);  // with whatever the wave still held of that class
                step(w, c == 0 ? K_OPAQUE : K_GLASS, m);
                for (auto& l : w.ln) if (l.st == ch) scatter_lane(l);
                did = true;
            }
            const int idle = count('I');
            if (idle && (!queue_empty || terminal || !did)) {
                int started = 0;
                for (auto& l : w.ln) if (l.st == 'I') {
                    if (next < last) l.path = (int)next++, l.seg = 0, l.pos = 0, l.st = next_state(paths[l.path], 0, 0), started++;
                    else if (pool[0].empty() && pool[1].empty()) l.st = 'X';
                }
                if (started || terminal) step(w, K_RESTART, std::max(started, terminal)), did = true;
            }
            ran = ran || did;
        }
        *max_pool = std::max(*max_pool, (double)(pool[0].size() + pool[1].size()));
        if (!ran) {
            if (count('X') == 64) { w.done = true; continue; }
            bool all_idle = true;
            for (auto& l : w.ln) all_idle = all_idle && (l.st == 'I' || l.st == 'X');
            if (all_idle && queue_empty && pool[0].empty() && pool[1].empty()) {
                for (auto& l : w.ln) l.st = 'X';
                w.done = true;
                continue;
            }
            if (w.force) w.clock += 10;  // nothing to do yet: wait for the others
            w.force = true;
        } else w.force = false;
    }
}

int main(int argc, char** argv) {
    if (argc < 4) return 1;
    Scene sc;
    std::string err;
    std::string text;
    if (!read_file(argv[1], text, err) || !parse_scene_ron(text, sc, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    const uint32_t W = atoi(argv[2]), H = atoi(argv[3]);
    SceneArrays a;
    a.spheres = sc.spheres.data(), a.n_spheres = sc.spheres.size();
    a.lights = sc.lights.data(), a.n_lights = sc.lights.size();
    a.lambertians = sc.lambertians.data(), a.n_lambertians = sc.lambertians.size();
    a.metals = sc.metals.data(), a.n_metals = sc.metals.size();
    a.glass = sc.glass.data(), a.n_glass = sc.glass.size();
    std::vector<char> blob;
    SceneLayout L;
    Bvh2 bvh;
    if (!pack_scene(a, false, nullptr, 0, blob, L, bvh, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    PtrMem mem{blob.data()};
    Camera64 cam64;
    camera_build(sc.camera.pos, sc.camera.yaw, sc.camera.pitch, sc.camera.vfov, (float)W / (float)H, cam64);
    CameraRegs cam = load_camera(cam64);

    const bool real_postpone = argc > 5 && !strcmp(argv[5], "postpone2");
    // trace every path of two frames in the kernel's queue order (8x8 tiles, row-major tiles)
    std::vector<Path> paths;
    const uint32_t tiles_x = (W + 7) / 8, tiles_y = (H + 7) / 8;
    size_t rays = 0;
    for (uint32_t frame = 0; frame < 2; frame++)
        for (uint32_t tile = 0; tile < tiles_x * tiles_y; tile++)
            for (uint32_t slot = 0; slot < 64; slot++) {
                const uint32_t px = (tile % tiles_x) * 8 + (slot & 7), py = (tile / tiles_x) * 8 + (slot >> 3);
                if (px >= W || py >= H) continue;
                PathState ps;
                ps.rng = pixel_seed(frame_seed(0x00C0FFEE, frame), pixel_hash(px, py, W, H));
                ps.d = camera_dir(cam, px, py, (float)W, (float)H, ps.rng);
                ps.o = cam.pos, ps.colour = mk(1, 1, 1), ps.bounce = 0;
                Path p;
                for (;;) {
                    Segment sg;
                    HostStack st;
                    Walk w;
                    Counters cnt = {0, 0, 0};
                    walk_begin<false>(w, L, ps.o, ps.d);
                    if (real_postpone) {
                        // a lane holds one leaf and keeps walking until it meets the next leaf or the end
                        int32_t pend = 0;  // 0 = none (leaf links are negative)
                        auto flush = [&]() {
                            const uint32_t ref = ~(uint32_t)pend, first = ref & LEAF_FIRST_MASK, count = (ref >> LEAF_COUNT_SHIFT) + 1u;
                            for (uint32_t k = 0; k < count; k++)
                                walk_test_sphere<PtrMem, false>(mem, L, w, ps.o, ps.d, first + k, cnt), sg.steps += 'L';
                            pend = 0;
                        };
                        for (;;) {
                            if (w.cur == WALK_FINISHED) {
                                if (pend) flush();
                                break;
                            }
                            if (walk_is_node(w.cur)) {
                                walk_node<PtrMem, HostStack, false, false>(mem, L, st, w, cnt), sg.steps += 'N';
                            } else {
                                if (pend) flush();
                                pend = w.cur;
                                walk_pop(w, st);
                            }
                        }
                    } else
                    while (w.cur != WALK_FINISHED) {
                        if (walk_is_node(w.cur)) walk_node<PtrMem, HostStack, false, false>(mem, L, st, w, cnt), sg.steps += 'N';
                        else walk_leaf_one<PtrMem, HostStack, false>(mem, L, st, w, ps.o, ps.d, cnt), sg.steps += 'L';
                    }
                    rays++;
                    Hit hit;
                    walk_result(w, hit);
                    const uint32_t mbits = hit_material(mem, L, hit);
                    const uint32_t cls = scatter_class(mbits, hit.back);
                    sg.cls = cls == SC_OPAQUE ? 'O' : (cls == SC_GLASS ? 'G' : 'T');
                    V3 radiance;
                    uint32_t mat;
                    sg.ends = scatter(mem, L, hit, ps, 50, true, radiance, mat);
                    p.push_back(sg);
                    if (sg.ends) break;
                }
                paths.push_back(std::move(p));
            }
    if (argc > 5 && !strcmp(argv[5], "postpone")) {
        // optimistic model of postponed leaves: a lane holds the sphere tests of one leaf and keeps
        // walking until it reaches the next leaf or the end (extra node visits from later pruning ignored)
        for (auto& p : paths)
            for (auto& sg : p) {
                std::string out, held;
                size_t i = 0;
                while (i < sg.steps.size()) {
                    if (sg.steps[i] == 'N') { out += 'N'; i++; continue; }
                    std::string run;
                    while (i < sg.steps.size() && sg.steps[i] == 'L') run += 'L', i++;
                    out += held;
                    held = run;
                }
                out += held;
                sg.steps = out;
            }
    }
    printf("%zu paths, %zu rays (%.3f rays/path)\n", paths.size(), rays, (double)rays / paths.size());

    // waves take chunks of 256 consecutive paths; model a wave as running chunk after chunk
    auto report = [&](const char* name, const Tally& tl, double extra = -1) {
        printf("%-34s %7.1f instr-slots/ray |", name, tl.total() / rays);
        for (int k = 0; k < K_COUNT; k++)
            if (tl.exec[k] > 0) printf(" %s %.4f @%.1f |", kind_name[k], tl.exec[k] / rays, tl.lanes[k] / tl.exec[k]);
        if (extra >= 0) printf(" max pool %.0f", extra);
        printf("\n");
    };
    const size_t per_wave = 4096;  // paths one simulated wave processes
    const bool quick = argc > 4 && !strcmp(argv[4], "owner");
    {
        Tally ideal;
        const int one[5] = {1, 1, 1, 1, 1};
        run_lane_owner(paths, 0, std::min<size_t>(paths.size(), 65536), one, ideal);
    }
    {
        std::vector<std::vector<int>> sets = {{16, 8, 24, 8, 32}, {16, 16, 24, 8, 32}, {24, 8, 24, 8, 32}, {1, 1, 1, 1, 1}};
        for (int n : {8, 12, 16, 20, 24}) for (int l : {4, 8, 12, 16}) for (int o : {16, 24, 32}) for (int g : {4, 8, 16}) for (int i : {16, 24, 32, 40})
            sets.push_back({n, l, o, g, i});
        double best = 1e30;
        for (auto& th : sets) {
            Tally tl;
            for (size_t f = 0; f < paths.size(); f += per_wave) run_lane_owner(paths, f, std::min(paths.size(), f + per_wave), th.data(), tl);
            char nm[64];
            snprintf(nm, sizeof nm, "owner %d,%d,%d,%d,%d", th[0], th[1], th[2], th[3], th[4]);
            if (&th - &sets[0] < 4 || tl.total() < best) report(nm, tl);
            best = std::min(best, tl.total());
        }
    }
    {
        const double ws[][5] = {{1, 1, 1, 1, 1}, {1.5, 1, 1, 1, 1}, {2, 1, 0.7, 0.7, 0.7}, {2, 1.5, 0.5, 0.5, 0.7}, {3, 1.5, 0.5, 0.5, 0.7},
                                {2, 1.5, 0.3, 0.3, 0.5}, {1.5, 1.2, 0.5, 0.5, 0.5}, {4, 2, 0.5, 0.5, 0.5}, {2, 1, 1, 1, 1}};
        for (int fl : {1, 8, 16})
            for (auto& w : ws) {
                Tally tl;
                for (size_t f = 0; f < paths.size(); f += per_wave) run_greedy(paths, f, std::min(paths.size(), f + per_wave), w, fl, tl);
                char nm[96];
                snprintf(nm, sizeof nm, "greedy floor %d w %.1f,%.1f,%.1f,%.1f,%.1f", fl, w[0], w[1], w[2], w[3], w[4]);
                report(nm, tl);
            }
    }
    if (!quick) {
        const int sets[][8] = {{32, 8, 48, 32, 16, 8, 8, 0}, {32, 16, 48, 32, 16, 8, 8, 0}, {24, 8, 48, 32, 16, 4, 4, 0}};
        for (int cap : {64, 128})
            for (auto& th : sets) {
                Tally tl;
                double mp = 0;
                for (size_t f = 0; f < paths.size(); f += per_wave) run_pool(paths, f, std::min(paths.size(), f + per_wave), th, cap, tl, &mp);
                char nm[96];
                snprintf(nm, sizeof nm, "pool cap %3d thr %d,%d,%d,%d,%d,%d,%d", cap, th[0], th[1], th[2], th[3], th[4], th[5], th[6]);
                report(nm, tl, mp);
            }
    }
    if (!quick) {
        const size_t per_group = 16 * per_wave;
        auto run_cfg = [&](const GroupCfg& g, const char* tag) {
            Tally tl;
            double mp = 0;
            for (size_t f = 0; f < paths.size(); f += per_group) run_group(paths, f, std::min(paths.size(), f + per_group), g, tl, &mp);
            char nm[128];
            snprintf(nm, sizeof nm, "group%s thr %d,%d,%d,%d,%d dep<%d,%d wmin %d,%d cap %d", tag, g.thr[0], g.thr[1], g.thr[2], g.thr[3],
                     g.thr[4], g.dep_max[0], g.dep_max[1], g.wmin[0], g.wmin[1], g.cap[0]);
            report(nm, tl, mp);
            return tl.total();
        };
        {
            GroupCfg base;  // deposits disabled: 16 independent waves sharing one queue
            base.dep_max[0] = base.dep_max[1] = 0;
            run_cfg(base, " (no exchange)");
        }
        double best = 1e30;
        if (const char* e = getenv("SIM_XCOST")) cost[K_RETIRE] *= atof(e), cost[K_REFILL] *= atof(e);
        const int cap_env = getenv("SIM_CAP") ? atoi(getenv("SIM_CAP")) : 128;
        const bool few = getenv("SIM_FEW") != nullptr;
        for (int nthr : {12, 16}) for (int lthr : {8}) for (int othr : {24, 32, 65}) for (int gthr : {12, 65})
            for (int rthr : {24, 32}) for (int wo : {32, 48, 64}) for (int wg : {16, 32, 64}) for (int cap : {cap_env}) {
                if (few && !(nthr == 12 && rthr == 32 && wo == 48 && wg == 32 && ((othr == 24 && gthr == 12) || (othr == 65 && gthr == 65)))) continue;
                GroupCfg g;
                g.thr[0] = nthr, g.thr[1] = lthr, g.thr[2] = othr, g.thr[3] = gthr, g.thr[4] = rthr;
                g.wmin[0] = std::min(wo, cap), g.wmin[1] = std::min(wg, cap), g.cap[0] = g.cap[1] = cap;
                Tally tl;
                double mp = 0;
                for (size_t f = 0; f < paths.size(); f += per_group) run_group(paths, f, std::min(paths.size(), f + per_group), g, tl, &mp);
                if (tl.total() < best || few) {
                    best = std::min(best, tl.total());
                    run_cfg(g, "");
                }
            }
    }
    {   // every step at 64 lanes: the floor of any policy
        double lane_steps[K_COUNT] = {0};
        for (auto& p : paths) {
            for (auto& sg : p) {
                for (char c : sg.steps) lane_steps[c == 'N' ? K_NODE : K_LEAF]++;
                if (sg.cls == 'O') lane_steps[K_OPAQUE]++;
                if (sg.cls == 'G') lane_steps[K_GLASS]++;
            }
            lane_steps[K_RESTART]++;
        }
        double t = 0;
        for (int k = 0; k < K_COUNT; k++) t += lane_steps[k] / 64.0 * cost[k];
        printf("floor (all steps at 64 lanes): %.2f instr-slots/ray; lane-steps/ray node %.2f leaf %.2f opaque %.3f glass %.3f restart %.3f\n",
               t / rays, lane_steps[K_NODE] / rays, lane_steps[K_LEAF] / rays, lane_steps[K_OPAQUE] / rays, lane_steps[K_GLASS] / rays, lane_steps[K_RESTART] / rays);
    }
    return 0;
}
