// mi355pt_offline -- the reference's headless renderer (src/bin/offline.rs:7-84) over libmi355pt.
// Same defaults (1280x720, 1000 frames, scene1.ron, settings.ron, img.png) and the same output
// lines; what the reference hard-codes is a flag here.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mi355pt.h"

static void usage() {
    fprintf(stderr,
            "usage: mi355pt_offline [--scene scene1.ron] [--settings settings.ron] [--width 1280] [--height 720]\n"
            "                       [--frames 1000] [--seed N] [--device 0 | --devices 0,1,...] [--bvh sah|reference|lbvh]\n"
            "                       [--out img.png] [--chunk 50]\n"
            "  --devices: the frames are split over one context per listed device (a device may be listed more than\n"
            "             once) and the accumulators are summed into the first one's before the image is written\n");
}

int main(int argc, char** argv) {
    std::string scene_path = "scene1.ron", settings_path = "settings.ron", out = "img.png", bvh = "sah";
    uint32_t width = 1280, height = 720, frames = 1000, seed = 0x00C0FFEEu, chunk = 50;
    std::vector<int> devices;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&](const char* name) -> const char* {
            if (i + 1 >= argc) {
                fprintf(stderr, "missing value for %s\n", name);
                exit(2);
            }
            return argv[++i];
        };
        if (a == "--scene") scene_path = next("--scene");
        else if (a == "--settings") settings_path = next("--settings");
        else if (a == "--width") width = (uint32_t)strtoul(next("--width"), nullptr, 0);
        else if (a == "--height") height = (uint32_t)strtoul(next("--height"), nullptr, 0);
        else if (a == "--frames") frames = (uint32_t)strtoul(next("--frames"), nullptr, 0);
        else if (a == "--seed") seed = (uint32_t)strtoul(next("--seed"), nullptr, 0);
        else if (a == "--device") devices.assign(1, atoi(next("--device")));
        else if (a == "--devices") {
            devices.clear();
            for (const char* p = next("--devices"); *p;) {
                char* end = nullptr;
                devices.push_back((int)strtol(p, &end, 10));
                if (end == p) {
                    fprintf(stderr, "--devices wants a comma-separated list of device ordinals\n");
                    return 2;
                }
                p = *end == ',' ? end + 1 : end;
            }
        }
        else if (a == "--bvh") bvh = next("--bvh");
        else if (a == "--out") out = next("--out");
        else if (a == "--chunk") chunk = (uint32_t)strtoul(next("--chunk"), nullptr, 0);
        else {
            usage();
            return 2;
        }
    }
    if (chunk == 0) chunk = 1;
    auto t0 = std::chrono::steady_clock::now();

    // offline.rs:37: load_ron("scene1.ron").unwrap_or_default()
    char err[512] = {0};
    mi355pt_scene* scene = nullptr;
    if (mi355pt_scene_load_ron(scene_path.c_str(), &scene, err, sizeof err) != MI355PT_OK) {
        fprintf(stderr, "warning: %s -- using the default scene\n", err);
        scene = mi355pt_scene_default();
    }
    // offline.rs:39: load_ron("settings.ron").unwrap()
    int32_t samples = 1, depth = 50;
    if (mi355pt_settings_load_ron(settings_path.c_str(), &samples, &depth, err, sizeof err) != MI355PT_OK) {
        fprintf(stderr, "error: %s\n", err);
        return 1;
    }

    if (devices.empty()) devices.push_back(0);
    const uint32_t n_ctx = (uint32_t)devices.size();
    std::vector<mi355pt_ctx*> ctxs(n_ctx, nullptr);
    for (uint32_t k = 0; k < n_ctx; k++) {
        mi355pt_desc desc;
        memset(&desc, 0, sizeof desc);
        desc.width = width, desc.height = height, desc.device = devices[k], desc.base_seed = seed;
        desc.bvh_mode = bvh == "reference" ? MI355PT_BVH_REFERENCE : (bvh == "lbvh" ? MI355PT_BVH_LBVH : MI355PT_BVH_SAH);
        if (mi355pt_create(&desc, &ctxs[k]) != MI355PT_OK) {
            fprintf(stderr, "error: %s\n", mi355pt_last_error(nullptr));
            return 1;
        }
        if (mi355pt_set_settings(ctxs[k], samples, depth) != MI355PT_OK || mi355pt_set_scene_handle(ctxs[k], scene) != MI355PT_OK) {
            fprintf(stderr, "error: %s\n", mi355pt_last_error(ctxs[k]));
            return 1;
        }
    }
    mi355pt_scene_free(scene);
    mi355pt_ctx* ctx = ctxs[0];

    // offline.rs:44-63: one frame per iteration there; here every context renders its block of the frame
    // range (mi355pt_partition_frames) a chunk at a time, all devices working on their chunks at once
    std::vector<uint32_t> first(n_ctx), count(n_ctx), done_k(n_ctx, 0);
    for (uint32_t k = 0; k < n_ctx; k++) mi355pt_partition_frames(frames, n_ctx, k, &first[k], &count[k]);
    std::vector<uint32_t> seeds;
    for (uint32_t done = 0; done < frames;) {
        for (uint32_t k = 0; k < n_ctx; k++) {
            const uint32_t n = count[k] - done_k[k] < chunk ? count[k] - done_k[k] : chunk;
            if (n == 0) continue;
            seeds.resize(n);  // frame f of the whole render has the seed a one-device render gives it
            for (uint32_t j = 0; j < n; j++) seeds[j] = mi355pt_frame_seed(seed, first[k] + done_k[k] + j);
            if (mi355pt_render_async(ctxs[k], n, seeds.data()) != MI355PT_OK) {
                fprintf(stderr, "\nerror: %s\n", mi355pt_last_error(ctxs[k]));
                return 1;
            }
            done_k[k] += n, done += n;
        }
        for (uint32_t k = 0; k < n_ctx; k++)
            if (mi355pt_wait(ctxs[k]) != MI355PT_OK) {
                fprintf(stderr, "\nerror: %s\n", mi355pt_last_error(ctxs[k]));
                return 1;
            }
        fprintf(stderr, "\r%u/%u | %.1f%% ", done, frames, (float)done / (float)frames * 100.0f);
        fflush(stderr);
    }
    fprintf(stderr, "\n");
    if (n_ctx > 1 && frames > 0) {
        // contexts that got no frame (more devices than frames) stay out of the sum
        std::vector<mi355pt_ctx*> used;
        for (uint32_t k = 0; k < n_ctx; k++)
            if (count[k]) used.push_back(ctxs[k]);
        if (mi355pt_reduce(used.data(), used.size(), 0) != MI355PT_OK) {
            fprintf(stderr, "error: %s\n", mi355pt_last_error(ctx));
            return 1;
        }
    }

    printf("Img size: %u, %u\n", width, height);
    printf("Samples: %lld\n", (long long)frames * samples);
    if (mi355pt_save_png(ctx, (int32_t)frames, out.c_str()) != MI355PT_OK) {
        fprintf(stderr, "error: %s\n", mi355pt_last_error(ctx));
        return 1;
    }
    mi355pt_stats st;
    if (mi355pt_stats_get(ctx, &st) == MI355PT_OK && st.render_ms > 0.0)
        fprintf(stderr, "device %d: %.1f ms in path kernels; %llu paths, %llu rays on %u context(s)\n", devices[0], st.render_ms,
                (unsigned long long)st.paths, (unsigned long long)st.rays, n_ctx);
    for (mi355pt_ctx* c : ctxs) mi355pt_destroy(c);
    auto secs = std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t0).count();
    printf("Time took: %llds\n", (long long)secs);
    return 0;
}
