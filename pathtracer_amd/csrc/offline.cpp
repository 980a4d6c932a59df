// mi355pt_offline -- the reference's headless renderer (src/bin/offline.rs:7-84) over libmi355pt.
// Same defaults (1280x720, 1000 frames, scene1.ron, settings.ron, img.png) and the same output
// lines; what the reference hard-codes is a flag here.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/mi355pt.h"

static void usage() {
    fprintf(stderr,
            "usage: mi355pt_offline [--scene scene1.ron] [--settings settings.ron] [--width 1280] [--height 720]\n"
            "                       [--frames 1000] [--seed N] [--device 0] [--bvh sah|reference|lbvh] [--out img.png]\n"
            "                       [--chunk 50]\n");
}

int main(int argc, char** argv) {
    std::string scene_path = "scene1.ron", settings_path = "settings.ron", out = "img.png", bvh = "sah";
    uint32_t width = 1280, height = 720, frames = 1000, seed = 0x00C0FFEEu, chunk = 50;
    int device = 0;
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
        else if (a == "--device") device = atoi(next("--device"));
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

    mi355pt_desc desc;
    memset(&desc, 0, sizeof desc);
    desc.width = width, desc.height = height, desc.device = device, desc.base_seed = seed;
    desc.bvh_mode = bvh == "reference" ? MI355PT_BVH_REFERENCE : (bvh == "lbvh" ? MI355PT_BVH_LBVH : MI355PT_BVH_SAH);
    mi355pt_ctx* ctx = nullptr;
    if (mi355pt_create(&desc, &ctx) != MI355PT_OK) {
        fprintf(stderr, "error: %s\n", mi355pt_last_error(nullptr));
        return 1;
    }
    if (mi355pt_set_settings(ctx, samples, depth) != MI355PT_OK || mi355pt_set_scene_handle(ctx, scene) != MI355PT_OK) {
        fprintf(stderr, "error: %s\n", mi355pt_last_error(ctx));
        return 1;
    }
    mi355pt_scene_free(scene);

    // offline.rs:44-63: one frame per iteration there; a chunk of frames per call here
    for (uint32_t done = 0; done < frames;) {
        uint32_t n = frames - done < chunk ? frames - done : chunk;
        if (mi355pt_render(ctx, n, nullptr) != MI355PT_OK) {
            fprintf(stderr, "\nerror: %s\n", mi355pt_last_error(ctx));
            return 1;
        }
        done += n;
        fprintf(stderr, "\r%u/%u | %.1f%% ", done, frames, (float)done / (float)frames * 100.0f);
        fflush(stderr);
    }
    fprintf(stderr, "\n");

    printf("Img size: %u, %u\n", width, height);
    printf("Samples: %lld\n", (long long)frames * samples);
    if (mi355pt_save_png(ctx, (int32_t)frames, out.c_str()) != MI355PT_OK) {
        fprintf(stderr, "error: %s\n", mi355pt_last_error(ctx));
        return 1;
    }
    mi355pt_stats st;
    if (mi355pt_stats_get(ctx, &st) == MI355PT_OK && st.render_ms > 0.0)
        fprintf(stderr, "device: %.1f ms in path kernels, %.1f Mpaths/s, %.1f Mrays/s\n", st.render_ms,
                (double)st.paths / st.render_ms / 1e3, (double)st.rays / st.render_ms / 1e3);
    mi355pt_destroy(ctx);
    auto secs = std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t0).count();
    printf("Time took: %llds\n", (long long)secs);
    return 0;
}
