// Host-side data model either side of the hot path: the reference's Scene / Settings structs, their
// RON files, the byte encodings it uploads (its `Bytes` trait), the camera uniform, the seed
// schedule and the image post-process + PNG of SaveInfo::finish.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "pt_types.h"

namespace pt {

struct CameraSettings {  // reference camera.rs:3-22 (angles in degrees)
    float pos[3] = {0.0f, 0.0f, -3.0f};
    float yaw = 0.0f, pitch = 0.0f, vfov = 60.0f;
};
struct Settings {  // reference mod.rs:26-30
    int32_t samples = 1;
    int32_t depth = 50;
};
struct Scene {  // reference scene.rs:5-14; element types already in wire format
    CameraSettings camera;
    std::vector<Sphere32> spheres;
    std::vector<Vec16> lights, lambertians, metals;  // metals: w = roughness, others w = 0
    std::vector<float> glass;
};

Scene default_scene();  // reference scene.rs:15-31

// RON readers.  Return false (and fill err) on any IO / syntax / schema error, which is where the
// reference's load_ron returns None (lib.rs:6-14).
bool parse_scene_ron(const std::string& text, Scene& out, std::string& err);
bool parse_settings_ron(const std::string& text, Settings& out, std::string& err);
bool read_file(const std::string& path, std::string& out, std::string& err);

// CameraConfig::new(settings, aspect).build().bytes()  (camera.rs:36-57, 71-113)
void camera_build(const float pos[3], float yaw_deg, float pitch_deg, float vfov_deg, float aspect,
                  Camera64& out);

// Seed of frame `frame` when the caller supplies none (the reference draws rand::random() per frame,
// mod.rs:128,242): a fixed 32-bit mix of the base seed and the frame counter.
uint32_t frame_seed(uint32_t base_seed, uint32_t frame);

// SaveInfo::finish (mod.rs:420-432): rgb * (1/frames), powf(1/2.2), round(clamp01 * 255).
void postprocess_rgb8(const float* accum_rgba, uint32_t w, uint32_t h, int32_t frames, uint8_t* rgb8);
bool write_png_rgb8(const std::string& path, const uint8_t* rgb8, uint32_t w, uint32_t h, std::string& err);

}  // namespace pt
