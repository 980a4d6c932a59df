#include "scene_io.h"

#include <zlib.h>

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>

namespace pt {

Scene default_scene() {
    Scene s;
    auto sph = [](float x, float y, float z, float r, uint32_t mt, uint32_t mi) {
        Sphere32 p;
        memset(&p, 0, sizeof p);
        p.pos[0] = x, p.pos[1] = y, p.pos[2] = z, p.rad = r, p.mat_type = mt, p.mat_index = mi;
        return p;
    };
    // -Vector3::X is (-1, -0, -0): Neg flips the sign of every component (vector3.rs:81-87)
    s.spheres = {sph(1.0f, 0.0f, 0.0f, 1.0f, MAT_LAMBERTIAN, 0), sph(-1.0f, -0.0f, -0.0f, 1.0f, MAT_LIGHT, 0),
                 sph(0.0f, 1.0f, 0.0f, 1.0f, MAT_GLASS, 0), sph(-0.0f, -1.0f, -0.0f, 1.0f, MAT_METAL, 0)};
    s.lights = {Vec16{1.0f, 1.0f, 1.0f, 0.0f}};
    s.lambertians = {Vec16{0.5f, 0.5f, 0.5f, 0.0f}};
    s.metals = {Vec16{0.5f, 0.5f, 0.5f, 0.5f}};
    s.glass = {1.5f};
    return s;
}

bool read_file(const std::string& path, std::string& out, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        err = "cannot open " + path + ": " + strerror(errno);
        return false;
    }
    out.clear();
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    bool ok = !ferror(f);
    fclose(f);
    if (!ok) err = "read error on " + path;
    return ok;
}

// ------------------------------------------------------------------ RON subset
namespace {

struct Value {
    enum Kind { NUM, STRUCT, LIST } kind = NUM;
    std::string num;                                       // literal text
    std::vector<std::pair<std::string, Value>> fields;     // STRUCT (named) in file order
    std::vector<Value> items;                              // LIST or positional tuple
    const Value* get(const char* name) const {
        for (auto& f : fields)
            if (f.first == name) return &f.second;
        return nullptr;
    }
};

struct Parser {
    const char* p;
    const char* end;
    std::string err;
    int depth = 0;

    void skip() {
        for (;;) {
            while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) p++;
            if (p + 1 < end && p[0] == '/' && p[1] == '/') {
                while (p < end && *p != '\n') p++;
            } else if (p + 1 < end && p[0] == '/' && p[1] == '*') {
                p += 2;
                while (p + 1 < end && !(p[0] == '*' && p[1] == '/')) p++;
                p = p + 1 < end ? p + 2 : end;
            } else {
                return;
            }
        }
    }
    bool fail(const std::string& m) {
        if (err.empty()) err = "RON: " + m;
        return false;
    }
    static bool ident_start(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }
    static bool ident_char(char c) { return ident_start(c) || (c >= '0' && c <= '9'); }
    std::string ident() {
        const char* s = p;
        while (p < end && ident_char(*p)) p++;
        return std::string(s, p);
    }
    bool value(Value& v) {
        if (++depth > 32) return fail("nesting too deep");
        skip();
        if (p >= end) return fail("unexpected end of input");
        bool ok;
        if (ident_start(*p)) {  // optional struct name, e.g. `Scene(`
            const char* save = p;
            ident();
            skip();
            if (p >= end || *p != '(') {
                p = save;
                return fail("unexpected identifier");
            }
        }
        if (*p == '(') {
            p++;
            ok = body(v);
        } else if (*p == '[') {
            p++;
            v.kind = Value::LIST;
            for (;;) {
                skip();
                if (p >= end) return fail("unterminated list");
                if (*p == ']') {
                    p++;
                    break;
                }
                Value it;
                if (!value(it)) return false;
                v.items.push_back(std::move(it));
                skip();
                if (p < end && *p == ',') p++;
                else if (p < end && *p != ']') return fail("expected ',' or ']'");
            }
            ok = true;
        } else {
            const char* s = p;
            if (p < end && (*p == '-' || *p == '+')) p++;
            while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '_' ||
                               ((*p == '-' || *p == '+') && (p[-1] == 'e' || p[-1] == 'E'))))
                p++;
            if (p == s || (p == s + 1 && (*s == '-' || *s == '+'))) return fail("expected a value");
            v.kind = Value::NUM;
            v.num.assign(s, p);
            ok = true;
        }
        depth--;
        return ok;
    }
    bool body(Value& v) {  // after '('
        v.kind = Value::STRUCT;
        for (;;) {
            skip();
            if (p >= end) return fail("unterminated struct");
            if (*p == ')') {
                p++;
                return true;
            }
            const char* save = p;
            bool named = false;
            std::string name;
            if (ident_start(*p)) {
                name = ident();
                skip();
                if (p < end && *p == ':') {
                    p++;
                    named = true;
                } else {
                    p = save;
                }
            }
            Value it;
            if (!value(it)) return false;
            if (named) v.fields.emplace_back(name, std::move(it));
            else v.items.push_back(std::move(it));
            skip();
            if (p < end && *p == ',') p++;
            else if (p < end && *p != ')') return fail("expected ',' or ')'");
        }
    }
};

bool to_f32(const Value* v, float& out) {
    if (!v || v->kind != Value::NUM) return false;
    std::string s;
    for (char c : v->num)
        if (c != '_') s.push_back(c);
    char* e = nullptr;
    errno = 0;
    out = strtof(s.c_str(), &e);
    return e && *e == 0 && e != s.c_str();
}
bool to_u32(const Value* v, uint32_t& out) {
    if (!v || v->kind != Value::NUM) return false;
    if (v->num.empty() || v->num.find_first_of(".eE-") != std::string::npos) return false;
    char* e = nullptr;
    errno = 0;
    unsigned long long x = strtoull(v->num.c_str(), &e, 10);
    if (!e || *e != 0 || errno || x > 0xffffffffull) return false;
    out = (uint32_t)x;
    return true;
}
bool to_i32(const Value* v, int32_t& out) {
    if (!v || v->kind != Value::NUM) return false;
    if (v->num.empty() || v->num.find_first_of(".eE") != std::string::npos) return false;
    char* e = nullptr;
    errno = 0;
    long long x = strtoll(v->num.c_str(), &e, 10);
    if (!e || *e != 0 || errno || x > INT32_MAX || x < INT32_MIN) return false;
    out = (int32_t)x;
    return true;
}
bool to_vec3(const Value* v, float out[3]) {
    if (!v || v->kind != Value::STRUCT) return false;
    return to_f32(v->get("x"), out[0]) && to_f32(v->get("y"), out[1]) && to_f32(v->get("z"), out[2]);
}
bool parse_root(const std::string& text, Value& root, std::string& err) {
    Parser ps{text.data(), text.data() + text.size(), {}, 0};
    if (!ps.value(root)) {
        err = ps.err;
        return false;
    }
    ps.skip();
    if (ps.p != ps.end) {
        err = "RON: trailing characters";
        return false;
    }
    if (root.kind != Value::STRUCT) {
        err = "RON: top level must be a struct";
        return false;
    }
    return true;
}
bool material_list(const Value* v, const char* vec_key, const char* w_key, std::vector<Vec16>& out) {
    if (!v || v->kind != Value::LIST) return false;
    for (auto& it : v->items) {
        if (it.kind != Value::STRUCT) return false;
        float a[3], w = 0.0f;
        if (!to_vec3(it.get(vec_key), a)) return false;
        if (w_key && !to_f32(it.get(w_key), w)) return false;
        out.push_back(Vec16{a[0], a[1], a[2], w});
    }
    return true;
}

}  // namespace

bool parse_scene_ron(const std::string& text, Scene& out, std::string& err) {
    Value root;
    if (!parse_root(text, root, err)) return false;
    Scene s;
    const Value* cam = root.get("camera");
    if (!cam || cam->kind != Value::STRUCT || !to_vec3(cam->get("pos"), s.camera.pos) ||
        !to_f32(cam->get("yaw"), s.camera.yaw) || !to_f32(cam->get("pitch"), s.camera.pitch) ||
        !to_f32(cam->get("vfov"), s.camera.vfov)) {
        err = "scene: missing or malformed `camera`";
        return false;
    }
    const Value* sp = root.get("spheres");
    if (!sp || sp->kind != Value::LIST) {
        err = "scene: missing `spheres`";
        return false;
    }
    for (auto& it : sp->items) {
        Sphere32 p;
        memset(&p, 0, sizeof p);
        if (it.kind != Value::STRUCT || !to_vec3(it.get("pos"), p.pos) || !to_f32(it.get("rad"), p.rad) ||
            !to_u32(it.get("mat_type"), p.mat_type) || !to_u32(it.get("mat_index"), p.mat_index)) {
            err = "scene: malformed sphere";
            return false;
        }
        s.spheres.push_back(p);
    }
    if (!material_list(root.get("lights"), "colour", nullptr, s.lights) ||
        !material_list(root.get("lambertians"), "albedo", nullptr, s.lambertians) ||
        !material_list(root.get("metals"), "albedo", "roughness", s.metals)) {
        err = "scene: missing or malformed material list";
        return false;
    }
    const Value* gl = root.get("glass");
    if (!gl || gl->kind != Value::LIST) {
        err = "scene: missing `glass`";
        return false;
    }
    for (auto& it : gl->items) {
        float ir;
        if (it.kind != Value::STRUCT || !to_f32(it.get("ir"), ir)) {
            err = "scene: malformed glass entry";
            return false;
        }
        s.glass.push_back(ir);
    }
    out = std::move(s);
    return true;
}

bool parse_settings_ron(const std::string& text, Settings& out, std::string& err) {
    Value root;
    if (!parse_root(text, root, err)) return false;
    Settings s;
    if (!to_i32(root.get("samples"), s.samples) || !to_i32(root.get("depth"), s.depth)) {
        err = "settings: need integer `samples` and `depth`";
        return false;
    }
    out = s;
    return true;
}

// ------------------------------------------------------------------ camera
void camera_build(const float pos[3], float yaw_deg, float pitch_deg, float vfov_deg, float aspect,
                  Camera64& out) {
    // f32::to_radians multiplies by (PI / 180) evaluated in f32
    const float k = 3.14159274101257324f / 180.0f;
    const float yaw = yaw_deg * k, pitch = pitch_deg * k, vfov = vfov_deg * k;
    // CameraConfig::dir (camera.rs:51-57)
    const float dir[3] = {sinf(yaw) * cosf(pitch), sinf(pitch), cosf(yaw) * cosf(pitch)};
    // Camera::new (camera.rs:71-89)
    const float viewport = tanf(vfov / 2.0f) * 2.0f;
    const float up[3] = {0.0f, 1.0f, 0.0f};
    auto cross = [](const float a[3], const float b[3], float r[3]) {
        r[0] = (a[1] * b[2]) - (a[2] * b[1]);
        r[1] = (a[2] * b[0]) - (a[0] * b[2]);
        r[2] = (a[0] * b[1]) - (a[1] * b[0]);
    };
    float h[3], v[3];
    cross(up, dir, h);
    const float len = sqrtf((h[0] * h[0]) + (h[1] * h[1]) + (h[2] * h[2]));
    const float inv_len = 1.0f / len;  // Vector3 / f32 multiplies by the reciprocal (vector3.rs:102-108)
    for (int a = 0; a < 3; a++) h[a] = h[a] * inv_len;
    cross(dir, h, v);
    memset(&out, 0, sizeof out);
    for (int a = 0; a < 3; a++) {
        out.pos[a] = pos[a];
        out.horizontal[a] = h[a] * viewport * aspect;
        out.vertical[a] = v[a] * viewport;
        out.center[a] = pos[a] + dir[a];
    }
}

uint32_t frame_seed(uint32_t base_seed, uint32_t frame) {
    uint32_t x = base_seed ^ (frame * 0x9E3779B9u + 0x7F4A7C15u);
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}

// ------------------------------------------------------------------ image output
void postprocess_rgb8(const float* accum_rgba, uint32_t w, uint32_t h, int32_t frames, uint8_t* rgb8) {
    const float gamma = 1.0f / 2.2f;
    const float inv_frames = 1.0f / (float)frames;
    const size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) {
            float v = powf(accum_rgba[4 * i + c] * inv_frames, gamma);
            // image 0.24 f32 -> u8: clamp to [0,1], scale by 255, round half away from zero
            float q = v != v ? 0.0f : (v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v));
            rgb8[3 * i + c] = (uint8_t)roundf(q * 255.0f);
        }
}

namespace {
void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(x >> 24), v.push_back(x >> 16), v.push_back(x >> 8), v.push_back(x);
}
void chunk(std::vector<uint8_t>& file, const char tag[4], const uint8_t* data, size_t n) {
    put_be32(file, (uint32_t)n);
    size_t start = file.size();
    file.insert(file.end(), tag, tag + 4);
    file.insert(file.end(), data, data + n);
    put_be32(file, (uint32_t)crc32(0L, file.data() + start, (uInt)(n + 4)));
}
}  // namespace

bool write_png_rgb8(const std::string& path, const uint8_t* rgb8, uint32_t w, uint32_t h, std::string& err) {
    std::vector<uint8_t> raw((size_t)h * (3ull * w + 1));
    for (uint32_t y = 0; y < h; y++) {
        raw[(size_t)y * (3ull * w + 1)] = 0;  // filter: none
        memcpy(&raw[(size_t)y * (3ull * w + 1) + 1], rgb8 + (size_t)y * 3 * w, 3ull * w);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) {
        err = "png: deflate failed";
        return false;
    }
    std::vector<uint8_t> file = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w);
    put_be32(ihdr, h);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});  // 8-bit, colour type 2 (RGB)
    chunk(file, "IHDR", ihdr.data(), ihdr.size());
    chunk(file, "IDAT", z.data(), zlen);
    chunk(file, "IEND", nullptr, 0);
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) {
        err = "png: cannot open " + path + ": " + strerror(errno);
        return false;
    }
    bool ok = fwrite(file.data(), 1, file.size(), f) == file.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok) err = "png: write failed";
    return ok;
}

}  // namespace pt
