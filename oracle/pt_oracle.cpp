// pt_oracle.cpp -- CPU restatement of the reference path tracer's hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or
// calls this file; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg use it, and only as the checker.
//
// What it restates (paths relative to the reference checkout):
//   src/renderer/compute.wgsl:1-422   the whole compute shader
//   src/renderer/bvh.rs:10-140        BVHTree::new / new_interior / flatten
//   src/renderer/bounding_box.rs:17-48
//   src/renderer/camera.rs:36-113     CameraConfig::new/build/dir, Camera::new
//   src/renderer/vector3.rs:29-128    f32 vector ops (Div = multiply by 1/rhs)
//   src/renderer/mod.rs:168-244       per-frame protocol (seed, accumulate)
//   src/renderer/mod.rs:396-435       SaveInfo::finish post-process
//
// PARITY UNPINNED against the reference executable: the reference is Rust +
// WGSL run through wgpu 0.14.2 / naga 0.10.0 / Vulkan and can be neither built
// nor run where this oracle is built; it ships no golden vectors for this
// path.  The only reference-produced artefact is fixtures/mdimg.png, which pins
// the converged image statistically.  Everything the WGSL leaves to the
// driver is DEFINED here (and identically in the GPU kernels):
//   D1  no FMA contraction; every + - * / sqrt is one IEEE binary32 operation
//       evaluated in WGSL source order (build with -ffp-contract=off).
//   D2  dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
//   D3  normalize(v)  = v * (1.0f / sqrtf(dot(v,v)))
//   D4  reflect(d,n)  = d - (2.0f*dot(n,d))*n
//   D5  pow(x,5.0)    = (x*x)*(x*x)*x
//   D6  sin(x)        = det_sin(x) below (Cody-Waite + Cephes polynomials)
//   D7  u32(f)        = saturating, NaN -> 0
//   D8  min/max       = fminf/fmaxf (a NaN operand is ignored)
//   D9  a pixel seed of 0 is replaced by 0x9E3779B9 (xorshift32 would stick
//       at 0 and rand_in_sphere would never terminate)
//   D10 the glass branch draws its random number on every glass hit, also
//       under total internal reflection (naga 0.10 lowers `||` without
//       short-circuit and hoists the call); orc_set_glass_always_draws(0)
//       selects the short-circuit reading.
//   D11 per-frame seeds are an input (the reference draws them from
//       rand::random()); the BVH child swap (bvh.rs:107) is a parameter.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------- byte layouts
struct Sphere32 {  // sphere.rs:26-43
    float pos[3];
    float rad;
    uint32_t mat_type;
    uint32_t mat_index;
    uint32_t pad[2];
};
struct Node48 {  // bvh.rs:20-36
    float mn[3];
    uint32_t pad0;
    float mx[3];
    uint32_t pad1;
    uint32_t type;
    uint32_t index;
    uint32_t pad2[2];
};
struct Vec16 {  // Light / Lambertian / Metal (w = roughness for Metal), materials.rs
    float x, y, z, w;
};
struct Camera64 {  // camera.rs:91-113
    float pos[3], p0;
    float horizontal[3], p1;
    float vertical[3], p2;
    float center[3], p3;
};
static_assert(sizeof(Sphere32) == 32 && sizeof(Node48) == 48 && sizeof(Camera64) == 64 &&
                  sizeof(Vec16) == 16,
              "layout");

struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator/(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }  // D2
inline V3 normalize(V3 v) {                                                   // D3
    float inv = 1.0f / sqrtf(dot(v, v));
    return v * inv;
}
inline V3 reflect(V3 d, V3 n) {  // D4
    float k = 2.0f * dot(n, d);
    return d - k * n;
}
inline V3 vmin(V3 a, V3 b) { return v3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return v3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }

bool g_glass_always_draws = true;  // D10

// ---------------------------------------------------------------- random (compute.wgsl:1-35)
// D6: deterministic single-precision sine.  k = round(x*2/pi), three-part
// Cody-Waite reduction by pi/2, Cephes sinf/cosf minimax polynomials on
// [-pi/4, pi/4].  Only IEEE + - * and int conversion: bit-identical anywhere.
float det_sin(float x) {
    float kf = x * 0.63661975f;
    float kr = kf + (kf < 0.0f ? -0.5f : 0.5f);
    int32_t k = (int32_t)kr;
    float fk = (float)k;
    float r = x - fk * 1.5703125f;
    r = r - fk * 4.837512969970703125e-4f;
    r = r - fk * 7.54978995489188216e-8f;
    float z = r * r;
    float s = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float c = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z -
              0.5f * z + 1.0f;
    switch (k & 3) {
        case 0: return s;
        case 1: return c;
        case 2: return -s;
        default: return -c;
    }
}
inline float fract(float x) { return x - floorf(x); }
// compute.wgsl:2-4
float vec2rand(float cx, float cy) {
    float d = cx * 12.9898f + cy * 78.233f;
    return fract(det_sin(d) * 43758.5453f);
}
// D7
inline uint32_t sat_u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}
// compute.wgsl:5-12
inline uint32_t randu(uint32_t& seed) {
    uint32_t r = seed;
    r ^= r << 13;
    r ^= r >> 17;
    r ^= r << 5;
    seed = r;
    return r;
}
// compute.wgsl:13-15.  f32(0xffffffffu) rounds to 2^32, so this is float(r) * 2^-32.
inline float randf(uint32_t& seed) { return (float)randu(seed) / 4294967296.0f; }
// compute.wgsl:19-35
inline V3 rand_in_sphere(uint32_t& seed) {
    for (;;) {
        float x = 2.0f * randf(seed) - 1.0f;
        float y = 2.0f * randf(seed) - 1.0f;
        float z = 2.0f * randf(seed) - 1.0f;
        float len_sqrd = x * x + y * y + z * z;
        if (len_sqrd < 1.0f) return v3(x, y, z);
    }
}
// compute.wgsl:392-400 (+ D9).  ndc of the pixel centre, then the sin hash.
uint32_t pixel_seed(uint32_t gseed, uint32_t px, uint32_t py, uint32_t w, uint32_t h) {
    float ux = ((float)px + 0.5f) / (float)w;
    float uy = ((float)py + 0.5f) / (float)h;
    float nx = ux * 2.0f - 1.0f;
    float ny = -(uy * 2.0f - 1.0f);
    uint32_t s = gseed + sat_u32(vec2rand(nx, ny) * 4294967296.0f);
    if (s == 0u) s = 0x9E3779B9u;
    return s;
}

// ---------------------------------------------------------------- geometry
struct Ray {  // compute.wgsl:39-53 (inv_dir is never read by the shader; not kept)
    V3 pos, dir;
};
struct HitRecord {  // compute.wgsl:71-78
    float t;
    V3 pos, norm;
    uint32_t sphere_index;
    bool back;
};
// compute.wgsl:85-96
inline bool ray_bbox_intersect(V3 mn, V3 mx, const Ray& ray) {
    V3 t0s = (mn - ray.pos) / ray.dir;
    V3 t1s = (mx - ray.pos) / ray.dir;
    V3 tsmaller = vmin(t0s, t1s);
    V3 tbigger = vmax(t0s, t1s);
    float tmin = fmaxf(0.0f, fmaxf(tsmaller.x, fmaxf(tsmaller.y, tsmaller.z)));
    float tmax = fminf(tbigger.x, fminf(tbigger.y, tbigger.z));
    return tmin < tmax;
}
// compute.wgsl:115-150
inline bool ray_sphere_intersect(const Sphere32& sp, const Ray& ray, float t_min, float t_max,
                                 HitRecord& rec) {
    V3 c = v3(sp.pos[0], sp.pos[1], sp.pos[2]);
    V3 dir = ray.pos - c;
    float half_b = dot(dir, ray.dir);
    float cc = dot(dir, dir) - (sp.rad * sp.rad);
    float d = half_b * half_b - cc;
    if (d < 0.0f) return false;
    float d_sqrt = sqrtf(d);
    float t = (-half_b - d_sqrt);
    bool back = false;
    if (t < t_min || t > t_max) {
        t = (-half_b + d_sqrt);
        if (t < t_min || t > t_max) return false;
        back = true;
    }
    V3 pos = ray.pos + (ray.dir * t);
    rec.t = t;
    rec.pos = pos;
    rec.norm = normalize(pos - c);
    if (back) rec.norm = -rec.norm;
    rec.back = back;
    return true;
}

struct Stats {
    uint64_t paths, rays, interior_visits, leaf_visits, accepted, bounces[4], misses, exhausted;
};

struct SceneView {
    const Node48* bvh;
    uint32_t n_nodes;
    const Sphere32* spheres;
    uint32_t n_spheres;
    const Vec16* lights;
    uint32_t n_lights;
    const Vec16* lambertians;
    uint32_t n_lambertians;
    const Vec16* metals;
    uint32_t n_metals;
    const float* glass;
    uint32_t n_glass;
};

// compute.wgsl:208-268 -- stackless pre-order walk with escape indices.
inline bool closet_hit(const SceneView& sc, const Ray& ray, float t_min, float t_max, HitRecord& out,
                       Stats& st) {
    HitRecord closest{};
    closest.t = t_max;
    bool has_hit = false;
    st.rays++;
    uint32_t i = 0;
    while (i < sc.n_nodes) {
        const Node48& node = sc.bvh[i];
        i += 1;
        if (node.type == 0u) {
            st.interior_visits++;
            if (!ray_bbox_intersect(v3(node.mn[0], node.mn[1], node.mn[2]),
                                    v3(node.mx[0], node.mx[1], node.mx[2]), ray))
                i = node.index;
        } else if (node.type == 1u) {
            st.leaf_visits++;
            HitRecord rec{};
            if (ray_sphere_intersect(sc.spheres[node.index], ray, t_min, closest.t, rec)) {
                closest = rec;
                closest.sphere_index = node.index;
                has_hit = true;
                st.accepted++;
            }
        } else {
            return false;
        }
    }
    if (has_hit) {
        out = closest;
        return true;
    }
    return false;
}

// compute.wgsl:197-207
inline V3 refract(V3 i, V3 n, float etai_over_etat) {
    float cos_theta = fminf(dot(-i, n), 1.0f);
    V3 r_out_perp = etai_over_etat * (i + cos_theta * n);
    V3 r_out_parallel = -sqrtf(fabsf(1.0f - dot(r_out_perp, r_out_perp))) * n;
    return r_out_perp + r_out_parallel;
}
inline float pow5(float x) {  // D5
    float x2 = x * x;
    return x2 * x2 * x;
}
inline float reflectance(float cosine, float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow5(1.0f - cosine);
}
// compute.wgsl:269-276
inline V3 miss(float dir_y) {
    float t = (dir_y + 1.0f) / 2.0f;
    return (1.0f - t) * v3(1.0f, 1.0f, 1.0f) + t * v3(0.5f, 0.7f, 1.0f);
}

// compute.wgsl:277-368
V3 trace_path(const SceneView& sc, Ray ray, uint32_t& seed, int32_t depth, Stats& st) {
    V3 colour = v3(1.0f, 1.0f, 1.0f);
    V3 light = v3(0.0f, 0.0f, 0.0f);
    const float t_min = 0.0f;
    const float t_max = INFINITY;
    const float dist = 0.001f;
    bool not_hit_light = true;
    int32_t i = 0;
    while (i <= depth && not_hit_light) {
        HitRecord hr{};
        if (closet_hit(sc, ray, t_min, t_max, hr, st)) {
            const Sphere32& sphere = sc.spheres[hr.sphere_index];
            V3 new_pos = hr.pos + hr.norm * dist;
            switch (sphere.mat_type) {
                case 0u: {
                    st.bounces[0]++;
                    const Vec16& m = sc.lights[sphere.mat_index];
                    light = v3(m.x, m.y, m.z);
                    not_hit_light = false;
                } break;
                case 1u: {
                    st.bounces[1]++;
                    if (hr.back) return v3(0, 0, 0);
                    const Vec16& m = sc.lambertians[sphere.mat_index];
                    V3 scattered = normalize(hr.norm + rand_in_sphere(seed));
                    ray = Ray{new_pos, scattered};
                    colour = colour * v3(m.x, m.y, m.z);
                } break;
                case 2u: {
                    st.bounces[2]++;
                    if (hr.back) return v3(0, 0, 0);
                    const Vec16& m = sc.metals[sphere.mat_index];
                    V3 reflected = normalize(reflect(ray.dir, hr.norm) + rand_in_sphere(seed) * m.w);
                    ray = Ray{new_pos, reflected};
                    colour = colour * v3(m.x, m.y, m.z);
                } break;
                case 3u: {
                    st.bounces[3]++;
                    float mir = sc.glass[sphere.mat_index];
                    float ir = hr.back ? mir : 1.0f / mir;
                    float cos_theta = fminf(dot(-ray.dir, hr.norm), 1.0f);
                    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
                    bool cannot_refract = ir * sin_theta > 1.0f;
                    bool do_reflect;
                    if (g_glass_always_draws) {
                        float refl = reflectance(cos_theta, ir);
                        float rnd = randf(seed);
                        do_reflect = cannot_refract || refl > rnd;
                    } else {
                        do_reflect = cannot_refract || reflectance(cos_theta, ir) > randf(seed);
                    }
                    V3 dir, np;
                    if (do_reflect) {
                        dir = reflect(ray.dir, hr.norm);
                        np = hr.pos + hr.norm * dist;
                    } else {
                        dir = refract(ray.dir, hr.norm, ir);
                        np = hr.pos - hr.norm * dist;
                    }
                    dir = normalize(dir);
                    ray = Ray{np, dir};
                } break;
                default:
                    return v3(0, 0, 0);
            }
        } else {
            st.misses++;
            light = miss(ray.dir.y);
            not_hit_light = false;
        }
        i += 1;
    }
    if (not_hit_light) st.exhausted++;
    return colour * light;
}

// compute.wgsl:63-68
inline Ray camera_get_ray(const Camera64& cam, float u, float v) {
    V3 pos = v3(cam.pos[0], cam.pos[1], cam.pos[2]);
    V3 hor = v3(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]);
    V3 ver = v3(cam.vertical[0], cam.vertical[1], cam.vertical[2]);
    V3 cen = v3(cam.center[0], cam.center[1], cam.center[2]);
    return Ray{pos, normalize(cen + hor * u + ver * v - pos)};
}

// compute.wgsl:375-422 for one in-range pixel; returns final_colour / samples.
V3 shade_pixel(const SceneView& sc, const Camera64& cam, uint32_t px, uint32_t py, uint32_t w,
               uint32_t h, uint32_t gseed, int32_t samples, int32_t depth, Stats& st) {
    uint32_t local_seed = pixel_seed(gseed, px, py, w, h);
    V3 final_colour = v3(0, 0, 0);
    for (int32_t i = 0; i < samples; i++) {
        float rx = randf(local_seed);
        float ry = randf(local_seed);
        float ux = ((float)px + rx) / (float)w;
        float uy = ((float)py + ry) / (float)h;
        float nx = ux * 2.0f - 1.0f;
        float ny = -(uy * 2.0f - 1.0f);
        Ray ray = camera_get_ray(cam, nx, ny);
        st.paths++;
        final_colour = final_colour + trace_path(sc, ray, local_seed, depth, st);
    }
    float fs = (float)samples;
    return v3(final_colour.x / fs, final_colour.y / fs, final_colour.z / fs);
}

// ---------------------------------------------------------------- BVH build (bvh.rs)
struct BBox {
    V3 mn, mx;
};
// bounding_box.rs:17-27
inline BBox sphere_bbox(const Sphere32& s) {
    V3 pos = v3(s.pos[0], s.pos[1], s.pos[2]);
    V3 rad = v3(s.rad, s.rad, s.rad);
    V3 off = v3(0.01f, 0.01f, 0.01f);
    return BBox{pos - rad - off, pos + rad + off};
}
// bounding_box.rs:44-48 (Rust f32::min/max ignore a NaN operand, as fminf/fmaxf)
inline BBox combine(const BBox& a, const BBox& b) { return BBox{vmin(a.mn, b.mn), vmax(a.mx, b.mx)}; }

struct Tree {
    bool leaf;
    uint32_t obj;
    BBox bbox;
    Tree *left, *right;
};
struct Builder {
    const Sphere32* spheres;
    int swap_mode;  // 0 never, 1 always, 2 xorshift32 stream
    uint32_t swap_state;
    std::vector<Tree*> pool;
    Tree* make() {
        pool.push_back(new Tree{});
        return pool.back();
    }
    Tree* leaf(uint32_t id) {
        Tree* t = make();
        t->leaf = true;
        t->obj = id;
        t->bbox = sphere_bbox(spheres[id]);
        return t;
    }
    bool coin() {
        if (swap_mode == 0) return false;
        if (swap_mode == 1) return true;
        return (randu(swap_state) >> 16) & 1u;
    }
    // bvh.rs:60-112
    Tree* interior(std::vector<uint32_t> scene) {
        Tree* t = make();
        t->leaf = false;
        size_t n = scene.size();
        if (n == 2) {
            t->left = leaf(scene[0]);
            t->right = leaf(scene[1]);
            t->bbox = combine(t->left->bbox, t->right->bbox);
        } else if (n == 3) {
            uint32_t obj = scene.back();
            scene.pop_back();
            t->left = leaf(obj);
            t->right = interior(std::move(scene));
            t->bbox = combine(t->left->bbox, t->right->bbox);
        } else {
            // bvh.rs:82-96: `diff = max - min` is taken from the +-f32::MAX
            // sentinels, not from the accumulated box, so diff = (-inf,-inf,-inf),
            // both comparisons are false and the sort key is always z.
            V3 mn = v3(3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f);
            V3 mx = v3(-3.40282347e+38f, -3.40282347e+38f, -3.40282347e+38f);
            V3 diff = mx - mn;
            int axis = (diff.x > diff.y && diff.x > diff.z) ? 0 : (diff.y > diff.z ? 1 : 2);
            const Sphere32* sp = spheres;
            std::stable_sort(scene.begin(), scene.end(), [sp, axis](uint32_t a, uint32_t b) {
                return sp[a].pos[axis] < sp[b].pos[axis];
            });
            std::vector<uint32_t> other(scene.begin() + n / 2, scene.end());
            scene.resize(n / 2);
            Tree* l = interior(std::move(scene));
            Tree* r = interior(std::move(other));
            t->bbox = combine(l->bbox, r->bbox);
            if (coin()) std::swap(l, r);
            t->left = l;
            t->right = r;
        }
        return t;
    }
    ~Builder() {
        for (Tree* t : pool) delete t;
    }
};
// bvh.rs:114-140
void flatten(const Tree* t, std::vector<Node48>& out) {
    if (t->leaf) {
        Node48 n{};
        n.mn[0] = t->bbox.mn.x, n.mn[1] = t->bbox.mn.y, n.mn[2] = t->bbox.mn.z;
        n.mx[0] = t->bbox.mx.x, n.mx[1] = t->bbox.mx.y, n.mx[2] = t->bbox.mx.z;
        n.type = 1u;
        n.index = t->obj;
        out.push_back(n);
        return;
    }
    size_t i = out.size();
    out.push_back(Node48{});
    flatten(t->left, out);
    flatten(t->right, out);
    Node48 n{};
    n.mn[0] = t->bbox.mn.x, n.mn[1] = t->bbox.mn.y, n.mn[2] = t->bbox.mn.z;
    n.mx[0] = t->bbox.mx.x, n.mx[1] = t->bbox.mx.y, n.mx[2] = t->bbox.mx.z;
    n.type = 0u;
    n.index = (uint32_t)out.size();
    out[i] = n;
}

}  // namespace

// ================================================================== C entry points (ctypes)
extern "C" {

void orc_set_glass_always_draws(int v) { g_glass_always_draws = v != 0; }

uint32_t orc_randu(uint32_t* seed) { return randu(*seed); }
float orc_randf(uint32_t* seed) { return randf(*seed); }
void orc_rand_in_sphere(uint32_t* seed, float out[3]) {
    V3 v = rand_in_sphere(*seed);
    out[0] = v.x, out[1] = v.y, out[2] = v.z;
}
float orc_det_sin(float x) { return det_sin(x); }
float orc_vec2rand(float x, float y) { return vec2rand(x, y); }
uint32_t orc_sat_u32(float f) { return sat_u32(f); }
uint32_t orc_pixel_seed(uint32_t gseed, uint32_t px, uint32_t py, uint32_t w, uint32_t h) {
    return pixel_seed(gseed, px, py, w, h);
}
void orc_miss(float dir_y, float out[3]) {
    V3 v = miss(dir_y);
    out[0] = v.x, out[1] = v.y, out[2] = v.z;
}
float orc_reflectance(float cosine, float ref_idx) { return reflectance(cosine, ref_idx); }
void orc_refract(const float i[3], const float n[3], float eta, float out[3]) {
    V3 v = refract(v3(i[0], i[1], i[2]), v3(n[0], n[1], n[2]), eta);
    out[0] = v.x, out[1] = v.y, out[2] = v.z;
}
void orc_reflect(const float d[3], const float n[3], float out[3]) {
    V3 v = reflect(v3(d[0], d[1], d[2]), v3(n[0], n[1], n[2]));
    out[0] = v.x, out[1] = v.y, out[2] = v.z;
}
void orc_normalize(const float a[3], float out[3]) {
    V3 v = normalize(v3(a[0], a[1], a[2]));
    out[0] = v.x, out[1] = v.y, out[2] = v.z;
}
// rec = {t, pos.xyz, norm.xyz, back}
int orc_ray_sphere(const void* sphere32, const float o[3], const float d[3], float t_min, float t_max,
                   float rec[8]) {
    HitRecord hr{};
    Ray r{v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])};
    bool ok = ray_sphere_intersect(*(const Sphere32*)sphere32, r, t_min, t_max, hr);
    if (ok) {
        rec[0] = hr.t;
        rec[1] = hr.pos.x, rec[2] = hr.pos.y, rec[3] = hr.pos.z;
        rec[4] = hr.norm.x, rec[5] = hr.norm.y, rec[6] = hr.norm.z;
        rec[7] = hr.back ? 1.0f : 0.0f;
    }
    return ok;
}
int orc_ray_bbox(const float mn[3], const float mx[3], const float o[3], const float d[3]) {
    Ray r{v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])};
    return ray_bbox_intersect(v3(mn[0], mn[1], mn[2]), v3(mx[0], mx[1], mx[2]), r);
}

// vector3.rs known answers: op 0 length, 1 length_sqrd, 2 normal (out[0..3]), 3 dot, 4 cross.
void orc_vector3(int op, const float a[3], const float b[3], float out[3]) {
    V3 A = v3(a[0], a[1], a[2]), B = v3(b[0], b[1], b[2]);
    switch (op) {
        case 0: out[0] = sqrtf((A.x * A.x) + (A.y * A.y) + (A.z * A.z)); break;
        case 1: out[0] = (A.x * A.x) + (A.y * A.y) + (A.z * A.z); break;
        case 2: {  // vector3.rs:35-37,102-108: multiply by the reciprocal of the length
            float inv = 1.0f / sqrtf((A.x * A.x) + (A.y * A.y) + (A.z * A.z));
            out[0] = A.x * inv, out[1] = A.y * inv, out[2] = A.z * inv;
        } break;
        case 3: out[0] = (A.x * B.x) + (A.y * B.y) + (A.z * B.z); break;
        default:
            out[0] = (A.y * B.z) - (A.z * B.y);
            out[1] = (A.z * B.x) - (A.x * B.z);
            out[2] = (A.x * B.y) - (A.y * B.x);
    }
}

// camera.rs:36-57,71-113.  Angles in degrees as stored in the scene file.
void orc_camera_build(const float pos[3], float yaw_deg, float pitch_deg, float vfov_deg, float aspect,
                      void* out64) {
    const float to_rad = 3.14159274f / 180.0f;  // f32::to_radians: self * (PI_f32 / 180.0)
    float yaw = yaw_deg * to_rad, pitch = pitch_deg * to_rad, vfov = vfov_deg * to_rad;
    float dy = sinf(pitch);
    float dz = cosf(yaw) * cosf(pitch);
    float dx = sinf(yaw) * cosf(pitch);
    float h = tanf(vfov / 2.0f);
    float viewport = h * 2.0f;
    // cross(UP, dir), UP = (0,1,0)
    float cx = (1.0f * dz) - (0.0f * dy);
    float cy = (0.0f * dx) - (0.0f * dz);
    float cz = (0.0f * dy) - (1.0f * dx);
    float inv = 1.0f / sqrtf((cx * cx) + (cy * cy) + (cz * cz));
    float hx = cx * inv, hy = cy * inv, hz = cz * inv;
    // cross(dir, h)
    float vx = (dy * hz) - (dz * hy);
    float vy = (dz * hx) - (dx * hz);
    float vz = (dx * hy) - (dy * hx);
    Camera64 c{};
    c.pos[0] = pos[0], c.pos[1] = pos[1], c.pos[2] = pos[2];
    c.horizontal[0] = hx * viewport * aspect;
    c.horizontal[1] = hy * viewport * aspect;
    c.horizontal[2] = hz * viewport * aspect;
    c.vertical[0] = vx * viewport, c.vertical[1] = vy * viewport, c.vertical[2] = vz * viewport;
    c.center[0] = pos[0] + dx, c.center[1] = pos[1] + dy, c.center[2] = pos[2] + dz;
    memcpy(out64, &c, 64);
}

// BVHTree::new + flatten.  Returns the node count (2n-1) or -1 (n < 2, bvh.rs:61-62 panics) /
// -2 (capacity).
int64_t orc_bvh_build(const void* spheres32, uint32_t n, int swap_mode, uint32_t swap_seed,
                      void* out_nodes48, uint64_t cap_nodes) {
    if (n < 2) return -1;
    if (cap_nodes < 2ull * n - 1) return -2;
    Builder b;
    b.spheres = (const Sphere32*)spheres32;
    b.swap_mode = swap_mode;
    b.swap_state = swap_seed ? swap_seed : 1u;
    std::vector<uint32_t> ids(n);
    for (uint32_t i = 0; i < n; i++) ids[i] = i;
    Tree* root = b.interior(std::move(ids));
    std::vector<Node48> out;
    out.reserve(2ull * n - 1);
    flatten(root, out);
    memcpy(out_nodes48, out.data(), out.size() * sizeof(Node48));
    return (int64_t)out.size();
}

struct OrcScene {
    const void* bvh;
    uint32_t n_nodes;
    const void* spheres;
    uint32_t n_spheres;
    const void* lights;
    uint32_t n_lights;
    const void* lambertians;
    uint32_t n_lambertians;
    const void* metals;
    uint32_t n_metals;
    const void* glass;
    uint32_t n_glass;
};
static SceneView view_of(const OrcScene* s) {
    SceneView v;
    v.bvh = (const Node48*)s->bvh, v.n_nodes = s->n_nodes;
    v.spheres = (const Sphere32*)s->spheres, v.n_spheres = s->n_spheres;
    v.lights = (const Vec16*)s->lights, v.n_lights = s->n_lights;
    v.lambertians = (const Vec16*)s->lambertians, v.n_lambertians = s->n_lambertians;
    v.metals = (const Vec16*)s->metals, v.n_metals = s->n_metals;
    v.glass = (const float*)s->glass, v.n_glass = s->n_glass;
    return v;
}

// One closet_hit.  rec = {t, pos.xyz, norm.xyz, back}; visits = {interior, leaf, accepted}.
int orc_closest_hit(const OrcScene* scene, const float o[3], const float d[3], float rec[8],
                    uint32_t* sphere_index, uint64_t visits[3]) {
    SceneView sc = view_of(scene);
    Stats st{};
    HitRecord hr{};
    Ray r{v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])};
    bool ok = closet_hit(sc, r, 0.0f, INFINITY, hr, st);
    if (ok) {
        rec[0] = hr.t;
        rec[1] = hr.pos.x, rec[2] = hr.pos.y, rec[3] = hr.pos.z;
        rec[4] = hr.norm.x, rec[5] = hr.norm.y, rec[6] = hr.norm.z;
        rec[7] = hr.back ? 1.0f : 0.0f;
        *sphere_index = hr.sphere_index;
    }
    if (visits) visits[0] = st.interior_visits, visits[1] = st.leaf_visits, visits[2] = st.accepted;
    return ok;
}

// Primary camera ray through (px + jx, py + jy): out = {o.xyz, d.xyz}.
void orc_camera_ray(const void* cam64, uint32_t px, uint32_t py, float jx, float jy, uint32_t w, uint32_t h,
                    float out[6]) {
    float ux = ((float)px + jx) / (float)w;
    float uy = ((float)py + jy) / (float)h;
    Ray r = camera_get_ray(*(const Camera64*)cam64, ux * 2.0f - 1.0f, -(uy * 2.0f - 1.0f));
    out[0] = r.pos.x, out[1] = r.pos.y, out[2] = r.pos.z;
    out[3] = r.dir.x, out[4] = r.dir.y, out[5] = r.dir.z;
}

// The per-frame protocol of mod.rs:168-244 + the kernel: for every frame f,
// accum[p] += vec4(shade_pixel(p, seeds[f]), 0).  accum is W*H RGBA f32, row-major, tight rows; the
// caller clears it to (0,0,0,1) like the reference's LoadOp::Clear.  Only rows [y0, y1) and
// columns [x0, x1) are rendered (crop for large configurations); pass 0,h,0,w for the whole image.
// stats (9+ u64): paths, rays, interior, leaf, accepted, bounce[0..3], misses, exhausted.
void orc_render(const OrcScene* scene, const void* cam64, uint32_t w, uint32_t h, int32_t samples,
                int32_t depth, uint32_t n_frames, const uint32_t* seeds, float* accum, uint32_t x0,
                uint32_t x1, uint32_t y0, uint32_t y1, int threads, uint64_t* stats_out) {
    SceneView sc = view_of(scene);
    Camera64 cam;
    memcpy(&cam, cam64, 64);
    if (threads < 1) threads = 1;
    std::vector<Stats> stats(threads);
    for (auto& s : stats) memset(&s, 0, sizeof(Stats));
    std::atomic<uint32_t> next_row{y0};
    auto work = [&](int tid) {
        Stats& st = stats[tid];
        for (;;) {
            uint32_t y = next_row.fetch_add(1);
            if (y >= y1) break;
            for (uint32_t x = x0; x < x1; x++) {
                float* texel = accum + 4ull * ((uint64_t)y * w + x);
                for (uint32_t f = 0; f < n_frames; f++) {
                    V3 c = shade_pixel(sc, cam, x, y, w, h, seeds[f], samples, depth, st);
                    texel[0] += c.x;
                    texel[1] += c.y;
                    texel[2] += c.z;
                    texel[3] += 0.0f;
                }
            }
        }
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(work, t);
        for (auto& t : pool) t.join();
    }
    if (stats_out) {
        uint64_t* o = stats_out;
        for (int i = 0; i < 11; i++) o[i] = 0;
        for (auto& s : stats) {
            o[0] += s.paths, o[1] += s.rays, o[2] += s.interior_visits, o[3] += s.leaf_visits;
            o[4] += s.accepted;
            for (int m = 0; m < 4; m++) o[5 + m] += s.bounces[m];
            o[9] += s.misses, o[10] += s.exhausted;
        }
    }
}

// One sample of one pixel for one frame seed: out = final_colour / samples (what main adds to the texel).
void orc_shade_pixel(const OrcScene* scene, const void* cam64, uint32_t w, uint32_t h, int32_t samples,
                     int32_t depth, uint32_t seed, uint32_t px, uint32_t py, float out[3]) {
    SceneView sc = view_of(scene);
    Stats st{};
    V3 c = shade_pixel(sc, *(const Camera64*)cam64, px, py, w, h, seed, samples, depth, st);
    out[0] = c.x, out[1] = c.y, out[2] = c.z;
}

// mod.rs:396-435: rgb *= 1/frames; rgb = powf(rgb, 1/2.2); to_rgb8 = round(clamp(x,0,1)*255)
// (image 0.24.5 f32 -> u8 conversion; Rust round() is half away from zero, as roundf).
void orc_postprocess(const float* accum_rgba, uint32_t w, uint32_t h, int32_t frames, uint8_t* rgb8) {
    const float gamma = 1.0f / 2.2f;
    const float scale = 1.0f / (float)frames;
    for (uint64_t i = 0; i < (uint64_t)w * h; i++) {
        for (int c = 0; c < 3; c++) {
            float v = accum_rgba[4 * i + c] * scale;
            v = powf(v, gamma);
            float q = v != v ? 0.0f : fminf(fmaxf(v, 0.0f), 1.0f);
            rgb8[3 * i + c] = (uint8_t)roundf(q * 255.0f);
        }
    }
}

}  // extern "C"
