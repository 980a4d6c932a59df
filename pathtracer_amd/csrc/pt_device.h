// Per-lane arithmetic of the path tracer: RNG, pixel hash, camera ray, closest hit over the
// internal BVH, material scatter, sky.  Included by the HIP kernels; every function is also
// compilable for the host so the per-lane logic can be exercised without a GPU.
//
// Arithmetic contract for everything that reaches a pixel value (follows the reference shader
// src/renderer/compute.wgsl in source order; what WGSL leaves to the driver is fixed here):
//   * one IEEE binary32 operation per + - * / sqrt, no FMA contraction (-ffp-contract=off),
//   * dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z,  normalize(v) = v * (1/sqrt(dot(v,v))),
//   * reflect(d,n) = d - (2*dot(n,d))*n,  pow(x,5) = (x*x)*(x*x)*x,  sin = det_sin,
//   * u32(f) saturates, a zero pixel seed becomes 0x9E3779B9.
// The box tests of the traversal do NOT reach a pixel value (they only cull, conservatively: all
// boxes carry the reference's 0.01 padding), so they are free to use FMA and a different tree.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include "pt_types.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_FN __host__ __device__ __forceinline__
#else
#define PT_FN inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define PT_FMA(a, b, c) __builtin_fmaf(a, b, c)
#define PT_RCP(x) __builtin_amdgcn_rcpf(x)
// 2*x - 1: the doubling is exact, so the fused form has the same value as multiply-then-subtract.
#define PT_TWICE_MINUS_ONE(x) __builtin_fmaf(2.0f, (x), -1.0f)
#else
#define PT_FMA(a, b, c) fmaf(a, b, c)
#define PT_RCP(x) (1.0f / (x))
#define PT_TWICE_MINUS_ONE(x) (2.0f * (x) - 1.0f)
#endif

namespace pt {

struct V3 {
    float x, y, z;
};
PT_FN V3 mk(float x, float y, float z) {
    V3 v;
    v.x = x, v.y = y, v.z = z;
    return v;
}
PT_FN V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_FN V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_FN V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
PT_FN V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
PT_FN V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
PT_FN V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_FN float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// Correctly rounded sqrt(x), 1 / x and 1 / sqrt(x) (the latter rounded twice, like the expression).
// On the device the compiler's general expansions spend a third of their instructions on operands
// that never occur here (denormal scaling, infinities); these versions run the same refinement
// without it whenever every active lane of the wave has an ordinary operand, and the general
// expansion otherwise.  tests/hip/exact_check.hip compares them with the general expansions over all
// 2^32 bit patterns.
#ifndef PT_RCP_STEPS
#define PT_RCP_STEPS 1
#endif
#if defined(__HIP_DEVICE_COMPILE__)
PT_FN float sqrt_fast_body(float x) {  // 2^-96 <= x < inf
    const float s = __builtin_amdgcn_sqrtf(x);  // within 1 ulp
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    float r = r_dn <= 0.0f ? s_dn : s;
    return r_up > 0.0f ? s_up : r;
}
PT_FN float rcp_fast_body(float x) {  // 2^-100 <= x < 2^100
    float y = __builtin_amdgcn_rcpf(x);  // within 1 ulp
    float e = __builtin_fmaf(-x, y, 1.0f);
    y = __builtin_fmaf(y, e, y);
#if PT_RCP_STEPS > 1
    e = __builtin_fmaf(-x, y, 1.0f);
    y = __builtin_fmaf(y, e, y);
#endif
    return y;
}
PT_FN bool wave_any(bool p) { return __ballot(p) != 0ull; }
PT_FN float sqrt_rn(float x) {
    if (__builtin_expect(wave_any(__float_as_uint(x) - 0x0F800000u >= 0x70000000u), 0)) return sqrtf(x);
    return sqrt_fast_body(x);
}
PT_FN float rcp_rn(float x) {
    if (__builtin_expect(wave_any(__float_as_uint(x) - 0x0D800000u >= 0x64000000u), 0)) return 1.0f / x;
    return rcp_fast_body(x);
}
PT_FN float rcp_sqrt_rn(float x) {  // 2^-96 <= x < 2^100
    if (__builtin_expect(wave_any(__float_as_uint(x) - 0x0F800000u >= 0x62000000u), 0)) return 1.0f / sqrtf(x);
    return rcp_fast_body(sqrt_fast_body(x));
}
#else
inline float sqrt_rn(float x) { return sqrtf(x); }
inline float rcp_rn(float x) { return 1.0f / x; }
inline float rcp_sqrt_rn(float x) { return 1.0f / sqrtf(x); }
#endif
PT_FN V3 normalize3(V3 v) {
    float inv = rcp_sqrt_rn(dot3(v, v));
    return v * inv;
}
PT_FN V3 reflect3(V3 d, V3 n) {
    float k = 2.0f * dot3(n, d);
    return d - k * n;
}

// ---- random: compute.wgsl:1-35 ---------------------------------------------------------------
PT_FN uint32_t randu(uint32_t& s) {
    uint32_t r = s;
    r ^= r << 13;
    r ^= r >> 17;
    r ^= r << 5;
    s = r;
    return r;
}
// f32(randu) / f32(0xffffffffu): the divisor rounds to 2^32, so this is an exact scaling.
PT_FN float randf(uint32_t& s) { return (float)randu(s) * 2.3283064365386963e-10f; }
// One iteration of rand_in_sphere's rejection loop (compute.wgsl:20-30): three draws in x, y, z
// order; true when the point lies inside the unit ball.
// 2 * randf - 1 is formed as fma(float(r), 2^-31, -1): both scalings are by powers of two and hence exact, so
// the one rounding is that of the shader's final subtraction and the bits are the same; one instruction
// instead of two per coordinate, in the loop that is a sixth of the kernel's vector work.
PT_FN float rand_signed(uint32_t& s) { return PT_FMA((float)randu(s), 4.656612873077392578125e-10f, -1.0f); }
PT_FN bool rand_in_sphere_try(uint32_t& s, V3& v) {
    float x = rand_signed(s);
    float y = rand_signed(s);
    float z = rand_signed(s);
    v = mk(x, y, z);
    float l2 = x * x + y * y + z * z;
    return l2 < 1.0f;
}
PT_FN V3 rand_in_sphere(uint32_t& s) {
    V3 v;
    while (!rand_in_sphere_try(s, v)) {
    }
    return v;
}
// Deterministic sine: nearest multiple of pi/2 removed in three Cody-Waite steps, then the
// Cephes single-precision minimax polynomials; plain IEEE ops only.
PT_FN float det_sin(float x) {
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
    int q = k & 3;
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
PT_FN uint32_t sat_u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}
// compute.wgsl:2-4, 392-400
PT_FN uint32_t pixel_hash(uint32_t px, uint32_t py, uint32_t w, uint32_t h) {
    float ux = ((float)px + 0.5f) / (float)w;
    float uy = ((float)py + 0.5f) / (float)h;
    float nx = ux * 2.0f - 1.0f;
    float ny = -(uy * 2.0f - 1.0f);
    float d = nx * 12.9898f + ny * 78.233f;
    float v = det_sin(d) * 43758.5453f;
    float fr = v - floorf(v);
    return sat_u32(fr * 4294967296.0f);
}
PT_FN uint32_t pixel_seed(uint32_t frame_seed, uint32_t hash) {
    uint32_t s = frame_seed + hash;
    return s ? s : 0x9E3779B9u;
}

// ---- camera: compute.wgsl:63-68, 405-413 -----------------------------------------------------
struct CameraRegs {
    V3 pos, horizontal, vertical, center;
};
PT_FN CameraRegs load_camera(const Camera64& c) {
    CameraRegs r;
    r.pos = mk(c.pos[0], c.pos[1], c.pos[2]);
    r.horizontal = mk(c.horizontal[0], c.horizontal[1], c.horizontal[2]);
    r.vertical = mk(c.vertical[0], c.vertical[1], c.vertical[2]);
    r.center = mk(c.center[0], c.center[1], c.center[2]);
    return r;
}
// Division of a jittered pixel coordinate by the image width or height (compute.wgsl:407-409).  The quotient that
// reaches a pixel is the correctly rounded one.  With `exact_short` it is formed as q = x * rcp, then one
// correction q + fma(-q, d, x) * rcp -- three instructions instead of the eleven of the general expansion -- which
// is only ever set after the device has compared the two for EVERY numerator a pixel coordinate can be (all
// binary32 values in [2^-32, 65536] and zero; pt_verify_div_kernel), for this very divisor.
struct PixelDiv {
    float d, rcp;
    bool exact_short;
    PT_FN PixelDiv(float divisor) : d(divisor), rcp(0.0f), exact_short(false) {}
    PT_FN PixelDiv(float divisor, float reciprocal, bool verified) : d(divisor), rcp(reciprocal), exact_short(verified) {}
    PT_FN float short_quotient(float x) const {
        const float q = x * rcp;
        return PT_FMA(PT_FMA(-q, d, x), rcp, q);
    }
    PT_FN float of(float x) const { return exact_short ? short_quotient(x) : x / d; }
};
// Draws the two jitter numbers (x first) and returns the primary direction.
PT_FN V3 camera_dir(const CameraRegs& cam, uint32_t px, uint32_t py, PixelDiv fw, PixelDiv fh, uint32_t& rng) {
    float rx = randf(rng);
    float ry = randf(rng);
    float ux = fw.of((float)px + rx);
    float uy = fh.of((float)py + ry);
    float nx = PT_TWICE_MINUS_ONE(ux);
    float ny = -PT_TWICE_MINUS_ONE(uy);
    return normalize3(cam.center + cam.horizontal * nx + cam.vertical * ny - cam.pos);
}

// ---- memory views ----------------------------------------------------------------------------
// The scene blob is read through a view so the same code serves a blob staged in LDS, a blob in
// global memory, and a host buffer.
struct PtrMem {
    static constexpr bool ORDERED_NODES = false;  // true: the view serves direction-ordered node records
    const char* base;
    PT_FN int32_t root() const { return 0; }  // link of node 0
    template <class T>
    PT_FN T load(uint32_t byte_off) const {
        return *reinterpret_cast<const T*>(base + byte_off);
    }
    // One interior node record.  A view may serve nodes from another place than the rest of the
    // blob (the kernels keep the top of a large tree in LDS), hence the separate entry points.
    PT_FN void node64(uint32_t off, f4& q0, f4& q1, f4& q2, u2& ch) const {
        q0 = load<f4>(off), q1 = load<f4>(off + 16), q2 = load<f4>(off + 32), ch = load<u2>(off + 48);
    }
    PT_FN void node32(uint32_t off, u4& a, u4& b) const { a = load<u4>(off), b = load<u4>(off + 16); }
};

// Direction-ordered form of one Node64 (see LDS_NODE_BYTES): q = the 16 words of the node, out = 20
// words; link0 / link1 are what the record stores as child links.
PT_FN void node_to_ordered(const uint32_t* q, uint32_t link0, uint32_t link1, uint32_t* out) {
    // Node64: c0.lo.xyz = q[0..2], c0.hi.xyz = q[3..5], c1.lo.xyz = q[6..8], c1.hi.xyz = q[9..11]
    for (int k = 0; k < 3; k++) {
        out[6 * k + 0] = q[k], out[6 * k + 1] = q[6 + k];      // lo0, lo1
        out[6 * k + 2] = q[3 + k], out[6 * k + 3] = q[9 + k];  // hi0, hi1
        out[6 * k + 4] = q[k], out[6 * k + 5] = q[6 + k];      // lo0, lo1 again
    }
    out[18] = link0, out[19] = link1;
}
// Host view of a blob whose nodes were converted to ordered records (links = byte offsets of the
// records): what the kernels keep in LDS, for the CPU harness.
struct PtrOrderedMem : PtrMem {
    static constexpr bool ORDERED_NODES = true;
    const char* nodes;
    PT_FN void node_ordered(uint32_t a, const uint32_t* sgn, f4& ax, f4& ay, f4& az, u2& ch) const {
        const char* p = nodes + a;
        memcpy(&ax, p + sgn[0], 16), memcpy(&ay, p + 24 + sgn[1], 16), memcpy(&az, p + 48 + sgn[2], 16);
        memcpy(&ch, p + 72, 8);
    }
};

struct Hit {
    float t;
    int32_t idx;  // leaf-order sphere index, -1 = none
    bool back;
};

struct Counters {
    uint32_t nodes, tests, accepted;
};


// Slab test of one child box against the ray (inv = 1/dir, ood = org/dir), clipped to [0, tcap].
// Scalar FMAs on purpose: the packed two-float form (v_pk_fma_f32) was measured 20 % slower for the
// whole kernel on MI355X.
PT_FN bool slab(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 inv, V3 ood, float tcap,
                float& tnear) {
    float x0 = PT_FMA(lox, inv.x, -ood.x), x1 = PT_FMA(hix, inv.x, -ood.x);
    float y0 = PT_FMA(loy, inv.y, -ood.y), y1 = PT_FMA(hiy, inv.y, -ood.y);
    float z0 = PT_FMA(loz, inv.z, -ood.z), z1 = PT_FMA(hiz, inv.z, -ood.z);
    float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
    float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tcap));
    tnear = tn;
    return tn <= tf;
}

// Closest hit (compute.wgsl:208-268 + 115-150) over the internal BVH, as a resumable walk: the
// kernels advance many rays one step at a time and choose, per wave, which kind of step to run.
// The result is the sphere with the smallest accepted root; on an exact tie the sphere that comes
// later in the reference's own flattened order wins (its walk replaces the record when
// t == closest.t), which is what the tie rank in `info` encodes.  Traversal order and tree shape
// therefore do not affect the result.
struct Walk {
    V3 inv, ood;       // reciprocal direction and origin * inv, for the slab tests only
    float best_t;
    // Leaf-order index of the closest sphere so far, with WALK_BACK_BIT set when the ray leaves it
    // from inside; -1 = none yet.  One vector register on purpose: a separate bool would live in a
    // scalar register pair as a lane mask and cost three scalar instructions per nested branch of
    // the sphere test, and the scalar unit is shared by the four SIMDs of a compute unit.
    int32_t best_code;
    int32_t cur;       // >= 0 interior node to visit; leaf link (see walk_is_leaf); or WALK_FINISHED
    int32_t sp;        // entries on the traversal stack
    uint32_t sgn[3];   // 8 where the direction component is negative, else 0 (direction-ordered nodes)
};
// Leaf links are ~(first | (count-1) << 28) and lie in [-2^30, -1]; values below are free for
// the walk's end marker and for the kernels' own lane states.
constexpr int32_t WALK_FINISHED = INT32_MIN;
constexpr uint32_t WALK_BACK_BIT = 1u << 30;  // sphere indices stay below 2^28 (LEAF_FIRST_MASK)
PT_FN uint32_t walk_best_idx(const Walk& w) { return (uint32_t)w.best_code & (WALK_BACK_BIT - 1u); }
PT_FN void walk_result(const Walk& w, Hit& hit) {  // t is set by the caller's own copy where it keeps one
    hit.t = w.best_t;
    hit.idx = w.best_code < 0 ? -1 : (int32_t)walk_best_idx(w);
    hit.back = ((uint32_t)w.best_code & WALK_BACK_BIT) != 0u && w.best_code >= 0;
}
PT_FN bool walk_is_node(int32_t cur) { return cur >= 0; }
PT_FN bool walk_is_leaf(int32_t cur) { return (uint32_t)cur >= 0xC0000000u; }

template <bool HALF>
PT_FN void walk_begin(Walk& w, const SceneLayout& L, V3 o, V3 d, int32_t root = 0) {
    // Tiny direction components are clamped so no inf/NaN enters the box arithmetic; the
    // reciprocal may be approximate.  Neither reaches a pixel value: boxes only cull, and every
    // box carries the reference's 0.01 padding.
    V3 dc = d;
    if (fabsf(dc.x) < 1e-20f) dc.x = copysignf(1e-20f, dc.x);
    if (fabsf(dc.y) < 1e-20f) dc.y = copysignf(1e-20f, dc.y);
    if (fabsf(dc.z) < 1e-20f) dc.z = copysignf(1e-20f, dc.z);
    w.inv = mk(PT_RCP(dc.x), PT_RCP(dc.y), PT_RCP(dc.z));
    w.sgn[0] = dc.x < 0.0f ? 8u : 0u, w.sgn[1] = dc.y < 0.0f ? 8u : 0u, w.sgn[2] = dc.z < 0.0f ? 8u : 0u;
    if (HALF) o = o - mk(L.box_org[0], L.box_org[1], L.box_org[2]);  // half boxes are relative to box_org
    w.ood = mk(o.x * w.inv.x, o.y * w.inv.y, o.z * w.inv.z);
    w.best_t = INFINITY;
    w.best_code = -1;
    w.cur = root;  // a scene has >= 2 spheres, so node 0 is interior; `root` is its link in the view's terms
    w.sp = 0;
}

template <class Stack>
PT_FN void walk_pop(Walk& w, Stack& stack) {
    if (w.sp > 0) {
        w.sp--;
        w.cur = stack.pop(w.sp);
    } else {
        w.cur = WALK_FINISHED;
    }
}

// One interior node: both child boxes, near child first, far child pushed.  Requires a node.
// (A straight-line variant with selects, an unconditional far-child store and an up-front pop was
// measured 9 % slower: the step is bound by vector instruction issue, and the branches skip work.)
// binary16 -> binary32 of the low / high half of a word.  On the device the conversion folds into
// the consuming FMA (v_fma_mix_f32), so half boxes cost no extra instructions.
inline float half_bits_to_float(uint32_t h) {  // host-side decoder (exact)
    const uint32_t e = (h >> 10) & 31u, m = h & 1023u;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf((float)(m | 1024u), (int)e - 25);
    return (h & 0x8000u) ? -v : v;
}
#if defined(__HIP_DEVICE_COMPILE__)
typedef _Float16 pt_half2 __attribute__((ext_vector_type(2)));
PT_FN float half_lo(uint32_t u) { return (float)__builtin_bit_cast(pt_half2, u).x; }
PT_FN float half_hi(uint32_t u) { return (float)__builtin_bit_cast(pt_half2, u).y; }
#else
inline float half_lo(uint32_t u) { return half_bits_to_float(u & 0xffffu); }
inline float half_hi(uint32_t u) { return half_bits_to_float(u >> 16); }
#endif

// fminf(a, b) for a `b` that is never a signalling NaN (the closest hit so far: infinity or a computed
// root).  The plain call makes the compiler quiet `b` first, one more vector instruction per node; the
// instruction itself returns the same value for every such pair.
#if defined(__HIP_DEVICE_COMPILE__)
PT_FN float fmin_quiet(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#else
inline float fmin_quiet(float a, float b) { return fminf(a, b); }
#endif

template <class Mem, class Stack, bool COUNT, bool HALF>
PT_FN void walk_node(const Mem& mem, const SceneLayout& L, Stack& stack, Walk& w, Counters& cnt) {
    float t0, t1;
    bool h0, h1;
    int32_t c0, c1;
    if constexpr (Mem::ORDERED_NODES) {
        // per axis (near0, near1, far0, far1): the record is read from where the ray's sign puts the
        // near planes first, so the slab test needs no min / max per axis
        f4 ax, ay, az;
        u2 ch;
        mem.node_ordered((uint32_t)w.cur, w.sgn, ax, ay, az, ch);
        t0 = fmaxf(fmaxf(PT_FMA(ax.x, w.inv.x, -w.ood.x), PT_FMA(ay.x, w.inv.y, -w.ood.y)),
                   fmaxf(PT_FMA(az.x, w.inv.z, -w.ood.z), 0.0f));
        t1 = fmaxf(fmaxf(PT_FMA(ax.y, w.inv.x, -w.ood.x), PT_FMA(ay.y, w.inv.y, -w.ood.y)),
                   fmaxf(PT_FMA(az.y, w.inv.z, -w.ood.z), 0.0f));
        const float f0 = fminf(fminf(PT_FMA(ax.z, w.inv.x, -w.ood.x), PT_FMA(ay.z, w.inv.y, -w.ood.y)),
                               fmin_quiet(PT_FMA(az.z, w.inv.z, -w.ood.z), w.best_t));
        const float f1 = fminf(fminf(PT_FMA(ax.w, w.inv.x, -w.ood.x), PT_FMA(ay.w, w.inv.y, -w.ood.y)),
                               fmin_quiet(PT_FMA(az.w, w.inv.z, -w.ood.z), w.best_t));
        h0 = t0 <= f0, h1 = t1 <= f1;
        c0 = (int32_t)ch.x, c1 = (int32_t)ch.y;
    } else if constexpr (HALF) {
        uint32_t off = L.off_nodes + (uint32_t)w.cur * 32u;
        u4 a, b;
        mem.node32(off, a, b);
        h0 = slab(half_lo(a.x), half_hi(a.x), half_lo(a.y), half_hi(a.y), half_lo(a.z), half_hi(a.z), w.inv, w.ood,
                  w.best_t, t0);
        h1 = slab(half_lo(a.w), half_hi(a.w), half_lo(b.x), half_hi(b.x), half_lo(b.y), half_hi(b.y), w.inv, w.ood,
                  w.best_t, t1);
        c0 = (int32_t)b.z, c1 = (int32_t)b.w;
    } else {
        uint32_t off = L.off_nodes + (uint32_t)w.cur * 64u;
        f4 q0, q1, q2;
        u2 ch;
        mem.node64(off, q0, q1, q2, ch);
        h0 = slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, w.inv, w.ood, w.best_t, t0);
        h1 = slab(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, w.inv, w.ood, w.best_t, t1);
        c0 = (int32_t)ch.x, c1 = (int32_t)ch.y;
    }
    if (COUNT) cnt.nodes++;
    // Two independent conditional regions (both hit: push one; none hit: pop) around a select, instead of a
    // four-way if / else chain: the same values with a third fewer exec-mask instructions, and the scalar unit
    // shares the bound of this loop (scene1, 256 frames: 17.73 -> 17.54 ms).  The same treatment of the sphere
    // test's nested acceptance was measured 0.4 % slower and is not applied.
    w.cur = h0 ? c0 : c1;
    if (h0 && h1) {
        bool swap = t1 < t0;
        stack.push(w.sp, swap ? c0 : c1);
        w.sp++;
        w.cur = swap ? c1 : c0;
    }
    if (!(h0 || h1)) walk_pop(w, stack);
}

// The reference's own box test (compute.wgsl:85-96), operation for operation: IEEE divisions by the
// direction (zero components give infinities or NaN, which min / max drop), no clipping against the
// closest hit so far, strict comparison.
PT_FN bool slab_reference(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 o, V3 d) {
    const float x0 = (lox - o.x) / d.x, y0 = (loy - o.y) / d.y, z0 = (loz - o.z) / d.z;
    const float x1 = (hix - o.x) / d.x, y1 = (hiy - o.y) / d.y, z1 = (hiz - o.z) / d.z;
    const float sx = fminf(x0, x1), sy = fminf(y0, y1), sz = fminf(z0, z1);
    const float bx = fmaxf(x0, x1), by = fmaxf(y0, y1), bz = fmaxf(z0, z1);
    const float tmin = fmaxf(0.0f, fmaxf(sx, fmaxf(sy, sz)));
    const float tmax = fminf(bx, fminf(by, bz));
    return tmin < tmax;
}
// One interior node of a reference-format tree, walked the reference's way (SceneLayout::reference_walk):
// the spheres that get tested are exactly those the reference's walk tests (compute.wgsl:236-268), so the
// result equals the reference's even where a computed root lies outside its sphere's box.  Children in
// stored order; the order of tests does not matter (ties resolve by rank).
template <class Mem, class Stack, bool COUNT>
PT_FN void walk_node_reference(const Mem& mem, const SceneLayout& L, Stack& stack, Walk& w, V3 o, V3 d, Counters& cnt) {
    const uint32_t off = L.off_nodes + (uint32_t)w.cur * 64u;
    f4 q0, q1, q2;
    u2 ch;
    mem.node64(off, q0, q1, q2, ch);
    const bool h0 = slab_reference(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, d);
    const bool h1 = slab_reference(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, d);
    if (COUNT) cnt.nodes++;
    if (h0 && h1) {
        stack.push(w.sp, (int32_t)ch.y);
        w.sp++;
        w.cur = (int32_t)ch.x;
    } else if (h0) {
        w.cur = (int32_t)ch.x;
    } else if (h1) {
        w.cur = (int32_t)ch.y;
    } else {
        walk_pop(w, stack);
    }
}

// One sphere against the running closest hit (compute.wgsl:116-135 with t_max = closest so far).
template <class Mem, bool COUNT>
PT_FN void walk_test_sphere(const Mem& mem, const SceneLayout& L, Walk& w, V3 o, V3 d, uint32_t si, Counters& cnt) {
    f4 s = mem.template load<f4>(L.off_spheres + si * 16u);
    if (COUNT) cnt.tests++;
    V3 oc = o - mk(s.x, s.y, s.z);
    float half_b = dot3(oc, d);
    float c = dot3(oc, oc) - s.w;  // s.w holds rad * rad, the product the shader forms here (compute.wgsl:121)
    float disc = half_b * half_b - c;
    if (!(disc < 0.0f)) {
        float sq = sqrt_rn(disc);
        float t = -half_b - sq;
        bool back = false;
        if (t < 0.0f) {
            t = -half_b + sq;
            back = true;
        }
        if (!(t < 0.0f)) {
            bool take = t < w.best_t;
            if (t == w.best_t && w.best_code >= 0) {
                uint32_t rank = mem.template load<u2>(L.off_info + si * 8u).y;
                uint32_t best_rank = mem.template load<u2>(L.off_info + walk_best_idx(w) * 8u).y;
                take = rank > best_rank;
            }
            if (take) {
                if (COUNT) cnt.accepted++;
                w.best_t = t;
                w.best_code = (int32_t)(si | (back ? WALK_BACK_BIT : 0u));
            }
        }
    }
}

// The sphere that is kept out of the tree (SceneLayout::n_hoisted), tested where a walk begins.
template <class Mem, bool COUNT>
PT_FN void walk_test_hoisted(const Mem& mem, const SceneLayout& L, Walk& w, V3 o, V3 d, Counters& cnt) {
    if (L.n_hoisted) walk_test_sphere<Mem, COUNT>(mem, L, w, o, d, L.n_spheres - 1u, cnt);
}

// Every sphere of the current leaf, then the next stack entry.  Requires a leaf.
template <class Mem, class Stack, bool COUNT>
PT_FN void walk_leaf(const Mem& mem, const SceneLayout& L, Stack& stack, Walk& w, V3 o, V3 d, Counters& cnt) {
    uint32_t ref = ~(uint32_t)w.cur;
    uint32_t first = ref & LEAF_FIRST_MASK;
    uint32_t count = (ref >> LEAF_COUNT_SHIFT) + 1u;
    for (uint32_t k = 0; k < count; k++) walk_test_sphere<Mem, COUNT>(mem, L, w, o, d, first + k, cnt);
    walk_pop(w, stack);
}

// The first sphere of the current leaf only; the lane keeps the rest of the leaf as its pending
// leaf, or takes the next stack entry after the last sphere.  Lanes of a wave then never wait for
// the longest leaf among them.  Requires a leaf.
template <class Mem, class Stack, bool COUNT>
PT_FN void walk_leaf_one(const Mem& mem, const SceneLayout& L, Stack& stack, Walk& w, V3 o, V3 d, Counters& cnt) {
    uint32_t ref = ~(uint32_t)w.cur;
    uint32_t si = ref & LEAF_FIRST_MASK;
    uint32_t more = ref >> LEAF_COUNT_SHIFT;
    walk_test_sphere<Mem, COUNT>(mem, L, w, o, d, si, cnt);
    // the rest of the leaf: first + 1, count - 1; under the complement that is one addition
    if (more) w.cur = (int32_t)((uint32_t)w.cur + ((1u << LEAF_COUNT_SHIFT) - 1u));
    else walk_pop(w, stack);
}

// Whole query, one ray at a time.
template <class Mem, class Stack, bool COUNT, bool HALF, bool REFERENCE_WALK = false>
PT_FN void closest_hit(const Mem& mem, const SceneLayout& L, Stack& stack, V3 o, V3 d, Hit& hit,
                       Counters& cnt) {
    Walk w;
    walk_begin<HALF>(w, L, o, d, mem.root());
    if constexpr (!REFERENCE_WALK) walk_test_hoisted<Mem, COUNT>(mem, L, w, o, d, cnt);
    for (;;) {
        while (walk_is_node(w.cur)) {
            if constexpr (REFERENCE_WALK) walk_node_reference<Mem, Stack, COUNT>(mem, L, stack, w, o, d, cnt);
            else walk_node<Mem, Stack, COUNT, HALF>(mem, L, stack, w, cnt);
        }
        if (w.cur == WALK_FINISHED) break;
        walk_leaf<Mem, Stack, COUNT>(mem, L, stack, w, o, d, cnt);
    }
    walk_result(w, hit);
}

// ---- materials: compute.wgsl:197-207, 269-368 ---------------------------------------------------
PT_FN V3 refract3(V3 i, V3 n, float eta) {
    float cos_theta = fminf(dot3(-i, n), 1.0f);
    V3 perp = eta * (i + cos_theta * n);
    V3 par = -sqrt_rn(fabsf(1.0f - dot3(perp, perp))) * n;
    return perp + par;
}
// compute.wgsl:203-207.  The part that depends on the index alone is split off: a glass material is only
// ever used with eta = ir (leaving) or 1 / ir (entering), so the scene blob carries both and their r0
// (glass_record), computed on the host with these same IEEE operations.
PT_FN float reflectance_r0(float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    return r0 * r0;
}
PT_FN float reflectance_from_r0(float cosine, float r0) {
    float x = 1.0f - cosine;
    float x2 = x * x;
    return r0 + (1.0f - r0) * (x2 * x2 * x);
}
PT_FN float reflectance(float cosine, float ref_idx) { return reflectance_from_r0(cosine, reflectance_r0(ref_idx)); }
// One glass material in the blob: ir, 1 / ir (compute.wgsl:331-335), r0(ir), r0(1 / ir).
PT_FN f4 glass_record(float ir) {
    const float inv = rcp_rn(ir);
    return f4{ir, inv, reflectance_r0(ir), reflectance_r0(inv)};
}
PT_FN V3 sky(float dir_y) {
    float t = (dir_y + 1.0f) / 2.0f;
    return (1.0f - t) * mk(1.0f, 1.0f, 1.0f) + t * mk(0.5f, 0.7f, 1.0f);
}

struct PathState {
    V3 o, d;        // current ray
    V3 colour;      // throughput
    uint32_t rng;   // xorshift32 state (carries across the samples of one pixel-frame)
    int32_t bounce; // loop index i of trace_path
};

// One iteration of trace_path's loop after the closest-hit query, split by what the hit needs so a
// wave can run each kind with many lanes at once:
//   SC_TERMINAL  the path ends here: sky, light, back face of an opaque sphere, unknown material
//   SC_OPAQUE    lambertian or metal front face: one rand_in_sphere, new direction, albedo
//   SC_GLASS     dielectric: reflect or refract
enum : uint32_t { SC_TERMINAL = 0, SC_OPAQUE = 1, SC_GLASS = 2 };

// Material word of the hit sphere (type | index << 3); MAT_SKY for a miss.
constexpr uint32_t MAT_UNKNOWN = 4, MAT_SKY = 5;
template <class Mem>
PT_FN uint32_t hit_material(const Mem& mem, const SceneLayout& L, const Hit& hit) {
    if (hit.idx < 0) return MAT_SKY;
    return mem.template load<u2>(L.off_info + (uint32_t)hit.idx * 8u).x;
}
PT_FN uint32_t scatter_class(uint32_t mbits, bool back) {
    uint32_t type = mbits & 7u;
    if (type == MAT_GLASS) return SC_GLASS;
    if ((type == MAT_LAMBERTIAN || type == MAT_METAL) && !back) return SC_OPAQUE;
    return SC_TERMINAL;
}

// compute.wgsl:296-300, 303-305, 313-315, 356-358, 361-363: what trace_path returns.
template <class Mem>
PT_FN V3 scatter_terminal(const Mem& mem, const SceneLayout& L, uint32_t mbits, const PathState& ps) {
    uint32_t type = mbits & 7u;
    if (type == MAT_SKY) return ps.colour * sky(ps.d.y);
    if (type == MAT_LIGHT) {
        f4 m = mem.template load<f4>(L.off_lights + (mbits >> 3) * 16u);
        return ps.colour * mk(m.x, m.y, m.z);
    }
    return mk(0.0f, 0.0f, 0.0f);
}

// compute.wgsl:138-147 (hit position, outward / flipped normal) for the sphere that was hit.
template <class Mem>
PT_FN void hit_frame(const Mem& mem, const SceneLayout& L, const Hit& hit, const PathState& ps, V3& pos, V3& norm) {
    f4 s = mem.template load<f4>(L.off_spheres + (uint32_t)hit.idx * 16u);
    pos = ps.o + (ps.d * hit.t);
    norm = normalize3(pos - mk(s.x, s.y, s.z));
    if (hit.back) norm = -norm;
}

// Loop bookkeeping after a scatter (compute.wgsl:365 and the loop condition at :289).  Returns true
// when the path ends because the depth ran out; trace_path then returns colour * (0,0,0).
PT_FN bool next_bounce(PathState& ps, int32_t depth, V3& radiance) {
    ps.bounce += 1;
    if (ps.bounce > depth) {
        radiance = ps.colour * mk(0.0f, 0.0f, 0.0f);
        return true;
    }
    return false;
}

// compute.wgsl:302-326.  Lambertian: normalize(norm + r).  Metal: normalize(reflect(d, norm) + r * roughness).
// Both are computed as normalize(base + r * scale) with scale = 1 for lambertian, which is the same
// value (r * 1.0f == r exactly).
// `r` is the rand_in_sphere point of this bounce; the kernels draw it in separately scheduled
// single-try steps (the only random numbers an opaque bounce consumes, so the order is unchanged).
template <class Mem>
PT_FN bool scatter_opaque_with(const Mem& mem, const SceneLayout& L, const Hit& hit, uint32_t mbits, V3 r,
                               PathState& ps, int32_t depth, V3& radiance) {
    V3 pos, norm;
    hit_frame(mem, L, hit, ps, pos, norm);
    const bool metal = (mbits & 7u) == MAT_METAL;
    f4 m = mem.template load<f4>((metal ? L.off_metals : L.off_lambertians) + (mbits >> 3) * 16u);
    V3 base = metal ? reflect3(ps.d, norm) : norm;
    float scale = metal ? m.w : 1.0f;
    V3 dir = normalize3(base + r * scale);
    ps.o = pos + norm * 0.001f;
    ps.d = dir;
    ps.colour = ps.colour * mk(m.x, m.y, m.z);
    return next_bounce(ps, depth, radiance);
}
template <class Mem>
PT_FN bool scatter_opaque(const Mem& mem, const SceneLayout& L, const Hit& hit, uint32_t mbits, PathState& ps,
                          int32_t depth, V3& radiance) {
    V3 r = rand_in_sphere(ps.rng);
    return scatter_opaque_with(mem, L, hit, mbits, r, ps, depth, radiance);
}

// compute.wgsl:328-355
template <class Mem>
PT_FN bool scatter_glass(const Mem& mem, const SceneLayout& L, const Hit& hit, uint32_t mbits, PathState& ps,
                         int32_t depth, bool glass_always_draws, V3& radiance) {
    V3 pos, norm;
    hit_frame(mem, L, hit, ps, pos, norm);
    const float dist = 0.001f;
    const f4 g = mem.template load<f4>(L.off_glass + (mbits >> 3) * 16u);
    float ir = hit.back ? g.x : g.y;
    const float r0 = hit.back ? g.z : g.w;
    float cos_theta = fminf(dot3(-ps.d, norm), 1.0f);
    float sin_theta = sqrt_rn(1.0f - cos_theta * cos_theta);
    bool cannot_refract = ir * sin_theta > 1.0f;
    bool do_reflect = cannot_refract;
    if (glass_always_draws || !cannot_refract) {
        float refl = reflectance_from_r0(cos_theta, r0);
        float rnd = randf(ps.rng);
        do_reflect = cannot_refract || refl > rnd;
    }
    V3 dir, np;
    if (do_reflect) {
        dir = reflect3(ps.d, norm);
        np = pos + norm * dist;
    } else {
        dir = refract3(ps.d, norm, ir);
        np = pos - norm * dist;
    }
    ps.o = np;
    ps.d = normalize3(dir);
    return next_bounce(ps, depth, radiance);
}

// All of the above for one ray.  Returns true when the path ends; `radiance` is then what
// trace_path returns.  mat_out: 0..3 material, 4 sky, 5 unknown.
template <class Mem>
PT_FN bool scatter(const Mem& mem, const SceneLayout& L, const Hit& hit, PathState& ps, int32_t depth,
                   bool glass_always_draws, V3& radiance, uint32_t& mat_out) {
    const uint32_t mbits = hit_material(mem, L, hit);
    const uint32_t type = mbits & 7u;
    mat_out = type == MAT_SKY ? 4u : (type == MAT_UNKNOWN ? 5u : type);
    const uint32_t cls = scatter_class(mbits, hit.back);
    if (cls == SC_OPAQUE) return scatter_opaque(mem, L, hit, mbits, ps, depth, radiance);
    if (cls == SC_GLASS) return scatter_glass(mem, L, hit, mbits, ps, depth, glass_always_draws, radiance);
    radiance = scatter_terminal(mem, L, mbits, ps);
    return true;
}

// n / d and n % d for n, d < 2^24 with inv = 1.0f / d: float quotient, then one step of exact
// integer correction either way (the float product is within one of the true quotient).
PT_FN void divmod24(uint32_t n, uint32_t d, float inv, uint32_t& q, uint32_t& r) {
    uint32_t qq = (uint32_t)((float)n * inv);
    int32_t rr = (int32_t)(n - qq * d);
    if (rr < 0) qq -= 1u, rr += (int32_t)d;
    if (rr >= (int32_t)d) qq += 1u, rr -= (int32_t)d;
    q = qq;
    r = (uint32_t)rr;
}

// 8x8 pixel tiles: work unit u of a frame covers tile (u % tiles_x, u / tiles_x); slot in [0,64).
PT_FN void tile_pixel(uint32_t unit, uint32_t slot, uint32_t tiles_x, uint32_t& px, uint32_t& py) {
    uint32_t tx = unit % tiles_x, ty = unit / tiles_x;
    px = tx * 8u + (slot & 7u);
    py = ty * 8u + (slot >> 3);
}

// ---- camera rays: per-tile candidate lists ------------------------------------------------------
// Every camera ray of an 8x8 tile starts at the camera position and points into the cone spanned by
// the tile's four corner directions, whatever the frame and its jitter (compute.wgsl:405-413: the
// jittered pixel position stays inside the pixel).  Which spheres such a ray can reach is therefore a
// property of camera, tile and scene, not of the frame: the cull that a wave-uniform walk of the tree
// would repeat identically in every frame is done once per camera instead, and its outcome -- the few
// spheres whose padded ball meets the cone -- is kept as a short list per tile.  A camera ray then tests
// exactly those spheres (the same sphere arithmetic, so the same closest hit: ties resolve by rank
// whatever the order) and walks no tree.  Like the box tests of the walk, the cull never reaches a pixel
// value; it keeps the reference's 0.01 padding (bounding_box.rs:17-27) plus a slack for the rounding of
// the direction arithmetic.  Tiles whose list would be long (the horizon of a large plane) keep the walk.
constexpr uint32_t TILE_LIST_WORDS = 16;                    // per tile: count, then up to 15 leaf-order sphere indices
constexpr uint32_t TILE_LIST_CAP = TILE_LIST_WORDS - 1u;
constexpr uint32_t TILE_LIST_WALK = 0xffffffffu;            // count word: camera rays of this tile walk the tree
struct TileCone {
    double n[4][3];   // unit normals of the four side planes through the camera position, pointing inwards
    double org[3];    // camera position
    double axis[3];   // unit direction through the middle of the tile
    double cos_a, sin_a;  // of the largest angle between the axis and a corner direction, widened by ang_err
    double ang_err;   // bound on the angle between a computed (binary32) direction and the exact one
    bool valid;
};
PT_FN void tile_cone(const Camera64& cam, uint32_t tx, uint32_t ty, uint32_t w, uint32_t h, TileCone& c) {
    const double margin = 0.01;  // pixels; covers the rounding of (px + r) / w and of 2u - 1 (< 1e-3 pixel at 65535)
    const double x0 = 8.0 * tx - margin, x1 = 8.0 * tx + 8.0 + margin, y0 = 8.0 * ty - margin, y1 = 8.0 * ty + 8.0 + margin;
    const double nx[2] = {2.0 * x0 / w - 1.0, 2.0 * x1 / w - 1.0}, ny[2] = {-(2.0 * y0 / h - 1.0), -(2.0 * y1 / h - 1.0)};
    double d[4][3], dc[3], mag = 0.0;
    for (int a = 0; a < 3; a++) {
        const double base = (double)cam.center[a] - (double)cam.pos[a], hx = cam.horizontal[a], vy = cam.vertical[a];
        d[0][a] = base + hx * nx[0] + vy * ny[0];  // corners in order around the tile
        d[1][a] = base + hx * nx[1] + vy * ny[0];
        d[2][a] = base + hx * nx[1] + vy * ny[1];
        d[3][a] = base + hx * nx[0] + vy * ny[1];
        dc[a] = base + hx * 0.5 * (nx[0] + nx[1]) + vy * 0.5 * (ny[0] + ny[1]);
        c.org[a] = cam.pos[a];
        mag += fabs((double)cam.center[a]) + fabs((double)cam.pos[a]) + fabs(hx) + fabs(vy);
    }
    bool ok = true;
    double min_len = INFINITY;
    for (int k = 0; k < 4; k++) {
        const double* A = d[k];
        const double* B = d[(k + 1) & 3];
        double n[3] = {A[1] * B[2] - A[2] * B[1], A[2] * B[0] - A[0] * B[2], A[0] * B[1] - A[1] * B[0]};
        const double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        const double la = sqrt(A[0] * A[0] + A[1] * A[1] + A[2] * A[2]);
        min_len = fmin(min_len, la);
        if (!(len > 1e-300) || !(la > 1e-30) || !(len < INFINITY)) {
            ok = false;
            break;
        }
        const double s = (n[0] * dc[0] + n[1] * dc[1] + n[2] * dc[2]) < 0.0 ? -1.0 / len : 1.0 / len;
        for (int a = 0; a < 3; a++) c.n[k][a] = n[a] * s;
        // the other two corners and the centre lie strictly inside: the four half-spaces bound a convex cone
        const double* C = d[(k + 2) & 3];
        const double* D = d[(k + 3) & 3];
        const double ic = c.n[k][0] * C[0] + c.n[k][1] * C[1] + c.n[k][2] * C[2];
        const double id = c.n[k][0] * D[0] + c.n[k][1] * D[1] + c.n[k][2] * D[2];
        const double icen = c.n[k][0] * dc[0] + c.n[k][1] * dc[1] + c.n[k][2] * dc[2];
        if (!(ic > 0.0) || !(id > 0.0) || !(icen > 0.0)) ok = false;
        if (!((A[0] * dc[0] + A[1] * dc[1] + A[2] * dc[2]) > 0.0)) ok = false;  // narrower than a half-space
    }
    // center + horizontal * nx + vertical * ny - pos in binary32: four roundings of terms of size `mag`
    c.ang_err = ok ? 4.0 * 1.2e-7 * mag / min_len + 1e-6 : 0.0;
    c.valid = ok && c.ang_err < 1e-3;
    // The circular cone around the tile: every direction of the tile lies within its angle of the axis (the
    // directions are convex combinations of the corner directions, and the angle to a fixed axis is largest at
    // a corner).  The four planes alone do not bound the cone behind the camera, where a huge ball -- the ground
    // sphere of every shipped scene -- crosses all of them although no ray of the tile can reach it.
    if (c.valid) {
        const double lc = sqrt(dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2]);
        double cos_min = 1.0;
        for (int a = 0; a < 3; a++) c.axis[a] = dc[a] / lc;
        for (int k = 0; k < 4; k++) {
            const double l = sqrt(d[k][0] * d[k][0] + d[k][1] * d[k][1] + d[k][2] * d[k][2]);
            cos_min = fmin(cos_min, (d[k][0] * c.axis[0] + d[k][1] * c.axis[1] + d[k][2] * c.axis[2]) / l);
        }
        // widen by the direction error and a little more: cos(alpha + e) >= cos(alpha) - e
        c.cos_a = fmax(-1.0, cos_min - c.ang_err - 1e-9);
        c.sin_a = sqrt(fmax(0.0, 1.0 - c.cos_a * c.cos_a));
        c.valid = c.cos_a > 0.0;  // narrower than a right angle (any sensible tile is far narrower)
    }
}
// May a camera ray of the tile reach the ball of radius `rad` around (cx, cy, cz)?  (Never false for one that can.)
PT_FN bool tile_cone_meets_ball(const TileCone& c, double cx, double cy, double cz, double rad) {
    const double ox = cx - c.org[0], oy = cy - c.org[1], oz = cz - c.org[2];
    const double dist = sqrt(ox * ox + oy * oy + oz * oz);
    // The ball a ray must miss for the sphere test to be sure to miss: the sphere, the reference's 0.01 padding, and
    // what the test's own rounding can add -- its discriminant hb^2 - (|oc|^2 - r^2) carries a few units in the last
    // place of |oc|^2, i.e. about 4e-7 dist^2 in the squared distance of the ray from the centre, which is
    // 4e-7 dist^2 / (2 r) in that distance (twice that here).  The direction error widens the cone, not the ball: as
    // an angle it is a length ang_err * dist at the centre's distance for the planes, and part of cos_a for the cone.
    const double reach = rad + 0.01 + 4e-7 * dist * dist / fmax(rad, 1e-9);
    const double reach_planes = reach + c.ang_err * dist;
    for (int k = 0; k < 4; k++)
        if (c.n[k][0] * ox + c.n[k][1] * oy + c.n[k][2] * oz < -reach_planes) return false;
    // against the circular cone: the ball (angular radius beta as seen from the camera, centre at angle phi from
    // the axis) is out of reach when phi > alpha + beta, i.e. cos(phi) < cos(alpha + beta); both angles are below
    // a right angle, so the comparison of cosines is the comparison of angles
    if (dist > reach) {
        const double sin_b = reach / dist, cos_b = sqrt(fmax(0.0, 1.0 - sin_b * sin_b));
        const double cos_phi = (ox * c.axis[0] + oy * c.axis[1] + oz * c.axis[2]) / dist;
        if (cos_phi < c.cos_a * cos_b - c.sin_a * sin_b - 1e-9) return false;
    }
    return true;
}
// The same for an axis-aligned box (a node's child box, which already holds its spheres' padding).
PT_FN bool tile_cone_meets_box(const TileCone& c, const float lo[3], const float hi[3]) {
    double far2 = 0.0, l[3], u[3];
    for (int a = 0; a < 3; a++) {
        l[a] = (double)lo[a] - c.org[a], u[a] = (double)hi[a] - c.org[a];
        const double m = fmax(fabs(l[a]), fabs(u[a]));
        far2 += m * m;
    }
    const double slack = c.ang_err * sqrt(far2) + 1e-9;
    for (int k = 0; k < 4; k++) {
        double best = 0.0;  // the corner furthest along the inward normal
        for (int a = 0; a < 3; a++) best += fmax(c.n[k][a] * l[a], c.n[k][a] * u[a]);
        if (best < -slack) return false;
    }
    return true;
}
// Candidate list of tile (tx, ty) into rec[TILE_LIST_WORDS]: rec[0] = count <= max_list and the leaf-order
// indices behind it, or rec[0] = TILE_LIST_WALK.  Walks the Node64 tree when the blob holds one, else (binary16
// nodes: scenes of at most a few thousand spheres) looks at every sphere.
template <class Mem>
PT_FN void tile_list_build(const Mem& mem, const SceneLayout& L, const Camera64& cam, uint32_t tx, uint32_t ty, uint32_t w,
                           uint32_t h, uint32_t max_list, uint32_t* rec) {
    TileCone cone;
    tile_cone(cam, tx, ty, w, h, cone);
    if (!cone.valid) {
        rec[0] = TILE_LIST_WALK;
        return;
    }
    uint32_t count = 0;
    bool overflow = false;
    auto offer = [&](uint32_t si) {
        const f4 s = mem.template load<f4>(L.off_spheres + si * 16u);  // x, y, z, rad * rad
        if (!tile_cone_meets_ball(cone, s.x, s.y, s.z, sqrt((double)s.w))) return;
        if (count >= max_list) {
            overflow = true;
            return;
        }
        rec[1u + count++] = si;
    };
    if (L.node_half) {
        for (uint32_t si = 0; si < L.n_spheres && !overflow; si++) offer(si);
    } else {
        if (L.n_hoisted) offer(L.n_spheres - 1u);  // in no leaf: the walk below cannot find it
        constexpr int STACK = 64;
        int32_t stack[STACK];
        int sp = 0;
        stack[sp++] = 0;
        while (sp > 0 && !overflow) {
            const int32_t link = stack[--sp];
            if (link < 0) {
                const uint32_t ref = ~(uint32_t)link, first = ref & LEAF_FIRST_MASK, n = (ref >> LEAF_COUNT_SHIFT) + 1u;
                for (uint32_t k = 0; k < n; k++) offer(first + k);
                continue;
            }
            f4 q0, q1, q2;
            u2 ch;
            mem.node64(L.off_nodes + (uint32_t)link * 64u, q0, q1, q2, ch);
            const float lo0[3] = {q0.x, q0.y, q0.z}, hi0[3] = {q0.w, q1.x, q1.y};
            const float lo1[3] = {q1.z, q1.w, q2.x}, hi1[3] = {q2.y, q2.z, q2.w};
            const bool in0 = tile_cone_meets_box(cone, lo0, hi0), in1 = tile_cone_meets_box(cone, lo1, hi1);
            if (sp + 2 > STACK) {  // deeper than any tree the builders make: keep the walk
                overflow = true;
                break;
            }
            if (in0) stack[sp++] = (int32_t)ch.x;
            if (in1) stack[sp++] = (int32_t)ch.y;
        }
    }
    rec[0] = overflow ? TILE_LIST_WALK : count;
}
// One frame of a pixel whose tile has an empty candidate list: the camera ray cannot reach a sphere, so the path is
// jitter, camera ray, sky (compute.wgsl:405-415 with trace_path returning colour * miss(dir.y), colour = 1).  The
// value is what the path kernels hand to the accumulation: final_colour = 0 + colour * sky.
PT_FN V3 sky_pixel_frame(const CameraRegs& cam, uint32_t px, uint32_t py, PixelDiv fw, PixelDiv fh, uint32_t frame_seed, uint32_t hash) {
    uint32_t rng = pixel_seed(frame_seed, hash);
    const V3 d = camera_dir(cam, px, py, fw, fh, rng);
    const V3 radiance = mk(1.0f, 1.0f, 1.0f) * sky(d.y);
    return mk(0.0f, 0.0f, 0.0f) + radiance;
}

// Closest hit of a camera ray over its tile's candidate list: what the walk finds, without the walk.
template <class Mem, bool COUNT>
PT_FN void walk_tile_list(const Mem& mem, const SceneLayout& L, Walk& w, V3 o, V3 d, const uint32_t* rec, Counters& cnt) {
    const uint32_t n = rec[0];
    for (uint32_t k = 0; k < n; k++) walk_test_sphere<Mem, COUNT>(mem, L, w, o, d, rec[1u + k], cnt);
    w.cur = WALK_FINISHED;
}

}  // namespace pt
