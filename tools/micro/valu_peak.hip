// Issue-rate probes for gfx950: how many wave-level vector instructions per cycle and per SIMD the
// hardware sustains for the instruction kinds the path kernel is made of.
//   hipcc --offload-arch=gfx950 -O3 -o valu_peak valu_peak.hip && ./valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void __launch_bounds__(256) probe(float* out, int iters, uint32_t lane_mask_lo, uint32_t lane_mask_hi) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    uint32_t h = 0x3c003c00u;
    uint32_t s0 = iters, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
    uint64_t m0 = 0, m1 = 0;
    const uint64_t mask = ((uint64_t)lane_mask_hi << 32) | lane_mask_lo;
    const bool on = (mask >> (threadIdx.x & 63)) & 1;
    if (on) {
        for (int i = 0; i < iters; i++) {
            if (KIND == 0) {
                REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                                   "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                                   "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                   : "v"(b), "v"(c));)
            } else if (KIND == 1) {
                REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n"
                                   "v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                                   "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n"
                                   "v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                                   : "v"(pb), "v"(pc));)
            } else if (KIND == 2) {
                REP16(asm volatile("v_fma_mix_f32 %0, %8, %0, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %1, %9 op_sel_hi:[1,0,0]\n"
                                   "v_fma_mix_f32 %2, %8, %2, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %3, %9 op_sel_hi:[1,0,0]\n"
                                   "v_fma_mix_f32 %4, %8, %4, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %5, %9 op_sel_hi:[1,0,0]\n"
                                   "v_fma_mix_f32 %6, %8, %6, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %7, %9 op_sel_hi:[1,0,0]\n"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                   : "v"(h), "v"(c));)
            } else if (KIND == 3) {  // min/max/cndmask mix as in the slab test
                REP16(asm volatile("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_min_f32 %2, %2, %9\n"
                                   "v_max_f32 %3, %3, %9\n v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n"
                                   "v_min_f32 %6, %6, %8\n v_max_f32 %7, %7, %9\n"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                   : "v"(b), "v"(c));)
            } else if (KIND == 5) {  // scalar ALU only, 8 independent registers
                REP16(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                                   "s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"
                                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7)
                                   :
                                   : "scc");)
            } else if (KIND == 6) {  // the path kernel's mix: two vector instructions, one scalar (counted: the 8 vector ones)
                REP16(asm volatile("v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n s_add_u32 %8, %8, 1\n"
                                   "v_fma_f32 %2, %2, %12, %13\n v_fma_f32 %3, %3, %12, %13\n s_add_u32 %9, %9, 1\n"
                                   "v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n s_add_u32 %10, %10, 1\n"
                                   "v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n s_add_u32 %11, %11, 1\n"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0),
                                     "+s"(s1), "+s"(s2), "+s"(s3)
                                   : "v"(b), "v"(c)
                                   : "scc");)
            } else if (KIND == 7) {  // one scalar per vector instruction
                REP16(asm volatile("v_fma_f32 %0, %0, %12, %13\n s_add_u32 %8, %8, 1\n v_fma_f32 %1, %1, %12, %13\n s_add_u32 %9, %9, 1\n"
                                   "v_fma_f32 %2, %2, %12, %13\n s_add_u32 %10, %10, 1\n v_fma_f32 %3, %3, %12, %13\n s_add_u32 %11, %11, 1\n"
                                   "v_fma_f32 %4, %4, %12, %13\n s_add_u32 %8, %8, 1\n v_fma_f32 %5, %5, %12, %13\n s_add_u32 %9, %9, 1\n"
                                   "v_fma_f32 %6, %6, %12, %13\n s_add_u32 %10, %10, 1\n v_fma_f32 %7, %7, %12, %13\n s_add_u32 %11, %11, 1\n"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0),
                                     "+s"(s1), "+s"(s2), "+s"(s3)
                                   : "v"(b), "v"(c)
                                   : "scc");)
            } else if (KIND == 8) {  // vector compare + exec-mask bookkeeping, as around every divergent branch
                REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n s_and_saveexec_b64 %2, vcc\n v_fma_f32 %0, %0, %4, %5\n s_or_b64 exec, exec, %2\n"
                                   "v_cmp_lt_f32 vcc, %1, %4\n s_and_saveexec_b64 %3, vcc\n v_fma_f32 %1, %1, %4, %5\n s_or_b64 exec, exec, %3\n"
                                   : "+v"(a0), "+v"(a1), "=s"(m0), "=s"(m1)
                                   : "v"(b), "v"(c)
                                   : "vcc", "scc");)
            } else if (KIND == 4) {  // one dependent chain: latency per instruction
                REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                                   "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                                   "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                                   : "+v"(a0)
                                   : "v"(b), "v"(c));)
            }
        }
    }
    a0 += (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7) + (float)(m0 + m1);
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int KIND>
static void run(const char* name, int waves_per_simd, uint64_t mask, float* out, int cus, double ghz_hint) {
    const int iters = 2000;
    const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, (uint32_t)mask, (uint32_t)(mask >> 32));
        hipEventRecord(e1);
        const hipError_t le = hipGetLastError(), se = hipEventSynchronize(e1);
        if (le != hipSuccess || se != hipSuccess) printf("  launch: %s, sync: %s\n", hipGetErrorString(le), hipGetErrorString(se));
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 16 * 8;
    const double per_simd = instr_per_wave * waves_per_simd;
    const double cycles = ms * 1e-3 * ghz_hint * 1e9;
    printf("%-28s waves/SIMD %d  lanes %2d  %8.3f ms  %.3f wave-instr/cycle/SIMD (at %.2f GHz)  %.2f cycles/instr/wave\n", name,
           waves_per_simd, __builtin_popcountll(mask), ms, per_simd / cycles, ghz_hint, cycles / instr_per_wave);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", prop.name, cus, ghz);
    float* out;
    hipMalloc(&out, 4096);
    const uint64_t all = ~0ull;
    for (int w : {1, 2, 4, 8}) run<0>("v_fma_f32 x8 independent", w, all, out, cus, ghz);
    for (int w : {1, 4}) run<0>("v_fma_f32, lanes 0-31 only", w, 0xffffffffull, out, cus, ghz);
    for (int w : {1, 4}) run<0>("v_fma_f32, lanes 0-15 only", w, 0xffffull, out, cus, ghz);
    for (int w : {1, 4}) run<0>("v_fma_f32, every 4th lane", w, 0x1111111111111111ull, out, cus, ghz);
    for (int w : {1, 2, 4, 8}) run<1>("v_pk_fma_f32 x4 independent", w, all, out, cus, ghz);
    for (int w : {1, 4}) run<2>("v_fma_mix_f32 x8", w, all, out, cus, ghz);
    for (int w : {1, 4}) run<3>("min/max/mul/add x8", w, all, out, cus, ghz);
    for (int w : {1, 2, 4, 8}) run<4>("v_fma_f32 dependent chain", w, all, out, cus, ghz);
    for (int w : {1, 2, 4, 6, 8}) run<5>("s_add_u32 x8 independent", w, all, out, cus, ghz);
    for (int w : {1, 2, 4, 6, 8}) run<6>("2 v_fma : 1 s_add (per v_fma)", w, all, out, cus, ghz);
    for (int w : {1, 2, 4, 6, 8}) run<7>("1 v_fma : 1 s_add (per v_fma)", w, all, out, cus, ghz);
    for (int w : {1, 2, 4, 6}) run<8>("cmp+saveexec+fma+or (per 8 = 2 groups)", w, all, out, cus, ghz);
    return 0;
}
