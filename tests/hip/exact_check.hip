// Exhaustive check of the device's short correctly rounded sqrt / reciprocal sequences
// (pt_device.h: sqrt_rn, rcp_rn, rcp_sqrt_rn) against the compiler's general expansions, over all
// 2^32 binary32 bit patterns.  Test infrastructure: built by __graft_entry__.build(), loaded by
// tests/test_gpu_exact_math.py through ctypes.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../pathtracer_amd/csrc/pt_device.h"

namespace {

template <int WHICH>
__global__ void check(unsigned long long* out) {  // out[0] = mismatches, out[1] = smallest mismatching pattern
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;  // 2^22 threads x 1024 patterns
    unsigned long long bad = 0, first = ~0ull;
    for (uint32_t k = 0; k < 1024u; k++) {
        // consecutive lanes take consecutive patterns, so whole waves sit inside the ordinary range
        const uint32_t u = (t & 63u) | (k << 6) | ((t >> 6) << 16);
        const float x = __uint_as_float(u);
        float a, b;
        if (WHICH == 0) a = pt::sqrt_rn(x), b = sqrtf(x);
        else if (WHICH == 1) a = pt::rcp_rn(x), b = 1.0f / x;
        else if (WHICH == 2) a = pt::rcp_sqrt_rn(x), b = 1.0f / sqrtf(x);
        else if (WHICH == 3) a = __builtin_amdgcn_rcpf(x), b = 1.0f / x;  // control: the raw instructions are
        else a = __builtin_amdgcn_sqrtf(x), b = sqrtf(x);                 // NOT correctly rounded
        const bool same = __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
        if (!same) {
            bad++;
            if (u < first) first = u;
        }
    }
    if (bad) {
        atomicAdd(&out[0], bad);
        atomicMin(&out[1], first);
    }
}

}  // namespace

// which: 0 sqrt, 1 reciprocal, 2 reciprocal square root; 3 / 4 controls: the raw v_rcp_f32 / v_sqrt_f32, which
// must show mismatches (they are accurate to 1 ulp only).  result[0] = number of mismatching patterns,
// result[1] = the smallest one.  Returns 0 on success, a HIP error code otherwise.
extern "C" int exact_check_run(int which, unsigned long long* result) {
    unsigned long long* dev = nullptr;
    hipError_t e = hipMalloc(&dev, 16);
    if (e != hipSuccess) return (int)e;
    const unsigned long long init[2] = {0ull, ~0ull};
    e = hipMemcpy(dev, init, 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const dim3 grid(1u << 14), block(256);
        if (which == 0) hipLaunchKernelGGL(check<0>, grid, block, 0, 0, dev);
        else if (which == 1) hipLaunchKernelGGL(check<1>, grid, block, 0, 0, dev);
        else if (which == 2) hipLaunchKernelGGL(check<2>, grid, block, 0, 0, dev);
        else if (which == 3) hipLaunchKernelGGL(check<3>, grid, block, 0, 0, dev);
        else hipLaunchKernelGGL(check<4>, grid, block, 0, 0, dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(result, dev, 16, hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    return (int)e;
}
