// Random-gather ceilings for gfx950: every lane follows a chain of dependent record fetches at
// pseudo-random indices of a table, the access pattern of a BVH walk.  Reports records/s and GB/s by
// table size (which level of the memory hierarchy serves it), record size and resident waves.
//   hipcc --offload-arch=gfx950 -O3 -o gather gather.hip && ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

typedef uint32_t w4 __attribute__((ext_vector_type(4)));
typedef uint32_t w2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}

// REC = 64: three 16-byte loads and one 8-byte load of one record; REC = 32: two 16-byte loads.
// LDS: the table (<= 64 KB) is first copied to LDS and gathered from there.
template <int REC, bool LDS>
__global__ void __launch_bounds__(256) gather(const char* table, uint32_t n_rec_mask, int steps, uint32_t* out) {
    extern __shared__ __align__(16) char smem[];
    if (LDS) {
        const uint32_t bytes = (n_rec_mask + 1u) * REC;
        for (uint32_t i = threadIdx.x * 16u; i < bytes; i += 256u * 16u)
            *reinterpret_cast<w4*>(smem + i) = *reinterpret_cast<const w4*>(table + i);
        __syncthreads();
    }
    const char* base = LDS ? smem : table;
    uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) & n_rec_mask;
    uint32_t acc = 0;
    for (int s = 0; s < steps; s++) {
        const char* p = base + (size_t)idx * REC;
        w4 a = *reinterpret_cast<const w4*>(p);
        w4 b = *reinterpret_cast<const w4*>(p + 16);
        uint32_t v = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
        if (REC == 64) {
            w4 c = *reinterpret_cast<const w4*>(p + 32);
            w2 d = *reinterpret_cast<const w2*>(p + 48);
            v ^= c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y;
        }
        acc += v;
        idx = mix(idx + v + (uint32_t)s) & n_rec_mask;  // next index depends on the data just loaded
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int REC, bool LDS>
static void run(const char* table, size_t table_bytes, int waves_per_simd, int cus, uint32_t* out) {
    const uint32_t n_rec = (uint32_t)(table_bytes / REC);
    const int steps = table_bytes > (256u << 20) ? 256 : 1024;
    const int blocks = cus * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((gather<REC, LDS>), dim3(blocks), dim3(256), LDS ? table_bytes : 0, 0, table, n_rec - 1u, steps, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double recs = (double)blocks * 256.0 * steps;
    printf("%-6s rec %2d B  table %10.3f MiB  waves/SIMD %d  %8.3f ms  %8.1f Grec/s  %8.1f GB/s\n", LDS ? "LDS" : "global", REC,
           table_bytes / 1048576.0, waves_per_simd, best, recs / best * 1e-6, recs * REC / best * 1e-6);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs\n", prop.name, cus);
    const size_t max_bytes = (size_t)1 << 30;
    std::vector<uint32_t> host(max_bytes / 4);
    uint32_t s = 12345u;
    for (size_t i = 0; i < host.size(); i++) {
        s ^= s << 13, s ^= s >> 17, s ^= s << 5;
        host[i] = s;
    }
    char* table;
    uint32_t* out;
    CHECK(hipMalloc(&table, max_bytes));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemcpy(table, host.data(), max_bytes, hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gather<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gather<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int w : {2, 4}) {   // two 32 KB LDS copies per CU at 2 workgroups
        run<64, true>(table, 32u << 10, w, cus, out);
        run<32, true>(table, 32u << 10, w, cus, out);
    }
    const size_t sizes[] = {16u << 10, 256u << 10, 2u << 20, 16u << 20, 128u << 20, (size_t)1 << 30};
    for (size_t bytes : sizes)
        for (int w : {4, 8}) {
            run<64, false>(table, bytes, w, cus, out);
            run<32, false>(table, bytes, w, cus, out);
        }
    return 0;
}
