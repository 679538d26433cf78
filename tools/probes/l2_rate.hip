// Per-CU L2 -> CU streaming rate probe (gfx950): LDS-DMA (global_load_lds_dwordx4) vs global_load_dwordx4 into VGPRs.
// Every block of an XCD streams the same 2-MiB region (L2 hits after the first pass), 512 threads, 1 block per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/l2_rate tools/probes/l2_rate.hip ; run: gpurun_out/l2_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int UNROLL>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, size_t region, int passes, u32x4* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)(blockIdx.x & 7) * region;  // one region per XCD
    u32x4 accv = {0, 0, 0, 0};
    const size_t step = 512 * 16;  // bytes per block-wide instruction
    for (int p = 0; p < passes; ++p) {
        for (size_t off = 0; off < region; off += step * UNROLL) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const char* g = base + off + u * step + tid * 16;
                if (MODE == 0) {
                    char* dst = smem + ((u & 7) * 512 + wave * 64) * 16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                } else {
                    u32x4 v = *(const u32x4*)g;
                    accv ^= v;
                }
            }
            if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 1 && accv.x == 0x12345678u) sink[0] = accv;
}

// GEMM-like: a [512 rows][ld bytes] operand panel per XCD; one k-tile = 512 rows x 128 B = 8 pieces per thread, each wave
// instruction covers 8 rows x 128 B (rows ld bytes apart); SW = 1 applies the chunk ^= (row >> 1) & 7 source swizzle
template <int SW>
__global__ __launch_bounds__(512) void probe_rows(const char* __restrict__ src, size_t ld, int ktiles, int passes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)(blockIdx.x & 7) * 512 * ld;
    const int r0 = wave * 8 + (lane >> 3);
    const int c = SW ? ((lane & 7) ^ ((r0 >> 1) & 7)) : (lane & 7);
    const char* p0 = base + (size_t)r0 * ld + c * 16;
    for (int p = 0; p < passes; ++p)
        for (int t = 0; t < ktiles; ++t) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                char* dst = smem + (t & 1) * 65536 + (i * 512 + wave * 64) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p0 + (size_t)i * 64 * ld + t * 128),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int SW>
static void run_rows(const char* name, const char* d, size_t ld, int grid) {
    const int passes = 16, ktiles = (int)(ld / 128);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)probe_rows<SW>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    probe_rows<SW><<<grid, 512, 131072>>>(d, ld, ktiles, 2);
    hipEventRecord(e0);
    probe_rows<SW><<<grid, 512, 131072>>>(d, ld, ktiles, passes);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * 512 * ld * passes;
    printf("%-20s ld=%6zu grid=%4d: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU @2.1GHz\n", name, ld, grid, ms, bytes / ms / 1e9,
           bytes / ms / 1e6 / (grid < 256 ? grid : 256) / 2.1);
}

template <int MODE, int UNROLL>
static void run(const char* name, const char* d, size_t region, int grid, u32x4* sink) {
    const int passes = 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)probe<MODE, UNROLL>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    probe<MODE, UNROLL><<<grid, 512, 65536>>>(d, region, 2, sink);
    hipEventRecord(e0);
    probe<MODE, UNROLL><<<grid, 512, 65536>>>(d, region, passes, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * region * passes;
    printf("%-28s grid=%4d: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU @2.1GHz\n", name, grid, ms, bytes / ms / 1e9,
           bytes / ms / 1e6 / (grid < 256 ? grid : 256) / 2.1);
}

int main() {
    const size_t region = 2u << 20;
    char* d;
    u32x4* sink;
    hipMalloc(&d, 8 * 512 * (size_t)24576 + 4096);
    hipMalloc(&sink, 64);
    hipMemset(d, 1, 8 * region);
    for (int grid : {64, 256}) {
        run<0, 4>("LDS-DMA dwordx4 unroll4", d, region, grid, sink);
        run<0, 8>("LDS-DMA dwordx4 unroll8", d, region, grid, sink);
        run<1, 4>("global_load_dwordx4 unroll4", d, region, grid, sink);
        run<1, 8>("global_load_dwordx4 unroll8", d, region, grid, sink);
    }
    for (int grid : {64, 256})
        for (size_t ld : {(size_t)6144, (size_t)6144 + 128, (size_t)24576, (size_t)24576 + 128}) {
            run_rows<0>("rows 8x128B", d, ld, grid);
            run_rows<1>("rows 8x128B swizzled", d, ld, grid);
        }
    return 0;
}
