// Back-to-back v_mfma_f32_32x32x16_bf16 on NCH independent accumulator chains: cycles per MFMA for one wave per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_chain tools/probes/mfma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NCH>
__global__ __launch_bounds__(256) void probe(float* out, int iters, long long* cyc) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
    f32x16 acc[NCH];
    for (int i = 0; i < NCH; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i % NCH], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NCH; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NCH> static void run() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&cyc, 64);
    probe<NCH><<<256, 256>>>(out, 10, cyc);
    probe<NCH><<<256, 256>>>(out, 2000, cyc);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d accumulator chain(s): %.2f cycles per MFMA\n", NCH, (double)c / 2000 / 16);
}
int main() { run<1>(); run<2>(); run<3>(); run<4>(); run<8>(); return 0; }
