// Price of candidate softmax fillers in the attn_q4 regime (ONE wave per SIMD, v_mfma_f32_32x32x16_bf16 back to back, NV fillers behind each
// MFMA): cycles per MFMA for a stream of NV x <op>, against NV x v_add_f32.  Candidates for cheaper row sums: packed f16 adds and the dot2 forms.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/filler_price tools/probes/filler_price.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define OPS(X) \
    X(0, "v_add_f32 %0, %0, %1", "v_add_f32") \
    X(1, "v_pk_add_f16 %0, %0, %1", "v_pk_add_f16") \
    X(2, "v_dot2_f32_f16 %0, %1, %1, %0", "v_dot2_f32_f16 (VOP3P)") \
    X(3, "v_dot2c_f32_f16 %0, %1, %1", "v_dot2c_f32_f16") \
    X(4, "v_dot2c_f32_bf16 %0, %1, %1", "v_dot2c_f32_bf16") \
    X(5, "v_dot2_f32_bf16 %0, %1, %1, %0", "v_dot2_f32_bf16 (VOP3P)") \
    X(6, "v_cvt_pk_f16_f32 %0, %0, %1", "v_cvt_pk_f16_f32") \
    X(7, "v_cvt_pk_bf16_f32 %0, %0, %1", "v_cvt_pk_bf16_f32") \
    X(8, "v_pk_add_f32 %0, %0, %1", "v_pk_add_f32 (one lane pair reg)") \
    X(9, "v_exp_f32 %0, %1", "v_exp_f32") \
    X(10, "v_pk_fma_f16 %0, %0, %1, %1", "v_pk_fma_f16") \
    X(11, "v_pk_mul_f16 %0, %0, %1", "v_pk_mul_f16") \
    X(12, "v_exp_f16 %0, %1", "v_exp_f16")
template <int OP, int NV>
__global__ __launch_bounds__(256) void probe(float* out, int iters, long long* cyc) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v[16];
    for (int e = 0; e < 16; ++e) v[e] = f2{0.01f * (lane + e), 0.02f};
    f2 c = {1e-6f * lane, 1e-6f};
    asm volatile("" : "+v"(c));
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int r = (i * NV + k) & 15;
#define EMIT(id, s, n) if (OP == id) { if (id == 8) asm volatile(s : "+v"(v[r]) : "v"(c)); else asm volatile(s : "+v"(v[r].x) : "v"(c.x)); }
                OPS(EMIT)
#undef EMIT
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int e = 0; e < 16; ++e) s += v[e].x + v[e].y;
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP, int NV>
static double run() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&cyc, 64);
    const int iters = 2000;
    probe<OP, NV><<<256, 256>>>(out, 10, cyc);
    probe<OP, NV><<<256, 256>>>(out, iters, cyc);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    (void)hipFree(out); (void)hipFree(cyc);
    return (double)c / iters / 8;
}
int main() {
    const double base = run<0, 0>();
    printf("bare MFMA stream: %.1f cycles per MFMA\n", base);
#define ROW(id, s, n) printf("%-34s NV=2 %6.1f  NV=3 %6.1f  NV=4 %6.1f  NV=5 %6.1f  NV=6 %6.1f  NV=8 %6.1f cycles per MFMA\n", n, run<id, 2>(), run<id, 3>(), run<id, 4>(), run<id, 5>(), run<id, 6>(), run<id, 8>());
    OPS(ROW)
    return 0;
}
