// Which arrangement of the softmax fillers of two consecutive MFMA gaps (4 exp2, 4 adds, 2 cvt_pk in total) is cheapest?  One wave per SIMD,
// independent registers (gap_pattern.hip showed the dependencies do not add cost).  Pattern strings: e = v_exp_f32, a = v_add_f32,
// c = v_cvt_pk_bf16_f32, | = the MFMA between the two gaps (a leading MFMA is implied).  Cycles per MFMA.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/gap_order tools/probes/gap_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int P>
struct Pat;
#define PATS(X) \
    X(0, "eeaac|eeaac") X(1, "eeee|aaaacc") X(2, "eee|eaaaacc") X(3, "eaeac|eaeac") X(4, "aaeec|aaeec") X(5, "aacee|aacee") \
    X(6, "eaace|eaace") X(7, "eeeeaa|aacc") X(8, "aaaacc|eeee") X(9, "ecaea|ecaea") X(10, "eeaa|eeaacc") X(11, "eea|eeaaacc") \
    X(12, "eeaacc|eeaa") X(13, "e|eeeaaaacc") X(14, "aeaec|aeaec") X(15, "aecea|aecea")
template <int P>
__global__ __launch_bounds__(256) void probe(float* out, int iters, long long* cyc, const char* pat) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
    f32x16 S[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) S[i][e] = 0.01f * e;
    float v[16];
    for (int e = 0; e < 16; ++e) v[e] = 0.01f * (lane + e);
    float c0 = 1e-6f * lane; asm volatile("" : "+v"(c0));
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#define STEP(ch, r) \
    if (ch == 'e') asm volatile("v_exp_f32 %0, %1" : "=v"(v[(r) & 15]) : "v"(v[((r) + 8) & 15])); \
    else if (ch == 'a') asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(r) & 15]) : "v"(c0)); \
    else if (ch == 'c') asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[(r) & 15]) : "v"(v[((r) + 5) & 15]), "v"(c0)); \
    else if (ch == '|') asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(S[2 * i + 1]) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(S[2 * i]) : "v"(a), "v"(b));
#define PATSTEPS(id, s) if (P == id) { constexpr const char* p = s; _Pragma("unroll") for (int k = 0; k < (int)sizeof(s) - 1; ++k) { const char ch = p[k]; STEP(ch, 3 * k + i) } }
            PATS(PATSTEPS)
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += S[i][e];
    for (int e = 0; e < 16; ++e) s += v[e];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int P>
static void run(const char* pat) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&cyc, 64);
    const int iters = 2000;
    probe<P><<<256, 256>>>(out, 10, cyc, pat);
    probe<P><<<256, 256>>>(out, iters, cyc, pat);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-14s %6.1f cycles per MFMA\n", pat, (double)c / iters / 8);
}
int main() {
#define RUN(id, s) run<id>(s);
    PATS(RUN)
    return 0;
}
