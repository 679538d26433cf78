// Operand layout and issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands and unit (E8M0 = 127) scales.
// Hypothesis checked here: lane l holds row (l & 31) of A / column (l & 31) of B, K elements (l >> 5) * 32 .. + 31 as 32
// consecutive bytes (8 VGPRs); C/D as for every 32x32 MFMA: D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31].
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_fp8 tools/probes/mfma_fp8.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned char to_e4m3(float v) {  // exact for the small integers / halves used here
    const int r = __builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false);
    return (unsigned char)(r & 0xff);
}
__global__ void layout(const float* A, const float* B, float* D, int sa, int sb) {  // A [32][64], B [64][32] row-major fp32, D [32][32]
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) {
        unsigned ua = 0, ub = 0;
        for (int e = 0; e < 4; ++e) {
            const int k = (l >> 5) * 32 + w * 4 + e;
            ua |= (unsigned)to_e4m3(A[(l & 31) * 64 + k]) << (8 * e);
            ub |= (unsigned)to_e4m3(B[k * 32 + (l & 31)]) << (8 * e);
        }
        a[w] = (int)ua; b[w] = (int)ub;
    }
    f32x16 c;
    for (int e = 0; e < 16; ++e) c[e] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ __launch_bounds__(256) void rate(float* out, int iters, long long* cyc) {
    const int l = threadIdx.x & 63;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) { a[w] = 0x38383838 + l; b[w] = 0x3c3c3c3c - l; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i & 3], 0, 0, 0, 127, 0, 127);
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    std::vector<float> A(32 * 64), B(64 * 32), D(32 * 32), R(32 * 32);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) A[i * 64 + k] = (float)(((i * 7 + k * 3) % 9) - 4) * 0.5f;      // asymmetric
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)(((k * 5 + j * 11) % 7) - 3) * 0.25f;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < 64; ++k) s += (double)A[i * 64 + k] * B[k * 32 + j]; R[i * 32 + j] = (float)s; }
    float *dA, *dB, *dD; long long* cyc;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dD, D.size() * 4); (void)hipMalloc(&cyc, 64);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int sa : {127, 128}) for (int sb : {127, 126}) {
        layout<<<1, 64>>>(dA, dB, dD, sa, sb);
        (void)hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        const double f = std::ldexp(1.0, (sa - 127) + (sb - 127));
        double maxd = 0; for (int i = 0; i < 1024; ++i) maxd = std::fmax(maxd, std::fabs(D[i] - R[i] * f));
        printf("layout check, scale bytes A %d B %d (expected factor %.2f): max |D - A.B * factor| = %.3g  %s\n", sa, sb, f, maxd, maxd < 1e-4 ? "layout confirmed" : "MISMATCH");
    }
    rate<<<256, 256>>>(dD, 10, cyc);
    rate<<<256, 256>>>(dD, 2000, cyc);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("issue rate: %.2f cycles per v_mfma_scale_f32_32x32x64_f8f6f4 (262144 flop) per SIMD, one wave per SIMD\n", (double)c / 2000 / 16);
    return 0;
}
