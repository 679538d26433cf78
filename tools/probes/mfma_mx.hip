// Per-lane block scales of v_mfma_scale_f32_32x32x64_f8f6f4 (MX operands).  Hypothesis: lane l supplies, for the A operand, the E8M0
// scale of (row l & 31, K block l >> 5) and likewise for B (column l & 31, K block l >> 5); opsel picks the byte of the scale VGPR.
// D[i][j] = sum_k A[i][k] 2^(sa[i][k / 32] - 127) B[k][j] 2^(sb[j][k / 32] - 127).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_mx tools/probes/mfma_mx.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned char to_e4m3(float v) { return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false) & 0xff); }
template <int OA, int OB>
__global__ void run(const float* A, const float* B, const int* SA, const int* SB, float* D) {  // SA / SB: one packed scale word per lane
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
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OA, SA[l], OB, SB[l]);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
int main() {
    std::vector<float> A(32 * 64), B(64 * 32), D(1024);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) A[i * 64 + k] = (float)(((i * 7 + k * 3) % 9) - 4) * 0.5f;
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)(((k * 5 + j * 11) % 7) - 3) * 0.25f;
    std::vector<int> SA(64), SB(64);
    auto sa = [](int l, int byte) { return 120 + ((l * 3 + byte * 5) % 13); };   // differs per lane and per byte
    auto sb = [](int l, int byte) { return 122 + ((l * 5 + byte * 7) % 11); };
    for (int l = 0; l < 64; ++l) {
        SA[l] = sa(l, 0) | (sa(l, 1) << 8) | (sa(l, 2) << 16) | (sa(l, 3) << 24);
        SB[l] = sb(l, 0) | (sb(l, 1) << 8) | (sb(l, 2) << 16) | (sb(l, 3) << 24);
    }
    float *dA, *dB, *dD; int *dSA, *dSB;
    (void)hipMalloc(&dA, 8192); (void)hipMalloc(&dB, 8192); (void)hipMalloc(&dD, 4096); (void)hipMalloc(&dSA, 256); (void)hipMalloc(&dSB, 256);
    (void)hipMemcpy(dA, A.data(), 8192, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), 8192, hipMemcpyHostToDevice);
    (void)hipMemcpy(dSA, SA.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dSB, SB.data(), 256, hipMemcpyHostToDevice);
    auto check = [&](int oa, int ob) {
        (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        double maxd = 0, maxr = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0;
            for (int k = 0; k < 64; ++k) {
                const int kb = k >> 5;
                s += (double)A[i * 64 + k] * std::ldexp(1.0, sa(kb * 32 + i, oa) - 127) * (double)B[k * 32 + j] * std::ldexp(1.0, sb(kb * 32 + j, ob) - 127);
            }
            maxd = std::fmax(maxd, std::fabs(D[i * 32 + j] - s)); maxr = std::fmax(maxr, std::fabs(s));
        }
        printf("opsel A %d B %d: max |D - expected| = %.3g (max |expected| %.3g)  %s\n", oa, ob, maxd, maxr, maxd <= 1e-5 * maxr ? "per-lane block scales confirmed" : "MISMATCH");
    };
    run<0, 0><<<1, 64>>>(dA, dB, dSA, dSB, dD); check(0, 0);
    run<1, 0><<<1, 64>>>(dA, dB, dSA, dSB, dD); check(1, 0);
    run<0, 2><<<1, 64>>>(dA, dB, dSA, dSB, dD); check(0, 2);
    run<3, 1><<<1, 64>>>(dA, dB, dSA, dSB, dD); check(3, 1);
    return 0;
}
