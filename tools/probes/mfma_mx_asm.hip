// The asm forms attn_q4f's generated body uses for v_mfma_scale_f32_32x32x64_f8f6f4: A / B operands in AGPRs, C in VGPRs (a separate
// tuple from the destination), scale bytes chosen by op_sel / op_sel_hi.  Prints which hypothesis about the byte select matches.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_mx_asm tools/probes/mfma_mx_asm.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned char to_e4m3(float v) { return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false) & 0xff); }
template <int MODE>
__global__ void run(const float* A, const float* B, const int* SA, const int* SB, float* D) {
    const int l = threadIdx.x, hi = l >> 5;
    i32x8 a, b;
    // operand = 16-byte chunks hi and 2 + hi of the 64-byte row (attn_q4f's layout)
    for (int w = 0; w < 8; ++w) {
        unsigned ua = 0, ub = 0;
        for (int e = 0; e < 4; ++e) {
            const int t = w * 4 + e;                       // byte of the operand
            const int k = ((t >> 4) * 2 + hi) * 16 + (t & 15);
            ua |= (unsigned)to_e4m3(A[(l & 31) * 64 + k]) << (8 * e);
            ub |= (unsigned)to_e4m3(B[k * 32 + (l & 31)]) << (8 * e);
        }
        a[w] = (int)ua; b[w] = (int)ub;
    }
    f32x16 c, d;
    for (int e = 0; e < 16; ++e) c[e] = 0.5f;
    const int sa = SA[l], sb = SB[l];
    if (MODE == 0)
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n\ts_nop 15\n\ts_nop 15" : "=v"(d) : "a"(a), "a"(b), "v"(c), "v"(sa), "v"(sb));
    if (MODE == 1)
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel:[1,0,0] op_sel_hi:[0,0,0]\n\ts_nop 15\n\ts_nop 15" : "=v"(d) : "a"(a), "a"(b), "v"(c), "v"(sa), "v"(sb));
    if (MODE == 2)
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel:[0,1,0] op_sel_hi:[0,0,0]\n\ts_nop 15\n\ts_nop 15" : "=v"(d) : "a"(a), "a"(b), "v"(c), "v"(sa), "v"(sb));
    if (MODE == 3)
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\ts_nop 15\n\ts_nop 15" : "=v"(d) : "a"(a), "a"(b), "v"(c), "v"(sa), "v"(sb));
    if (MODE == 4)  // VGPR sources, as gemm_g4f
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel:[1,0,0] op_sel_hi:[0,0,0]\n\ts_nop 15\n\ts_nop 15" : "=v"(d) : "v"(a), "v"(b), "v"(c), "v"(sa), "v"(sb));
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = d[r];
}
int main() {
    std::vector<float> A(32 * 64), B(64 * 32), D(1024);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) A[i * 64 + k] = (float)(((i * 7 + k * 3) % 9) - 4) * 0.5f + 0.25f;
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)(((k * 5 + j * 11) % 7) - 3) * 0.25f + 0.125f;
    auto sa = [](int l, int byte) { return 120 + ((l * 3 + byte * 5) % 13); };
    auto sb = [](int l, int byte) { return 122 + ((l * 5 + byte * 7) % 11); };
    std::vector<int> SA(64), SB(64);
    for (int l = 0; l < 64; ++l) {
        SA[l] = sa(l, 0) | (sa(l, 1) << 8) | (sa(l, 2) << 16) | (sa(l, 3) << 24);
        SB[l] = sb(l, 0) | (sb(l, 1) << 8) | (sb(l, 2) << 16) | (sb(l, 3) << 24);
    }
    float *dA, *dB, *dD; int *dSA, *dSB;
    (void)hipMalloc(&dA, 8192); (void)hipMalloc(&dB, 8192); (void)hipMalloc(&dD, 4096); (void)hipMalloc(&dSA, 256); (void)hipMalloc(&dSB, 256);
    (void)hipMemcpy(dA, A.data(), 8192, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), 8192, hipMemcpyHostToDevice);
    (void)hipMemcpy(dSA, SA.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dSB, SB.data(), 256, hipMemcpyHostToDevice);
    auto check = [&](const char* what) {
        (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        for (int oa = 0; oa < 4; ++oa) for (int ob = 0; ob < 4; ++ob) {
            double maxd = 0, maxr = 0;
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
                double s = 0.5;
                for (int k = 0; k < 64; ++k) {
                    const int blk = k >> 5;  // MX block of head-dim element k: its scale comes from lane half blk
                    s += (double)A[i * 64 + k] * std::ldexp(1.0, sa(blk * 32 + i, oa) - 127) * (double)B[k * 32 + j] * std::ldexp(1.0, sb(blk * 32 + j, ob) - 127);
                }
                maxd = std::fmax(maxd, std::fabs(D[i * 32 + j] - s)); maxr = std::fmax(maxr, std::fabs(s));
            }
            if (maxd <= 1e-5 * maxr) printf("%s: matches scale bytes A %d, B %d (chunk layout [hi | 2 + hi], blocks = memory-contiguous 32)\n", what, oa, ob);
        }
    };
    run<0><<<1, 64>>>(dA, dB, dSA, dSB, dD); check("AGPR src, op_sel [0,0,0] op_sel_hi [0,0,0]");
    run<1><<<1, 64>>>(dA, dB, dSA, dSB, dD); check("AGPR src, op_sel [1,0,0] op_sel_hi [0,0,0]");
    run<2><<<1, 64>>>(dA, dB, dSA, dSB, dD); check("AGPR src, op_sel [0,1,0] op_sel_hi [0,0,0]");
    run<3><<<1, 64>>>(dA, dB, dSA, dSB, dD); check("AGPR src, op_sel [0,0,0] op_sel_hi [1,0,0]");
    run<4><<<1, 64>>>(dA, dB, dSA, dSB, dD); check("VGPR src, op_sel [1,0,0] op_sel_hi [0,0,0]");
    printf("done\n");
    return 0;
}
