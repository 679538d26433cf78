// The softmax gap of attn_q4 in isolation (one wave per SIMD): per MFMA [exp, exp, add, add, cvt] with the real kernel's dependencies, to find
// which of them costs the ~50 cycles per MFMA the kernel runs at (filler_price.hip: five INDEPENDENT plain fillers cost 34).
//   MODE 0: fillers on VALU-written registers only (independent)            1: the two exp2 read MFMA-written accumulators (S of 8 MFMAs earlier)
//   MODE 2: 1 + add / cvt consume the exp2 results of the previous gap       3: 2 + the MFMAs accumulate in VGPRs that the exp2 read (S double buffer)
//   MODE 4: 3 + cvt results feed the next MFMAs' B operand                   5: 4 with the accumulators in AGPRs and v_accvgpr_read copies... (not built)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/gap_pattern tools/probes/gap_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, long long* cyc) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
    f32x16 S[8];   // "scores": MFMA destinations
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) S[i][e] = 0.01f * e;
    float tmp[8], sum[4] = {0, 0, 0, 0}, src[16];
    u32x4 pk[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int e = 0; e < 8; ++e) tmp[e] = 0.1f * e;
    for (int e = 0; e < 16; ++e) src[e] = -0.01f * (lane + e);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // MFMA i writes S[i]; the exp2 of this gap read S[(i + 4) & 7] (written four MFMAs ago + one loop trip for half of them)
            if (MODE >= 4) {
                bf16x8 bb = __builtin_bit_cast(bf16x8, pk[i & 1]);
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(S[i]) : "v"(a), "v"(bb));
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(S[i]) : "v"(a), "v"(b));
            }
            const int g = i & 1;             // tmp group written by this gap; the other group is consumed
            float* rd = MODE >= 1 ? nullptr : src;
            (void)rd;
#define EXP(dst, e) \
    if (MODE >= 1) asm volatile("v_exp_f32 %0, %1" : "=v"(dst) : "v"(S[(i + 4) & 7][e])); \
    else asm volatile("v_exp_f32 %0, %1" : "=v"(dst) : "v"(src[(2 * i + e) & 15]));
            EXP(tmp[4 * g + 0], 2 * (i & 7) % 16)
            EXP(tmp[4 * g + 1], (2 * (i & 7) + 1) % 16)
            if (MODE >= 2) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum[0]) : "v"(tmp[4 * (g ^ 1) + 0]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum[1]) : "v"(tmp[4 * (g ^ 1) + 1]));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[i & 1][i >> 1]) : "v"(tmp[4 * (g ^ 1) + 0]), "v"(tmp[4 * (g ^ 1) + 1]));
            } else {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum[0]) : "v"(src[0]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum[1]) : "v"(src[1]));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[i & 1][i >> 1]) : "v"(src[2]), "v"(src[3]));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = sum[0] + sum[1] + sum[2] + sum[3];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += S[i][e];
    for (int e = 0; e < 8; ++e) s += tmp[e];
    s += __builtin_bit_cast(float, pk[0][0]) + __builtin_bit_cast(float, pk[1][3]);
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
static void run(const char* what) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&cyc, 64);
    const int iters = 2000;
    probe<MODE><<<256, 256>>>(out, 10, cyc);
    probe<MODE><<<256, 256>>>(out, iters, cyc);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("mode %d: %6.1f cycles per MFMA  (%s)\n", MODE, (double)c / iters / 8, what);
}
int main() {
    run<0>("[exp exp add add cvt] on VALU-written registers, independent");
    run<1>("the exp2 read MFMA-written accumulators (written four MFMAs earlier)");
    run<2>("+ add / cvt consume the previous gap's exp2 results");
    run<4>("+ the cvt results are the B operand of later MFMAs");
    return 0;
}
