// Schedule probe for the attention softmax at head_dim 64: per KV tile a wave owes 16 MFMA (32x32x16), 32 v_exp_f32, 32 adds,
// 16 v_cvt_pk.  Two waves per SIMD (A = waves 0-3, B = waves 4-7).  Compared here, in cycles per (tile of A + tile of B):
//   PP : two phases  -- [A: 16 MFMA + 16 cvt | B: 16 x (exp, exp, add, add)] [roles swapped]            (the shipped attn_pp_k)
//   3P : three phases -- [A: 16 MFMA + 16 cvt | B: 32 adds] [roles swapped] [A and B: 32 exp each, no MFMA in flight]
// Registers only (no LDS, no memory): an upper bound on what the split can give.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/three_phase tools/probes/three_phase.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int iters, long long* cyc) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float v[32], p[32];
    unsigned pk[16];
    for (int e = 0; e < 32; ++e) { v[e] = -0.01f * (lane + e); p[e] = 0.5f; }
    float s0 = 0.f, s1 = 0.f;
    auto mseg = [&]() {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(p[2 * i]), "v"(p[2 * i + 1]));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto adds = [&]() {
#pragma unroll
        for (int e = 0; e < 32; e += 2) {
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(s0) : "v"(p[e]));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(s1) : "v"(p[e + 1]));
        }
    };
    auto exps = [&]() {
#pragma unroll
        for (int e = 0; e < 32; ++e) asm volatile("v_exp_f32 %0, %1" : "=v"(p[e]) : "v"(v[e]));
    };
    auto quads = [&]() {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            asm volatile("v_exp_f32 %0, %1" : "=v"(p[2 * g]) : "v"(v[2 * g]));
            asm volatile("v_exp_f32 %0, %1" : "=v"(p[2 * g + 1]) : "v"(v[2 * g + 1]));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(s0) : "v"(p[(2 * g + 30) & 31]));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(s1) : "v"(p[(2 * g + 31) & 31]));
        }
    };
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // PP
            if (grp == 0) mseg(); else quads();
            __builtin_amdgcn_s_barrier();
            if (grp == 0) quads(); else mseg();
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 1) {          // 3P
            if (grp == 0) mseg(); else adds();
            __builtin_amdgcn_s_barrier();
            if (grp == 0) adds(); else mseg();
            __builtin_amdgcn_s_barrier();
            exps();
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 2) {   // matrix segment beside an idle partner
            if (grp == 0) mseg();
            __builtin_amdgcn_s_barrier();
            if (grp == 1) mseg();
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 3) {   // matrix segment beside 32 adds
            if (grp == 0) mseg(); else adds();
            __builtin_amdgcn_s_barrier();
            if (grp == 0) adds(); else mseg();
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 4) {   // exps only, both waves
            exps();
            __builtin_amdgcn_s_barrier();
            exps();
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 5) {   // matrix segment without the own cvt, partner idle
            if (grp == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
            }
            __builtin_amdgcn_s_barrier();
            if (grp == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
            }
            __builtin_amdgcn_s_barrier();
        } else if (MODE == 6) {   // matrix segment beside exps only (no adds)
            if (grp == 0) mseg(); else exps();
            __builtin_amdgcn_s_barrier();
            if (grp == 0) exps(); else mseg();
            __builtin_amdgcn_s_barrier();
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = s0 + s1;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int e = 0; e < 16; ++e) s += __uint_as_float(pk[e]);
    if (s == 12345.678f) out[0] = s;
    if (lane == 0 && blockIdx.x == 0 && (wave & 3) == 0) cyc[grp] = t1 - t0;
}
template <int MODE> static void run(const char* name) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&cyc, 64);
    probe<MODE><<<256, 512>>>(out, 10, cyc);
    probe<MODE><<<256, 512>>>(out, 2000, cyc);
    (void)hipDeviceSynchronize();
    long long c[2]; (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-14s %7.1f cycles per (tile of A + tile of B)  = %6.1f per 16 MFMA  (matrix-pipe floor 512)\n", name, (double)c[0] / 2000, (double)c[0] / 4000);
}
int main() {
    run<0>("two phases"); run<1>("three phases"); run<2>("M | idle"); run<5>("bare M | idle"); run<3>("M | 32 adds"); run<6>("M | 32 exps"); run<4>("exps, exps");
    return 0;
}
