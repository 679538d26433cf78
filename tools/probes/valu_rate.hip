// Issue cost of the VALU instructions an attention softmax is made of, on gfx950: cycles per wave-instruction for a stream of
// independent instructions, with 1 or 2 waves per SIMD, and beside a partner wave that streams MFMAs.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/valu_rate tools/probes/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// OP: 0 v_exp_f32  1 v_add_f32  2 v_cvt_pk_bf16_f32  3 v_max3_f32  4 v_pk_add_f32  5 v_fma_f32  6 v_pk_fma_f32  7 v_exp_f16
//     8 v_ldexp_f32  9 v_pk_mul_f32  10 v_mul_f32  11 v_exp_f32 + v_add_f32 alternating  12 v_fract_f32  13 v_cvt_i32_f32  14 v_exp_legacy? (v_mov)
template <int OP, int PARTNER, int SWAP = 0, int PRIO = 0, int NF = 0, int M16 = 0>
__global__ __launch_bounds__(512) void probe(float* out, int iters, long long* cyc, int nw) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long long t0 = 0, t1 = 0;
    float s = 0.f;
    // SWAP: the VALU waves are the OLDER half (waves 0-3).  PRIO: 1 = s_setprio 3 on the VALU waves, 2 = on the MFMA waves
    if (PARTNER && (SWAP ? wave >= 4 : wave < 4)) {
        if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        t0 = __builtin_amdgcn_s_memtime();
        float fv[8];
        for (int e = 0; e < 8; ++e) fv[e] = 0.001f * (lane + e);
        float fc = 1e-6f * lane;
        asm volatile("" : "+v"(fc));
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 acc4[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) acc4[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
            if (M16) {  // 32 x v_mfma_f32_16x16x32_bf16 = the same flops as 16 x 32x32x16
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    acc4[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[i & 7], 0, 0, 0);
                    if (NF && (i & 1))
#pragma unroll
                        for (int k = 0; k < NF; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(fv[(i * NF + k) & 7]) : "v"(fc));
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < NF; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(fv[(i * NF + k) & 7]) : "v"(fc));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) s += acc4[i][e];
        for (int e = 0; e < 8; ++e) s += fv[e];
    } else {
        if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
        float v[16], w[16], u[16], x[16];
        f2 p[16];
        for (int e = 0; e < 16; ++e) { v[e] = -0.01f * (lane + e); w[e] = 0.5f; u[e] = 0.25f; x[e] = 0.125f; p[e] = f2{v[e], w[e]}; }
        float c0 = 1e-6f * lane, c1 = -1e-6f * lane;
        f2 pc = {c0, c1};
        asm volatile("" : "+v"(c0), "+v"(c1), "+v"(pc));
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#define X_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define X_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c0));
#define X_CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c0));
#define X_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c0), "v"(c1));
#define X_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
#define X_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c0), "v"(c1));
#define X_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pc));
#define X_EXPH(i) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
#define X_LDEXP(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c0));
#define X_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
#define X_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c0));
#define X_EXPADD(i) asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %1, %1, %2" : "+v"(v[i]), "+v"(w[i]) : "v"(c0));
#define X_FRACT(i) asm volatile("v_fract_f32 %0, %0" : "+v"(v[i]));
#define X_CVTI(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(v[i]));
#define X_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(c0));
#define X_EXP2ADD(i) asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %3, %3, %2" : "+v"(v[i]), "+v"(w[i]), "+v"(u[i]) : "v"(c0));
#define X_EEA(i) asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_add_f32 %2, %2, %3" : "+v"(v[i]), "+v"(w[i]), "+v"(u[i]) : "v"(c0));
#define X_EEAA(i) asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %4, %4, %3" : "+v"(v[i]), "+v"(w[i]), "+v"(u[i]), "+v"(x[i]) : "v"(c0));
#define X_EAE(i) asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %2, %2, %3\n\tv_exp_f32 %1, %1" : "+v"(v[i]), "+v"(w[i]), "+v"(u[i]) : "v"(c0));
            if (OP == 16) { REP16(X_EEA) }
            if (OP == 17) { REP16(X_EEAA) }
            if (OP == 18) { REP16(X_EAE) }
            if (OP == 0) { REP16(X_EXP) REP16(X_EXP) }
            if (OP == 1) { REP16(X_ADD) REP16(X_ADD) }
            if (OP == 2) { REP16(X_CVT) REP16(X_CVT) }
            if (OP == 3) { REP16(X_MAX3) REP16(X_MAX3) }
            if (OP == 4) { REP16(X_PKADD) REP16(X_PKADD) }
            if (OP == 5) { REP16(X_FMA) REP16(X_FMA) }
            if (OP == 6) { REP16(X_PKFMA) REP16(X_PKFMA) }
            if (OP == 7) { REP16(X_EXPH) REP16(X_EXPH) }
            if (OP == 8) { REP16(X_LDEXP) REP16(X_LDEXP) }
            if (OP == 9) { REP16(X_PKMUL) REP16(X_PKMUL) }
            if (OP == 10) { REP16(X_MUL) REP16(X_MUL) }
            if (OP == 11) { REP16(X_EXPADD) }
            if (OP == 12) { REP16(X_FRACT) REP16(X_FRACT) }
            if (OP == 13) { REP16(X_CVTI) REP16(X_CVTI) }
            if (OP == 14) { REP16(X_MOV) REP16(X_MOV) }
            if (OP == 15) { REP16(X_EXP2ADD) }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int e = 0; e < 16; ++e) s += v[e] + w[e] + u[e] + x[e] + p[e][0] + p[e][1];
    }
    if (s == 12345.678f) out[0] = s;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int OP, int PARTNER, int SWAP = 0, int PRIO = 0, int NF = 0, int M16 = 0>
static void run(const char* name, int per_iter) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&cyc, 64);
    const int iters = 2000;
    for (int nw : {4, 8}) {
        if (PARTNER && nw == 4) continue;
        (void)hipMemset(cyc, 0, 64);
        probe<OP, PARTNER, SWAP, PRIO, NF, M16><<<256, nw * 64>>>(out, 10, cyc, nw);
        probe<OP, PARTNER, SWAP, PRIO, NF, M16><<<256, nw * 64>>>(out, iters, cyc, nw);
        (void)hipDeviceSynchronize();
        long long c[8];
        (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
        if (PARTNER)
            printf("%-28s beside an MFMA wave (%s, %d own v_add per 32 MFMA-cycles): %6.2f cycles/instr   (MFMA wave: %6.1f cycles per 16 MFMA)\n", name, M16 ? "16x16x32" : "32x32x16", NF, (double)c[SWAP ? 0 : 4] / iters / per_iter, (double)c[SWAP ? 4 : 0] / iters);
        else
            printf("%-28s %d wave(s)/SIMD: %6.2f cycles/instr per wave\n", name, nw / 4, (double)c[0] / iters / per_iter);
    }
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
#define ALL(P) \
    run<0, P>("v_exp_f32", 32); run<7, P>("v_exp_f16", 32); run<1, P>("v_add_f32", 32); run<10, P>("v_mul_f32", 32); run<5, P>("v_fma_f32", 32); \
    run<2, P>("v_cvt_pk_bf16_f32", 32); run<3, P>("v_max3_f32", 32); run<4, P>("v_pk_add_f32", 32); run<9, P>("v_pk_mul_f32", 32); run<6, P>("v_pk_fma_f32", 32); \
    run<8, P>("v_ldexp_f32", 32); run<12, P>("v_fract_f32", 32); run<13, P>("v_cvt_i32_f32", 32); run<14, P>("v_mov_b32", 32); \
    run<11, P>("exp+add pairs (per pair)", 16); run<15, P>("exp+add+add (per triple)", 16);
#define SOME(S, P) \
    run<0, 1, S, P>("v_exp_f32", 32); run<1, 1, S, P>("v_add_f32", 32); run<4, 1, S, P>("v_pk_add_f32", 32); run<11, 1, S, P>("exp+add pairs (per pair)", 16); run<15, 1, S, P>("exp+add+add (per triple)", 16);
#define NFS(NF, M16) \
    run<0, 1, 0, 0, NF, M16>("v_exp_f32", 32); run<1, 1, 0, 0, NF, M16>("v_add_f32", 32); run<2, 1, 0, 0, NF, M16>("v_cvt_pk_bf16_f32", 32); run<4, 1, 0, 0, NF, M16>("v_pk_add_f32", 32);
#define MIX(NF, M16) \
    run<11, 1, 0, 0, NF, M16>("exp,add (per pair)", 16); run<16, 1, 0, 0, NF, M16>("exp,exp,add (per triple)", 16); run<18, 1, 0, 0, NF, M16>("exp,add,exp (per triple)", 16); run<17, 1, 0, 0, NF, M16>("exp,exp,add,add (per quad)", 16); run<15, 1, 0, 0, NF, M16>("exp,add,add (per triple)", 16);
    MIX(0, 0) MIX(1, 0) MIX(2, 0) MIX(3, 0) MIX(0, 1) MIX(2, 1)
    return 0;
}
