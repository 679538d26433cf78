// Can an MFMA-only wave and a VALU-only (softmax-like) wave on the SAME SIMD overlap on gfx950?
// Block = 8 waves; waves 0-3 play the matrix role (16 x v_mfma_f32_32x32x16_bf16 per iteration, 4 accumulators), waves 4-7 the
// softmax role (per iteration 32 v_exp_f32 + row-max + row-sum + 16 v_cvt_pk_bf16_f32, the work of one 32q x 64kv attention
// unit at head_dim 64).  Each role is timed alone and together, with s_memtime per wave.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/pingpong tools/probes/pingpong.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE bit0: matrix role active, bit1: softmax role active.  SUMK: 0 = 32 v_add_f32, 1 = 16 v_pk_add_f32, 2 = no sum.
// MAXK: 0 = 32 v_max_f32, 1 = 16 v_max3_f32.  PRIO: s_setprio 1 on the matrix waves.  SYNC: s_barrier per iteration.
template <int MODE, int SUMK, int MAXK, int PRIO, int SYNC>
__global__ __launch_bounds__(512) void probe(float* out, int iters, long long* cyc, unsigned* hwid) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long long t0 = 0, t1 = 0;
    float s = 0.f;
    if (lane == 0 && blockIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        hwid[wave] = id;
    }
    if (wave < 4) {
        if (MODE & 1) {
            bf16x8 a, b;
            for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
            f32x16 acc[4];
            for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
            if (PRIO) __builtin_amdgcn_s_setprio(1);
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
                if (SYNC) __builtin_amdgcn_s_barrier();
            }
            t1 = __builtin_amdgcn_s_memtime();
            for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
        } else if (SYNC) {
            for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_barrier();
        }
    } else {
        if (MODE & 2) {
            float v[32];
            unsigned pk[16];
            for (int e = 0; e < 32; ++e) v[e] = -0.01f * (lane + e);
            float sum = 0.f, sum2 = 0.f, mx = 0.f;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
                // row max of the 32 scores
                if (MAXK == 0) {
#pragma unroll
                    for (int e = 0; e < 32; ++e) asm volatile("v_max_f32 %0, %0, %1" : "+v"(mx) : "v"(v[e]));
                } else {
#pragma unroll
                    for (int e = 0; e < 32; e += 2) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(v[e]), "v"(v[e + 1]));
                }
                float p[32];
#pragma unroll
                for (int e = 0; e < 32; ++e) asm volatile("v_exp_f32 %0, %1" : "=v"(p[e]) : "v"(v[e]));
                if (SUMK == 0) {
#pragma unroll
                    for (int e = 0; e < 32; e += 2) {
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(p[e]));
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum2) : "v"(p[e + 1]));
                    }
                } else if (SUMK == 1) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 acc2 = {sum, sum2};
#pragma unroll
                    for (int e = 0; e < 32; e += 2) {
                        f2 pe = {p[e], p[e + 1]};
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc2) : "v"(pe));
                    }
                    sum = acc2[0]; sum2 = acc2[1];
                }
#pragma unroll
                for (int e = 0; e < 32; e += 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[e >> 1]) : "v"(p[e]), "v"(p[e + 1]));
#pragma unroll
                for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(pk[e]));
                if (SYNC) __builtin_amdgcn_s_barrier();
            }
            t1 = __builtin_amdgcn_s_memtime();
            s = sum + sum2 + mx;
        } else if (SYNC) {
            for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_barrier();
        }
    }
    if (s == 12345.678f) out[0] = s;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int MODE, int SUMK, int MAXK, int PRIO, int SYNC>
static void run(const char* name) {
    float* out; long long* cyc; unsigned* hw;
    hipMalloc(&out, 64); hipMalloc(&cyc, 64); hipMalloc(&hw, 64);
    hipMemset(cyc, 0, 64);
    const int iters = 2000;
    probe<MODE, SUMK, MAXK, PRIO, SYNC><<<256, 512>>>(out, 10, cyc, hw);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<MODE, SUMK, MAXK, PRIO, SYNC><<<256, 512>>>(out, iters, cyc, hw);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[8]; unsigned h[8];
    hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    hipMemcpy(h, hw, 32, hipMemcpyDeviceToHost);
    printf("%-46s matrix wave: %7.1f ticks/iter (16 MFMA = 512 pipe cycles)   softmax wave: %7.1f ticks/iter   %.3f ms\n", name,
           (double)c[0] / iters, (double)c[4] / iters, ms);
    static bool once = false;
    if (!once) {
        once = true;
        printf("  HW_ID per wave (simd = bits 5:4, wave slot = bits 3:0):");
        for (int w = 0; w < 8; ++w) printf(" w%d:simd%u/slot%u", w, (h[w] >> 4) & 3, h[w] & 15);
        printf("\n");
    }
    hipFree(out); hipFree(cyc); hipFree(hw);
}

int main() {
    run<1, 0, 0, 0, 0>("matrix only");
    run<2, 0, 0, 0, 0>("softmax only (add, max)");
    run<2, 0, 1, 0, 0>("softmax only (add, max3)");
    run<2, 1, 1, 0, 0>("softmax only (pk_add, max3)");
    run<2, 2, 1, 0, 0>("softmax only (no sum, max3)");
    run<3, 0, 0, 0, 0>("both (add, max)");
    run<3, 0, 1, 0, 0>("both (add, max3)");
    run<3, 1, 1, 0, 0>("both (pk_add, max3)");
    run<3, 2, 1, 0, 0>("both (no sum, max3)");
    run<3, 0, 1, 1, 0>("both (add, max3) prio");
    run<3, 1, 1, 1, 0>("both (pk_add, max3) prio");
    run<3, 0, 1, 0, 1>("both (add, max3) barrier/iter");
    run<3, 1, 1, 0, 1>("both (pk_add, max3) barrier/iter");
    run<3, 1, 1, 1, 1>("both (pk_add, max3) prio barrier/iter");
    return 0;
}
