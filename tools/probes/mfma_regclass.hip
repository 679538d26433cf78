// v_mfma_f32_32x32x16_bf16 issue rate by operand register class (one wave per SIMD, four accumulator chains), hand-written asm:
//   0: A, B, C/D all VGPR      1: A, B in AGPR, C/D VGPR      2: A AGPR, B VGPR, C/D AGPR      3: all AGPR
//   4: as 1 but the first MFMA of every chain takes C from ANOTHER VGPR block (the -m block of attn_q4)
//   5: the exact MFMA sequence of segment 1 of attn_q4 (csrc/attn_q4_loop.inc)      6: segment 2
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_regclass tools/probes/mfma_regclass.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V>
__global__ __launch_bounds__(256, 1) void probe(float* out, int iters, long long* cyc) {
    // register contents are whatever the wave starts with: the issue rate does not depend on the data
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (V == 0)
            asm volatile(
                ".rept 4\n\t"
                "v_mfma_f32_32x32x16_bf16 v[0:15], v[128:131], v[132:135], v[0:15]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[16:31], v[128:131], v[136:139], v[16:31]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[32:47], v[140:143], v[132:135], v[32:47]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[48:63], v[140:143], v[136:139], v[48:63]\n\t"
                ".endr\n\t" ::: "memory");
        if (V == 1)
            asm volatile(
                ".rept 4\n\t"
                "v_mfma_f32_32x32x16_bf16 v[0:15], a[96:99], a[64:67], v[0:15]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[16:31], a[96:99], a[80:83], v[16:31]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[32:47], a[100:103], a[64:67], v[32:47]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[48:63], a[100:103], a[80:83], v[48:63]\n\t"
                ".endr\n\t" ::: "memory");
        if (V == 2)
            asm volatile(
                ".rept 4\n\t"
                "v_mfma_f32_32x32x16_bf16 a[0:15], a[128:131], v[160:163], a[0:15]\n\t"
                "v_mfma_f32_32x32x16_bf16 a[16:31], a[132:135], v[160:163], a[16:31]\n\t"
                "v_mfma_f32_32x32x16_bf16 a[32:47], a[128:131], v[176:179], a[32:47]\n\t"
                "v_mfma_f32_32x32x16_bf16 a[48:63], a[132:135], v[176:179], a[48:63]\n\t"
                ".endr\n\t" ::: "memory");
        if (V == 3)
            asm volatile(
                ".rept 4\n\t"
                "v_mfma_f32_32x32x16_bf16 a[0:15], a[128:131], a[160:163], a[0:15]\n\t"
                "v_mfma_f32_32x32x16_bf16 a[16:31], a[132:135], a[160:163], a[16:31]\n\t"
                "v_mfma_f32_32x32x16_bf16 a[32:47], a[128:131], a[176:179], a[32:47]\n\t"
                "v_mfma_f32_32x32x16_bf16 a[48:63], a[132:135], a[176:179], a[48:63]\n\t"
                ".endr\n\t" ::: "memory");
        if (V == 4)
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 v[0:15], a[96:99], a[64:67], v[128:143]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[16:31], a[96:99], a[80:83], v[144:159]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[32:47], a[100:103], a[64:67], v[128:143]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[48:63], a[100:103], a[80:83], v[144:159]\n\t"
                ".rept 3\n\t"
                "v_mfma_f32_32x32x16_bf16 v[0:15], a[96:99], a[64:67], v[0:15]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[16:31], a[96:99], a[80:83], v[16:31]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[32:47], a[100:103], a[64:67], v[32:47]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[48:63], a[100:103], a[80:83], v[48:63]\n\t"
                ".endr\n\t" ::: "memory");
        if (V == 5)
            asm volatile(
                "v_mfma_f32_32x32x16_bf16 v[64:79], a[96:99], a[64:67], v[128:143]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[96:111], a[96:99], a[80:83], v[144:159]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[80:95], a[100:103], a[64:67], v[128:143]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[112:127], a[100:103], a[80:83], v[144:159]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[64:79], a[104:107], a[68:71], v[64:79]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[96:111], a[104:107], a[84:87], v[96:111]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[80:95], a[108:111], a[68:71], v[80:95]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[112:127], a[108:111], a[84:87], v[112:127]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[64:79], a[112:115], a[72:75], v[64:79]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[96:111], a[112:115], a[88:91], v[96:111]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[80:95], a[116:119], a[72:75], v[80:95]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[112:127], a[116:119], a[88:91], v[112:127]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[64:79], a[120:123], a[76:79], v[64:79]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[96:111], a[120:123], a[92:95], v[96:111]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[80:95], a[124:127], a[76:79], v[80:95]\n\t"
                "v_mfma_f32_32x32x16_bf16 v[112:127], a[124:127], a[92:95], v[112:127]\n\t" ::: "memory");
        if (V == 6)
            asm volatile(
                ".rept 4\n\t"
                "v_mfma_f32_32x32x16_bf16 a[0:15], a[128:131], v[160:163], a[0:15]\n\t"
                "v_mfma_f32_32x32x16_bf16 a[16:31], a[132:135], v[160:163], a[16:31]\n\t"
                "v_mfma_f32_32x32x16_bf16 a[32:47], a[128:131], v[176:179], a[32:47]\n\t"
                "v_mfma_f32_32x32x16_bf16 a[48:63], a[132:135], v[176:179], a[48:63]\n\t"
                ".endr\n\t" ::: "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (iters < 0) out[0] = 1.f;
}
template <int V> static void run(const char* what) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&cyc, 64);
    probe<V><<<256, 256>>>(out, 10, cyc);
    probe<V><<<256, 256>>>(out, 2000, cyc);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%d %-70s %.2f cycles per MFMA\n", V, what, (double)c / 2000 / 16);
}
int main() {
    run<0>("A, B, C/D VGPR");
    run<1>("A, B AGPR; C/D VGPR");
    run<2>("A AGPR, B VGPR; C/D AGPR");
    run<3>("all AGPR");
    run<4>("A, B AGPR; C/D VGPR; first MFMA of a chain with C from another VGPR block");
    run<5>("segment 1 of attn_q4 (MFMAs only)");
    run<6>("segment 2 of attn_q4 (MFMAs only)");
    return 0;
}
