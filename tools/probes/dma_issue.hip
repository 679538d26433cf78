// Issue cost of one LDS-DMA instruction (1 KiB per wave) on gfx950 for the encodings a kernel can choose from:
//   0 global_load_lds_dwordx4 v, s[base:base+1]          (saddr + 32-bit voffset, what the kernels use)
//   1 global_load_lds_dwordx4 v[lo:hi], off               (64-bit vaddr)
//   2 buffer_load_dwordx4 v, s[rsrc:+3], soff offen lds   (MUBUF: SRD + 32-bit voffset)
//   3 global_load_dwordx4 to VGPRs (no LDS), saddr form   (for comparison: register staging)
//   4 buffer_load_dwordx4 off, s[rsrc:+3], soff lds       (MUBUF, NO address VGPR: ADD_TID_ENABLE + stride 16 in the descriptor: lane l reads base + soff + 16 l --
//                                                          needs the source laid out exactly as the LDS image)
//   5 buffer_load_dword off, s[rsrc:+3], soff lds          (the same with 4 bytes per lane, stride 4: 256 B per instruction)
// One wave per SIMD (or two), NB back-to-back instructions, source = a 64-KiB L2-resident window; cycles per instruction
// from s_memtime around the issue burst (not including the final vmcnt wait) and including it.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/dma_issue tools/probes/dma_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void probe(const char* src, float* out, int iters, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)blockIdx.x * 65536;
    const unsigned voff = (unsigned)(((lane >> 3) * 512) + (((lane & 7) ^ ((lane >> 4) & 7)) * 16));
    const char* vaddr = base + voff;
    u32x4 rs;  // raw buffer descriptor: base, stride 0, 64 KiB, dword format
    rs.x = __builtin_amdgcn_readfirstlane((unsigned)(size_t)base);
    rs.y = __builtin_amdgcn_readfirstlane((unsigned)((size_t)base >> 32) & 0xffff);
    rs.z = 65536;
    rs.w = 0x00020000;
    u32x4 rt = rs;  // ADD_TID_ENABLE (word 3 bit 23), stride 16 (word 1 bits 29:16); num_records counts strides
    rt.y = rs.y | (16u << 16); rt.z = 65536 / 16; rt.w = (1u << 23);  // with ADD_TID_ENABLE the DATA_FORMAT bits are stride[17:14]: keep them zero
    u32x4 rt4 = rs;
    rt4.y = rs.y | (4u << 16); rt4.z = 65536 / 4; rt4.w = (1u << 23);
    long long tissue = 0, tall = 0;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            char* dst = smem + wave * 8192 + i * 1024;
            const unsigned m0v = (unsigned)(size_t)(__attribute__((address_space(3))) char*)dst;
            const char* sb = base + i * 4096;
            if (MODE == 0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sb), "v"(voff), "s"(__builtin_amdgcn_readfirstlane(m0v)) : "memory", "m0");
            if (MODE == 1) { const char* va = vaddr + i * 4096; asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(va), "s"(__builtin_amdgcn_readfirstlane(m0v)) : "memory", "m0"); }
            if (MODE == 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %0, %3 offen lds" ::"s"(rs), "v"(voff), "s"(__builtin_amdgcn_readfirstlane(m0v)), "s"(i * 4096) : "memory", "m0");
            if (MODE == 4) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 off, %0, %2 lds" ::"s"(rt), "s"(__builtin_amdgcn_readfirstlane(m0v)), "s"(i * 4096) : "memory", "m0");
            if (MODE == 5) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword off, %0, %2 lds" ::"s"(rt4), "s"(__builtin_amdgcn_readfirstlane(m0v)), "s"(i * 4096) : "memory", "m0");
            if (MODE == 3) { u32x4 r; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sb) : "memory"); asm volatile("" ::"v"(r)); }
        }
        const long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t2 = __builtin_amdgcn_s_memtime();
        tissue += t1 - t0;
        tall += t2 - t0;
    }
    if (acc.x == 0x12345) out[0] = 1.f;
    if (lane == 0 && blockIdx.x == 0 && wave == 0) { cyc[0] = tissue; cyc[1] = tall; }
}
template <int MODE> static void run(const char* name, const char* src) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&cyc, 64);
    for (int nw : {4, 8}) {
        probe<MODE><<<256, nw * 64, 65536>>>(src, out, 5, cyc);
        probe<MODE><<<256, nw * 64, 65536>>>(src, out, 200, cyc);
        (void)hipDeviceSynchronize();
        long long c[2]; (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
        printf("%-44s %d wave(s)/SIMD: issue %6.1f cycles per instruction, with the final wait %6.1f\n", name, nw / 4, (double)c[0] / 200 / 8, (double)c[1] / 200 / 8);
    }
}
int main() {
    char* src; (void)hipMalloc(&src, 256 * 65536); (void)hipMemset(src, 1, 256 * 65536);
    run<0>("global_load_lds_dwordx4 saddr + voffset", src);
    run<1>("global_load_lds_dwordx4 64-bit vaddr", src);
    run<2>("buffer_load_dwordx4 offen lds (MUBUF)", src);
    run<3>("global_load_dwordx4 to VGPRs (saddr)", src);
    run<4>("buffer_load_dwordx4 off lds, ADD_TID stride 16", src);
    run<5>("buffer_load_dword off lds, ADD_TID stride 4", src);
    return 0;
}
