// gemm_g4f: C = (A W^T) * scales (+ epilogue) on e4m3 operands and v_mfma_scale_f32_32x32x64_f8f6f4 -- the four-wave 256 x 256 tiling of
// gemm_g4.hip for the fp8 engine (weight_format = 1, BASELINE configs[4]; the reference has no fp8 path: parity unpinned).  K loop = one
// generated asm statement (gen_gemm_g4f.py: schedule, hazards, register map); this file computes the addresses and runs the shared
// vector epilogue with the dequantisation scales (gemm_epi.h, SC = true) on the wave tile's two 64-column halves.
// MX = false: per-token scales a_scale[m] (activations quantised by ln_modulate_k / quant_rows_fp8_k), unit block scales in the MFMA,
//             three A stages.  MX = true: A is an MX image (mx_a_s: one E8M0 scale per row and 32 elements, written by the attention /
//             GELU epilogues), the lane's block scales come straight into registers (one 16-byte load per K-tile).
#define S2V_HOST
#include "common.h"
#include "kernels.h"
#include "gemm_epi.h"
#include "gemm_g4f_regs.h"

typedef __attribute__((ext_vector_type(32))) float f32x32;
typedef __attribute__((ext_vector_type(16))) unsigned int u32x16;

template <int EPI, bool MX>
__global__ __launch_bounds__(256, 1) void gemm_g4f(const GemmArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_STRIDE = G4F_A3_A_STRIDE, W_BASE = G4F_A3_W_BASE, W_STRIDE = G4F_A3_W_STRIDE;
    const int tid = threadIdx.x, lane = tid & 63;
    clk_stamp(a.clk, gridDim.x >> 1, 0);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, hi = lane >> 5;

    // tile order as gemm_g4 / gemm_bf16_pp64: XCD x owns a contiguous range of the GM-grouped order
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int GM = a.gm > 0 ? a.gm : 4;
    const int per_group = GM * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_g = wg - group * per_group;
    const int m0 = (first_m + in_g % gsz) * 256, n0 = (in_g / gsz) * 256;

    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_base_u32(smem));
    // fragment addresses [A | W][stage parity][2 s + kk]: row (w * 128 + block * 32 + fr), 16-byte chunk 4 s + 2 kk + hi XOR-swizzled by
    // (row >> 1) & 7; the 32-row block is the instruction's immediate
    u32x16 vaddr, voff;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const unsigned inrow = (unsigned)((((x >> 1) * 4 + (x & 1) * 2 + hi) ^ ((fr >> 1) & 7)) << 4);
            vaddr[4 * g + x] = lds0 + g * A_STRIDE + (wm * 128 + fr) * 128 + inrow;
            vaddr[8 + 4 * g + x] = lds0 + W_BASE + g * W_STRIDE + (wn * 128 + fr) * 128 + inrow;
        }
    // staging as gemm_g4: piece p = rows p * 32 + wave * 8 + (lane >> 3), chunk XOR on the source address; operands are BYTES
    const int srow = wave * 8 + (lane >> 3);
    const int scol = ((lane & 7) ^ ((srow >> 1) & 7)) * 16;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        voff[p] = (unsigned)((int64_t)(srow + p * 32) * a.lda + scol);
        voff[8 + p] = (unsigned)((int64_t)(srow + p * 32) * a.ldw + scol);
    }
    const unsigned long long ap = (unsigned long long)((const char*)a.A + (int64_t)m0 * a.lda);
    const unsigned long long wp = (unsigned long long)((const char*)a.W + (int64_t)n0 * a.ldw);
    u32x4 ptr;
    ptr[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ap);
    ptr[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ap >> 32));
    ptr[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wp);
    ptr[3] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wp >> 32));
    u32x2 sin = {lds0 + wave * 1024, (unsigned)((a.K / 128 - 4) / 2)};
    // MX: the K-tile's scale dword of row (wave * 64 + lane) of the tile; the lane reads the dwords of its A-fragment rows wm * 128 + j * 32 + fr
    u32x4 vmx = {(unsigned)(8 * hi), 0u, 0u, 0u};
    u32x2 ssc = {0u, 0u};
    u32x2 ssm0 = {0u, 0u};  // {LDS address of the wave's 64 scale dwords in scale stage 0, bytes between the scale rows of consecutive K-tiles}
    if (MX) {  // K-tile major scales with the rows of a 128-row half permuted (kernels.h): the lane's four dwords are 16 consecutive bytes
        vmx[1] = (unsigned)((wm * 128 + fr * 4) * 4);
        const unsigned long long sp = (unsigned long long)((const char*)a.mx_a_s + (size_t)m0 * 4);
        ssc[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sp);
        ssc[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sp >> 32));
        ssm0[1] = (unsigned)(a.mx_rows * 4);
    }

    f32x32 AC[8];
    if (MX) {
        asm volatile(
#include "gemm_g4f_body_mx.inc"
            : "=" G4F_ACC0(AC[0]), "=" G4F_ACC1(AC[1]), "=" G4F_ACC2(AC[2]), "=" G4F_ACC3(AC[3]), "=" G4F_ACC4(AC[4]), "=" G4F_ACC5(AC[5]),
              "=" G4F_ACC6(AC[6]), "=" G4F_ACC7(AC[7]), "+" G4F_PTR(ptr), "+" G4F_SIN(sin), "+" G4F_VADDR(vaddr), "+" G4F_SSC(ssc)
            : G4F_VOFF(voff), G4F_VMX(vmx), G4F_SSM0(ssm0)
            : G4F_CLOBBERS);
    } else {
        asm volatile(
#include "gemm_g4f_body_a3.inc"
            : "=" G4F_ACC0(AC[0]), "=" G4F_ACC1(AC[1]), "=" G4F_ACC2(AC[2]), "=" G4F_ACC3(AC[3]), "=" G4F_ACC4(AC[4]), "=" G4F_ACC5(AC[5]),
              "=" G4F_ACC6(AC[6]), "=" G4F_ACC7(AC[7]), "+" G4F_PTR(ptr), "+" G4F_SIN(sin), "+" G4F_VADDR(vaddr)
            : G4F_VOFF(voff)
            : G4F_CLOBBERS, "v164", "v165", "v166", "v167", "s44", "s45", "s46", "s47");
    }
    __builtin_amdgcn_s_barrier();  // every wave is done with the stages: the epilogue patches alias them

    char* patch = smem + wave * 16384;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = AC[2 * (2 * h + i) + (j >> 1)][(j & 1) * 16 + e];
        epilogue_wave<EPI, 4, true>(a, acc, m0 + wm * 128, n0 + wn * 128 + h * 64, patch, lane);
    }
    clk_stamp(a.clk, gridDim.x >> 1, 1);
}

template <int EPI, bool MX>
static int launch_g4f_t(const GemmArgs& a_in, hipStream_t st) {
    GemmArgs a = a_in;
    const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 255) / 256;
    if (a.gm <= 0) a.gm = (tiles_n <= 16 && tiles_m >= 32 && a.K >= 16384) ? 1 : 4;  // as gemm_g4, K in bytes
    const int lds = MX ? G4F_MX_LDS_BYTES : G4F_A3_LDS_BYTES;
    const void* fn = (const void*)gemm_g4f<EPI, MX>;
    S2V_TRY(ensure_lds_attr(fn, lds));
    void* args[] = {(void*)&a, (void*)&tiles_m, (void*)&tiles_n};
    S2V_CHECK_HIP(hipLaunchKernel(fn, dim3(tiles_m * tiles_n), dim3(256), args, lds, st));
    return 0;
}

// e4m3 operands padded to whole 256-row tiles (launch_gemm_fp8 has checked the rest), an even number >= 4 of 128-byte K-tiles, the
// vector epilogue; the epilogue / scale combinations the fp8 engine uses
bool gemm_g4f_ok(const GemmArgs& a, int epi) {
    const bool mx = a.mx_a_s != nullptr;
    if (a.conv || a.splitk > 1 || a.m_begin != 0 || a.K % 256 != 0 || a.K / 128 < 4) return false;
    if (a.lda % 16 != 0 || a.ldw % 16 != 0 || !epi_vec_ok(a, epi)) return false;
    if (a.a_rows_padded < ((a.M + 255) / 256) * 256 || a.w_rows_padded < ((a.N + 255) / 256) * 256) return false;
    if (mx) return epi == EPI_BIAS || epi == EPI_BIAS_GATE_RES;
    return epi == EPI_BIAS || epi == EPI_BIAS_GELU || epi == EPI_BIAS_GATE_RES || epi == EPI_BIAS_QKNORM;
}

int launch_gemm_g4f(const GemmArgs& a, int epi, hipStream_t st) {
    const bool mx = a.mx_a_s != nullptr;
    switch (epi) {
        case EPI_BIAS: return mx ? launch_g4f_t<EPI_BIAS, true>(a, st) : launch_g4f_t<EPI_BIAS, false>(a, st);
        case EPI_BIAS_GATE_RES: return mx ? launch_g4f_t<EPI_BIAS_GATE_RES, true>(a, st) : launch_g4f_t<EPI_BIAS_GATE_RES, false>(a, st);
        case EPI_BIAS_GELU: return launch_g4f_t<EPI_BIAS_GELU, false>(a, st);
        case EPI_BIAS_QKNORM: return launch_g4f_t<EPI_BIAS_QKNORM, false>(a, st);
        default: return s2v_fail(__FILE__, __LINE__, "gemm_g4f: bad epilogue", -1);
    }
}
