// gemm_g4t: the four-wave 256 x 256 bf16 GEMM of gemm_g4.hip as a PERSISTENT kernel -- one workgroup per CU walks its share of the
// output tiles -- whose epilogue is trickled through the MFMA gaps of the next tile's K loop (gen_gemm_g4t.py: design, hazards, register
// map).  Replaces the nn.Linear + GELU call site attention.py:1237-1243 (FF1) where the shape qualifies (gemm_g4t_ok); everything else
// stays on gemm_g4 / gemm_bf16_pp64.  Bit-identical to them: the asm restates gemm_epi.h's arithmetic instruction for instruction.
// This file: tile order (the XCD-contiguous, GM-grouped order of gemm_g4, dealt round by round to the 32 workgroups of an XCD), the
// per-workgroup tile records in LDS, the per-lane addresses, and the C++ epilogue of a workgroup's LAST tile.
#define S2V_HOST
#include "common.h"
#include "kernels.h"
#include "gemm_epi.h"
#include "gemm_g4t_regs.h"

typedef __attribute__((ext_vector_type(32))) float f32x32;
typedef __attribute__((ext_vector_type(16))) unsigned int u32x16;

// tile index (position in the XCD-contiguous order) -> (m0, n0), as gemm_g4
__device__ __forceinline__ void g4t_tile_origin(int wg, int tiles_m, int tiles_n, int GM, int& m0, int& n0) {
    const int per_group = GM * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_g = wg - group * per_group;
    m0 = (first_m + in_g % gsz) * 256;
    n0 = (in_g / gsz) * 256;
}

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_g4t(const GemmArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 operand stages 128 KiB | 4 patches x 4 KiB | tile records]
    const int tid = threadIdx.x, lane = tid & 63;
    clk_stamp(a.clk, 0, 0);  // persistent: workgroup 0 lives for the whole launch
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, hi = lane >> 5;

    // this workgroup's tiles: XCD x owns a contiguous range of the order; its P workgroups take it round by round
    const int nwg = gridDim.x, bid = blockIdx.x, ntile = tiles_m * tiles_n;
    const int P = nwg >> 3, xcd = bid & 7, slot = bid >> 3;
    const int q = ntile >> 3, r = ntile & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, len = q + (xcd < r ? 1 : 0);
    const int cnt = slot < len ? (len - slot + P - 1) / P : 0;
    if (cnt == 0) return;
    const int GM = a.gm > 0 ? a.gm : 4;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_base_u32(smem));
    if (tid < cnt) {  // record tid: {A tile base, W tile base, C tile base, bias byte offset}
        int m0, n0;
        g4t_tile_origin(start + slot + tid * P, tiles_m, tiles_n, GM, m0, n0);
        unsigned long long* rec = (unsigned long long*)(smem + G4T_TABLE_BASE + tid * 32);
        rec[0] = (unsigned long long)((const char*)a.A + 2 * (int64_t)m0 * a.lda);
        rec[1] = (unsigned long long)((const char*)a.W + 2 * (int64_t)n0 * a.ldw);
        rec[2] = (unsigned long long)((char*)a.C + 2 * ((int64_t)m0 * a.ldc + n0));
        // qknorm: + (m0 << 2 | kind of the tile's heads: 0 q, 1 k, 2 v) for the trickle of the NEXT tile's K loop
        const unsigned kind = EPI == EPI_BIAS_QKNORM ? (n0 >= 2 * a.qk_D ? 2u : n0 >= a.qk_D ? 1u : 0u) : 0u;
        rec[3] = (unsigned long long)(unsigned)(2 * n0) | ((unsigned long long)(((unsigned)m0 << 2) | kind) << 32);
    }
    if constexpr (EPI == EPI_BIAS_QKNORM) {
        // the trickle's constant block: rotary table, position arithmetic, eps; the LayerNorm weights / biases of q and k (4 x 64 bf16)
        unsigned* cb = (unsigned*)(smem + G4T_QK_CONST_BASE);
        if (tid == 0) {
            const unsigned long long cs = (unsigned long long)a.qk_cs;
            cb[0] = (unsigned)cs; cb[1] = (unsigned)(cs >> 32);
            cb[2] = (unsigned)a.tok_per_batch; cb[3] = (unsigned)a.text_len;
            cb[4] = __float_as_uint(1.0f / (float)a.tok_per_batch); cb[5] = __float_as_uint(a.qk_eps);
        }
        if (tid < 128) {
            const int which = tid >> 5, e = tid & 31;  // 0: q weight, 1: k weight, 2: q bias, 3: k bias
            const unsigned* src = (const unsigned*)(which < 2 ? a.qk_w[which] : a.qk_b[which - 2]);
            cb[G4T_QK_LN_OFF / 4 + tid] = src[e];
        }
    }
    __syncthreads();

    u32x16 vaddr, voff;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned inrow = (unsigned)(((s * 2 + hi) ^ ((fr >> 1) & 7)) << 4);
            vaddr[4 * g + s] = lds0 + g * G4T_A_STRIDE + (wm * 128 + fr) * 128 + inrow;
            vaddr[8 + 4 * g + s] = lds0 + G4T_W_BASE + g * G4T_W_STRIDE + (wn * 128 + fr) * 128 + inrow;
        }
    const int srow = wave * 8 + (lane >> 3);
    const int scol = ((lane & 7) ^ ((srow >> 1) & 7)) * 8;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        voff[p] = (unsigned)(2 * ((int64_t)(srow + p * 32) * a.lda + scol));
        voff[8 + p] = (unsigned)(2 * ((int64_t)(srow + p * 32) * a.ldw + scol));
    }
    // the trickle's lane constants: store offset, patch write / read addresses (gemm_epi.h's swizzle: 16-byte chunk ^ (row & 7)), bias offset
    const unsigned pbase = lds0 + G4T_PATCH_BASE + wave * G4T_PATCH_WAVE;
    u32x4 vlane;
    vlane[0] = (unsigned)(2 * ((int64_t)(lane >> 3) * a.ldc + (lane & 7) * 8));
    vlane[1] = pbase + fr * 128 + ((fr & 7) << 4) + hi * 8;
    vlane[2] = pbase + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);
    vlane[3] = hi * 8;
    if constexpr (EPI == EPI_BIAS_QKNORM) vlane[3] |= (unsigned)(wm * 128 + (lane >> 3)) << 8;  // the q/k-norm trickle's first token row of the lane
    unsigned vtab = lds0 + G4T_TABLE_BASE;
    u32x4 ptr = {0, 0, 0, 0};
    const unsigned sin0 = lds0 + wave * 1024;
    u32x2 sin1 = {(unsigned)((a.K / 64 - 4) / 2), (unsigned)cnt};
    const unsigned sin2 = (unsigned)(16 * a.ldc);
    u32x2 sin3 = {(unsigned)(2 * (wm * 128 * a.ldc + wn * 128)), (unsigned)(wn * 256)};
    const unsigned long long bp = (unsigned long long)a.bias;
    u32x2 sin4 = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bp), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bp >> 32))};

    f32x32 AC[8];
    if constexpr (EPI == EPI_BIAS_QKNORM) {
        const unsigned qcb = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + G4T_QK_CONST_BASE));
        asm volatile(
#include "gemm_g4t_body_qknorm.inc"
            : "=" G4T_ACC0(AC[0]), "=" G4T_ACC1(AC[1]), "=" G4T_ACC2(AC[2]), "=" G4T_ACC3(AC[3]), "=" G4T_ACC4(AC[4]), "=" G4T_ACC5(AC[5]),
              "=" G4T_ACC6(AC[6]), "=" G4T_ACC7(AC[7]), "+" G4T_PTR(ptr), "+" G4T_SIN1(sin1), "+" G4T_VTAB(vtab)
            : G4T_VADDR(vaddr), G4T_VOFF(voff), G4T_VLANE(vlane), G4T_SIN0(sin0), G4T_SIN2(sin2), G4T_SIN3(sin3), G4T_SIN4(sin4), G4T_QK_CB(qcb)
            : G4T_CLOBBERS, G4T_QK_CLOBBERS);
    } else if (EPI == EPI_BIAS_GELU) {
        asm volatile(
#include "gemm_g4t_body_gelu.inc"
            : "=" G4T_ACC0(AC[0]), "=" G4T_ACC1(AC[1]), "=" G4T_ACC2(AC[2]), "=" G4T_ACC3(AC[3]), "=" G4T_ACC4(AC[4]), "=" G4T_ACC5(AC[5]),
              "=" G4T_ACC6(AC[6]), "=" G4T_ACC7(AC[7]), "+" G4T_PTR(ptr), "+" G4T_SIN1(sin1), "+" G4T_VTAB(vtab)
            : G4T_VADDR(vaddr), G4T_VOFF(voff), G4T_VLANE(vlane), G4T_SIN0(sin0), G4T_SIN2(sin2), G4T_SIN3(sin3), G4T_SIN4(sin4)
            : G4T_CLOBBERS);
    } else {
        asm volatile(
#include "gemm_g4t_body_bias.inc"
            : "=" G4T_ACC0(AC[0]), "=" G4T_ACC1(AC[1]), "=" G4T_ACC2(AC[2]), "=" G4T_ACC3(AC[3]), "=" G4T_ACC4(AC[4]), "=" G4T_ACC5(AC[5]),
              "=" G4T_ACC6(AC[6]), "=" G4T_ACC7(AC[7]), "+" G4T_PTR(ptr), "+" G4T_SIN1(sin1), "+" G4T_VTAB(vtab)
            : G4T_VADDR(vaddr), G4T_VOFF(voff), G4T_VLANE(vlane), G4T_SIN0(sin0), G4T_SIN2(sin2), G4T_SIN3(sin3), G4T_SIN4(sin4)
            : G4T_CLOBBERS);
    }
    __builtin_amdgcn_s_barrier();  // every wave is done with the stages: the C++ epilogue's patches alias them

    // the workgroup's last tile: the shared vector epilogue, as gemm_g4
    int m0, n0;
    g4t_tile_origin(start + slot + (cnt - 1) * P, tiles_m, tiles_n, GM, m0, n0);
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
        epilogue_wave<EPI, 4>(a, acc, m0 + wm * 128, n0 + wn * 128 + h * 64, patch, lane);
    }
    clk_stamp(a.clk, 0, 1);
}

template <int EPI>
static int launch_g4t_t(const GemmArgs& a_in, hipStream_t st) {
    GemmArgs a = a_in;
    const int tiles_m = a.M / 256, tiles_n = a.N / 256;
    if (a.gm <= 0) a.gm = (tiles_n <= 16 && tiles_m >= 32 && a.K >= 8192) ? 1 : 4;  // as gemm_g4
    int dev = 0, ncu = 256;
    S2V_CHECK_HIP(hipGetDevice(&dev));
    S2V_CHECK_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const int grid = (ncu / 8) * 8;
    const void* fn = (const void*)gemm_g4t<EPI>;
    const int lds = G4T_LDS_BYTES + (EPI == EPI_BIAS_QKNORM ? G4T_QK_CONST_BYTES : 0);
    S2V_TRY(ensure_lds_attr(fn, lds));
    void* args[] = {(void*)&a, (void*)&tiles_m, (void*)&tiles_n};
    S2V_CHECK_HIP(hipLaunchKernel(fn, dim3(grid), dim3(256), args, lds, st));
    return 0;
}

// whole 256 x 256 tiles only (the engine splits a partial last row tile off before it gets here), a bias, enough K-tiles to carry the
// trickle, at least two rounds of tiles (one round has nothing to hide an epilogue behind) and no more than the record table holds
bool gemm_g4t_ok(const GemmArgs& a, int epi, int ncu) {
    if (epi != EPI_BIAS_GELU && epi != EPI_BIAS && epi != EPI_BIAS_QKNORM) return false;
    if (a.conv || a.splitk > 1 || a.mx_out_q || !a.bias || a.m_begin != 0) return false;
    if (a.M % 256 != 0 || a.N % 256 != 0 || a.K % 128 != 0) return false;
    // the trickled q/k-norm: whole tiles of q, k or v heads, a rotary table (the 2B model's no-rotary form stays on the C++ epilogue), rows
    // that convert exactly to float
    if (epi == EPI_BIAS_QKNORM && (a.qk_cs == nullptr || a.qk_D % 256 != 0 || a.N != 3 * a.qk_D || a.tok_per_batch <= 0 || a.M >= (1 << 24))) return false;
    const int tk = epi == EPI_BIAS_GELU ? G4T_TK_GELU : epi == EPI_BIAS_QKNORM ? G4T_TK_QKNORM : G4T_TK_BIAS;
    if (a.K / 64 < tk + 4) return false;
    if (a.lda % 8 != 0 || a.ldw % 8 != 0 || a.ldc % 8 != 0 || !epi_vec_ok(a, epi)) return false;
    const int64_t ntile = (int64_t)(a.M / 256) * (a.N / 256), grid = (ncu / 8) * 8;
    if (grid < 8 || ntile < 2 * grid) return false;
    if ((ntile / 8 + 1 + grid / 8 - 1) / (grid / 8) > G4T_TABLE_RECORDS) return false;
    if (a.w_rows_padded && a.w_rows_padded < a.N) return false;
    return true;
}

int launch_gemm_g4t(const GemmArgs& a, int epi, hipStream_t st) {
    switch (epi) {
        case EPI_BIAS: return launch_g4t_t<EPI_BIAS>(a, st);
        case EPI_BIAS_GELU: return launch_g4t_t<EPI_BIAS_GELU>(a, st);
        case EPI_BIAS_QKNORM: return launch_g4t_t<EPI_BIAS_QKNORM>(a, st);
        default: return s2v_fail(__FILE__, __LINE__, "gemm_g4t: bad epilogue", -1);
    }
}
