// gemm_g4: C = A W^T (+ epilogue) on v_mfma_f32_32x32x16_bf16 for the plain (non-convolution) bf16 GEMMs of the denoise step
// (replaces the nn.Linear call sites attention_processor.py:2049-2051,2090; attention.py:1241-1243).
// 256 x 256 output tile per workgroup, FOUR waves (2 x 2), one per SIMD, 128 x 128 wave tiles; the K loop is one generated asm
// statement (gen_gemm_g4.py -> gemm_g4_body.inc: schedule, hazards and register map in its header); this file computes the
// addresses, and after the loop runs the shared vector epilogue (gemm_epi.h) on the wave tile's two 64-column halves through a
// 16-KiB LDS patch that aliases the (then idle) operand stages.
#define S2V_HOST
#include "common.h"
#include "kernels.h"
#include "gemm_epi.h"
#include "gemm_g4_regs.h"
#include <cstdlib>
#include <type_traits>

typedef __attribute__((ext_vector_type(32))) float f32x32;
typedef __attribute__((ext_vector_type(16))) unsigned int u32x16;
typedef __attribute__((ext_vector_type(8))) unsigned int u32x8;

#ifdef S2V_DIAG
// cycle accounting (tools/stall_g4.py): sums over all workgroups (wave 0) of [prologue, K loop, epilogue] in s_memtime cycles, the number of
// workgroups, and s_memtime / s_memrealtime spans of workgroup 100 (shader clock)
__device__ unsigned long long g_g4_dbg[8];
extern "C" __attribute__((visibility("default"))) int s2v_g4_debug_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_g4_dbg), sizeof(g_g4_dbg)) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_g4_dbg), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define G4_STAMP(i) const long long g4_t##i = (long long)__builtin_amdgcn_s_memtime()
#else
#define G4_STAMP(i) do { } while (0)
#endif

// T16 = f16_t (round 5): the fp16 model dtype's linears -- the K loop with v_mfma_f32_32x32x16_f16 (gemm_g4_body_f16.inc: the same generated
// statement, one mnemonic changed) and the shared vector epilogue decoding / packing fp16 (gemm_epi.h H2<T16>); no split K, no fused q/k norm
template <int EPI, typename T16 = bf16_t>
__global__ __launch_bounds__(256, 1) void gemm_g4(const GemmArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x [A 256 rows x 128 B | W 256 rows x 128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    clk_stamp(a.clk, gridDim.x >> 1, 0);
    G4_STAMP(0);
#ifdef S2V_DIAG
    const long long g4_r0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, hi = lane >> 5;

    // tile order as gemm_bf16_pp64: XCD x owns a contiguous range of the GM-grouped order (neighbouring tiles share A / W panels in
    // that XCD's L2)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int wg_all = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    // split K: workgroups [z * tiles, (z + 1) * tiles) of the order reduce K range z of every tile
    const int S = a.splitk > 1 ? a.splitk : 1, ntile = tiles_m * tiles_n;
    const int z = wg_all / ntile, wg = wg_all - z * ntile;
    const int Ks = a.K / S;
    const int GM = a.gm > 0 ? a.gm : 4;
    const int per_group = GM * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_g = wg - group * per_group;
    const int m0 = (first_m + in_g % gsz) * 256, n0 = (in_g / gsz) * 256;

    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_base_u32(smem));
    // fragment addresses [A | W][stage][k-step]: row (w * 128 + block * 32 + fr) of the operand image, 16-B chunk (2 s + hi) XOR-swizzled
    // by (row >> 1) & 7 = (fr >> 1) & 7; the 32-row block is the instruction's immediate
    u32x16 vaddr, voff;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned inrow = (unsigned)(((s * 2 + hi) ^ ((fr >> 1) & 7)) << 4);
            vaddr[4 * g + s] = lds0 + g * G4_A_STRIDE + (wm * 128 + fr) * 128 + inrow;
            vaddr[8 + 4 * g + s] = lds0 + G4_W_BASE + g * G4_W_STRIDE + (wn * 128 + fr) * 128 + inrow;
        }
    // staging: piece p (0..7) of an operand image = rows p * 32 + wave * 8 + (lane >> 3), 8 chunks of 16 B, chunk XOR on the SOURCE
    // address (the XOR term does not depend on p); global address = K-tile base (SGPR pair) + per-lane 32-bit offset
    const int srow = wave * 8 + (lane >> 3);
    const int scol = ((lane & 7) ^ ((srow >> 1) & 7)) * 8;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        voff[p] = (unsigned)(2 * ((int64_t)(srow + p * 32) * a.lda + scol));
        voff[8 + p] = (unsigned)(2 * ((int64_t)(srow + p * 32) * a.ldw + scol));
    }
    // L2 prefetch: lane l touches the cache line of row (l & 7) * 32 + wave * 8 + (l >> 3) -- the 64 rows of this wave's eight pieces
    u32x2 vpf;
    vpf[0] = (unsigned)(2 * (int64_t)((lane & 7) * 32 + srow) * a.lda);
    vpf[1] = (unsigned)(2 * (int64_t)((lane & 7) * 32 + srow) * a.ldw);
    const unsigned long long ap = (unsigned long long)((const char*)a.A + 2 * ((int64_t)m0 * a.lda + (int64_t)z * Ks));
    const unsigned long long wp = (unsigned long long)((const char*)a.W + 2 * ((int64_t)n0 * a.ldw + (int64_t)z * Ks));
    u32x4 ptr;
    ptr[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ap);
    ptr[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ap >> 32));
    ptr[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wp);
    ptr[3] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wp >> 32));
    u32x2 sin = {lds0 + wave * 1024, (unsigned)((Ks / 64 - 2 - (G4_PF > 2 ? G4_PF : 2)) / 2)};  // pairs of K-tiles in the asm loop

    // split K: slot (tile, z) of the fp32 partials, [wave][64 register quads][lane] x 16 B
    const int tile = (m0 >> 8) * tiles_n + (n0 >> 8);
    const unsigned long long ub = (unsigned long long)(a.sk_ws + ((size_t)tile * S * 4 + wave) * 16384), zb = ub + (unsigned long long)z * 262144;
    const unsigned voff16 = lane * 16;
    u32x4 sk;
    sk[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)zb);
    sk[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(zb >> 32));
    sk[2] = (unsigned)__builtin_amdgcn_readfirstlane(S);
    sk[3] = 0;

    G4_STAMP(1);
    f32x32 AC[8];  // acc[i][j] (i: 32-column block of W rows, j: 32-row block of A rows) = registers 64 i + 16 j of a[0:255]
    if constexpr (std::is_same<T16, f16_t>::value) {
        asm volatile(
#include "gemm_g4_body_f16.inc"
            : "=" G4_ACC0(AC[0]), "=" G4_ACC1(AC[1]), "=" G4_ACC2(AC[2]), "=" G4_ACC3(AC[3]), "=" G4_ACC4(AC[4]), "=" G4_ACC5(AC[5]),
              "=" G4_ACC6(AC[6]), "=" G4_ACC7(AC[7]), "+" G4_PTR(ptr), "+" G4_SIN(sin), "+" G4_VADDR(vaddr)
            : G4_VOFF(voff), G4_VPF(vpf), G4_SK(sk), G4_VSK(voff16)
            : G4_CLOBBERS);
    } else {
        asm volatile(
#include "gemm_g4_body.inc"
            : "=" G4_ACC0(AC[0]), "=" G4_ACC1(AC[1]), "=" G4_ACC2(AC[2]), "=" G4_ACC3(AC[3]), "=" G4_ACC4(AC[4]), "=" G4_ACC5(AC[5]),
              "=" G4_ACC6(AC[6]), "=" G4_ACC7(AC[7]), "+" G4_PTR(ptr), "+" G4_SIN(sin), "+" G4_VADDR(vaddr)
            : G4_VOFF(voff), G4_VPF(vpf), G4_SK(sk), G4_VSK(voff16)
            : G4_CLOBBERS);
    }
    G4_STAMP(2);
    __builtin_amdgcn_s_barrier();  // every wave is done with the stages: the epilogue patches alias them

    if (S > 1) {
        // Hand-over of the fp32 partial tiles (gen_gemm_g4.py gen_sk_store / gen_sk_sum: layout [wave][64 register quads][lane] x 16 B,
        // slot (tile, z)).  Every access to them is sc1 (device scope: written through, never served from an L2) -- with plain
        // accesses the hand-over needs __threadfence(), an L2 write-back + invalidate per wave that also evicts the operands of the
        // workgroups still in their K loops on the same XCD (measured: 145 us against 130 us unsplit at 80 tiles x K 7680).
        // the partial of this workgroup was stored at the end of the K-loop statement
        __syncthreads();
        unsigned* flag = (unsigned*)smem;
        if (tid == 0) *flag = __hip_atomic_fetch_add(a.sk_cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != (unsigned)(S - 1)) return;
        __syncthreads();  // flag is read by everyone before the patches overwrite it
        if (tid == 0) __hip_atomic_store(a.sk_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the next launch finds zero
        // the last workgroup to arrive adds the S partials in split order (the sum must not depend on who was last): the others' from
        // memory, its own from its registers at its place in the order
        {
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ub), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ub >> 32));
            const unsigned ns = (unsigned)__builtin_amdgcn_readfirstlane(S), zs = (unsigned)__builtin_amdgcn_readfirstlane(z);
            asm volatile(
#include "gemm_g4_sk_sum.inc"
                : "+" G4_ACC0(AC[0]), "+" G4_ACC1(AC[1]), "+" G4_ACC2(AC[2]), "+" G4_ACC3(AC[3]), "+" G4_ACC4(AC[4]), "+" G4_ACC5(AC[5]),
                  "+" G4_ACC6(AC[6]), "+" G4_ACC7(AC[7])
                : [lo] "s"(lo), [hi] "s"(hi), [voff] "v"(voff16), [ns] "s"(ns), [z] "s"(zs)
                : G4_SK_CLOBBERS);
        }
    }

    char* patch = smem + wave * 16384;
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // the wave tile's two 64-column halves through the 128 x 64 epilogue of the eight-wave kernels
        f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = AC[2 * (2 * h + i) + (j >> 1)][(j & 1) * 16 + e];
        epilogue_wave<EPI, 4, false, T16>(a, acc, m0 + wm * 128, n0 + wn * 128 + h * 64, patch, lane);
    }
#ifdef S2V_DIAG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G4_STAMP(3);
    if (tid == 0) {
        atomicAdd(&g_g4_dbg[0], (unsigned long long)(g4_t1 - g4_t0));
        atomicAdd(&g_g4_dbg[1], (unsigned long long)(g4_t2 - g4_t1));
        atomicAdd(&g_g4_dbg[2], (unsigned long long)(g4_t3 - g4_t2));
        atomicAdd(&g_g4_dbg[3], 1ull);
        if (blockIdx.x == 100) {
            g_g4_dbg[4] = (unsigned long long)(g4_t3 - g4_t0);
            g_g4_dbg[5] = (unsigned long long)((long long)__builtin_amdgcn_s_memrealtime() - g4_r0);
        }
    }
#endif
    clk_stamp(a.clk, gridDim.x >> 1, 1);
}

template <int EPI, typename T16 = bf16_t>
static int launch_g4_t(const GemmArgs& a_in, hipStream_t st) {
    GemmArgs a = a_in;
    const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 255) / 256;
    if (a.gm <= 0) {
        a.gm = (tiles_n <= 16 && tiles_m >= 32 && a.K >= 8192) ? 1 : 4;  // measured (tools/stall_g4.py, S2V_G4_GM): N = 3072, K = 12288 is 7 % faster with 1, everything else with >= 4
#ifdef S2V_DIAG
        if (const char* e = getenv("S2V_G4_GM")) a.gm = atoi(e);
#endif
    }
    const void* fn = (const void*)gemm_g4<EPI, T16>;
    S2V_TRY(ensure_lds_attr(fn, G4_LDS_BYTES));
    void* args[] = {(void*)&a, (void*)&tiles_m, (void*)&tiles_n};
    const int S = a.splitk > 1 ? a.splitk : 1;
    S2V_REQUIRE(S == 1 || (a.sk_ws && a.sk_cnt), "gemm_g4: split K needs the partial workspace and the arrival counters");
    S2V_CHECK_HIP(hipLaunchKernel(fn, dim3(tiles_m * tiles_n * S), dim3(256), args, G4_LDS_BYTES, st));
    return 0;
}

// plain bf16 operands, whole 256-row tiles present behind A (a_rows_padded) and W, an even number of K-tiles and enough of them for the
// prefetch distance, vector epilogue
bool gemm_g4_ok(const GemmArgs& a, int epi) {
    const int S = a.splitk > 1 ? a.splitk : 1;
    const int nT = a.K / S / 64;
    return !a.conv && a.K % (128 * S) == 0 && nT >= (G4_PF > 2 ? G4_PF : 2) + 2 && a.lda % 8 == 0 && a.ldw % 8 == 0 && epi_vec_ok(a, epi) && a.m_begin == 0 &&
           a.a_rows_padded >= ((a.M + 255) / 256) * 256;
}
int gemm_choose_splitk(int64_t tiles, int K, int64_t ncu) {
    if (tiles * 2 > ncu) return 1;
    int S = (int)(ncu / tiles < 4 ? ncu / tiles : 4);  // the sum adds at most four partials
    while (S > 1 && !(K % (128 * S) == 0 && K / (64 * S) >= 16)) --S;
    return S;
}
// fp16 operands: the same conditions as gemm_g4_ok without split K
bool gemm_g4_f16_ok(const GemmArgs& a, int epi) { return a.splitk <= 1 && a.mx_out_q == nullptr && gemm_g4_ok(a, epi); }
int launch_gemm_g4_f16(const GemmArgs& a, int epi, hipStream_t st) {
    S2V_REQUIRE(gemm_g4_f16_ok(a, epi), "gemm_g4 (fp16): shape / epilogue not supported");
    switch (epi) {
        case EPI_BIAS: return launch_g4_t<EPI_BIAS, f16_t>(a, st);
        case EPI_BIAS_GELU: return launch_g4_t<EPI_BIAS_GELU, f16_t>(a, st);
        case EPI_BIAS_GATE_RES: return launch_g4_t<EPI_BIAS_GATE_RES, f16_t>(a, st);
        case EPI_BIAS_ADD: return launch_g4_t<EPI_BIAS_ADD, f16_t>(a, st);
        case EPI_BIAS_QKNORM: return launch_g4_t<EPI_BIAS_QKNORM, f16_t>(a, st);
        default: return s2v_fail(__FILE__, __LINE__, "gemm_g4 (fp16): bad epilogue", -1);
    }
}
int launch_gemm_g4(const GemmArgs& a, int epi, hipStream_t st) {
    switch (epi) {
        case EPI_BIAS: return launch_g4_t<EPI_BIAS>(a, st);
        case EPI_BIAS_GELU: return launch_g4_t<EPI_BIAS_GELU>(a, st);
        case EPI_BIAS_GATE_RES: return launch_g4_t<EPI_BIAS_GATE_RES>(a, st);
        case EPI_BIAS_ADD: return launch_g4_t<EPI_BIAS_ADD>(a, st);
        case EPI_BIAS_QKNORM: return launch_g4_t<EPI_BIAS_QKNORM>(a, st);
        default: return s2v_fail(__FILE__, __LINE__, "gemm_g4: bad epilogue", -1);
    }
}
