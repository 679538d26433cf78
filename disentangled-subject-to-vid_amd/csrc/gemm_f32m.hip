// fp32 GEMM on the matrix pipe: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; 64 cycles per instruction and SIMD = the fp32 vector
// rate, 157 TFLOP/s on the part -- MI355X_MICROARCH.md "FP32-input MFMA").  The instruction is bit-for-bit a k-ordered fmaf chain
// (D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)), one rounding per product), so with the K index fed in ascending order this kernel returns
// EXACTLY the bits of gemm_simple_k<float> (gemm.hip: acc = fmaf(a, w, acc) for k = 0 .. K-1, same zero fill of the K tail, same
// epilogue4) -- tests/test_gpu_f32m.py holds it to that.  It is the CPU-reference-parity mode (fp32 model dtype, <= 1e-3 against the
// oracle) made fast enough to serve as the on-GPU reference of a whole run at the headline geometry: the VALU
// kernel runs the C3 step in minutes, this one in seconds.
//
// Tile 128 (m) x 128 (n) x 16 (k), four waves (2 x 2) of 64 x 64 = 2 x 2 MFMA blocks, two workgroups per CU.  Operands are staged
// through registers into TRANSPOSED LDS images [k][row] (pitch rows + 32 floats: the two k-rows a wave reads per instruction -- lanes 0-31
// take k, lanes 32-63 take k + 1 -- fall into the two halves of the banks), so a fragment read is one ds_read_b32 of 64 consecutive
// floats per half.
// Round 6: the tile is a template parameter (128 or 64 rows x 128 or 64 columns, still 2 x 2 waves): a GEMM whose 128 x 128
// tiles leave the CUs unevenly loaded (configs[0]: M = 2500, N = 1920 -> 300 tiles on 256 CUs, i.e. the time of TWO tiles per CU for 1.17 tiles
// of work) runs on smaller tiles -- the fp32 MFMA is slow enough (64 cycles for 32 x 32 x 2) that the fragment traffic of a 64 x 64 tile is no
// bound, and the bits do not depend on the tile (every output element is the same k-ordered chain): f32m_pick_tile, tests/test_gpu_f32m.py.
// The MFMA is issued swapped (first operand = weight rows): a lane then owns four consecutive output columns of one
// token row, which is what epilogue4 takes.  Tile order: XCD-aware (each XCD walks a contiguous range of tiles) and grouped by 8 row
// tiles so the 64 workgroups resident on an XCD share A / W tiles through its L2.
#define S2V_HOST
#include "common.h"
#include "kernels.h"
#include "gemm_epi.h"

#define FBK 16

// element offsets of A (plain or implicit-GEMM convolution; the same arithmetic as gemm.hip's a_row_base / a_k_off)
__device__ __forceinline__ int64_t f32m_row_base(const GemmArgs& a, int m) {
    if (!a.conv) return (int64_t)m * a.lda;
    const int hw = a.oH * a.oW;
    const int f = m / hw, rem = m - f * hw;
    const int y = rem / a.oW, x = rem - y * a.oW;
    const int cs = a.cstride > 1 ? a.cstride : 1;
    return (((int64_t)f * a.Hp + y * cs) * a.Wp + x * cs) * a.cin;
}
__device__ __forceinline__ int64_t f32m_k_off(const GemmArgs& a, int k) {
    if (!a.conv) return k;
    const int tap = k / a.cin, ci = k - tap * a.cin;
    const int dt = a.kt == 3 ? tap / 9 : 0;
    const int r9 = tap - dt * 9;
    const int dy = r9 / 3, dx = r9 - dy * 3;
    return (((int64_t)dt * a.Hp + dy) * a.Wp + dx) * a.cin + ci;
}

template <int EPI, int FBM, int FBN>
__global__ __launch_bounds__(256, (FBM * FBN <= 64 * 64 ? 4 : 2)) void gemm_f32m_k(const GemmArgs a, int tiles_m, int tiles_n) {
    static_assert((FBM == 128 || FBM == 64) && (FBN == 128 || FBN == 64), "gemm_f32m: tile sides are 64 or 128");
    constexpr int WM = FBM / 2, WN = FBN / 2, MJ = WM / 32, NI = WN / 32;  // wave tile and its 32 x 32 MFMA blocks
    constexpr int CA = FBM * 4 / 256, CW = FBN * 4 / 256;                    // 4-float chunks of a 16-float K-tile row each thread stages
    __shared__ float sA[2][FBK][FBM + 32];
    __shared__ float sW[2][FBK][FBN + 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // tile of this workgroup: hardware deals workgroup ids round-robin over the 8 XCDs -> XCD x gets the contiguous range x of tiles
    const int total = tiles_m * tiles_n;
    int v = blockIdx.x;
    {
        const int per = total >> 3, rem = total & 7;  // the first `rem` XCDs own one tile more
        const int x = v & 7, slot = v >> 3;
        v = x * per + min(x, rem) + slot;
    }
    const int GM = 8;
    const int group = v / (GM * tiles_n), first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_g = v - group * GM * tiles_n;
    const int tm_i = first_m + in_g % gsz, tn_i = in_g / gsz;
    const int m0 = tm_i * FBM, n0 = tn_i * FBN;

    const float* A = (const float*)a.A;
    const float* W = (const float*)a.W;
    // staging: thread t copies chunks c, c + 256 / rows, ... (of the 4 four-float chunks of a 16-float K-tile) of row t % rows of either operand:
    // two chunks per thread for a 128-row side, one for a 64-row side
    const int rA = tid % FBM, cA = tid / FBM, rW = tid % FBN, cW = tid / FBN;
    const int64_t baseA = f32m_row_base(a, min(m0 + rA, a.M - 1));
    const int64_t baseW = (int64_t)min(n0 + rW, a.N - 1) * a.ldw;
    f32x4 ra[CA], rw[CW];
    auto gload = [&](int kt) {
#pragma unroll
        for (int j = 0; j < CA; ++j) {
            const int k = kt * FBK + (cA + (256 / FBM) * j) * 4;
            ra[j] = k < a.K ? *(const f32x4*)(A + baseA + f32m_k_off(a, k)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < CW; ++j) {
            const int k = kt * FBK + (cW + (256 / FBN) * j) * 4;
            rw[j] = k < a.K ? *(const f32x4*)(W + baseW + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < CA; ++j) {
            const int kc = (cA + (256 / FBM) * j) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) sA[buf][kc + e][rA] = ra[j][e];
        }
#pragma unroll
        for (int j = 0; j < CW; ++j) {
            const int kc = (cW + (256 / FBN) * j) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) sW[buf][kc + e][rW] = rw[j][e];
        }
    };
    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
    const int fr = lane & 31, hi = lane >> 5;
    f32x16 acc[NI][MJ];  // [n block][m block]: D[i = n][j = m]
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nkt = (a.K + FBK - 1) / FBK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
#pragma unroll
        for (int kk = 0; kk < FBK; kk += 2) {
            float wv[NI], av[MJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) wv[i] = sW[buf][kk + hi][wn + 32 * i + fr];
#pragma unroll
            for (int j = 0; j < MJ; ++j) av[j] = sA[buf][kk + hi][wm + 32 * j + fr];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[i], av[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nkt) lstore(buf ^ 1);
        __syncthreads();
    }
    // D register e of a lane: i = (e & 3) + 8 * (e >> 2) + 4 * hi (output column), j = fr (token row)
#pragma unroll
    for (int bj = 0; bj < MJ; ++bj) {
        const int m = m0 + wm + bj * 32 + fr;
#pragma unroll
        for (int bi = 0; bi < NI; ++bi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn + bi * 32 + 8 * g + 4 * hi;
                const float vv[4] = {acc[bi][bj][4 * g], acc[bi][bj][4 * g + 1], acc[bi][bj][4 * g + 2], acc[bi][bj][4 * g + 3]};
                if (n < a.N) epilogue4<float, EPI>(a, m, n, vv);
            }
    }
}

bool gemm_f32m_ok(const GemmArgs& a) {
    if (a.valu_only || a.m_begin != 0 || a.splitk > 1) return false;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return false;
    if (a.K % 4 != 0 || a.ldw % 4 != 0) return false;
    if (a.conv ? (a.cin % 4 != 0) : (a.lda % 4 != 0)) return false;
    if (((uintptr_t)a.A & 15) != 0 || ((uintptr_t)a.W & 15) != 0) return false;
    const int64_t tiles = (int64_t)((a.M + 63) / 64) * ((a.N + 63) / 64);
    return tiles < (1 << 30);
}

// Tile choice (round 6).  The matrix pipe of a CU is what the workgroups on it share, and the dispatcher deals tiles to CUs as evenly as their count
// allows: a launch takes about ceil(tiles / CUs) tiles' time.  Cost of a candidate = ceil(tiles / CUs) x tile area x (1 + a small penalty for the
// shorter sides: more operand traffic per MFMA and more prologues); the cheapest wins, ties go to the larger tile.  At C3 sizes (thousands of tiles)
// every candidate costs the same within the penalty and 128 x 128 stays; at configs[0] (M = 2500) the N = 1920 GEMMs go from 300 tiles (2 per CU on
// 44 CUs, 1 on the rest) to 1200 tiles of 64 x 64 (5 quarter tiles per CU at most: 1.25 tile times instead of 2).
static int f32m_cus() {
    static int cus[64] = {0};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
    if (!cus[d]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
        cus[d] = n;
    }
    return cus[d];
}
int f32m_pick_tile(int M, int N, int cus) {
    static const int bm[4] = {128, 128, 64, 64}, bn[4] = {128, 64, 128, 64};
    static const double pen[4] = {1.0, 1.04, 1.04, 1.08};
    int best = 0;
    double best_cost = 0;
    for (int t = 0; t < 4; ++t) {
        const int64_t tiles = (int64_t)((M + bm[t] - 1) / bm[t]) * ((N + bn[t] - 1) / bn[t]);
        const double cost = (double)((tiles + cus - 1) / cus) * bm[t] * bn[t] * pen[t];
        if (t == 0 || cost < best_cost) { best = t; best_cost = cost; }
    }
    return best;
}

template <int FBM, int FBN>
static int launch_f32m_t(const GemmArgs& a, int epi, hipStream_t st) {
    const int tiles_m = (a.M + FBM - 1) / FBM, tiles_n = (a.N + FBN - 1) / FBN;
    const dim3 grid(tiles_m * tiles_n);
    switch (epi) {
        case EPI_BIAS: hipLaunchKernelGGL((gemm_f32m_k<EPI_BIAS, FBM, FBN>), grid, dim3(256), 0, st, a, tiles_m, tiles_n); break;
        case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm_f32m_k<EPI_BIAS_GELU, FBM, FBN>), grid, dim3(256), 0, st, a, tiles_m, tiles_n); break;
        case EPI_BIAS_GATE_RES: hipLaunchKernelGGL((gemm_f32m_k<EPI_BIAS_GATE_RES, FBM, FBN>), grid, dim3(256), 0, st, a, tiles_m, tiles_n); break;
        case EPI_BIAS_ADD: hipLaunchKernelGGL((gemm_f32m_k<EPI_BIAS_ADD, FBM, FBN>), grid, dim3(256), 0, st, a, tiles_m, tiles_n); break;
        default: return s2v_fail(__FILE__, __LINE__, "gemm_f32m: bad epilogue", -1);
    }
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_f32m(const GemmArgs& a, int epi, hipStream_t st) {
    S2V_REQUIRE(gemm_f32m_ok(a), "gemm_f32m: shape / alignment not supported (K, lda, ldw multiples of 4 floats, 16-byte aligned operands)");
    // GemmArgs::tile 30 .. 33 (s2v_op_linear impl 30 .. 33: tests and tools/f32m_bench.py) forces a candidate; otherwise the cost model picks
    const int t = (a.tile >= 30 && a.tile <= 33) ? a.tile - 30 : f32m_pick_tile(a.M, a.N, f32m_cus());
    switch (t) {
        case 1: return launch_f32m_t<128, 64>(a, epi, st);
        case 2: return launch_f32m_t<64, 128>(a, epi, st);
        case 3: return launch_f32m_t<64, 64>(a, epi, st);
        default: return launch_f32m_t<128, 128>(a, epi, st);
    }
}
