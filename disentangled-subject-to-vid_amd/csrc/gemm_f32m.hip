// fp32 GEMM on the matrix pipe: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; 64 cycles per instruction and SIMD = the fp32 vector
// rate, 157 TFLOP/s on the part -- MI355X_MICROARCH.md "FP32-input MFMA").  The instruction is bit-for-bit a k-ordered fmaf chain
// (D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)), one rounding per product), so with the K index fed in ascending order this kernel returns
// EXACTLY the bits of gemm_simple_k<float> (gemm.hip: acc = fmaf(a, w, acc) for k = 0 .. K-1, same zero fill of the K tail, same
// epilogue4) -- tests/test_gpu_f32m.py holds it to that.  It is the CPU-reference-parity mode (fp32 model dtype, <= 1e-3 against the
// oracle) made fast enough to serve as the on-GPU reference of a whole run at the headline geometry: the VALU
// kernel runs the C3 step in minutes, this one in seconds.
//
// Tile 128 (m) x 128 (n) x 16 (k), four waves (2 x 2) of 64 x 64 = 2 x 2 MFMA blocks, two workgroups per CU.  Operands are staged
// through registers into TRANSPOSED LDS images [k][row] (pitch 160 floats: the two k-rows a wave reads per instruction -- lanes 0-31
// take k, lanes 32-63 take k + 1 -- fall into the two halves of the banks), so a fragment read is one ds_read_b32 of 64 consecutive
// floats per half.  The MFMA is issued swapped (first operand = weight rows): a lane then owns four consecutive output columns of one
// token row, which is what epilogue4 takes.  Tile order: XCD-aware (each XCD walks a contiguous range of tiles) and grouped by 8 row
// tiles so the 64 workgroups resident on an XCD share A / W tiles through its L2.
#define S2V_HOST
#include "common.h"
#include "kernels.h"
#include "gemm_epi.h"

#define FBM 128
#define FBN 128
#define FBK 16
#define FPITCH 160

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

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32m_k(const GemmArgs a, int tiles_m, int tiles_n) {
    __shared__ float sA[2][FBK][FPITCH];
    __shared__ float sW[2][FBK][FPITCH];
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
    // staging: thread t copies the 4-float chunks c0 and c0 + 2 (of the 4 chunks of a 16-float K-tile) of row t & 127 of either operand
    const int r = tid & 127, c0 = tid >> 7;
    const int64_t baseA = f32m_row_base(a, min(m0 + r, a.M - 1));
    const int64_t baseW = (int64_t)min(n0 + r, a.N - 1) * a.ldw;
    f32x4 ra[2], rw[2];
    auto gload = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = kt * FBK + (c0 + 2 * j) * 4;
            if (k < a.K) {
                ra[j] = *(const f32x4*)(A + baseA + f32m_k_off(a, k));
                rw[j] = *(const f32x4*)(W + baseW + k);
            } else {
                ra[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                rw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kc = (c0 + 2 * j) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sA[buf][kc + e][r] = ra[j][e];
                sW[buf][kc + e][r] = rw[j][e];
            }
        }
    };
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int fr = lane & 31, hi = lane >> 5;
    f32x16 acc[2][2];  // [n block][m block]: D[i = n][j = m]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
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
            const float w0 = sW[buf][kk + hi][wn + fr], w1 = sW[buf][kk + hi][wn + 32 + fr];
            const float a0 = sA[buf][kk + hi][wm + fr], a1 = sA[buf][kk + hi][wm + 32 + fr];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, a0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, a1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, a0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, a1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nkt) lstore(buf ^ 1);
        __syncthreads();
    }
    // D register e of a lane: i = (e & 3) + 8 * (e >> 2) + 4 * hi (output column), j = fr (token row)
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
        const int m = m0 + wm + bj * 32 + fr;
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
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
    const int64_t tiles = (int64_t)((a.M + FBM - 1) / FBM) * ((a.N + FBN - 1) / FBN);
    return tiles < (1 << 30);
}

int launch_gemm_f32m(const GemmArgs& a, int epi, hipStream_t st) {
    S2V_REQUIRE(gemm_f32m_ok(a), "gemm_f32m: shape / alignment not supported (K, lda, ldw multiples of 4 floats, 16-byte aligned operands)");
    const int tiles_m = (a.M + FBM - 1) / FBM, tiles_n = (a.N + FBN - 1) / FBN;
    const dim3 grid(tiles_m * tiles_n);
    switch (epi) {
        case EPI_BIAS: hipLaunchKernelGGL(gemm_f32m_k<EPI_BIAS>, grid, dim3(256), 0, st, a, tiles_m, tiles_n); break;
        case EPI_BIAS_GELU: hipLaunchKernelGGL(gemm_f32m_k<EPI_BIAS_GELU>, grid, dim3(256), 0, st, a, tiles_m, tiles_n); break;
        case EPI_BIAS_GATE_RES: hipLaunchKernelGGL(gemm_f32m_k<EPI_BIAS_GATE_RES>, grid, dim3(256), 0, st, a, tiles_m, tiles_n); break;
        case EPI_BIAS_ADD: hipLaunchKernelGGL(gemm_f32m_k<EPI_BIAS_ADD>, grid, dim3(256), 0, st, a, tiles_m, tiles_n); break;
        default: return s2v_fail(__FILE__, __LINE__, "gemm_f32m: bad epilogue", -1);
    }
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}
