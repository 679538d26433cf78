// GEMM kernels for the CogVideoX denoise path (replaces the nn.Linear call sites
// attention_processor.py:2049-2051,2090; attention.py:1241-1243; activations.py:87-90).
//
//  gemm_bf16_128   : bf16 MFMA (v_mfma_f32_32x32x16_bf16), 128x128x64 block tile, 4 waves (2x2, 64x64 each),
//                    operands staged HBM -> LDS with global_load_lds (16 B/lane), LDS image XOR-swizzled on the
//                    SOURCE address (chunk ^= (row>>1)&7) so every ds_read_b128 lane-group hits 16 distinct
//                    16-B slots; double buffered; XCD-aware + grouped tile order for L2 reuse.
//                    MFMA is issued "swapped" (first operand = weight rows) so each lane owns 4 consecutive
//                    output columns of one token row -> 8-byte stores and lane-local bias/gate epilogue.
//  gemm_bf16_stag  : 256x128x64 tile, three-stage LDS-DMA ring, two wave groups half a k-step apart (N not fitting 256-column tiles)
//  gemm_bf16_w8    : 256x256 tile, K32 half-steps, four-stage ring, lock-step (reference schedule of the race-screen test)
//  gemm_bf16_pp64  : 256x256x64 tile, ping-pong of the two waves of a SIMD, saddr-form LDS-DMA -- the default
//  gemm_simple<T>  : LDS-tiled VALU kernel for fp32 (the CPU-reference-parity mode) and odd shapes.
#define S2V_HOST
#include "common.h"
#include "kernels.h"
#include <type_traits>
#include <algorithm>
#include <cstdlib>

#include "gemm_epi.h"

// ---------------------------------------------------------------------------------------------------
// bf16 MFMA kernel
#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (BM * BK * 2)  // 16 KiB per operand per stage

// element offset of A[m][0] (plain: m*lda; conv: the top-left-front input pixel of output pixel m)
__device__ __forceinline__ int64_t a_row_base(const GemmArgs& a, int m) {
    if (!a.conv) return (int64_t)m * a.lda;
    m = min(m, a.M - 1);
    const int hw = a.oH * a.oW;
    const int f = m / hw, rem = m - f * hw;
    const int y = rem / a.oW, x = rem - y * a.oW;
    const int cs = a.cstride > 1 ? a.cstride : 1;
    return (((int64_t)f * a.Hp + y * cs) * a.Wp + x * cs) * a.cin;
}
// element offset added for column k (plain: k; conv: tap displacement + channel)
__device__ __forceinline__ int64_t a_k_off(const GemmArgs& a, int k) {
    if (!a.conv) return k;
    const int tap = k / a.cin, ci = k - tap * a.cin;
    const int dt = a.kt == 3 ? tap / 9 : 0;
    const int r9 = tap - dt * 9;
    const int dy = r9 / 3, dx = r9 - dy * 3;
    return (((int64_t)dt * a.Hp + dy) * a.Wp + dx) * a.cin + ci;
}

// A operand: per-thread row bases (fixed for the whole K loop) + a per-k-step displacement
__device__ __forceinline__ void stage_tile_a(const bf16_t* __restrict__ g, const int64_t rb[4], int64_t koff, char* lds,
                                             int tid) {
    const int wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gi = i * 256 + tid;
        const int row = gi >> 3;
        const int cp = gi & 7;
        const int c = cp ^ ((row >> 1) & 7);
        const bf16_t* src = g + rb[i] + koff + c * 8;
        char* dst = lds + (i * 256 + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, int ld, int row0, int k0, char* lds, int tid) {
    // 128 rows x 128 B; thread t of round i owns LDS chunk (i*256+t): row = g>>3, physical 16-B chunk = g&7.
    const int wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gi = i * 256 + tid;
        const int row = gi >> 3;
        const int cp = gi & 7;
        const int c = cp ^ ((row >> 1) & 7);
        const bf16_t* src = g + (size_t)(row0 + row) * ld + k0 + c * 8;
        char* dst = lds + (i * 256 + wave * 64) * 16;  // wave-uniform base; HW adds lane*16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 lds_frag(const char* tile, int row, int cl) {
    const int off = row * 128 + ((cl ^ ((row >> 1) & 7)) << 4);
    return *(const bf16x8*)(tile + off);
}

// one 32 x 32 x 16 MFMA on 16-bit operands of either encoding (the fragments are bits: bf16x8 is just their carrier type)
template <typename T16>
__device__ __forceinline__ f32x16 mfma16(bf16x8 x, bf16x8 y, f32x16 acc) {
    if constexpr (std::is_same<T16, f16_t>::value) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), acc, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
}
// T16 = bf16_t (v_mfma_f32_32x32x16_bf16) or f16_t (v_mfma_f32_32x32x16_f16: the fp16 model dtype, round 5 -- the staging, the swizzle and the
// fragment reads move 16-bit elements whatever they encode; the fp16 form takes the lane-local epilogue4 below)
template <int EPI, typename T16 = bf16_t>
__global__ __launch_bounds__(256, 2) void gemm_bf16_128(const GemmArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap (block b runs on XCD b%8), then grouped (8 tile rows) order
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int GM = 8;
    const int per_group = GM * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_g = wg - group * per_group;
    const int tile_m = first_m + in_g % gsz;
    const int tile_n = in_g / gsz;
    const int m0 = a.m_begin + tile_m * BM, n0 = tile_n * BN;

    const bf16_t* A = (const bf16_t*)a.A;
    const bf16_t* W = (const bf16_t*)a.W;
    // stage s: A tile at s*2*TILE_BYTES, W tile right behind it

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nt = a.K / BK;
    int64_t rb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = a_row_base(a, m0 + ((i * 256 + tid) >> 3));
    stage_tile_a(A, rb, a_k_off(a, 0), smem, tid);
    stage_tile(W, a.ldw, n0, 0, smem + TILE_BYTES, tid);
    __syncthreads();

    const int fr = lane & 31, hi = lane >> 5;
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            stage_tile_a(A, rb, a_k_off(a, (t + 1) * BK), smem + (cur ^ 1) * 2 * TILE_BYTES, tid);
            stage_tile(W, a.ldw, n0, (t + 1) * BK, smem + (cur ^ 1) * 2 * TILE_BYTES + TILE_BYTES, tid);
        }
        const char* tA = smem + cur * 2 * TILE_BYTES;
        const char* tW = tA + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 wf[2], af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) wf[i] = lds_frag(tW, wn * 64 + i * 32 + fr, kk * 2 + hi);
#pragma unroll
            for (int j = 0; j < 2; ++j) af[j] = lds_frag(tA, wm * 64 + j * 32 + fr, kk * 2 + hi);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (std::is_same<T16, f16_t>::value)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[i]), __builtin_bit_cast(f16x8, af[j]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    if (epi_vec_ok(a, EPI)) {  // the k-loop ended with a barrier: the operand stages are free
        epilogue_wave<EPI, 2, false, T16>(a, acc, m0 + wm * 64, n0 + wn * 64, smem + wave * 8192, lane);
        return;
    }
    // epilogue: D[i = n][j = m]; lane: m = fr, n = (reg&3) + 8*(reg>>2) + 4*hi
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + wm * 64 + j * 32 + fr;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int n = n0 + wn * 64 + i * 32 + 8 * rq + 4 * hi;
                float v[4] = {acc[i][j][rq * 4 + 0], acc[i][j][rq * 4 + 1], acc[i][j][rq * 4 + 2], acc[i][j][rq * 4 + 3]};
                if (n < a.N) epilogue4<T16, EPI>(a, m, n, v);
            }
        }
}

// the fp16 model dtype's linears: K % 64 == 0, 16-byte aligned rows, operands with the rows of whole 128-row tiles present behind them
bool gemm_f16_ok(const GemmArgs& a, int epi) {
    if (epi == EPI_BIAS_QKNORM || a.splitk > 1 || a.m_begin != 0) return false;
    if (a.K % BK != 0 || a.ldw % 8 != 0 || (a.conv ? a.cin % 64 != 0 : a.lda % 8 != 0)) return false;
    const int64_t mp = (int64_t)((a.M + BM - 1) / BM) * BM, np = (int64_t)((a.N + BN - 1) / BN) * BN;
    return (a.conv || a.a_rows_padded >= mp) && (a.w_rows_padded >= np || a.N % BN == 0);
}
int launch_gemm_f16(const GemmArgs& a0, int epi, hipStream_t st) {
    S2V_REQUIRE(gemm_f16_ok(a0, epi), "gemm_f16: shape / padding not supported");
    GemmArgs a = a0;
    a.f16 = 1;  // the dispatcher of the bf16 kernels on their fp16 instantiations
    return launch_gemm_bf16(a, epi, st);
}

// ---------------------------------------------------------------------------------------------------
// 256(M) x 128(N) x 64 block tile, 8 waves (4 x 2, 64x64 each), THREE LDS stages (3 x 48 KiB) filled by global_load_lds two
// k-steps ahead, counted s_waitcnt vmcnt(6): the tile of gemm_bf16_stag below (its lock-step predecessor was removed).
#define RBM 256
#define RBN 128
#define RA_BYTES (RBM * BK * 2)   // 32 KiB
#define RW_BYTES (RBN * BK * 2)   // 16 KiB
#define RSTAGE (RA_BYTES + RW_BYTES)

// ---------------------------------------------------------------------------------------------------
// gemm_bf16_stag: same 256x128x64 tile / 3-stage LDS-DMA ring, but the two wave groups (waves 0-3 / 4-7, one wave of
// each per SIMD) run half a k-step apart: a k-step is split into a LOAD slot (16 ds_read_b128 -> fragments in VGPRs)
// and a COMPUTE slot (16 MFMA), slots are separated by s_barrier, and group B starts one barrier late.  At any time
// one wave of every SIMD is in its MFMA slot (s_setprio 1) while its partner issues LDS reads / LDS-DMA, so the
// matrix pipe and the LDS/VMEM pipes overlap instead of alternating.
//   odd slot of step t : everybody issues the DMA of tile t+2 (stage (t-1)%3: its last reads were consumed >= 1
//                        barrier ago), A computes tile t, B loads tile t; both end with the counted vmcnt that
//                        retires tile t+1, then barrier
//   even slot          : A loads tile t+1, B computes tile t
template <int EPI, typename T16 = bf16_t>
__global__ __launch_bounds__(512, 2) void gemm_bf16_stag(const GemmArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave & 3, wn = grp;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int GM = 4;
    const int per_group = GM * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_g = wg - group * per_group;
    const int m0 = (first_m + in_g % gsz) * RBM, n0 = (in_g / gsz) * RBN;

    // saddr-form LDS-DMA (see glds16_saddr): wave-uniform tile base + loop-invariant 32-bit lane offsets
    const int64_t rbase0 = a_row_base(a, m0);
    const char* Abase = (const char*)a.A + 2 * rbase0;
    const char* Wbase = (const char*)a.W + 2 * (int64_t)n0 * a.ldw;
    unsigned offA[4], offW[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gi = i * 512 + tid, row = gi >> 3;
        offA[i] = (unsigned)(2 * (a_row_base(a, m0 + row) - rbase0 + ((gi & 7) ^ ((row >> 1) & 7)) * 8));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int gi = i * 512 + tid, row = gi >> 3;
        offW[i] = (unsigned)(2 * ((int64_t)row * a.ldw + ((gi & 7) ^ ((row >> 1) & 7)) * 8));
    }
    auto stage = [&](int t, int s) {
        char* base = smem + s * RSTAGE;
        const char* ta = Abase + 2 * a_k_off(a, t * BK);
        const char* tw = Wbase + 2 * (int64_t)t * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_saddr(ta, offA[i], base + (i * 512 + wave * 64) * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_saddr(tw, offW[i], base + RA_BYTES + (i * 512 + wave * 64) * 16);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 wf[4][2], af[4][2];
    const int fr = lane & 31, hi = lane >> 5;

    auto load_frags = [&](int t) {
        const char* tA = smem + (t % 3) * RSTAGE;
        const char* tW = tA + RA_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) wf[kk][i] = lds_frag(tW, wn * 64 + i * 32 + fr, kk * 2 + hi);
#pragma unroll
            for (int j = 0; j < 2; ++j) af[kk][j] = lds_frag(tA, wm * 64 + j * 32 + fr, kk * 2 + hi);
        }
    };
    auto compute = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<T16>(wf[kk][i], af[kk][j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };

    const int nt = a.K / BK;
    stage(0, 0);
    if (nt > 1) {
        stage(1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    if (grp == 0) {
        for (int t = 0; t < nt; ++t) {
            load_frags(t);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if (t + 2 < nt) stage(t + 2, (t + 2) % 3);
            compute();
            if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else {
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nt; ++t) {
            if (t + 2 < nt) stage(t + 2, (t + 2) % 3);
            load_frags(t);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // last step: group A starts overwriting the stages (epilogue patches) right behind this barrier
            if (t + 1 == nt) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            compute();
            if (t + 1 < nt) __builtin_amdgcn_s_barrier();
        }
    }

    if (epi_vec_ok(a, EPI)) {
        // group A left the loop through a barrier that B passed after its last LDS reads: the stages are free
        epilogue_wave<EPI, 2, false, T16>(a, acc, m0 + wm * 64, n0 + wn * 64, smem + wave * 8192, lane);
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + wm * 64 + j * 32 + fr;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int n = n0 + wn * 64 + i * 32 + 8 * rq + 4 * hi;
                float v[4] = {acc[i][j][rq * 4 + 0], acc[i][j][rq * 4 + 1], acc[i][j][rq * 4 + 2], acc[i][j][rq * 4 + 3]};
                if (n < a.N) epilogue4<T16, EPI>(a, m, n, v);
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// 256 x 256 block tiles (gemm_bf16_w8, gemm_bf16_pp64).  A four-wave 128 x 128-wave-tile form, a ping-pong on K32 stages and a
// sixteen-wave ping-pong were measured (HISTORY.md section 3) and removed.
#define WBM 256
#define WBN 256
#define K32 32
#define WH_A (WBM * K32 * 2)   // 16 KiB
#define WH_STAGE (2 * WH_A)    // 32 KiB
#define WH_NST 4
// fragment read of a K32 stage (rows of 64 B: 4 chunks of 16 B, chunk ^= (row >> 2) & 3 on the DMA source and here)
__device__ __forceinline__ bf16x8 lds_frag32(const char* tile, int row, int cl) {
    return *(const bf16x8*)(tile + row * 64 + ((cl ^ ((row >> 2) & 3)) << 4));
}

// ---------------------------------------------------------------------------------------------------
#ifdef S2V_DIAG  // the lock-step reference schedule of the race-screen test lives in libs2v_hip_diag.so only
// gemm_bf16_w8: the same 256 x 256 block tile, BK32 half-steps, four-stage LDS-DMA ring and fragment register
// prefetch of the removed four-wave form, with EIGHT waves (2 x 4) of 128(m) x 64(n) wave tiles = two waves per SIMD.  A wave that is
// stuck issuing an LDS-DMA piece (60-180 cycles each, per MI355X_MICROARCH.md) no longer idles the matrix pipe: its
// SIMD partner issues MFMAs meanwhile.  Per half-step and wave: 16 MFMA 32x32x16, 12 ds_read_b128 (next half-tile),
// 4 LDS-DMA pieces (half-tile h+4); counted vmcnt(8) keeps two half-tiles in flight across the single barrier.
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_w8(const GemmArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;  // 2 (m) x 4 (n); waves w and w+4 (same SIMD) differ in wm

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int GM = 4;
    const int per_group = GM * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_g = wg - group * per_group;
    const int m0 = (first_m + in_g % gsz) * WBM, n0 = (in_g / gsz) * WBN;

    // staging: 1024 chunks per operand per half-step, 2 per thread: gi = i*512 + tid -> row = i*128 + (tid>>2)
    const int srow = tid >> 2;
    const int scol = ((tid & 3) ^ ((srow >> 2) & 3)) * 8;
    // saddr-form LDS-DMA (see glds16_saddr): wave-uniform half-tile base + loop-invariant 32-bit lane offsets
    const int64_t rbase0 = a_row_base(a, m0);
    const char* Abase = (const char*)a.A + 2 * rbase0;
    const char* Wbase = (const char*)a.W + 2 * (int64_t)n0 * a.ldw;
    unsigned offA[2], offW[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        offA[i] = (unsigned)(2 * (a_row_base(a, m0 + srow + i * 128) - rbase0 + scol));
        offW[i] = (unsigned)(2 * ((int64_t)(srow + i * 128) * a.ldw + scol));
    }
    int64_t ka_cur = 0;
    auto glds_one = [&](int t, int s, int idx4) {  // 0,1 -> A pieces, 2,3 -> W pieces
        char* base = smem + s * WH_STAGE;
        const int i = idx4 & 1;
        if (idx4 < 2) glds16_saddr(Abase + 2 * ka_cur, offA[i], base + (i * 512 + wave * 64) * 16);
        else glds16_saddr(Wbase + 2 * (int64_t)t * K32, offW[i], base + WH_A + (i * 512 + wave * 64) * 16);
    };

    f32x16 acc[2][4];  // [n block][m block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nh = a.K / K32;
    const int fr = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int tp = min(p, nh - 1);
        ka_cur = a.conv ? a_k_off(a, tp * K32) : (int64_t)tp * K32;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) glds_one(tp, p, g4);
    }
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    bf16x8 wfA[2][2], afA[2][4], wfB[2][2], afB[2][4];
    auto read_w = [&](int t, bf16x8 (&wf)[2][2], int kk, int i) {
        wf[kk][i] = lds_frag32(smem + (t & 3) * WH_STAGE + WH_A, wn * 64 + i * 32 + fr, kk * 2 + hi);
    };
    auto read_a = [&](int t, bf16x8 (&af)[2][4], int kk, int j) {
        af[kk][j] = lds_frag32(smem + (t & 3) * WH_STAGE, wm * 128 + j * 32 + fr, kk * 2 + hi);
    };
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) read_w(0, wfA, kk, i);
#pragma unroll
        for (int j = 0; j < 4; ++j) read_a(0, afA, kk, j);
    }

    auto half_step = [&](int h, bf16x8 (&wf)[2][2], bf16x8 (&af)[2][4], bf16x8 (&wfn)[2][2], bf16x8 (&afn)[2][4]) {
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int tl = min(h + 4, nh - 1);
        ka_cur = a.conv ? a_k_off(a, tl * K32) : (int64_t)tl * K32;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                // 3 fragment reads + 1 DMA piece per 4 MFMA
                read_w(h + 1, wfn, kk, i);
                read_a(h + 1, afn, kk, i * 2);
                read_a(h + 1, afn, kk, i * 2 + 1);
                glds_one(tl, h & 3, kk * 2 + i);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk][i], af[kk][j], acc[i][j], 0, 0, 0);
            }
    };
    for (int h = 0; h < nh; h += 2) {  // nh even (K % 64 == 0)
        half_step(h, wfA, afA, wfB, afB);
        half_step(h + 1, wfB, afB, wfA, afA);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    char* patch = smem + wave * 8192;
    if (epi_vec_ok(a, EPI)) {
#pragma unroll
        for (int qj = 0; qj < 2; ++qj) {
            f32x16 sub[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) sub[i][j] = acc[i][qj * 2 + j];
            epilogue_wave64<EPI>(a, sub, m0 + wm * 128 + qj * 64, n0 + wn * 64, patch, lane);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 128 + j * 32 + fr;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int n = n0 + wn * 64 + i * 32 + 8 * rq + 4 * hi;
                float v[4] = {acc[i][j][rq * 4 + 0], acc[i][j][rq * 4 + 1], acc[i][j][rq * 4 + 2], acc[i][j][rq * 4 + 3]};
                if (n < a.N) epilogue4<bf16_t, EPI>(a, m, n, v);
            }
        }
}

#endif  // S2V_DIAG

// ---------------------------------------------------------------------------------------------------
#ifdef S2V_DIAG
__device__ long long g_pp_dbg[64];  // diagnostics (ABL == 4)
__device__ long long g_blk_times[2048];  // [launch parity][workgroup]: s_memrealtime at entry / exit, cycles, output tiles (gemm_q4 ACCT)
__device__ int g_launch_no;
extern "C" __attribute__((visibility("default"))) int s2v_debug_read_blocks(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blk_times), sizeof(long long) * 2048) == hipSuccess ? 0 : -1; }
extern "C" __attribute__((visibility("default"))) int s2v_debug_read(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp_dbg), sizeof(long long) * 64) == hipSuccess ? 0 : -1; }
#endif

// gemm_bf16_pp64: the ping-pong schedule of gemm_bf16_pp on K-tiles of 64 with 128-byte LDS rows, so every LDS-DMA lane
// group fetches a FULL 128-B line (gemm_bf16_pp / _w8 fetch 64-B half lines: twice the L2 requests for the same bytes, and
// the measured DMA-only time of those kernels equals their MFMA-only time).  Two 64-KiB stages; a K-tile is four
// 16-KiB operand halves  A-lo | A-hi | W-lo | W-hi  (128 rows x 128 B, chunk ^= (row >> 1) & 7).  Group 0 (waves 0-3, A rows
// 0-127) stages A-lo and W-lo, group 1 (waves 4-7) stages A-hi and W-hi: 4 pieces per thread per half-step, like pp.
// Barrier intervals: group 0 loads half-step h = 2T + s of K-tile T in I_2h and computes in I_2h+1, group 1 one later.
//   group 0, tile T:  I_4T   reads(T,0) + DMA A-lo(T+1)      I_4T+1 MFMA
//                     I_4T+2 reads(T,1) + DMA W-lo(T+1)      I_4T+3 MFMA, vmcnt(0)
//   group 1, tile T:  I_4T+1 reads(T,0) + DMA W-hi(T+1)      I_4T+2 MFMA
//                     I_4T+3 reads(T,1) + DMA A-hi(T+1), vmcnt(4)   I_4T+4 MFMA, vmcnt(0)
//   WAR  stage (T+1)&1 held tile T-1: A-lo(T-1) is last read in I_4T-2 (group 0 only), W-*(T-1) and A-hi(T-1) in I_4T-1;
//        each read's lgkmcnt(0) sits after the next barrier, and one more barrier precedes the DMA issue above.
//   RAW  tile T+1 is first read in I_4T+4 (A-lo, W-lo, W-hi) and I_4T+5 (A-hi): the waits above sit in I_4T+3 / I_4T+4.
// FP8: the same schedule byte for byte on e4m3 operands -- a K-tile is 128 elements = the same 128-byte rows, a half-step (64 bytes
// of K) is ONE v_mfma_scale_f32_32x32x64_f8f6f4 per 32x32 block (64 cycles, twice the bf16 rate).  The instruction's LOGICAL K order
// (tools/probes/mfma_mx.hip): block b (32 elements, one E8M0 scale) = bytes 16 b .. 16 b + 15 of the operand registers of BOTH lane
// halves, its scale is taken from lanes 32 b + row.  With the bf16 chunk order -- a lane's two 16-byte fragments are chunks
// s*4 + kk*2 + hi, kk = 0, 1 -- the logical blocks of half-step s are therefore the memory-contiguous 32-element blocks 2 s and
// 2 s + 1 of the K-tile, and lane (row, hi) supplies the scale of block 2 s + hi.  Per-token / per-channel scales are applied by the
// epilogue; the block scales are unit unless A is an MX image (GemmArgs::mx_a_s).
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x8v cat16(bf16x8 lo, bf16x8 hi) {
    const i32x4v a = __builtin_bit_cast(i32x4v, lo), b = __builtin_bit_cast(i32x4v, hi);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <int EPI, int ABL = 0, bool FP8 = false, typename T16 = bf16_t>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pp64(const GemmArgs a, int tiles_m, int tiles_n) {
    constexpr int ES = FP8 ? 1 : 2;         // bytes per operand element
    constexpr int BKE = 128 / ES;           // elements per 128-byte K-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long tk0 = (ABL >= 4) ? (long long)__builtin_amdgcn_s_memtime() : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int g = wm, w4 = wave & 3;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    // rows of the tile order that share a column sweep: 4 for wide outputs; narrow outputs (N = 3072: 12 column tiles) measured
    // 2-3 % faster with 3 (K = 3072) and 1 (K = 12288) -- an XCD's 32 concurrent tiles then span the whole width
    const int GM = a.gm > 0 ? a.gm : 4;
    const int per_group = GM * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_g = wg - group * per_group;
    const int m0 = (first_m + in_g % gsz) * WBM, n0 = (in_g / gsz) * WBN;

    // staging: piece i (0..3) of this wave covers rows g*128 + i*32 + w4*8 + (lane>>3), 8 chunks of 16 B each
    const int srow = g * 128 + w4 * 8 + (lane >> 3);
    const int scol = ((lane & 7) ^ ((srow >> 1) & 7)) * (16 / ES);
    // addresses = wave-uniform tile base (SGPR pair) + per-lane 32-bit byte offset (loop-invariant VGPR): the LDS-DMA takes
    // the saddr + voffset form and the load segment carries no address VALU
    const int64_t rbase0 = a_row_base(a, m0);
    const char* Abase = (const char*)a.A + ES * rbase0;
    const char* Wbase = (const char*)a.W + ES * (int64_t)n0 * a.ldw;
    unsigned offA[4], offW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        offA[i] = (unsigned)(ES * (a_row_base(a, m0 + srow + i * 32) - rbase0 + scol));
        offW[i] = (unsigned)(ES * ((int64_t)(srow + i * 32) * a.ldw + scol));
    }
    const int nT = a.K / BKE;
    const bool mx = FP8 && a.mx_a_s != nullptr;                 // block-scaled A (GemmArgs::mx_a_s)
    const unsigned offS = (unsigned)((w4 * 64 + lane) * 4);     // scale dword of lane: rows w4 * 64 + lane of the tile (K-tile major: contiguous)
    const int ldst = (g * 128 + w4 * 8) * 128;  // byte offset of the wave's piece 0 inside an operand image
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_base_u32(smem));  // LDS destinations as integers: no null-check SALU per piece
    auto dma_a = [&](int t) {
        if (ABL == 1 && t > 0) return;
        const int tc = min(t, nT - 1);
        const int64_t ka = a.conv ? a_k_off(a, tc * BKE) : (int64_t)tc * BKE;
        const char* tb = Abase + ES * ka;
        const unsigned base = lds0 + (t & 1) * 65536 + ldst;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            glds16_saddr_m0(tb, offA[i], base + i * 4096);
        if (mx && !g) {  // the K-tile's four block scales of the tile's 256 rows: one dword per row, 64 rows per wave of group 0
            const char* sb = (const char*)a.mx_a_s + ((size_t)tc * a.mx_rows + m0) * 4;
            glds4_saddr_m0(sb, offS, lds0 + 131072 + (t & 1) * 1024 + w4 * 256);
        }
    };
    auto dma_w = [&](int t) {
        if (ABL == 1 && t > 0) return;
        const int tc = min(t, nT - 1);
        const char* tb = Wbase + ES * (int64_t)tc * BKE;
        const unsigned base = lds0 + (t & 1) * 65536 + 32768 + ldst;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            glds16_saddr_m0(tb, offW[i], base + i * 4096);
    };

    f32x16 acc[2][4];  // [n block][m block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int fr = lane & 31, hi = lane >> 5;
    dma_a(0);
    dma_w(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // opens I_0
    if (g) __builtin_amdgcn_s_barrier();  // group 1 idles through I_0

    bf16x8 wf[2][2], af[2][4];
    int sb_[4] = {0x007f007f, 0x007f007f, 0x007f007f, 0x007f007f};  // E8M0 scales of the lane's A blocks: byte 0 for half-step 0, byte 2 for half-step 1 (unit unless MX)
    auto reads = [&](int t, int s) {
        if (ABL == 6 && t > 0) return;
        const char* tA = smem + (t & 1) * 65536;
        const char* tW = tA + 32768;
        if (FP8 && mx && s == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                sb_[j] = (int)(*(const unsigned*)(smem + 131072 + (t & 1) * 1024 + (wm * 128 + fr * 4 + j) * 4) >> (8 * hi));  // rows permuted inside a 128-row half (kernels.h)
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) wf[kk][i] = lds_frag(tW, wn * 64 + i * 32 + fr, s * 4 + kk * 2 + hi);
#pragma unroll
            for (int j = 0; j < 4; ++j) af[kk][j] = lds_frag(tA, wm * 128 + j * 32 + fr, s * 4 + kk * 2 + hi);
        }
    };
    // (ABL >= 4): per-wave stall accounting with s_memtime (diagnostics; totals of block 100 go to g_pp_dbg[8][6])
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto now = [&]() -> long long { return (ABL >= 4) ? (long long)__builtin_amdgcn_s_memtime() : 0; };
    auto cluster = [&](bool tile_end) {
        __builtin_amdgcn_sched_barrier(0);
        const long long t0 = now();
        __builtin_amdgcn_s_barrier();
        const long long t1 = now();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long t2 = now();
        __builtin_amdgcn_sched_barrier(0);
        if (ABL != 7) __builtin_amdgcn_s_setprio(1);
        if (FP8) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = tile_end ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat16(wf[0][i], wf[1][i]), cat16(af[0][j], af[1][j]), acc[i][j], 0, 0, 0, 127, 2, sb_[j])
                                         : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat16(wf[0][i], wf[1][i]), cat16(af[0][j], af[1][j]), acc[i][j], 0, 0, 0, 127, 0, sb_[j]);
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (ABL != 3 && ABL != 5) acc[i][j] = mfma16<T16>(wf[kk][i], af[kk][j], acc[i][j]);
                        else asm volatile("" ::"v"(wf[kk][i]), "v"(af[kk][j]));
                    }
        }
        if (ABL != 7) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        const long long t3 = now();
        if (tile_end) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t4 = now();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        const long long t5 = now();
        if ((ABL >= 4)) {
            tacc[0] += t1 - t0;  // barrier after the load segment
            tacc[1] += t2 - t1;  // lgkmcnt(0)
            tacc[2] += t3 - t2;  // MFMA cluster issue
            tacc[3] += t4 - t3;  // vmcnt(0) at the tile end
            tacc[4] += t5 - t4;  // barrier after the compute segment
        }
    };
    const long long tl0 = now();
    for (int t = 0; t < nT; ++t) {
        const long long u0 = now();
        reads(t, 0);
        __builtin_amdgcn_sched_barrier(0);
        const long long u1 = now();
        if (g) dma_w(t + 1);
        else dma_a(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        const long long u2 = now();
        cluster(false);
        const long long u3 = now();
        reads(t, 1);
        __builtin_amdgcn_sched_barrier(0);
        const long long u4 = now();
        if (g) {
            dma_a(t + 1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            dma_w(t + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        const long long u5 = now();
        cluster(true);
        if ((ABL >= 4)) {
            tacc[6] += (u1 - u0) + (u4 - u3);  // ds_read issue
            tacc[7] += (u2 - u1) + (u5 - u4);  // LDS-DMA issue (+ group 1's vmcnt(4))
        }
    }
    if ((ABL >= 4)) {
        tacc[5] = now() - tl0;
        tacc[1] = tl0 - tk0;  // prologue (replaces the lgkmcnt slot)
    }
    const long long te0 = now();
    if (!g) __builtin_amdgcn_s_barrier();  // pairs with group 1's last compute segment
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    char* patch = smem + wave * 16384;
    if (epi_vec_ok(a, EPI)) {
        epilogue_wave<EPI, 4, FP8, T16>(a, acc, m0 + wm * 128, n0 + wn * 64, patch, lane);
        if ((ABL >= 4)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tacc[3] = now() - te0;  // epilogue incl. store drain (replaces the vmcnt slot)
#ifdef S2V_DIAG
            if (blockIdx.x == 100 && lane == 0)
                for (int e = 0; e < 8; ++e) g_pp_dbg[wave * 8 + e] = tacc[e];
#endif
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 128 + j * 32 + fr;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int n = n0 + wn * 64 + i * 32 + 8 * rq + 4 * hi;
                float v[4] = {acc[i][j][rq * 4 + 0], acc[i][j][rq * 4 + 1], acc[i][j][rq * 4 + 2], acc[i][j][rq * 4 + 3]};
                if (n < a.N) epilogue4<T16, EPI>(a, m, n, v);
            }
        }
}

// ---------------------------------------------------------------------------------------------------
#ifdef S2V_DIAG  // an experiment kept for the record (HISTORY.md section 3, "four-wave persistent form"): it equals gemm_bf16_pp64, it does
                 // not beat it, and it serves two epilogues only -- libs2v_hip_diag.so / tools/stall_q4.py / tools/check_q4.py
// gemm_q4: PERSISTENT 256 x 256 block tile, FOUR waves (2 x 2) of 128 x 128 wave tiles = ONE wave per SIMD, K-tiles of 128 bytes
// in two 64-KiB LDS stages.  A 128 x 128 wave tile reads 8 fragments per 16 MFMA where the 128 x 64 tiles of the eight-wave kernels
// read 12: gemm_bf16_pp64 moves 256 KiB through the LDS per K-tile of 2048 MFMA cycles (125 of the 128 B/clk the LDS has), this
// kernel 192 KiB.  One instruction stream per wave, written in issue order with a fence per MFMA slot:
// { MFMA, one fragment read of the NEXT K16 step (first 8 slots), one LDS-DMA piece (every second slot of steps 0 and 3), at most
// two or three instructions of the PREVIOUS output tile's epilogue }.
//
// One workgroup per CU walks its XCD's share of the output tiles; the K-tiles of successive output tiles form ONE stream through the
// two stages, so the LDS-DMA of the next output tile's first K-tiles is issued in the last steps of this one (a workgroup per tile
// paid ~2800 cycles of exposed prologue, ~2 % of a K = 3072 tile).
//
// Epilogue.  With one wave per SIMD nothing hides an epilogue: written after the K loop it cost 12 700 of a tile's 125 000 cycles
// (7 000 of them the 128 KiB of stores, which leave a CU at ~18 B/clk), and GELU would add ~2 500 VALU instructions per wave at
// ~5 cycles each.  So the tile is only CONVERTED after its K loop -- accumulators + bias -> 128 registers of packed bf16 (~650 VALU
// instructions, the part that stays exposed) -- and everything else TRICKLES through the MFMA slots of the next tile's K loop, one
// 32-row x 64-column unit (16 of the 128 registers) per five K-tiles: GELU two VALU instructions per slot, the LDS transposition
// (8 ds_write_b64 + 4 ds_read_b128 through the wave's private 4-KiB patch), the gate / residual loads and arithmetic, 4 full-line
// stores, then the 128 registers rotate down by one unit so that ONE set of five K-tile bodies serves all eight units.  Loads and
// stores of the trickle are issued right after a K-tile's barrier, so the vmcnt(0) of the next barrier finds them complete.  The
// last tile's epilogue runs the same micro-op sequence once without MFMAs.
//
// The LDS-DMA of a K-tile is issued as EARLY as its stage allows -- three to four steps (>= 1500 cycles) before its first use.
// ONE barrier per K-tile, at the start of its last step (s = 3):
//   RAW  K-tile g+1 (A half issued in step 3 of K-tile g-1, W half in step 0 of K-tile g): vmcnt(0) + barrier before step 3 of
//        K-tile g, whose fragment prefetch is the first read of K-tile g+1;
//   WAR  K-tile g+2 overwrites the stage of K-tile g from step 3 of K-tile g on, i.e. after that same barrier; the last fragment
//        reads of K-tile g are issued in its step 2 and waited for (lgkmcnt(0)) before the barrier.
// Plain (non-conv) operands, K >= 43 K-tiles, N % 64 == 0, vector epilogue (epi_vec_ok) only; launch_gemm_bf16 checks.
template <int EPI, int ACCT = 0>
__global__ __launch_bounds__(256, 1) void gemm_q4(const GemmArgs a, int tiles_m, int tiles_n) {
    constexpr int ES = 2, BKE = 64, NSTEP = 4, NRD = 8, GM = 4;
    static_assert(EPI == EPI_BIAS || EPI == EPI_BIAS_GELU, "gemm_q4: bias / bias + GELU only (the register file has no room for gate and residual rows)");
    constexpr bool GELU = EPI == EPI_BIAS_GELU, GATE = false, ADDR = false, HAS_LD = false;
    constexpr int PH_ST = EPI == EPI_BIAS ? 0 : 3, NPH = GELU ? 4 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    char* smem = smem_all + 32768;  // [0, 32 KiB): the four epilogue patches; the stages above them, so that M0 minus an instruction
                                    // offset (glds16_saddr_m0_imm) never drops below the workgroup's LDS base
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, hi = lane >> 5;

    // this workgroup's output tiles: XCD x owns a contiguous range of the (GM-grouped) tile order, its workgroups take every
    // nslots-th tile of it, so at any time the CUs of an XCD work on neighbouring tiles (shared A / W panels in that XCD's L2)
    const int T = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int q = T >> 3, r = T & 7;
    const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int cnt = q + (xcd < r ? 1 : 0);
    if (slot >= cnt) return;
    auto tile_origin = [&](int local, int& m0, int& n0) {
        const int wg = first + min(local, cnt - 1);  // past the end: the last tile again (its loads land in dead stages)
        const int per_group = GM * tiles_n;
        const int group = wg / per_group;
        const int first_m = group * GM;
        const int gsz = min(tiles_m - first_m, GM);
        const int in_g = wg - group * per_group;
        m0 = (first_m + in_g % gsz) * WBM;
        n0 = (in_g / gsz) * WBN;
    };

    // staging: piece p (0..7) of an operand = rows p*32 + wave*8 + (lane >> 3); the chunk XOR depends on row bits 1-3 only, so ONE
    // per-lane offset serves every piece; the piece displacement goes on the SGPR base (advanced once per K-tile), the K-tile
    // displacement in the instruction's immediate, the LDS destination in M0 = stage base + immediate: five instructions per piece
    // (s_add_u32, s_addc_u32, s_add m0, s_nop, global_load_lds) and no VGPR beyond the two offsets -- the register file is full.
    const int srow = wave * 8 + (lane >> 3);
    const int scol = ((lane & 7) ^ ((srow >> 1) & 7)) * (16 / ES);
    const unsigned offA = (unsigned)(ES * ((int64_t)srow * a.lda + scol));
    const unsigned offW = (unsigned)(ES * ((int64_t)srow * a.ldw + scol));
    const int64_t pstrideA = (int64_t)ES * 32 * a.lda, pstrideW = (int64_t)ES * 32 * a.ldw;  // piece p: + p * pstride on the SGPR base
    const int nT = a.K / BKE;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_base_u32(smem)) + wave * 1024;
    int m0, n0, m1, n1;
    tile_origin(slot, m0, n0);
    tile_origin(slot + nslots, m1, n1);
    const char* AK = (const char*)a.A + ES * (int64_t)m0 * a.lda;  // K-tile t of the current output tile
    const char* WK = (const char*)a.W + ES * (int64_t)n0 * a.ldw;
    const char* Anxt = (const char*)a.A + ES * (int64_t)m1 * a.lda;  // K-tile 0 of the next output tile
    const char* Wnxt = (const char*)a.W + ES * (int64_t)n1 * a.ldw;
    unsigned so = 0;  // byte offset of the stage that holds K-tile t (0 / 65536); the stream of K-tiles never resets it

    f32x16 acc[4][4];           // [n block][m block]
    bf16x8 wf[2][4], af[2][4];  // fragment registers, double-buffered by step parity
    auto rd = [&](bool other, int s, int buf, int n) {  // the n-th fragment read of step s of K-tile t (other: of K-tile t+1)
        const char* tA = smem + (other ? so ^ 65536u : so);
        const char* tW = tA + 32768;
        if (n >> 2) af[buf][n & 3] = lds_frag(tA, wm * 128 + (n & 3) * 32 + fr, s * 2 + hi);
        else wf[buf][n & 3] = lds_frag(tW, wn * 128 + (n & 3) * 32 + fr, s * 2 + hi);
    };
    auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
    auto now2 = [&]() -> long long { return ACCT ? (long long)__builtin_amdgcn_s_memtime() : 0; };
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long tk0 = now2(), tr0 = ACCT ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
#ifdef S2V_DIAG
    const int launch_no = ACCT ? g_launch_no : 0;
#endif

    // ---- the trickled epilogue of the PREVIOUS output tile ------------------------------------------------------------------
    // Its 128 x 128 wave tile as packed bf16 pairs, unit-major: unit u = h * 4 + j (h: 64-column half, j: 32-row block); within a unit
    // register (i2 * 4 + rq) * 2 + e holds columns i2*32 + 8 rq + 4 hi + 2 e + {0, 1} of row fr -- the accumulator layout.  Units 0
    // and 1 go straight into the wave's two 4-KiB LDS patches at conversion time, units 2..7 wait in pk[16 (u - 2) ..]: 96 registers
    // are what is left beside 256 accumulators and 64 fragment registers (with 128 the compiler spilled 81).
    unsigned pk[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) pk[i] = 0u;
    u32x4 rb[4] = {}, gl[4] = {}, xl[4] = {};  // the unit's 32 rows read back row-major (16 B per lane), their gates, their residuals
    float ga = 0.f, gb = 0.f, ta = 0.f, tb = 0.f;
    int mp = 0, np = 0;       // top-left corner of the previous tile's wave tile
    bool have_prev = false;  // no stores while pk holds nothing (first tile)
    char* patch = smem_all + wave * 8192;
    const int c16 = lane & 7, rl = lane >> 3;
    const float inv_tok = GATE ? 1.0f / (float)a.tok_per_batch : 0.f;
    auto trk = [&](int PH, int s, int k, int u) {  // the trickle's micro-ops of slot (s, k) of the K-tile with phase PH of unit u
        const int sq = s * 16 + k;
        const int mu = mp + (u & 3) * 32, nu = np + (u >> 2) * 64;  // the unit's corner
        const int n = nu + c16 * 8;
        const bool n_ok = n < a.N;
        const int nc = n_ok ? n : 0;
        if (PH == 0 && s == 1 && k == 8) {  // accumulator layout -> patch (rows of 128 B, 16-B chunks XOR-swizzled by row & 7).  The ONLY
            // unit-specific instructions of the trickle: eight ds_write_b64 behind one wave-uniform switch (this slot overruns its MFMA
            // by ~100 cycles, once per unit); everything after the transposition works on the read-back registers
            char* wp = patch + (u & 1) * 4096 + fr * 128 + hi * 8;
            const int x7 = fr & 7;
#define S2V_Q4_WR(U)                                                                                                          \
    case U:                                                                                                                   \
        _Pragma("unroll") for (int w = 0; w < 8; ++w)                                                                        \
            *(u32x2*)(wp + ((((w >> 2) * 4 + (w & 3)) ^ x7) << 4)) = u32x2{pk[(U - 2) * 16 + 2 * w], pk[(U - 2) * 16 + 2 * w + 1]}; \
        break;
            switch (u) {
                S2V_Q4_WR(2) S2V_Q4_WR(3) S2V_Q4_WR(4) S2V_Q4_WR(5) S2V_Q4_WR(6) S2V_Q4_WR(7)
                default: break;  // units 0 and 1 are in their patches since the conversion
            }
#undef S2V_Q4_WR
        }
        if (PH == 0 && s == 2 && k >= 8 && k < 12) {  // read back row-major: row it*8 + lane/8, 16 B at column chunk lane%8
            const int it = k - 8, row = it * 8 + rl;
            rb[it] = *(const u32x4*)(patch + (u & 1) * 4096 + row * 128 + ((c16 ^ (row & 7)) << 4));
        }
        if (GELU && ((PH == 0 && s == 3) || PH == 1 || PH == 2 || (PH == 3 && s < 2))) {  // x * sigmoid(2u) as in gelu_tanh_fast() on the
            // read-back rows, two elements (one packed register) per eleven slots, two VALU instructions per slot
            const int qq = PH * 64 + sq - 48, pr = qq / 11, st = qq % 11, it = pr >> 2, e = pr & 3;
            if (pr < 16) {
                if (st == 0) { ga = __uint_as_float(rb[it][e] << 16); gb = __uint_as_float(rb[it][e] & 0xffff0000u); }
                if (st == 1) { ta = 0.044715f * ga; tb = 0.044715f * gb; }
                if (st == 2) { ta = ta * ga; tb = tb * gb; }
                if (st == 3) { ta = ta * ga + ga; tb = tb * gb + gb; }
                if (st == 4) { ta = 1.5957691216057308f * ta; tb = 1.5957691216057308f * tb; }
                if (st == 5) { ta = -1.4426950408889634f * ta; tb = -1.4426950408889634f * tb; }
                if (st == 6) { ta = __builtin_amdgcn_exp2f(ta); tb = __builtin_amdgcn_exp2f(tb); }
                if (st == 7) { ta = 1.0f + ta; tb = 1.0f + tb; }
                if (st == 8) { ta = __builtin_amdgcn_rcpf(ta); tb = __builtin_amdgcn_rcpf(tb); }
                if (st == 9) { ta = ga * ta; tb = gb * tb; }
                if (st == 10) rb[it][e] = pack2bf(ta, tb);
            }
        }
        if (HAS_LD && PH == 0 && s == 3 && (k & 1)) {  // residual / gate rows, right after the barrier (complete by the next one)
            const int it = (k >> 1) & 3, m = min(mu + it * 8 + rl, a.M - 1);
            if (k < 8) {
                xl[it] = GATE ? *(const u32x4*)((const bf16_t*)a.X + (size_t)m * a.ldx + nc) : *(const u32x4*)((const bf16_t*)a.R + (size_t)m * a.ldr + nc);
            } else if (GATE) {
                int b = (int)((float)m * inv_tok);  // m / tok_per_batch without the integer division (corrected below)
                int rr = m - b * a.tok_per_batch;
                if (rr < 0) { rr += a.tok_per_batch; --b; }
                if (rr >= a.tok_per_batch) { rr -= a.tok_per_batch; ++b; }
                const void* gsel = rr < a.text_len ? a.gate_txt : (a.gate_ref != nullptr && rr < a.text_len + a.ref_len) ? a.gate_ref : a.gate_vid;
                gl[it] = *(const u32x4*)((const bf16_t*)gsel + (size_t)b * a.gate_stride + nc);
            }
        }
        if (HAS_LD && ((PH == 1 && s == 3) || PH == 2 || (PH == 3 && s < 3))) {  // first use after the barrier of PH 1: loads are complete
            const int qq = (PH - 1) * 64 + sq - 48;  // 0 .. 127
            if (GATE) {  // rnd(x + rnd(gate * v)), one bf16 pair per eight slots
                const int pr = qq >> 3, st = qq & 7, it = pr >> 2, e = pr & 3;
                if (st == 0) { ga = __uint_as_float(gl[it][e] << 16); gb = __uint_as_float(gl[it][e] & 0xffff0000u); }
                if (st == 1) { ta = __uint_as_float(rb[it][e] << 16); tb = __uint_as_float(rb[it][e] & 0xffff0000u); }
                if (st == 2) { ta = ga * ta; tb = gb * tb; }
                if (st == 3) rb[it][e] = pack2bf(ta, tb);
                if (st == 4) { ta = __uint_as_float(rb[it][e] << 16); tb = __uint_as_float(rb[it][e] & 0xffff0000u); }
                if (st == 5) { ga = __uint_as_float(xl[it][e] << 16); gb = __uint_as_float(xl[it][e] & 0xffff0000u); }
                if (st == 6) { ta = ga + ta; tb = gb + tb; }
                if (st == 7) rb[it][e] = pack2bf(ta, tb);
            } else if (qq < 64) {  // rnd(v + r), one pair per four slots
                const int pr = qq >> 2, st = qq & 3, it = pr >> 2, e = pr & 3;
                if (st == 0) { ta = __uint_as_float(rb[it][e] << 16); tb = __uint_as_float(rb[it][e] & 0xffff0000u); }
                if (st == 1) { ga = __uint_as_float(xl[it][e] << 16); gb = __uint_as_float(xl[it][e] & 0xffff0000u); }
                if (st == 2) { ta = ga + ta; tb = gb + tb; }
                if (st == 3) rb[it][e] = pack2bf(ta, tb);
            }
        }
        if (PH == PH_ST && s == 3 && (k & 1) && k < 8) {  // four full-line stores, right after the barrier
            const int it = k >> 1, m = mu + it * 8 + rl;
            if (have_prev && m < a.M && n_ok) {
                if (GATE) *(u32x4*)((bf16_t*)a.X + (size_t)m * a.ldx + n) = rb[it];
                else *(u32x4*)((bf16_t*)a.C + (size_t)m * a.ldc + n) = rb[it];
            }
        }
    };

    // one K-tile: 4 steps of 16 MFMA.  MODE 1: first K-tile of an output tile (accumulators start from the inline constant 0, no
    // 256-instruction clear); MODE 2 / 3: the last two, whose LDS-DMA belongs to the next output tile (K-tiles 0 and 1 of it).
    // PH >= 0: the K-tile carries phase PH of the trickle for unit u
    auto ktile = [&](auto mode_tag, auto ph_tag, int u) {
        constexpr int MODE = decltype(mode_tag)::value, PH = decltype(ph_tag)::value;
        const unsigned stg = lds0 + so, stgo = lds0 + (so ^ 65536u);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s == NSTEP - 1) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                fence();
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = k >> 2, j = k & 3;
                if (MODE == 1 && s == 0) {
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cur][i], af[cur][j], z, 0, 0, 0);
                } else {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cur][i], af[cur][j], acc[i][j], 0, 0, 0);
                }
                if (k < NRD) {  // fragments of the next step (of the next K-tile after the last step)
                    if (s < NSTEP - 1) rd(false, s + 1, nxt, k);
                    else rd(true, 0, nxt, k);
                }
                if ((k & 1) == 0) {
                    const int p = k >> 1;
                    if (s == 0) {  // W half of K-tile t+1 -> the other stage
                        if (MODE == 3) glds16_saddr_m0_imm(Wnxt + p * pstrideW, offW, stgo, 32768 + p * 4096, 0);
                        else glds16_saddr_m0_imm(WK + p * pstrideW, offW, stgo, 32768 + p * 4096, 128);
                    }
                    if (s == NSTEP - 1) {  // A half of K-tile t+2 -> this stage (its reads ended before the barrier above)
                        if (MODE == 2) glds16_saddr_m0_imm(Anxt + p * pstrideA, offA, stg, p * 4096, 0);
                        else if (MODE == 3) glds16_saddr_m0_imm(Anxt + p * pstrideA, offA, stg, p * 4096, 128);
                        else glds16_saddr_m0_imm(AK + p * pstrideA, offA, stg, p * 4096, 256);
                    }
                }
                if (PH >= 0) trk(PH, s, k, u);
                fence();
            }
        }
        if (!(ACCT & 4)) {  // ablation 4: every K-tile re-reads K-tile 0 (cache-resident operands: the loop without memory latency)
            AK += ES * BKE;
            WK += ES * BKE;
        }
        so ^= 65536u;
    };
    using C0 = std::integral_constant<int, 0>;
    using CN = std::integral_constant<int, -1>;

    {  // K-tile 0 whole, A half of K-tile 1
        const unsigned stg = lds0, stgo = lds0 + 65536u;
#pragma unroll
        for (int p = 0; p < 8; ++p) glds16_saddr_m0_imm(AK + p * pstrideA, offA, stg, p * 4096, 0);
#pragma unroll
        for (int p = 0; p < 8; ++p) glds16_saddr_m0_imm(WK + p * pstrideW, offW, stg, 32768 + p * 4096, 0);
#pragma unroll
        for (int p = 0; p < 8; ++p) glds16_saddr_m0_imm(AK + p * pstrideA, offA, stgo, p * 4096, 128);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // K-tile 0 landed; the A half of K-tile 1 stays in flight
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int n = 0; n < NRD; ++n) rd(false, 0, 0, n);
    fence();
    if (ACCT) tacc[4] = now2() - tk0;

    for (int local = slot; local < cnt; local += nslots) {
        const long long tl0 = now2();
        ktile(std::integral_constant<int, 1>{}, CN{}, 0);
        for (int u = 0; u < 8; ++u) {  // 8 * NPH K-tiles carry the previous tile's epilogue, one unit after the other
            ktile(C0{}, std::integral_constant<int, 0>{}, u);
            if (NPH > 1) {
                ktile(C0{}, std::integral_constant<int, 1>{}, u);
                ktile(C0{}, std::integral_constant<int, 2>{}, u);
                ktile(C0{}, std::integral_constant<int, 3>{}, u);
            }
        }
        for (int t = 1 + 8 * NPH; t < nT - 2; ++t) ktile(C0{}, CN{}, 0);
        ktile(std::integral_constant<int, 2>{}, CN{}, 0);
        // bias: ONE more K step whose W fragment holds bias[n] at k = 0 and whose A fragment holds 1.0 at k = 0 -- the matrix pipe adds
        // acc + bias * 1 in fp32 (exact product, one rounding, after the last real K step: the order of the plain epilogue) for 16
        // MFMAs per tile instead of 32 bias registers and 256 v_add on a wave that issues one VALU instruction per ~5 cycles
        unsigned short bz[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nb = min(n0 + wn * 128 + i * 32, a.N - 32) + fr;
            bz[i] = (a.bias && hi == 0) ? ((const unsigned short*)a.bias)[nb] : (unsigned short)0;
        }
        ktile(std::integral_constant<int, 3>{}, CN{}, 0);
        {
            bf16x8 one = {0, 0, 0, 0, 0, 0, 0, 0};
            one[0] = hi == 0 ? (__bf16)1.0f : (__bf16)0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf16x8 bw = {0, 0, 0, 0, 0, 0, 0, 0};
                bw[0] = __builtin_bit_cast(__bf16, bz[i]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw, one, acc[i][j], 0, 0, 0);
            }
        }
        const long long te0 = now2();
        if (ACCT) tacc[3] += te0 - tl0;

        // conversion: rnd_bf16(accumulator) -> pk (the only part of the epilogue that is not hidden)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int un = (i >> 1) * 4 + j, w = (i & 1) * 4 + rq;
                    const unsigned p0 = pack2bf(acc[i][j][rq * 4 + 0], acc[i][j][rq * 4 + 1]);
                    const unsigned p1 = pack2bf(acc[i][j][rq * 4 + 2], acc[i][j][rq * 4 + 3]);
                    if (un < 2) {
                        *(u32x2*)(patch + un * 4096 + fr * 128 + hi * 8 + ((((w >> 2) * 4 + (w & 3)) ^ (fr & 7)) << 4)) = u32x2{p0, p1};
                    } else {
                        pk[(un - 2) * 16 + 2 * w] = p0;
                        pk[(un - 2) * 16 + 2 * w + 1] = p1;
                    }
                }
        mp = m0 + wm * 128;
        np = n0 + wn * 128;
        have_prev = true;
        fence();
        if (ACCT) tacc[0] += now2() - te0;

        // the stream moves on: the K-tiles nT, nT+1 issued above are K-tiles 0, 1 of the next output tile
        m0 = m1;
        n0 = n1;
        AK = Anxt;
        WK = Wnxt;
        tile_origin(local + 2 * nslots, m1, n1);
        Anxt = (const char*)a.A + ES * (int64_t)m1 * a.lda;
        Wnxt = (const char*)a.W + ES * (int64_t)n1 * a.ldw;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // LDS-DMA of the (unused) K-tiles past the last output tile
    const long long tf0 = now2();
#pragma unroll
    for (int u = 0; u < 8; ++u) {  // the last tile's epilogue, unit by unit through the same patches, nothing to hide behind
        const int mu = mp + (u & 3) * 32, nu = np + (u >> 2) * 64;
        const int n = nu + c16 * 8;
        char* pu = patch + (u & 1) * 4096;
        if (u >= 2) {
#pragma unroll
            for (int w = 0; w < 8; ++w)
                *(u32x2*)(pu + fr * 128 + hi * 8 + ((((w >> 2) * 4 + (w & 3)) ^ (fr & 7)) << 4)) = u32x2{pk[(u - 2) * 16 + 2 * w], pk[(u - 2) * 16 + 2 * w + 1]};
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + rl, m = mu + row;
            u32x4 v = *(const u32x4*)(pu + row * 128 + ((c16 ^ (row & 7)) << 4));
            if (GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = pack2bf(gelu_tanh_fast(__uint_as_float(v[e] << 16)), gelu_tanh_fast(__uint_as_float(v[e] & 0xffff0000u)));
            }
            if (have_prev && m < a.M && n < a.N) *(u32x4*)((bf16_t*)a.C + (size_t)m * a.ldc + n) = v;
        }
    }
#ifdef S2V_DIAG
    if (ACCT) {
        tacc[1] = now2() - tf0;                                         // final flush
        tacc[5] = now2() - tk0;                                         // whole workgroup
        tacc[6] = (long long)__builtin_amdgcn_s_memrealtime() - tr0;    // the same in 100 MHz ticks
        tacc[7] = (cnt - slot + nslots - 1) / nslots;                   // output tiles of this workgroup
        if (blockIdx.x == 100 && lane == 0)
            for (int e = 0; e < 8; ++e) g_pp_dbg[wave * 8 + e] = tacc[e];
        if (tid == 0 && blockIdx.x < 256) {
            long long* o = g_blk_times + (launch_no & 1) * 1024 + blockIdx.x * 4;
            o[0] = tr0;
            o[1] = tr0 + tacc[6];
            o[2] = tacc[5];
            o[3] = tacc[7];
            if (blockIdx.x == 0) g_launch_no = launch_no + 1;
        }
    }
#endif
}

template <int EPI>
static int launch_q4_t(const GemmArgs& a, hipStream_t st) {
    const int tiles_m = (a.M + WBM - 1) / WBM, tiles_n = (a.N + WBN - 1) / WBN;
    const void* fn = (const void*)gemm_q4<EPI, 0>;
    if (EPI == EPI_BIAS) switch (a.ablate) {  // stall accounting
            case 5: fn = (const void*)gemm_q4<EPI_BIAS, 2>; break;
            case 6: fn = (const void*)gemm_q4<EPI_BIAS, 2 | 4>; break;
            default: break;
        }
    S2V_TRY(ensure_lds_attr(fn, 163840));
    int dev = 0, ncu = 256;
    S2V_CHECK_HIP(hipGetDevice(&dev));
    S2V_CHECK_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const int grid = std::min((ncu / 8) * 8, ((tiles_m * tiles_n + 7) / 8) * 8);
    void* args[] = {(void*)&a, (void*)&tiles_m, (void*)&tiles_n};
    S2V_CHECK_HIP(hipLaunchKernel(fn, dim3(grid), dim3(256), args, 163840, st));
    return 0;
}
#endif

// compute units of the current device, queried once per device (launch-path heuristics count tile ROUNDS in these)
static int device_cus() {
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

template <int EPI, typename T16 = bf16_t>
static int launch_pp64_t(const GemmArgs& a_in, hipStream_t st) {
    GemmArgs a = a_in;
    const int tiles_m = (a.M + WBM - 1) / WBM, tiles_n = (a.N + WBN - 1) / WBN;
    if (a.gm <= 0) a.gm = (!a.conv && tiles_n <= 16 && tiles_m >= 32) ? (a.K >= 8192 ? 1 : 3) : 4;
    S2V_TRY(ensure_lds_attr((const void*)gemm_bf16_pp64<EPI, 0, false, T16>, 131072));
#ifdef S2V_DIAG
    if (EPI == EPI_BIAS && a.ablate) {  // diagnostics only (tools/ablate_gemm.py)
        const void* fn = a.ablate == 1 ? (const void*)gemm_bf16_pp64<EPI_BIAS, 1> : a.ablate == 4 ? (const void*)gemm_bf16_pp64<EPI_BIAS, 4> : a.ablate == 5 ? (const void*)gemm_bf16_pp64<EPI_BIAS, 5> : a.ablate == 6 ? (const void*)gemm_bf16_pp64<EPI_BIAS, 6> : a.ablate == 7 ? (const void*)gemm_bf16_pp64<EPI_BIAS, 7> : (const void*)gemm_bf16_pp64<EPI_BIAS, 3>;
        S2V_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        void* args[] = {(void*)&a, (void*)&tiles_m, (void*)&tiles_n};
        S2V_CHECK_HIP(hipLaunchKernel(fn, dim3(tiles_m * tiles_n), dim3(512), args, 131072, st));
        return 0;
    }
#endif
    hipLaunchKernelGGL((gemm_bf16_pp64<EPI, 0, false, T16>), dim3(tiles_m * tiles_n), dim3(512), 131072, st, a, tiles_m, tiles_n);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

#ifdef S2V_DIAG
template <int EPI>
static int launch_w8_t(const GemmArgs& a, hipStream_t st) {
    const int tiles_m = (a.M + WBM - 1) / WBM, tiles_n = (a.N + WBN - 1) / WBN;
    S2V_TRY(ensure_lds_attr((const void*)gemm_bf16_w8<EPI>, WH_NST * WH_STAGE));
    hipLaunchKernelGGL(gemm_bf16_w8<EPI>, dim3(tiles_m * tiles_n), dim3(512), WH_NST * WH_STAGE, st, a, tiles_m, tiles_n);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}
#endif

template <int EPI, typename T16 = bf16_t>
static int launch_stag_t(const GemmArgs& a, hipStream_t st) {
    const int tiles_m = (a.M + RBM - 1) / RBM, tiles_n = (a.N + RBN - 1) / RBN;
    S2V_TRY(ensure_lds_attr((const void*)gemm_bf16_stag<EPI, T16>, 3 * RSTAGE));
    hipLaunchKernelGGL((gemm_bf16_stag<EPI, T16>), dim3(tiles_m * tiles_n), dim3(512), 3 * RSTAGE, st, a, tiles_m, tiles_n);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// 7 = 256x256x64 eight-wave ping-pong (the product schedule), 2 = staggered 256x128 ring, 0 = 128x128 double buffer; 7 falls back
// to 2 (then 0) when the shape does not fit its tiles.  libs2v_hip_diag.so (S2V_DIAG) can also select 5 = the 256x256 eight-wave
// lock-step ring and the compile-time ablations of gemm_bf16_pp64 (tools/ablate_gemm.py, tools/stall_pp64.py).
#ifdef S2V_DIAG
int g_gemm_ablate = 0;
int g_gemm_impl = 9;  // 9: gemm_g4 where it qualifies, gemm_bf16_pp64 otherwise (the product's choice); 7: gemm_bf16_pp64; 5 / 8: A/B references
int g_gemm_g4t = 1;   // the persistent trickled-epilogue form of gemm_g4 where it qualifies (s2v_set_gemm_g4t: A/B switch of the diagnostics build)
extern "C" __attribute__((visibility("default"))) int s2v_set_gemm_g4t(int on) { g_gemm_g4t = on; return 0; }
extern "C" __attribute__((visibility("default"))) int s2v_set_gemm_impl(int impl) { g_gemm_impl = impl & 0xff; g_gemm_ablate = impl >> 8; return 0; }
#else
static constexpr int g_gemm_ablate = 0;
static constexpr int g_gemm_impl = 9;  // 9: gemm_g4 where it qualifies, gemm_bf16_pp64 otherwise (the product's choice); 7: gemm_bf16_pp64; 5 / 8: A/B references
static constexpr int g_gemm_g4t = 1;
#endif

template <int EPI>
static int launch_fp8_t(const GemmArgs& a, hipStream_t st) {
    const int tiles_m = (a.M + WBM - 1) / WBM, tiles_n = (a.N + WBN - 1) / WBN;
    S2V_TRY(ensure_lds_attr((const void*)gemm_bf16_pp64<EPI, 0, true>, 131072 + 2048));  // + two stages of A block scales (MX)
    hipLaunchKernelGGL((gemm_bf16_pp64<EPI, 0, true>), dim3(tiles_m * tiles_n), dim3(512), 131072 + 2048, st, a, tiles_m, tiles_n);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_gemm_fp8(const GemmArgs& a, int epi, hipStream_t st) {
    S2V_REQUIRE(!a.conv && (a.a_scale || a.mx_a_s) && a.w_scale, "gemm_fp8: plain mode with the weight scales and row or block scales of A only");
    S2V_REQUIRE(!a.mx_out_q || (epi == EPI_BIAS_GELU && a.mx_out_s && a.N % 64 == 0), "gemm_fp8: MX output is the GELU epilogue's, N a multiple of 64");
    S2V_REQUIRE(!(a.mx_out_q || a.mx_a_s) || a.mx_rows >= ((a.M + WBM - 1) / WBM) * WBM, "gemm_fp8: MX block scales are K-tile major, mx_rows must cover the padded M");
    S2V_REQUIRE(a.K % 128 == 0 && a.lda % 16 == 0 && a.ldw % 16 == 0, "gemm_fp8: K must be a multiple of 128, rows 16-byte aligned");
    S2V_REQUIRE(a.a_rows_padded >= ((a.M + WBM - 1) / WBM) * WBM && a.w_rows_padded >= ((a.N + WBN - 1) / WBN) * WBN,
                "gemm_fp8: operands must be padded to whole 256-row tiles");
    S2V_REQUIRE(epi_vec_ok(a, epi), "gemm_fp8: output rows must be 16-byte aligned and N a multiple of 8");
    if (g_gemm_impl == 9 && gemm_g4f_ok(a, epi)) return launch_gemm_g4f(a, epi, st);  // four-wave generated-asm loop (gemm_g4f.hip)
    switch (epi) {
        case EPI_BIAS: return launch_fp8_t<EPI_BIAS>(a, st);
        case EPI_BIAS_GELU: return launch_fp8_t<EPI_BIAS_GELU>(a, st);
        case EPI_BIAS_GATE_RES: return launch_fp8_t<EPI_BIAS_GATE_RES>(a, st);
        case EPI_BIAS_QKNORM: return launch_fp8_t<EPI_BIAS_QKNORM>(a, st);
        default: return s2v_fail(__FILE__, __LINE__, "gemm_fp8: bad epilogue", -1);
    }
}

// the 256-column kernels take N that is not a multiple of 256 when the weight buffer physically holds the padded rows and
// the last tile is at least half full (N = 1920 / 5760 of the 2B model); narrower outputs go to the 128-column kernels
static bool w_tile_ok(const GemmArgs& a) {
    if (a.N % WBN == 0) return true;
    return a.N > WBN && a.w_rows_padded >= ((a.N + WBN - 1) / WBN) * WBN && (a.N % WBN) >= WBN / 2;
}

int launch_gemm_bf16(const GemmArgs& a0, int epi, hipStream_t st) {
    GemmArgs a = a0;
    a.ablate = g_gemm_ablate;
    if (epi == EPI_BIAS_QKNORM) {  // only the vector epilogue of the 256- and 128-row kernels implements it
        S2V_REQUIRE(!a.conv && epi_vec_ok(a, epi) && a.qk_D > 0 && a.qk_D % 64 == 0 && a.N == 3 * a.qk_D && a.tok_per_batch > 0 && a.qk_w[0] && a.qk_w[1] &&
                        a.qk_b[0] && a.qk_b[1],
                    "gemm_bf16: fused qk-norm needs the vector epilogue, N = 3 * qk_D and the LayerNorm parameters");
        if (!(a.m_begin > 0) && !(w_tile_ok(a) && a.a_rows_padded >= ((a.M + WBM - 1) / WBM) * WBM)) {
            const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
            if (a.f16) hipLaunchKernelGGL((gemm_bf16_128<EPI_BIAS_QKNORM, f16_t>), dim3(tiles_m * tiles_n), dim3(256), 4 * TILE_BYTES, st, a, tiles_m, tiles_n);
            else hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS_QKNORM>, dim3(tiles_m * tiles_n), dim3(256), 4 * TILE_BYTES, st, a, tiles_m, tiles_n);
            S2V_CHECK_HIP(hipGetLastError());
            return 0;
        }
    }
    if (a.m_begin > 0) {  // row tail of a split GEMM: the 128 x 128 kernel on rows [m_begin, M)
        S2V_REQUIRE(!a.conv && a.K % BK == 0 && a.lda % 8 == 0 && a.ldw % 8 == 0, "gemm_bf16: bad tail launch");
        const int tiles_m = (a.M - a.m_begin + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
        const size_t shmem = 4 * TILE_BYTES;
        const dim3 grid(tiles_m * tiles_n);
#define S2V_TAIL128(E)                                                                                                              \
    if (a.f16) hipLaunchKernelGGL((gemm_bf16_128<E, f16_t>), grid, dim3(256), shmem, st, a, tiles_m, tiles_n);                         \
    else hipLaunchKernelGGL(gemm_bf16_128<E>, grid, dim3(256), shmem, st, a, tiles_m, tiles_n);                                        \
    break;
        switch (epi) {  // a.f16: the fp16 instantiation (until round 5's last day this branch launched the bf16 kernel on fp16 operands)
            case EPI_BIAS: S2V_TAIL128(EPI_BIAS)
            case EPI_BIAS_GELU: S2V_TAIL128(EPI_BIAS_GELU)
            case EPI_BIAS_GATE_RES: S2V_TAIL128(EPI_BIAS_GATE_RES)
            case EPI_BIAS_ADD: S2V_TAIL128(EPI_BIAS_ADD)
            case EPI_BIAS_QKNORM: S2V_TAIL128(EPI_BIAS_QKNORM)
            default: return s2v_fail(__FILE__, __LINE__, "gemm_bf16: bad epilogue", -1);
        }
#undef S2V_TAIL128
        S2V_CHECK_HIP(hipGetLastError());
        return 0;
    }
    S2V_REQUIRE(a.K % BK == 0, "gemm_bf16: K must be a multiple of 64");
    // a.f16 (round 5): the operands are fp16 -- the same dispatch on the kernels' fp16 instantiations (the four-wave asm loop with the fp16 mnemonic,
    // the eight-wave ping-pong, the staggered 256 x 128 ring, the 128 x 128 kernel), the fused q/k-norm epilogue included; not for fp16: gemm_g4t, split K
    S2V_REQUIRE(!a.f16 || (a.splitk <= 1 && a.mx_out_q == nullptr), "gemm (fp16): no split K or MX output");
#ifdef S2V_DIAG
    if (g_gemm_impl == 8 && !a.conv && (epi == EPI_BIAS || epi == EPI_BIAS_GELU) && w_tile_ok(a) && a.N % 64 == 0 && a.K >= 36 * 64 &&
        epi_vec_ok(a, epi) && a.a_rows_padded >= ((a.M + WBM - 1) / WBM) * WBM && a.lda % 8 == 0 && a.ldw % 8 == 0) {
        return epi == EPI_BIAS ? launch_q4_t<EPI_BIAS>(a, st) : launch_q4_t<EPI_BIAS_GELU>(a, st);
    }
#endif
    bool g4_epi = true;
#ifdef S2V_DIAG
    if (const char* e = getenv("S2V_G4_EPI_MASK")) g4_epi = (atoi(e) >> epi) & 1;  // bisecting aid: g4 for the epilogues of the mask only
#endif
    // One round of tiles (<= one per CU) leaves the epilogue fully exposed, and the fused q/k-norm + rotary epilogue is the longest:
    // eight waves run it faster than four (C1 QKV, 230 tiles: 65 us on the ping-pong kernel, 72 us on gemm_g4)
    const int64_t out_tiles = (int64_t)((a.M + 255) / 256) * ((a.N + 255) / 256), cus = device_cus();  // rounds are counted in THIS device's CUs
    if (epi == EPI_BIAS_QKNORM && out_tiles <= cus) g4_epi = false;
    // the bias + GELU epilogue is the next longest: up to two rounds of tiles with a short reduction (C1 FF1: 300 tiles of 30 K-tiles) also run
    // faster on eight waves -- same-box A/B of the C1 step, 11.50 -> 11.12 ms (tools/c1_attn_kernel_probe.py, S2V_G4_EPI_MASK); same epilogue
    // code, bit-identical results
    if (epi == EPI_BIAS_GELU && a.splitk <= 1 && a.K <= 2048 && out_tiles <= 2 * cus) g4_epi = false;
    // a convolution whose 256 x 256 tiles would leave half the CUs idle (the VAE's latent-resolution layers: M = 10800, N = 512 -> 86 tiles of
    // 216 K-tiles, 0.52 PF, profiles/r04_vae_conv_rates.txt) runs on 256 x 128 tiles instead
    const bool conv_few = a.conv && out_tiles * 2 <= cus && epi != EPI_BIAS_QKNORM;
    const bool big_tiles = (a.tile == 0 && !conv_few) || (a.conv && !conv_few) || epi == EPI_BIAS_QKNORM;  // GemmArgs::tile: the caller asks for smaller tiles
    bool g4t_epi = true;
#ifdef S2V_DIAG
    if (const char* e = getenv("S2V_G4T_EPI_MASK")) g4t_epi = (atoi(e) >> epi) & 1;  // same-box A/B: gemm_g4t for the epilogues of the mask only (tools/epi_mask_probe.py)
#endif
    if (g_gemm_impl == 9 && g_gemm_g4t && g4t_epi && !a.f16 && big_tiles && w_tile_ok(a) && gemm_g4_ok(a, epi) && gemm_g4t_ok(a, epi, (int)cus)) return launch_gemm_g4t(a, epi, st);
    if (g_gemm_impl == 9 && big_tiles && g4_epi && w_tile_ok(a) && gemm_g4_ok(a, epi)) return a.f16 ? launch_gemm_g4_f16(a, epi, st) : launch_gemm_g4(a, epi, st);  // four-wave generated-asm K loop
    if ((g_gemm_impl == 7 || g_gemm_impl == 8 || g_gemm_impl == 9) && big_tiles && w_tile_ok(a) && (a.conv || a.a_rows_padded >= ((a.M + WBM - 1) / WBM) * WBM)) {
        S2V_REQUIRE((a.conv ? a.cin % 64 == 0 : a.lda % 8 == 0) && a.ldw % 8 == 0, "gemm_bf16: bad leading dims");
        switch (epi) {
            case EPI_BIAS: return a.f16 ? launch_pp64_t<EPI_BIAS, f16_t>(a, st) : launch_pp64_t<EPI_BIAS>(a, st);
            case EPI_BIAS_GELU: return a.f16 ? launch_pp64_t<EPI_BIAS_GELU, f16_t>(a, st) : launch_pp64_t<EPI_BIAS_GELU>(a, st);
            case EPI_BIAS_GATE_RES: return a.f16 ? launch_pp64_t<EPI_BIAS_GATE_RES, f16_t>(a, st) : launch_pp64_t<EPI_BIAS_GATE_RES>(a, st);
            case EPI_BIAS_ADD: return a.f16 ? launch_pp64_t<EPI_BIAS_ADD, f16_t>(a, st) : launch_pp64_t<EPI_BIAS_ADD>(a, st);
            case EPI_BIAS_QKNORM: return a.f16 ? launch_pp64_t<EPI_BIAS_QKNORM, f16_t>(a, st) : launch_pp64_t<EPI_BIAS_QKNORM>(a, st);
            default: return s2v_fail(__FILE__, __LINE__, "gemm_bf16: bad epilogue", -1);
        }
    }
#ifdef S2V_DIAG
    if (g_gemm_impl == 5 && w_tile_ok(a) && (a.conv || a.a_rows_padded >= ((a.M + WBM - 1) / WBM) * WBM)) {
        S2V_REQUIRE((a.conv ? a.cin % 64 == 0 : a.lda % 8 == 0) && a.ldw % 8 == 0, "gemm_bf16: bad leading dims");
        switch (epi) {
            case EPI_BIAS: return launch_w8_t<EPI_BIAS>(a, st);
            case EPI_BIAS_GELU: return launch_w8_t<EPI_BIAS_GELU>(a, st);
            case EPI_BIAS_GATE_RES: return launch_w8_t<EPI_BIAS_GATE_RES>(a, st);
            case EPI_BIAS_ADD: return launch_w8_t<EPI_BIAS_ADD>(a, st);
            default: return s2v_fail(__FILE__, __LINE__, "gemm_bf16: bad epilogue", -1);
        }
    }
#endif
    if ((g_gemm_impl == 2 || g_gemm_impl >= 4) && (a.tile != 2 || a.conv) && epi != EPI_BIAS_QKNORM && (a.conv || a.a_rows_padded >= ((a.M + RBM - 1) / RBM) * RBM)) {
        S2V_REQUIRE((a.conv ? a.cin % 64 == 0 : a.lda % 8 == 0) && a.ldw % 8 == 0, "gemm_bf16: bad leading dims");
        switch (epi) {
            case EPI_BIAS: return a.f16 ? launch_stag_t<EPI_BIAS, f16_t>(a, st) : launch_stag_t<EPI_BIAS>(a, st);
            case EPI_BIAS_GELU: return a.f16 ? launch_stag_t<EPI_BIAS_GELU, f16_t>(a, st) : launch_stag_t<EPI_BIAS_GELU>(a, st);
            case EPI_BIAS_GATE_RES: return a.f16 ? launch_stag_t<EPI_BIAS_GATE_RES, f16_t>(a, st) : launch_stag_t<EPI_BIAS_GATE_RES>(a, st);
            case EPI_BIAS_ADD: return a.f16 ? launch_stag_t<EPI_BIAS_ADD, f16_t>(a, st) : launch_stag_t<EPI_BIAS_ADD>(a, st);
            default: return s2v_fail(__FILE__, __LINE__, "gemm_bf16: bad epilogue", -1);
        }
    }
    S2V_REQUIRE((a.conv ? a.cin % 64 == 0 : a.lda % 8 == 0) && a.ldw % 8 == 0, "gemm_bf16: bad leading dims");
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int grid = tiles_m * tiles_n;
    const size_t shmem = 4 * TILE_BYTES;
    switch (epi) {
        case EPI_BIAS:
            if (a.f16) hipLaunchKernelGGL((gemm_bf16_128<EPI_BIAS, f16_t>), dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            else hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS>, dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            break;
        case EPI_BIAS_GELU:
            if (a.f16) hipLaunchKernelGGL((gemm_bf16_128<EPI_BIAS_GELU, f16_t>), dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            else hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS_GELU>, dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            break;
        case EPI_BIAS_GATE_RES:
            if (a.f16) hipLaunchKernelGGL((gemm_bf16_128<EPI_BIAS_GATE_RES, f16_t>), dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            else hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS_GATE_RES>, dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            break;
        case EPI_BIAS_ADD:
            if (a.f16) hipLaunchKernelGGL((gemm_bf16_128<EPI_BIAS_ADD, f16_t>), dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            else hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS_ADD>, dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            break;
        case EPI_BIAS_QKNORM:
            if (a.f16) hipLaunchKernelGGL((gemm_bf16_128<EPI_BIAS_QKNORM, f16_t>), dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            else hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS_QKNORM>, dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            break;
        default:
            return s2v_fail(__FILE__, __LINE__, "gemm_bf16: bad epilogue", -1);
    }
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// simple tiled kernel: 64x64 tile, BK 16, 256 threads, 4x4 outputs / thread (n contiguous)
template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm_simple_k(const GemmArgs a) {
    __shared__ float sA[16][64 + 4];
    __shared__ float sW[16][64 + 4];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tm = (tid >> 4) * 4, tn = (tid & 15) * 4;
    const T* A = (const T*)a.A;
    const T* W = (const T*)a.W;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < a.K; k0 += 16) {
        // 64 rows x 16 k per operand = 1024 elements, 4 per thread
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * 256 + tid;
            const int row = e >> 4, k = e & 15;
            float va = 0.f, vw = 0.f;
            if (k0 + k < a.K) {
                if (m0 + row < a.M) va = ET<T>::ld(A + a_row_base(a, m0 + row) + a_k_off(a, k0 + k));
                if (n0 + row < a.N) vw = ET<T>::ld(W + (size_t)(n0 + row) * a.ldw + k0 + k);
            }
            sA[k][row] = va;
            sW[k][row] = vw;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float av[4], wv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = sA[k][tm + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = sW[k][tn + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], wv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (n0 + tn < a.N) epilogue4<T, EPI>(a, m0 + tm + i, n0 + tn, acc[i]);
    }
}

template <typename T>
static int launch_simple_t(const GemmArgs& a, int epi, hipStream_t st) {
    dim3 grid((a.N + 63) / 64, (a.M + 63) / 64);
    switch (epi) {
        case EPI_BIAS: hipLaunchKernelGGL((gemm_simple_k<T, EPI_BIAS>), grid, dim3(256), 0, st, a); break;
        case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm_simple_k<T, EPI_BIAS_GELU>), grid, dim3(256), 0, st, a); break;
        case EPI_BIAS_GATE_RES:
            hipLaunchKernelGGL((gemm_simple_k<T, EPI_BIAS_GATE_RES>), grid, dim3(256), 0, st, a);
            break;
        case EPI_BIAS_ADD: hipLaunchKernelGGL((gemm_simple_k<T, EPI_BIAS_ADD>), grid, dim3(256), 0, st, a); break;
        default: return s2v_fail(__FILE__, __LINE__, "gemm_simple: bad epilogue", -1);
    }
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_simple(const GemmArgs& a, int epi, int dtype, hipStream_t st) {
    if (dtype == S2V_F32 && gemm_f32m_ok(a)) return launch_gemm_f32m(a, epi, st);
    S2V_REQUIRE((a.M + 63) / 64 <= 65535, "gemm_simple: M too large for the generic kernel");
    S2V_DT_DISPATCH(dtype, return launch_simple_t<T>(a, epi, st))
    return 0;
}

// generic strided fp32 GEMM-accumulate (load-time only; LoRA merge W += alpha * B.A)
__global__ void gemm_strided_f32_k(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk,
                                   float* C, int64_t ldc, int M, int N, int K, float alpha) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    if (n >= N || m >= M) return;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(A[m * sam + k * sak], B[n * sbn + k * sbk], acc);
    C[m * ldc + n] += alpha * acc;
}
int launch_gemm_strided_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk,
                            float* C, int64_t ldc, int M, int N, int K, float alpha, hipStream_t st) {
    dim3 grid((N + 255) / 256, M);
    hipLaunchKernelGGL(gemm_strided_f32_k, grid, dim3(256), 0, st, A, sam, sak, B, sbn, sbk, C, ldc, M, N, K, alpha);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}
