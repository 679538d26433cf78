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

// ---------------------------------------------------------------------------------------------------
// shared epilogue: 4 consecutive columns n..n+3 of row m
template <typename T, int EPI>
__device__ __forceinline__ void epilogue4(const GemmArgs& a, int m, int n, const float v[4]) {
    if (m >= a.M) return;
    const T* bias = (const T*)a.bias;
    float y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float bv = (bias && n + i < a.N) ? ET<T>::ld(bias + n + i) : 0.f;
        y[i] = ET<T>::rnd(v[i] + bv);
    }
    if (EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = ET<T>::rnd(gelu_tanh_f(y[i]));
    }
    if (EPI == EPI_BIAS_GATE_RES) {
        const int b = m / a.tok_per_batch;
        const int r = m - b * a.tok_per_batch;
        const void* gsel = r < a.text_len ? a.gate_txt : (a.gate_ref != nullptr && r < a.text_len + a.ref_len) ? a.gate_ref : a.gate_vid;
        const T* gate = (const T*)gsel + (size_t)b * a.gate_stride;
        T* x = (T*)a.X + (size_t)m * a.ldx + n;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (n + i < a.N) {
                float t = ET<T>::rnd(ET<T>::ld(gate + n + i) * y[i]);
                ET<T>::st(x + i, ET<T>::ld(x + i) + t);
            }
        }
        return;
    }
    if (EPI == EPI_BIAS_ADD) {
        const T* r = (const T*)a.R + (size_t)m * a.ldr + n;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (n + i < a.N) y[i] = y[i] + ET<T>::ld(r + i);
    }
    T* c = (T*)a.C + (size_t)m * a.ldc + n;
    if (n + 3 < a.N && (a.ldc & 3) == 0) {
        if (sizeof(T) == 2) {
            u32x2 p;
            p.x = pack2bf(y[0], y[1]);
            p.y = pack2bf(y[2], y[3]);
            *(u32x2*)c = p;
        } else {
            *(f32x4*)c = (f32x4){y[0], y[1], y[2], y[3]};
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (n + i < a.N) ET<T>::st(c + i, y[i]);
    }
}

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

// Epilogue of one wave's 64(m) x 64(n) accumulator tile (2x2 MFMA 32x32 blocks, D[i = n][j = m]).
// Accumulator layout -> +bias, (GELU), round to bf16 in registers -> the wave's private 8 KiB LDS patch (rows of
// 128 B, 16-B chunks XOR-swizzled by row&7) -> read back row-major, 16 B per lane, 8 lanes per 128-B line -> gate /
// residual in that layout -> full-line global stores.  A row-per-lane epilogue (8-B stores at a row stride) was
// store-issue bound: ~0.7 ms of a 3.4 ms FF1 launch.
template <int EPI, int MB, bool SC = false>
__device__ __forceinline__ void epilogue_wave(const GemmArgs& a, const f32x16 (&acc)[2][MB], int mw, int nw, char* patch, int lane) {
    // MB 32-row blocks: the patch holds MB*32 rows of 128 B (8 KiB for MB = 2, 16 KiB for MB = 4); all accumulator blocks are
    // written first, then read back, so the LDS round trip and the bias loads are paid once per wave tile
    const int fr = lane & 31, hi = lane >> 5;
    const bf16_t* bias = (const bf16_t*)a.bias;
    u32x2 bvec[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int nl = (q >> 2) * 32 + 8 * (q & 3) + 4 * hi;
        bvec[q] = u32x2{0u, 0u};
        if (bias && nw + nl + 3 < a.N) bvec[q] = *(const u32x2*)(bias + nw + nl);  // 8-byte aligned: nw % 64 == 0, nl % 4 == 0
        else if (bias) {
            unsigned short t[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (nw + nl + e < a.N) t[e] = ((const unsigned short*)bias)[nw + nl + e];
            bvec[q] = u32x2{(unsigned)t[0] | ((unsigned)t[1] << 16), (unsigned)t[2] | ((unsigned)t[3] << 16)};
        }
    }
    // SC (fp8 operands): acc * a_scale[row] * w_scale[column] first -- the dequantisation of the per-token / per-channel scales
    float sa[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) sa[j] = SC ? a.a_scale[min(mw + j * 32 + fr, a.M - 1)] : 1.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int nl = i * 32 + 8 * rq + 4 * hi;  // local column of 4 consecutive outputs
            const u32x2 bq = bvec[i * 4 + rq];
            f32x4 sw = {1.f, 1.f, 1.f, 1.f};
            if (SC) sw = *(const f32x4*)(a.w_scale + nw + nl);  // the scale array is padded to the 256-column tile
            const float bv[4] = {__uint_as_float(bq.x << 16), __uint_as_float(bq.x & 0xffff0000u), __uint_as_float(bq.y << 16),
                                 __uint_as_float(bq.y & 0xffff0000u)};
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float av = SC ? acc[i][j][rq * 4 + e] * (sa[j] * sw[e]) : acc[i][j][rq * 4 + e];
                    y[e] = bf2f(f2bf(av + bv[e]));
                    if (EPI == EPI_BIAS_GELU) y[e] = gelu_tanh_fast(y[e]);
                }
                const int row = j * 32 + fr;
                u32x2 p;
                p.x = pack2bf(y[0], y[1]);
                p.y = pack2bf(y[2], y[3]);
                *(u32x2*)(patch + row * 128 + ((((nl >> 3) ^ (row & 7))) << 4) + (nl & 4) * 2) = p;
            }
        }
    // same-wave LDS accesses are ordered; the compiler inserts the lgkmcnt wait for the dependent reads
    const int c16 = lane & 7;
    const int n = nw + c16 * 8;
#pragma unroll
    for (int it = 0; it < MB * 4; ++it) {
        const int row = it * 8 + (lane >> 3);
        const int m = mw + row;
        u32x4 v = *(const u32x4*)(patch + row * 128 + ((c16 ^ (row & 7)) << 4));
        if (m >= a.M || n >= a.N) continue;
        if (EPI == EPI_BIAS_GATE_RES) {
            const int b = m / a.tok_per_batch;
            const int r = m - b * a.tok_per_batch;
            const void* gsel = r < a.text_len ? a.gate_txt : (a.gate_ref != nullptr && r < a.text_len + a.ref_len) ? a.gate_ref : a.gate_vid;
            const bf16_t* gate = (const bf16_t*)gsel + (size_t)b * a.gate_stride + n;
            bf16_t* x = (bf16_t*)a.X + (size_t)m * a.ldx + n;
            const u32x4 g = *(const u32x4*)gate;
            const u32x4 xo = *(const u32x4*)x;
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t0 = bf2f(f2bf(__uint_as_float(g[e] << 16) * __uint_as_float(v[e] << 16)));
                const float t1 = bf2f(f2bf(__uint_as_float(g[e] & 0xffff0000u) * __uint_as_float(v[e] & 0xffff0000u)));
                o[e] = pack2bf(__uint_as_float(xo[e] << 16) + t0, __uint_as_float(xo[e] & 0xffff0000u) + t1);
            }
            *(u32x4*)x = o;
        } else {
            if (EPI == EPI_BIAS_ADD) {
                const u32x4 rr = *(const u32x4*)((const bf16_t*)a.R + (size_t)m * a.ldr + n);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = pack2bf(__uint_as_float(v[e] << 16) + __uint_as_float(rr[e] << 16),
                                   __uint_as_float(v[e] & 0xffff0000u) + __uint_as_float(rr[e] & 0xffff0000u));
            }
            *(u32x4*)((bf16_t*)a.C + (size_t)m * a.ldc + n) = v;
        }
    }
}
template <int EPI>
__device__ __forceinline__ void epilogue_wave64(const GemmArgs& a, const f32x16 (&acc)[2][2], int mw, int nw, char* patch,
                                                int lane) {
    epilogue_wave<EPI, 2>(a, acc, mw, nw, patch, lane);
}
// vectorised epilogue is usable when whole 8-column groups exist and rows are 16-byte aligned
__host__ __device__ __forceinline__ bool epi_vec_ok(const GemmArgs& a, int epi) {
    if ((a.N & 7) != 0) return false;
    if (epi == EPI_BIAS_GATE_RES) return (a.ldx & 7) == 0 && (a.gate_stride & 7) == 0;
    if (epi == EPI_BIAS_ADD && (a.ldr & 7) != 0) return false;
    return (a.ldc & 7) == 0;
}

template <int EPI>
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
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    if (epi_vec_ok(a, EPI)) {  // the k-loop ended with a barrier: the operand stages are free
        epilogue_wave64<EPI>(a, acc, m0 + wm * 64, n0 + wn * 64, smem + wave * 8192, lane);
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
                if (n < a.N) epilogue4<bf16_t, EPI>(a, m, n, v);
            }
        }
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
template <int EPI>
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
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk][i], af[kk][j], acc[i][j], 0, 0, 0);
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
        epilogue_wave64<EPI>(a, acc, m0 + wm * 64, n0 + wn * 64, smem + wave * 8192, lane);
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
                if (n < a.N) epilogue4<bf16_t, EPI>(a, m, n, v);
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// 256 x 256 block tiles (gemm_bf16_w8, gemm_bf16_pp64).  A four-wave 128 x 128-wave-tile form, a ping-pong on K32 stages and a
// sixteen-wave ping-pong were measured and removed.
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
// of K) is ONE v_mfma_scale_f32_32x32x64_f8f6f4 per 32x32 block (64 cycles, twice the bf16 rate) whose 32-byte operand is the
// two 16-byte fragments a lane reads for its K half (chunks s*4 + hi*2 + {0, 1}); block scales are unit (E8M0 127), the
// per-token / per-channel scales are applied by the epilogue.
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x8v cat16(bf16x8 lo, bf16x8 hi) {
    const i32x4v a = __builtin_bit_cast(i32x4v, lo), b = __builtin_bit_cast(i32x4v, hi);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <int EPI, int ABL = 0, bool FP8 = false>
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
    const int GM = 4;
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
    auto reads = [&](int t, int s) {
        if (ABL == 6 && t > 0) return;
        const char* tA = smem + (t & 1) * 65536;
        const char* tW = tA + 32768;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) wf[kk][i] = lds_frag(tW, wn * 64 + i * 32 + fr, FP8 ? s * 4 + hi * 2 + kk : s * 4 + kk * 2 + hi);
#pragma unroll
            for (int j = 0; j < 4; ++j) af[kk][j] = lds_frag(tA, wm * 128 + j * 32 + fr, FP8 ? s * 4 + hi * 2 + kk : s * 4 + kk * 2 + hi);
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
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat16(wf[0][i], wf[1][i]), cat16(af[0][j], af[1][j]), acc[i][j], 0, 0, 0, 127, 0, 127);
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (ABL != 3 && ABL != 5) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk][i], af[kk][j], acc[i][j], 0, 0, 0);
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
        epilogue_wave<EPI, 4, FP8>(a, acc, m0 + wm * 128, n0 + wn * 64, patch, lane);
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
                if (n < a.N) epilogue4<bf16_t, EPI>(a, m, n, v);
            }
        }
}

template <int EPI>
static int launch_pp64_t(const GemmArgs& a, hipStream_t st) {
    const int tiles_m = (a.M + WBM - 1) / WBM, tiles_n = (a.N + WBN - 1) / WBN;
    S2V_TRY(ensure_lds_attr((const void*)gemm_bf16_pp64<EPI>, 131072));
#ifdef S2V_DIAG
    if (EPI == EPI_BIAS && a.ablate) {  // diagnostics only (tools/ablate_gemm.py)
        const void* fn = a.ablate == 1 ? (const void*)gemm_bf16_pp64<EPI_BIAS, 1> : a.ablate == 4 ? (const void*)gemm_bf16_pp64<EPI_BIAS, 4> : a.ablate == 5 ? (const void*)gemm_bf16_pp64<EPI_BIAS, 5> : a.ablate == 6 ? (const void*)gemm_bf16_pp64<EPI_BIAS, 6> : a.ablate == 7 ? (const void*)gemm_bf16_pp64<EPI_BIAS, 7> : (const void*)gemm_bf16_pp64<EPI_BIAS, 3>;
        S2V_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        void* args[] = {(void*)&a, (void*)&tiles_m, (void*)&tiles_n};
        S2V_CHECK_HIP(hipLaunchKernel(fn, dim3(tiles_m * tiles_n), dim3(512), args, 131072, st));
        return 0;
    }
#endif
    hipLaunchKernelGGL(gemm_bf16_pp64<EPI>, dim3(tiles_m * tiles_n), dim3(512), 131072, st, a, tiles_m, tiles_n);
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

template <int EPI>
static int launch_stag_t(const GemmArgs& a, hipStream_t st) {
    const int tiles_m = (a.M + RBM - 1) / RBM, tiles_n = (a.N + RBN - 1) / RBN;
    S2V_TRY(ensure_lds_attr((const void*)gemm_bf16_stag<EPI>, 3 * RSTAGE));
    hipLaunchKernelGGL(gemm_bf16_stag<EPI>, dim3(tiles_m * tiles_n), dim3(512), 3 * RSTAGE, st, a, tiles_m, tiles_n);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// 7 = 256x256x64 eight-wave ping-pong (the product schedule), 2 = staggered 256x128 ring, 0 = 128x128 double buffer; 7 falls back
// to 2 (then 0) when the shape does not fit its tiles.  libs2v_hip_diag.so (S2V_DIAG) can also select 5 = the 256x256 eight-wave
// lock-step ring and the compile-time ablations of gemm_bf16_pp64 (tools/ablate_gemm.py, tools/stall_pp64.py).
#ifdef S2V_DIAG
int g_gemm_ablate = 0;
int g_gemm_impl = 7;
extern "C" __attribute__((visibility("default"))) int s2v_set_gemm_impl(int impl) { g_gemm_impl = impl & 0xff; g_gemm_ablate = impl >> 8; return 0; }
#else
static constexpr int g_gemm_ablate = 0;
static constexpr int g_gemm_impl = 7;
#endif

template <int EPI>
static int launch_fp8_t(const GemmArgs& a, hipStream_t st) {
    const int tiles_m = (a.M + WBM - 1) / WBM, tiles_n = (a.N + WBN - 1) / WBN;
    S2V_TRY(ensure_lds_attr((const void*)gemm_bf16_pp64<EPI, 0, true>, 131072));
    hipLaunchKernelGGL((gemm_bf16_pp64<EPI, 0, true>), dim3(tiles_m * tiles_n), dim3(512), 131072, st, a, tiles_m, tiles_n);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_gemm_fp8(const GemmArgs& a, int epi, hipStream_t st) {
    S2V_REQUIRE(!a.conv && a.a_scale && a.w_scale, "gemm_fp8: plain mode with both scale vectors only");
    S2V_REQUIRE(a.K % 128 == 0 && a.lda % 16 == 0 && a.ldw % 16 == 0, "gemm_fp8: K must be a multiple of 128, rows 16-byte aligned");
    S2V_REQUIRE(a.a_rows_padded >= ((a.M + WBM - 1) / WBM) * WBM && a.w_rows_padded >= ((a.N + WBN - 1) / WBN) * WBN,
                "gemm_fp8: operands must be padded to whole 256-row tiles");
    S2V_REQUIRE(epi_vec_ok(a, epi), "gemm_fp8: output rows must be 16-byte aligned and N a multiple of 8");
    switch (epi) {
        case EPI_BIAS: return launch_fp8_t<EPI_BIAS>(a, st);
        case EPI_BIAS_GELU: return launch_fp8_t<EPI_BIAS_GELU>(a, st);
        case EPI_BIAS_GATE_RES: return launch_fp8_t<EPI_BIAS_GATE_RES>(a, st);
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
    if (a.m_begin > 0) {  // row tail of a split GEMM: the 128 x 128 kernel on rows [m_begin, M)
        S2V_REQUIRE(!a.conv && a.K % BK == 0 && a.lda % 8 == 0 && a.ldw % 8 == 0, "gemm_bf16: bad tail launch");
        const int tiles_m = (a.M - a.m_begin + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
        const size_t shmem = 4 * TILE_BYTES;
        const dim3 grid(tiles_m * tiles_n);
        switch (epi) {
            case EPI_BIAS: hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS>, grid, dim3(256), shmem, st, a, tiles_m, tiles_n); break;
            case EPI_BIAS_GELU: hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS_GELU>, grid, dim3(256), shmem, st, a, tiles_m, tiles_n); break;
            case EPI_BIAS_GATE_RES: hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS_GATE_RES>, grid, dim3(256), shmem, st, a, tiles_m, tiles_n); break;
            case EPI_BIAS_ADD: hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS_ADD>, grid, dim3(256), shmem, st, a, tiles_m, tiles_n); break;
            default: return s2v_fail(__FILE__, __LINE__, "gemm_bf16: bad epilogue", -1);
        }
        S2V_CHECK_HIP(hipGetLastError());
        return 0;
    }
    S2V_REQUIRE(a.K % BK == 0, "gemm_bf16: K must be a multiple of 64");
    if (g_gemm_impl == 7 && w_tile_ok(a) && (a.conv || a.a_rows_padded >= ((a.M + WBM - 1) / WBM) * WBM)) {
        S2V_REQUIRE((a.conv ? a.cin % 64 == 0 : a.lda % 8 == 0) && a.ldw % 8 == 0, "gemm_bf16: bad leading dims");
        switch (epi) {
            case EPI_BIAS: return launch_pp64_t<EPI_BIAS>(a, st);
            case EPI_BIAS_GELU: return launch_pp64_t<EPI_BIAS_GELU>(a, st);
            case EPI_BIAS_GATE_RES: return launch_pp64_t<EPI_BIAS_GATE_RES>(a, st);
            case EPI_BIAS_ADD: return launch_pp64_t<EPI_BIAS_ADD>(a, st);
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
    if ((g_gemm_impl == 2 || g_gemm_impl >= 4) && (a.conv || a.a_rows_padded >= ((a.M + RBM - 1) / RBM) * RBM)) {
        S2V_REQUIRE((a.conv ? a.cin % 64 == 0 : a.lda % 8 == 0) && a.ldw % 8 == 0, "gemm_bf16: bad leading dims");
        switch (epi) {
            case EPI_BIAS: return launch_stag_t<EPI_BIAS>(a, st);
            case EPI_BIAS_GELU: return launch_stag_t<EPI_BIAS_GELU>(a, st);
            case EPI_BIAS_GATE_RES: return launch_stag_t<EPI_BIAS_GATE_RES>(a, st);
            case EPI_BIAS_ADD: return launch_stag_t<EPI_BIAS_ADD>(a, st);
            default: return s2v_fail(__FILE__, __LINE__, "gemm_bf16: bad epilogue", -1);
        }
    }
    S2V_REQUIRE((a.conv ? a.cin % 64 == 0 : a.lda % 8 == 0) && a.ldw % 8 == 0, "gemm_bf16: bad leading dims");
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int grid = tiles_m * tiles_n;
    const size_t shmem = 4 * TILE_BYTES;
    switch (epi) {
        case EPI_BIAS:
            hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS>, dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            break;
        case EPI_BIAS_GELU:
            hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS_GELU>, dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            break;
        case EPI_BIAS_GATE_RES:
            hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS_GATE_RES>, dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            break;
        case EPI_BIAS_ADD:
            hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS_ADD>, dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
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
    S2V_REQUIRE((a.M + 63) / 64 <= 65535, "gemm_simple: M too large for the generic kernel");
    return dtype == S2V_BF16 ? launch_simple_t<bf16_t>(a, epi, st) : launch_simple_t<float>(a, epi, st);
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
