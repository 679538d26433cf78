// Joint [text | ref-image | video] self-attention, head_dim 64, no mask
// (replaces F.scaled_dot_product_attention at attention_processor.py:2083-2087).
//
//  attn_bf16_k : flash-attention forward on v_mfma_f32_32x32x16_bf16.
//     block = NW (8; 4 as diagnostic) waves x 32 query rows; KV tile = 64 keys; K tile [64 kv][64 d] and V^T tile [64 d][64 kv] are
//     staged with global_load_lds into double-buffered, XOR-swizzled LDS (same image as gemm.hip).
//     QK^T is issued swapped (S^T = K . Q^T) so every lane owns ONE query row: row max / row sum / rescale are
//     lane-local (one cross-half exchange per tile).  P^T feeds the PV MFMA straight from the S^T accumulator
//     registers: the MFMA k-slot <-> key assignment is free, so V^T is stored in HBM with the keys of every
//     16-group permuted [0-3, 8-11, 4-7, 12-15] (done by qk_norm_rope) and no lane exchange is needed.
//     O^T = V^T . P^T accumulates with the query again on the lane axis.
//     Softmax in the exp2 domain: Q is pre-multiplied by scale * log2(e) and the S^T accumulators start at -m_run (a
//     loop-carried register block), so p = exp2(MFMA result) with no per-score fma (measured +3.5 % at C3).
//  attn_simple_k<T> : one wave per query row, fp32 math, any dtype (CPU-reference-parity mode / cross-check).
#define S2V_HOST
#include "common.h"
#include "kernels.h"

#define KV_TILE 64
#define Q_BLOCK 128
#define ATT_TILE_BYTES (64 * 64 * 2)  // 8 KiB

template <int NT>
__device__ __forceinline__ void stage64(const bf16_t* __restrict__ g, size_t ld, char* lds, int tid) {
    // 64 rows x 128 B; 512 chunks of 16 B; 512 / NT rounds of NT threads.  The address is written as
    // (wave-uniform tile base) + zext(32-bit lane offset) so the LDS-DMA can take the saddr + voffset form
    const int wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 512 / NT; ++i) {
        const int gi = i * NT + tid;
        const int row = gi >> 3;
        const int cp = gi & 7;
        const int c = cp ^ ((row >> 1) & 7);
        const unsigned off = (unsigned)(2 * ((unsigned)row * (unsigned)ld + c * 8));
        const char* src = (const char*)g + (size_t)off;
        char* dst = lds + (i * NT + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}
template <int NW>
__device__ __forceinline__ void stage_kv(const bf16_t* __restrict__ kg, size_t ldk, const bf16_t* __restrict__ vg, size_t ldv,
                                         char* lds, int tid) {
    stage64<NW * 64>(kg, ldk, lds, tid);
    stage64<NW * 64>(vg, ldv, lds + ATT_TILE_BYTES, tid);
}
__device__ __forceinline__ bf16x8 frag64(const char* tile, int row, int cl) {
    return *(const bf16x8*)(tile + row * 128 + ((cl ^ ((row >> 1) & 7)) << 4));
}

// ABL (diagnostics, tools/ablate_attn.py; results are wrong on purpose): 1 = no exp2, 2 = no row-sum adds, 3 = no row max
template <int ABL, int NW = 8>
__global__ __launch_bounds__(NW * 64, 2) void attn_bf16_k(const AttnArgs a, int nqb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][K tile | VT tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, hi = lane >> 5;

    // XCD-aware order: all q-blocks of one (b,h) run on one XCD so its K/V stay in that XCD's L2
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qq = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    const int bh = wg / nqb, qb = wg - bh * nqb;
    const int b = bh / a.H, h = bh - b * a.H;
    const int D = a.H * 64;

    const bf16_t* qkv = (const bf16_t*)a.qkv + (size_t)b * a.Ntok * a.ld_qkv;
    const bf16_t* Kg = qkv + D + h * 64;
    const bf16_t* VTg = (const bf16_t*)a.vt + (size_t)(b * a.H + h) * 64 * a.ntok_pad;

    // Q fragments (B operand of S^T = K.Q^T): lane (q = fr, hi) holds Q[q][16kk + 8hi .. +8], pre-multiplied by
    // scale * log2(e) and rounded to bf16 once (the reference's math path rounds its scaled q and k to bf16 as well), so the
    // MFMA result is already the exp2 argument
    const int q_row = qb * (NW * 32) + wave * 32 + fr;
    const int q_ld = min(q_row, a.Ntok - 1);
    const float c0 = a.scale * 1.4426950408889634f;
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        qf[kk] = *(const bf16x8*)(qkv + (size_t)q_ld * a.ld_qkv + h * 64 + kk * 16 + hi * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[kk][e] = (__bf16)((float)qf[kk][e] * c0);
    }

    f32x16 ot[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) ot[i][e] = 0.f;
    // the S^T accumulators START at -m_run: this 16-register block is loop-carried and only rewritten when a row maximum
    // grows, so in the steady state a score costs exp2 + add (+ half a cvt_pk) and no fma
    f32x16 negm16;
#pragma unroll
    for (int e = 0; e < 16; ++e) negm16[e] = 0.f;
    float m_run = 0.f, l_run = 0.f;

    const int nt = (a.Ntok + KV_TILE - 1) / KV_TILE;
    stage_kv<NW>(Kg, a.ld_qkv, VTg, a.ntok_pad, smem, tid);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const int kv0 = t * KV_TILE;
        if (t + 1 < nt) {
            char* nb = smem + (cur ^ 1) * 2 * ATT_TILE_BYTES;
            stage_kv<NW>(Kg + (size_t)(kv0 + KV_TILE) * a.ld_qkv, a.ld_qkv, VTg + kv0 + KV_TILE, a.ntok_pad, nb, tid);
        }
        const char* tK = smem + cur * 2 * ATT_TILE_BYTES;
        const char* tV = tK + ATT_TILE_BYTES;

        // S^T[kv][q] - m_run for two 32-key blocks
        f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag64(tK, kb * 32 + fr, hi), qf[0], negm16, 0, 0, 0);
#pragma unroll
            for (int kk = 1; kk < 4; ++kk)
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag64(tK, kb * 32 + fr, kk * 2 + hi), qf[kk], st[kb], 0, 0, 0);
        }
        if (kv0 + KV_TILE > a.Ntok) {  // tail tile: mask keys >= Ntok
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kv = kv0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                    if (kv >= a.Ntok) st[kb][e] = -INFINITY;
                }
        }
        float mx = st[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (ABL != 3) mx = fmaxf(mx, st[kb][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));  // (row max of this tile) - m_run
        // The first tile adopts its own maximum (m_run starts at 0); later tiles only raise it, and the branch is skipped
        // exactly when no row of the wave grew (alpha == 1 for every lane).
        if (t == 0 || __any(mx > 0.f)) {
            const float d = (t == 0) ? mx : fmaxf(mx, 0.f);
            // t == 0: O and l are still zero, and exp2(-d) would overflow to inf (0 * inf = NaN) for scores below -128
            const float alpha = (t == 0) ? 1.0f : __builtin_amdgcn_exp2f(-d);
            m_run += d;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) ot[i][e] *= alpha;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) st[kb][e] -= d;
#pragma unroll
            for (int e = 0; e < 16; ++e) negm16[e] = -m_run;
        }
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = (ABL == 1) ? st[kb][e] : __builtin_amdgcn_exp2f(st[kb][e]);
                st[kb][e] = p;
                if (ABL != 2) psum += p;
            }
        l_run += psum;

        // O^T[d][q] += V^T[d][kv] . P^T[kv][q]; k-step s covers keys 16s..16s+15 of the tile
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kb = s >> 1, r0 = (s & 1) * 8;
            bf16x8 pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = (__bf16)st[kb][r0 + e];
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                bf16x8 vf = frag64(tV, db * 32 + fr, s * 2 + hi);
                ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, ot[db], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < a.Ntok) {
        bf16_t* o = (bf16_t*)a.out + (size_t)(b * a.Ntok + q_row) * a.ld_out + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int d = db * 32 + 8 * rq + 4 * hi;
                u32x2 p;
                p.x = pack2bf(ot[db][rq * 4 + 0] * inv, ot[db][rq * 4 + 1] * inv);
                p.y = pack2bf(ot[db][rq * 4 + 2] * inv, ot[db][rq * 4 + 3] * inv);
                *(u32x2*)(o + d) = p;
            }
    }
}

int g_attn_variant = 0;  // diagnostics (tools/microbench.py): timing-only ablations of the softmax VALU work
extern "C" int s2v_set_attn_variant(int v) { g_attn_variant = v; return 0; }

int launch_attn_bf16(const AttnArgs& a, hipStream_t st) {
    S2V_REQUIRE(a.vt != nullptr, "attn_bf16: V^T buffer missing");
    S2V_REQUIRE(a.ld_qkv % 8 == 0 && a.ntok_pad % 64 == 0, "attn_bf16: bad leading dims");
    S2V_REQUIRE(a.ntok_pad >= ((a.Ntok + 63) / 64) * 64, "attn_bf16: ntok_pad too small");
    const int nqb = (a.Ntok + Q_BLOCK - 1) / Q_BLOCK;
    const int grid = nqb * a.B * a.H;
    // default: eight waves (256 query rows) per block -- each thread moves one K and one V^T piece per KV tile, half the
    // LDS-DMA issue work per wave of the four-wave form (measured +2 % at C3); variants are diagnostics (tools/ablate_attn.py)
    const int nqb8 = (a.Ntok + 255) / 256;
    const dim3 g8(nqb8 * a.B * a.H), b8(512);
    switch (g_attn_variant) {
        case 1: hipLaunchKernelGGL((attn_bf16_k<1, 8>), g8, b8, 4 * ATT_TILE_BYTES, st, a, nqb8); break;
        case 2: hipLaunchKernelGGL((attn_bf16_k<2, 8>), g8, b8, 4 * ATT_TILE_BYTES, st, a, nqb8); break;
        case 3: hipLaunchKernelGGL((attn_bf16_k<3, 8>), g8, b8, 4 * ATT_TILE_BYTES, st, a, nqb8); break;
        case 4: hipLaunchKernelGGL((attn_bf16_k<0, 4>), dim3(grid), dim3(256), 4 * ATT_TILE_BYTES, st, a, nqb); break;
        default: hipLaunchKernelGGL((attn_bf16_k<0, 8>), g8, b8, 4 * ATT_TILE_BYTES, st, a, nqb8); break;
    }
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_simple_k(const AttnArgs a) {
    __shared__ float sq[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
    const int D = a.H * 64;
    const bool active = q < a.Ntok;
    const T* base = (const T*)a.qkv + (size_t)b * a.Ntok * a.ld_qkv;
    const int ql = active ? q : a.Ntok - 1;
    sq[wave][lane] = ET<T>::ld(base + (size_t)ql * a.ld_qkv + h * 64 + lane) * a.scale;
    __syncthreads();
    float m = -INFINITY, l = 0.f, o = 0.f;
    for (int kv0 = 0; kv0 < a.Ntok; kv0 += 64) {
        const int kv = kv0 + lane;
        float s = -INFINITY;
        if (kv < a.Ntok) {
            const T* kp = base + (size_t)kv * a.ld_qkv + D + h * 64;
            float acc = 0.f;
#pragma unroll 8
            for (int d = 0; d < 64; ++d) acc = fmaf(sq[wave][d], ET<T>::ld(kp + d), acc);
            s = acc;
        }
        const float mx = wave_max(s);
        const float mn = fmaxf(m, mx);
        const float alpha = expf(m - mn);
        const float p = expf(s - mn);
        l = l * alpha + wave_sum(p);
        o *= alpha;
        m = mn;
        const int nv = min(64, a.Ntok - kv0);
        for (int j = 0; j < nv; ++j) {
            const float pj = __shfl(p, j, 64);
            o = fmaf(pj, ET<T>::ld(base + (size_t)(kv0 + j) * a.ld_qkv + 2 * D + h * 64 + lane), o);
        }
    }
    if (active) ET<T>::st((T*)a.out + (size_t)(b * a.Ntok + q) * a.ld_out + h * 64 + lane, o / l);
}

int launch_attn_simple(const AttnArgs& a, int dtype, hipStream_t st) {
    dim3 grid((a.Ntok + 3) / 4, a.H, a.B);
    if (dtype == S2V_BF16)
        hipLaunchKernelGGL(attn_simple_k<bf16_t>, grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(attn_simple_k<float>, grid, dim3(256), 0, st, a);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}
