// Joint [text | ref-image | video] self-attention, head_dim 64, no mask
// (replaces F.scaled_dot_product_attention at attention_processor.py:2083-2087).
//
//  The bf16 product kernel is attn_q4 (attention_q4.hip: four waves, one per SIMD, generated asm): launch_attn_bf16 below dispatches to
//     it -- one workgroup per 256-row q-block (op-level launches) or one persistent workgroup per CU pulling q-blocks from per-XCD
//     queues (the engine's launches).  What follows in this file is compiled into the DIAGNOSTICS library only, as A/B references:
//  attn_pp_k / attn_pp_persist_k : the round-2 product kernel, flash-attention forward on v_mfma_f32_32x32x16_bf16, eight waves in a
//     two-group ping-pong (below); same launch forms; 5-6 % slower than attn_q4 on the same box (profiles/r03_attn_harness.txt).
//     block = 8 waves x 32 query rows; KV tile = 64 keys; K tile [64 kv][64 d] and V^T tile [64 d][64 kv] are staged with
//     LDS-DMA (global_load_lds) into a ring of XOR-swizzled LDS slots (same image as gemm.hip).
//     QK^T is issued swapped (S^T = K . Q^T) so every lane owns ONE query row: row max / row sum / rescale are
//     lane-local (one cross-half exchange per tile).  P^T feeds the PV MFMA straight from the S^T accumulator
//     registers: the MFMA k-slot <-> key assignment is free, so V^T is stored in HBM with the keys of every
//     16-group permuted [0-3, 8-11, 4-7, 12-15] (done by qk_norm_rope) and no lane exchange is needed.
//     O^T = V^T . P^T accumulates with the query again on the lane axis.
//     Softmax in the exp2 domain: Q is pre-multiplied by scale * log2(e) and the S^T accumulators start at -m_run (a
//     loop-carried register block), so p = exp2(MFMA result) with no per-score fma.
//  attn_bf16_k (S2V_DIAG builds only): the round-1 lock-step kernel with a per-tile row maximum, kept as the A/B reference.
//  attn_simple_k<T> : one wave per query row, fp32 math, any dtype (CPU-reference-parity mode / cross-check).
#define S2V_HOST
#include "common.h"
#include "kernels.h"
#include <type_traits>

#define KV_TILE 64
#define Q_BLOCK 128
#define ATT_TILE_BYTES (64 * 64 * 2)  // 8 KiB
#ifndef ATTN_PP_MAX_TOKENS
#define ATTN_PP_MAX_TOKENS 4608  // launch_attn_bf16: sequences up to this length run attn_pp, longer ones attn_q4 (profiles/r03_attn_short_sequences.txt:
                                 // attn_pp 9 % ahead at 4000 tokens, attn_q4 2 % ahead at 6000, 6 % at 19126)
#endif

// v_max3_f32 written out so that the max tree keeps the order it is given (the scores are never NaN: no canonicalisation)
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// ---- the lock-step kernel: A/B reference in bf16 (diagnostics library), the fp16 model dtype's attention in the product (T16 = f16_t) ----
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

// round-1 lock-step kernel (A/B reference of tools/attn_harness); ABL is unused.
// T16 = f16_t (round 5): the attention of the fp16 model dtype (src/inference.py:191) -- q, k, V^T and P in fp16 on v_mfma_f32_32x32x16_f16, output
// fp16.  This kernel takes a true per-tile row maximum, so p <= 1 and nothing can leave the fp16 range; staging, swizzle and fragment reads move
// 16-bit elements whatever they encode.  ~1 PFLOP/s at C3 in bf16 (9.2-9.4 ms, profiles/r02_attn_harness.txt): not the tuned asm kernel, but an
// order of magnitude above the fp32-pipe kernel the fp16 dtype started on.
template <int ABL, int NW = 8, typename T16 = bf16_t>
__global__ __launch_bounds__(NW * 64, 2) void attn_bf16_k(const AttnArgs a, int nqb) {
    constexpr bool H16 = std::is_same<T16, f16_t>::value;
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
        if constexpr (H16) {
            f16x8 qh = __builtin_bit_cast(f16x8, qf[kk]);
#pragma unroll
            for (int e = 0; e < 8; ++e) qh[e] = (_Float16)((float)qh[e] * c0);
            qf[kk] = __builtin_bit_cast(bf16x8, qh);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[kk][e] = (__bf16)((float)qf[kk][e] * c0);
        }
    }
    auto mfma16 = [](bf16x8 x, bf16x8 y, f32x16 acc) __attribute__((always_inline)) {
        if constexpr (H16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), acc, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
    };

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
            st[kb] = mfma16(frag64(tK, kb * 32 + fr, hi), qf[0], negm16);
#pragma unroll
            for (int kk = 1; kk < 4; ++kk) st[kb] = mfma16(frag64(tK, kb * 32 + fr, kk * 2 + hi), qf[kk], st[kb]);
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
            if constexpr (H16) {
                f16x8 ph;
#pragma unroll
                for (int e = 0; e < 8; ++e) ph[e] = (_Float16)st[kb][r0 + e];
                pf = __builtin_bit_cast(bf16x8, ph);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (__bf16)st[kb][r0 + e];
            }
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                bf16x8 vf = frag64(tV, db * 32 + fr, s * 2 + hi);
                ot[db] = mfma16(vf, pf, ot[db]);
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
                if constexpr (H16) {
                    p.x = pack2h(ot[db][rq * 4 + 0] * inv, ot[db][rq * 4 + 1] * inv);
                    p.y = pack2h(ot[db][rq * 4 + 2] * inv, ot[db][rq * 4 + 3] * inv);
                } else {
                    p.x = pack2bf(ot[db][rq * 4 + 0] * inv, ot[db][rq * 4 + 1] * inv);
                    p.y = pack2bf(ot[db][rq * 4 + 2] * inv, ot[db][rq * 4 + 3] * inv);
                }
                *(u32x2*)(o + d) = p;
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// attn_pp_k: 8-wave PING-PONG flash attention.  Waves w and w + 4 share a SIMD; group A = waves 0-3, group B = waves 4-7, one
// barrier interval apart.  Every interval one group is in its MATRIX segment M(t+1) (QK^T of tile t+1, P.V of tile t) while
// its SIMD partner is in its SOFTMAX segment S(t).
//
// What shapes the split (tools/probes/valu_rate.hip, MI355X): a wave issues plain VALU every 4.9 cycles and v_exp_f32 every 8.9
// on its own, but beside a partner that streams v_mfma_f32_32x32x16_bf16 back to back the SIMD grants the partner one VALU
// per 16 cycles (exp2 or not, any priority, either wave older); (exp2, exp2, add, add) per MFMA is the densest partner stream
// that still fits (33 cycles per MFMA), and the MFMA wave can place two plain VALU of its own behind each MFMA.  Head
// dimension 64 needs 32 exp2 per 16 MFMA, so the softmax is cut down to what those slots hold:
//   * NO per-tile row maximum.  The running maximum is adopted from the first tile; afterwards a tile only checks its row
//     SUMS (psum > 2^13, wave-wide vote) and takes the slow path -- true row max, rescale of O and l, exp2 again -- when some
//     p exceeded ~2^8 (deferred maximum: p stays far inside the fp32 / bf16 range, the result is the same up to rounding);
//   * S(t): 32 exp2 + 32 row-sum adds as 16 (exp2, exp2, add, add) groups, the vote, the wait for this wave's LDS-DMA;
//   * M(t+1): per MFMA one fragment read; the 16 v_cvt_pk (bf16 packing of P) behind the 8 QK^T MFMAs, the K fragments of
//     tile t+2 behind the 8 P.V MFMAs, and the wave's two LDS-DMA pieces AFTER its last MFMA (an LDS-DMA instruction keeps its
//     wave from issuing for 60-100 cycles: between MFMAs that is a bubble of the matrix pipe, after them it is barrier slack).
// Phases (one per barrier): A runs M(t) in phase 2t and S(t) in phase 2t+1, B one phase later.
// LDS ring of 4 slots x [K tile 8 KiB | V^T tile 8 KiB]; in M(t+1) every wave issues ONE K piece (8 rows) of tile t+4 and ONE
// V^T piece of tile t+2 and waits for them (vmcnt(0)) at the end of S(t+1):
//   RAW  K(t+4): A's pieces land by the end of phase 2t+3, B's by 2t+4; first read (fragment prefetch in M(t+3)) in phase 2t+6.
//        V(t+2): first read in M(t+3) = phase 2t+6.
//   WAR  K(t+4) replaces K(t), last read in phase 2t-1 (B's M(t-1) prefetch; K(0) and K(1): M(0), phases 0-1); first write in
//        phase 2t+2.  V(t+2) replaces V(t-2), last read by B's M(t-1) in phase 2t-1.
// Measured at C3 (B 2, H 48, N 19126; tools/attn_harness): 8.1 ms against 9.2-9.4 for the lock-step kernel; variants that
// lost the A/B (kept out of the tree): row sums with v_pk_add_f32 (+5 %), no s_setprio in M (+6 %), LDS-DMA between the P.V
// MFMAs (+3 %) or in S (+4 %), dedicated loader waves (three waves per SIMD force 168 registers and drop the fragment
// prefetch: +2 %), bounded scores with no maximum at all and the row sums on the matrix pipe (20 MFMA per tile: -1.5 %, not
// worth its precondition).
// (attn_pp below: the bf16 short-sequence kernel, see launch_attn_bf16)
__device__ long long g_attn_dbg[64];  // ACCT: per-wave s_memtime totals of block 100
__device__ long long g_attn_blk[2 * 8192];  // ACCT: s_memrealtime at entry / exit of every workgroup (timeline of a launch)
#ifdef S2V_DIAG
extern "C" __attribute__((visibility("default"))) int s2v_attn_debug_read_blocks(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_blk), sizeof(long long) * 2 * 8192) == hipSuccess ? 0 : -1; }
extern "C" __attribute__((visibility("default"))) int s2v_attn_debug_read(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_dbg), sizeof(long long) * 64) == hipSuccess ? 0 : -1; }
#endif
// one work item = the 256 query rows (wg % nqb) of (sample, head) wg / nqb; all eight waves enter and leave it together
template <bool ACCT>
__device__ __forceinline__ void attn_pp_item(const AttnArgs& a, int nqb, int wg, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int fr = lane & 31, hi = lane >> 5;
    if (ACCT && tid == 0 && wg < 8192) g_attn_blk[2 * wg] = (long long)__builtin_amdgcn_s_memrealtime();
    const int bh = wg / nqb, qb = wg - bh * nqb;
    const int b = bh / a.H, h = bh - b * a.H;
    const int D = a.H * 64;

    const bf16_t* qkv = (const bf16_t*)a.qkv + (size_t)b * a.Ntok * a.ld_qkv;
    const char* Kg = (const char*)(qkv + D + h * 64);
    const char* VTg = (const char*)((const bf16_t*)a.vt + (size_t)(b * a.H + h) * 64 * a.ntok_pad);
    const int nt = (a.Ntok + KV_TILE - 1) / KV_TILE;

    // staging: this wave's piece = rows wave*8 .. +7 of a tile, lane = (row, 16-B chunk), chunk XOR on the SOURCE address;
    // global address = wave-uniform tile base (SGPR pair) + per-lane 32-bit offset
    const int srow = wave * 8 + (lane >> 3);
    const int sc = (lane & 7) ^ ((srow >> 1) & 7);
    const unsigned offK = (unsigned)(2 * (srow * a.ld_qkv + sc * 8));
    const unsigned offV = (unsigned)(2 * (srow * a.ntok_pad + sc * 8));
    const size_t k_tile_stride = (size_t)KV_TILE * a.ld_qkv * 2;
    auto dma_k = [&](int t) {  // tiles past the end are clamped (re-staged into a dead slot) so that every wave issues the same count
        const int tc = min(t, nt - 1);
        glds16_saddr(Kg + (size_t)tc * k_tile_stride, offK, smem + (t & 3) * 16384 + wave * 1024);
    };
    auto dma_v = [&](int t) {
        const int tc = min(t, nt - 1);
        glds16_saddr(VTg + (size_t)tc * (KV_TILE * 2), offV, smem + (t & 3) * 16384 + 8192 + wave * 1024);
    };
    dma_k(0); dma_v(0); dma_k(1); dma_v(1); dma_k(2); dma_k(3);

    // Q fragments (B operand of S^T = K.Q^T): lane (q = fr, hi) holds Q[q][16kk + 8hi .. +8], pre-multiplied by
    // scale * log2(e) and rounded to bf16 once (the reference's math path rounds its scaled q and k to bf16 as well), so the
    // MFMA result is already the exp2 argument
    const int q_row = qb * 256 + wave * 32 + fr;
    const int q_ld = min(q_row, a.Ntok - 1);
    const float c0 = a.scale * 1.4426950408889634f;
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        qf[kk] = *(const bf16x8*)(qkv + (size_t)q_ld * a.ld_qkv + h * 64 + kk * 16 + hi * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[kk][e] = (__bf16)((float)qf[kk][e] * c0);
    }

    // the S^T accumulators START at -m_run: this 16-register block is loop-carried and only rewritten on the slow path
    f32x16 ot[2], st[2], negm16;
#pragma unroll
    for (int e = 0; e < 16; ++e) { ot[0][e] = 0.f; ot[1][e] = 0.f; negm16[e] = 0.f; }
    float l_run = 0.f;
    bf16x8 pk[4], kf[8], vf[8];
    f32x16 pp[2];  // exp2 results of the tile in flight between S(t) and M(t+1)

    // fragment byte offsets inside a tile: K (kk, kb) and V^T (s, db) use the same (row, chunk) pattern
    int foff[4][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int row = kb * 32 + fr, cl = kk * 2 + hi;
            foff[kk][kb] = row * 128 + ((cl ^ ((row >> 1) & 7)) << 4);
        }
    auto k_frag = [&](int t, int i) -> bf16x8 { return *(const bf16x8*)(smem + (t & 3) * 16384 + foff[i >> 1][i & 1]); };
    auto v_frag = [&](int t, int i) -> bf16x8 { return *(const bf16x8*)(smem + (t & 3) * 16384 + 8192 + foff[i >> 1][i & 1]); };
    auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp) __builtin_amdgcn_s_barrier();  // group B runs one phase behind
    fence();
    // M(0): QK^T of tile 0 and the K fragments of tile 1
#pragma unroll
    for (int i = 0; i < 8; ++i) kf[i] = k_frag(0, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i], qf[i >> 1], i < 2 ? negm16 : st[i & 1], 0, 0, 0);
    fence();
#pragma unroll
    for (int i = 0; i < 8; ++i) kf[i] = k_frag(min(1, nt - 1), i);
    fence();
    __builtin_amdgcn_s_barrier();
    fence();

    long long tacc[6] = {0, 0, 0, 0, 0, 0};
    auto now = [&]() -> long long { return ACCT ? (long long)__builtin_amdgcn_s_memtime() : 0; };
    const long long tl0 = now();
    for (int t = 0; t < nt; ++t) {
        // ---- softmax segment S(t)
        const long long u0 = now();
        const int kv0 = t * KV_TILE;
        if (kv0 + KV_TILE > a.Ntok) {  // tail tile: mask keys >= Ntok
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kv = kv0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                    if (kv >= a.Ntok) st[kb][e] = -INFINITY;
                }
        }
        float ps0 = 0.f, ps1 = 0.f;
        // 16 groups of (exp2, exp2, add, add): the adds of a group consume the exp2 results of the group before.  Written as
        // asm so that neither the SLP vectoriser (it packs the adds) nor the scheduler reorders the stream.
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int kb = g >> 3, e = (g & 7) * 2;
            asm volatile("v_exp_f32 %0, %1" : "=v"(pp[kb][e]) : "v"(st[kb][e]));
            asm volatile("v_exp_f32 %0, %1" : "=v"(pp[kb][e + 1]) : "v"(st[kb][e + 1]));
            if (g > 0) {
                const int gp = g - 1, kbp = gp >> 3, ep = (gp & 7) * 2;
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(ps0) : "v"(pp[kbp][ep]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(ps1) : "v"(pp[kbp][ep + 1]));
            }
        }
        ps0 += pp[1][14];
        ps1 += pp[1][15];
        float psum = ps0 + ps1;
        // Deferred maximum: the slow path runs for the first tile (adopts its maximum; its first exp2 pass may overflow, the
        // vote is NaN-safe) and whenever some p of the wave grew past ~2^8: m_run rises to the true maximum, everything at
        // the old scale is rescaled once, and the tile's exp2 is redone at the new scale.
        if (t == 0 || __any(!(psum <= 18446744073709551616.f))) {
            float m0 = max3f(st[0][0], st[0][1], st[0][2]), m1 = max3f(st[1][0], st[1][1], st[1][2]);
#pragma unroll
            for (int e = 3; e < 15; e += 2) { m0 = max3f(m0, st[0][e], st[0][e + 1]); m1 = max3f(m1, st[1][e], st[1][e + 1]); }
            const float mxp = max3f(m0, m1, fmaxf(st[0][15], st[1][15]));
            // cross-half exchange without LDS: sw[0] = lower half's value in both halves, sw[1] = upper half's
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mxp), __float_as_uint(mxp), false, false);
            const float mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            const float d = (t == 0) ? mx : fmaxf(mx, 0.f);
            // t == 0: O and l are still zero, and exp2(-d) would overflow to inf (0 * inf = NaN) for scores below -128
            const float alpha = (t == 0) ? 1.0f : __builtin_amdgcn_exp2f(-d);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) ot[i][e] *= alpha;
#pragma unroll
            for (int e = 0; e < 16; ++e) negm16[e] -= d;  // -(m + d) == (-m) - d exactly
            psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    pp[kb][e] = __builtin_amdgcn_exp2f(st[kb][e] - d);
                    psum += pp[kb][e];
                }
        }
        l_run += psum;
        asm volatile("" : "+v"(l_run));
        fence();
        const long long u1 = now();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long u2 = now();
        __builtin_amdgcn_s_barrier();
        const long long u3 = now();
        fence();

        // ---- matrix segment M(t+1), in issue order, one fenced step per MFMA.
        // QK^T of tile t+1 (clamped: the last one is computed twice and dropped): K fragments were prefetched; behind each MFMA
        // the V^T fragment of tile t for the matching P.V step and two v_cvt_pk of P.
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            st[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i], qf[i >> 1], i < 2 ? negm16 : st[i & 1], 0, 0, 0);
            vf[i] = v_frag(t, i);
            // P values 4i .. 4i+3: pk[s] element e = pp[s >> 1][(s & 1) * 8 + e]
            const int s = i >> 1, kb = s >> 1, r0 = (s & 1) * 8 + (i & 1) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[s][(i & 1) * 4 + e] = (__bf16)pp[kb][r0 + e];
            fence();
        }
        // P.V of tile t; behind the MFMAs the K fragments of tile t+2
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ot[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i], pk[i >> 1], ot[i & 1], 0, 0, 0);
            kf[i] = k_frag(min(t + 2, nt - 1), i);
            fence();
        }
        dma_k(t + 4);
        dma_v(t + 2);
        __builtin_amdgcn_s_setprio(0);
        fence();
        const long long u4 = now();
        __builtin_amdgcn_s_barrier();
        const long long u5 = now();
        fence();
        if (ACCT) {
            tacc[0] += u1 - u0;  // softmax segment
            tacc[1] += u2 - u1;  // vmcnt(0)
            tacc[2] += u3 - u2;  // barrier after S
            tacc[3] += u4 - u3;  // matrix segment
            tacc[4] += u5 - u4;  // barrier after M
        }
    }
    if (ACCT) {
        tacc[5] = now() - tl0;
        if (wg == 100 && lane == 0)
            for (int e = 0; e < 6; ++e) g_attn_dbg[wave * 8 + e] = tacc[e];
    }
    if (!grp) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    // A lane holds dims 8g + 4hi .. + 3 of its row for the eight groups g: one half-wave exchange per pair of groups (g, g + 1)
    // gives the lower half 16 contiguous bytes of group g and the upper half those of group g + 1 -> 4 x 16-B stores per lane
    // instead of 8 x 8-B (the store tail of a block is issue-bound)
    u32x2 og[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int db = g >> 2, rq = g & 3;
        og[g].x = pack2bf(ot[db][rq * 4 + 0] * inv, ot[db][rq * 4 + 1] * inv);
        og[g].y = pack2bf(ot[db][rq * 4 + 2] * inv, ot[db][rq * 4 + 3] * inv);
    }
    bf16_t* o = (bf16_t*)a.out + (size_t)(b * a.Ntok + min(q_row, a.Ntok - 1)) * a.ld_out + h * 64 + hi * 8;
#pragma unroll
    for (int g = 0; g < 8; g += 2) {
        const auto rx = __builtin_amdgcn_permlane32_swap(og[g].x, og[g + 1].x, false, false);
        const auto ry = __builtin_amdgcn_permlane32_swap(og[g].y, og[g + 1].y, false, false);
        u32x4 v = {rx[0], ry[0], rx[1], ry[1]};
        if (q_row < a.Ntok) *(u32x4*)(o + 8 * g) = v;
    }
    if (ACCT && tid == 0 && wg < 8192) g_attn_blk[2 * wg + 1] = (long long)__builtin_amdgcn_s_memrealtime();
}

// XCD-aware order of the work items: XCD x owns a contiguous range, so all q-blocks of one (sample, head) run on one XCD and its
// K / V stay in that XCD's L2
__device__ __forceinline__ void attn_xcd_range(int total, int x, int& first, int& cnt) {
    const int q = total >> 3, r = total & 7;
    first = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    cnt = q + (x < r ? 1 : 0);
}
// one workgroup per work item (the op-level entry point: no per-launch state)
template <bool ACCT>
__global__ __launch_bounds__(512, 2) void attn_pp_k(const AttnArgs a, int nqb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [4 slots][K tile | VT tile]
    int first, cnt;
    attn_xcd_range((int)gridDim.x, blockIdx.x & 7, first, cnt);
    attn_pp_item<ACCT>(a, nqb, first + (int)(blockIdx.x >> 3), smem);
}
// PERSISTENT form (the engine's launches): one workgroup per CU pulls items from its XCD's queue and, when that is empty, from the
// other XCDs' queues.  The hardware hands every XCD exactly 1/8 of a grid, but the XCDs of one launch run a few percent apart in
// clock: with static shares the launch ended 95-100 % apart per XCD plus a last round of 32 items on 256 slots (4.6 % of the
// slot-time idle, tools/attn_harness timeline); pulled work ends within one item.  queue[0..7]: next item of XCD x, queue[8]:
// workgroups done -- the last one to leave zeroes the nine counters for the next launch (they must be zero at the first).
template <bool ACCT>
__global__ __launch_bounds__(512, 2) void attn_pp_persist_k(const AttnArgs a, int nqb, int total, int* __restrict__ queue) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_item;
    const int xcd = blockIdx.x & 7;
    for (;;) {
        if (threadIdx.x == 0) {
            int wg = -1;
            for (int k = 0; k < 8 && wg < 0; ++k) {
                const int y = (xcd + k) & 7;
                int first, cnt;
                attn_xcd_range(total, y, first, cnt);
                if (__hip_atomic_load(&queue[y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= cnt) continue;  // drained: do not hammer it
                const int i = atomicAdd(&queue[y], 1);
                if (i < cnt) wg = first + i;
            }
            s_item = wg;
        }
        __syncthreads();  // also fences the previous item's LDS traffic from the next item's prologue
        const int wg = s_item;
        __syncthreads();
        if (wg < 0) break;
        attn_pp_item<ACCT>(a, nqb, wg, smem);
    }
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&queue[8], 1) == (int)gridDim.x - 1) {
            for (int i = 0; i < 9; ++i) __hip_atomic_store(&queue[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
        }
    }
}

#ifdef S2V_DIAG
int g_attn_variant = 0;  // see launch_attn_bf16
extern "C" __attribute__((visibility("default"))) int s2v_set_attn_variant(int v) { g_attn_variant = v; return 0; }
static int* g_attn_queue = nullptr;  // harness: a queue for op-level launches (variants 4, 5 = persistent product / accounting kernel)
static int g_attn_ncu = 0;
extern "C" __attribute__((visibility("default"))) int s2v_set_attn_queue(int* q, int ncu) { g_attn_queue = q; g_attn_ncu = ncu; return 0; }
#endif

int launch_attn_bf16(const AttnArgs& a_in, hipStream_t st) {
    AttnArgs a = a_in;
    S2V_REQUIRE(a.vt != nullptr, "attn_bf16: V^T buffer missing");
    S2V_REQUIRE(a.ld_qkv % 8 == 0 && a.ntok_pad % 64 == 0, "attn_bf16: bad leading dims");
    S2V_REQUIRE(a.ntok_pad >= ((a.Ntok + 63) / 64) * 64, "attn_bf16: ntok_pad too small");
    const int nqb8 = (a.Ntok + 255) / 256;  // 256 query rows per work item
    const int total = nqb8 * a.B * a.H;
#ifdef S2V_DIAG
    // variants: 0 = product (attn_q4; persistent when the caller brought a queue), 3 = product, one workgroup per item,
    //   4 = product, persistent on the harness queue (0 / 3 / 4: attn_pp up to ATTN_PP_MAX_TOKENS, attn_q4 beyond), 6 / 7 = attn_q4 per item /
    //   persistent at any length, 8 / 9 = attn_q8 (the same stream, eight
    //   waves x 32 rows), 10 / 11 = attn_pp (round-2 kernel) per item / persistent, 1 / 5 = attn_pp with stall accounting, 2 = round-1 kernel
    const int v = g_attn_variant;
    if ((v == 4 || v == 5 || v == 7 || v == 9 || v == 11) && !a.queue) { a.queue = g_attn_queue; a.num_cus = g_attn_ncu; }
    if (v == 3 || v == 6 || v == 8 || v == 10 || v == 1 || v == 2) a.queue = nullptr;
    if (v == 8 || v == 9) return launch_attn_q8(a, v == 9, st);
    if (v == 6 || v == 7) return launch_attn_q4(a, v == 7 && a.queue != nullptr, st);  // attn_q4 whatever the sequence length
    if (v == 1 || v == 2 || v == 5 || v == 10 || v == 11) {
        S2V_REQUIRE(a.mx_q == nullptr, "attn_bf16: the MX output of the fp8 engine is attn_q4 / attn_q8's (reference variants write bf16 only)");
        const dim3 g8(total), b8(512);
        const void* fn = (v == 1) ? (const void*)attn_pp_k<true> : (const void*)attn_pp_k<false>;
        size_t lds = 65536;
        if (v == 2) { fn = (const void*)attn_bf16_k<0, 8>; lds = 4 * ATT_TILE_BYTES; }
        if ((v == 5 || v == 11) && a.queue != nullptr) {
            fn = (v == 5) ? (const void*)attn_pp_persist_k<true> : (const void*)attn_pp_persist_k<false>;
            S2V_TRY(ensure_lds_attr(fn, 65536));
            int* queue = a.queue;
            void* args[] = {(void*)&a, (void*)&nqb8, (void*)&total, (void*)&queue};
            S2V_CHECK_HIP(hipLaunchKernel(fn, dim3((a.num_cus / 8) * 8), b8, args, lds, st));
            return 0;
        }
        S2V_TRY(ensure_lds_attr(fn, 65536));
        void* args[] = {(void*)&a, (void*)&nqb8};
        S2V_CHECK_HIP(hipLaunchKernel(fn, g8, b8, args, lds, st));
        return 0;
    }
#endif
    // persistent (work-pulling) launch when the caller owns a queue and there is more than two rounds of work; else one workgroup per item
    const bool persist = a.queue != nullptr && total > 2 * a.num_cus && a.num_cus >= 8;
    // Short sequences run the eight-wave ping-pong kernel (attn_pp, round 2's product kernel): attn_q4's one-statement body pays a long
    // prologue (first-tile maxima, fragment pipeline fill) and walks its last five KV tiles through the rare-path handler, which is
    // nothing at 299 tiles (C3: attn_q4 5-6 % ahead) and a quarter of the iterations at 20 (C1 step 12.01 -> 11.27 ms on one box, DESIGN section 3;
    // tools/attn_small_probe.py).  Same deferred-maximum rule (2^64), results within bf16 rounding of attn_q4's, not bit-identical.
    if (a.p16) {
        S2V_REQUIRE(attn_runs_q4(a.Ntok, a.mx_q != nullptr), "attn_bf16: fp16 P / V^T is the four-wave kernel's; short sequences run attn_pp on bf16 V^T");
        return launch_attn_q4h(a, persist, st);
    }
    if (a.mx_q == nullptr && a.Ntok <= ATTN_PP_MAX_TOKENS) {
        const void* fn = persist ? (const void*)attn_pp_persist_k<false> : (const void*)attn_pp_k<false>;
        S2V_TRY(ensure_lds_attr(fn, 65536));
        if (persist) {
            int* queue = a.queue;
            void* args[] = {(void*)&a, (void*)&nqb8, (void*)&total, (void*)&queue};
            S2V_CHECK_HIP(hipLaunchKernel(fn, dim3((a.num_cus / 8) * 8), dim3(512), args, 65536, st));
        } else {
            void* args[] = {(void*)&a, (void*)&nqb8};
            S2V_CHECK_HIP(hipLaunchKernel(fn, dim3(total), dim3(512), args, 65536, st));
        }
        return 0;
    }
    return launch_attn_q4(a, persist, st);
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

// fp16 model dtype: qkv / out fp16, a.vt = V^T [B][H][64][ntok_pad] fp16 in the k-slot order (launch_v_transpose moves 16-bit elements: the bf16
// pass serves fp16 input unchanged)
int launch_attn_f16(const AttnArgs& a, hipStream_t st) {
    S2V_REQUIRE(a.vt != nullptr && a.ntok_pad % 64 == 0 && a.ld_qkv % 8 == 0, "attn_f16: V^T scratch / padding / leading dimension");
    // long sequences: the four-wave generated-asm kernel (fp16 q / k / P: attn_q4hh), persistent when the caller brought a queue; short ones
    // (where attn_q4's long pipeline does not pay, as in bf16) the lock-step kernel below
    if (a.Ntok > ATTN_PP_MAX_TOKENS) return launch_attn_q4hh(a, a.queue != nullptr && a.num_cus >= 8, st);
    const int nqb = (a.Ntok + 255) / 256, total = nqb * a.B * a.H;
    hipLaunchKernelGGL((attn_bf16_k<0, 8, f16_t>), dim3(total), dim3(512), 4 * ATT_TILE_BYTES, st, a, nqb);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_attn_simple(const AttnArgs& a, int dtype, hipStream_t st) {
    if ((dtype == S2V_F32 || dtype == S2V_F16) && !a.valu_only) return launch_attn_f32m(a, dtype, st);
    dim3 grid((a.Ntok + 3) / 4, a.H, a.B);
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(attn_simple_k<T>, grid, dim3(256), 0, st, a))
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}
