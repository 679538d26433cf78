// Joint [text | ref-image | video] self-attention, head_dim 64, no mask (replaces F.scaled_dot_product_attention at
// attention_processor.py:2083-2087): the four-wave, one-wave-per-SIMD form.  Compiled with -fno-slp-vectorize (build.py): the SLP
// vectoriser merges element accesses of neighbouring accumulators into 32-wide vectors, which keeps SROA from promoting them.
#define S2V_HOST
#include "common.h"
#include "kernels.h"
#include <type_traits>

#define KV_TILE 64
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// XCD-aware order of the work items (as attention.hip): XCD x owns a contiguous range of q-blocks
__device__ __forceinline__ void attn_xcd_range(int total, int x, int& first, int& cnt) {
    const int q = total >> 3, r = total & 7;
    first = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    cnt = q + (x < r ? 1 : 0);
}

// ---------------------------------------------------------------------------------------------------
// attn_q4: FOUR waves, one per SIMD, 64 query rows per wave (two 32-row blocks j = 0, 1), the whole 512-entry register file.
// A workgroup is the same 256-row work item as attn_pp_item (attention.hip).  What changes against the eight-wave ping-pong:
//   * no SIMD partner: the wave fills the shadow of its OWN MFMAs.  A 32x32x16 MFMA occupies the matrix pipe for 32 cycles and
//     leaves room for about five single-issue instructions behind it (MI355X_MICROARCH, per-instruction constants); the
//     eight-wave form gave the softmax wave one VALU per 16 cycles beside its partner's MFMA stream;
//   * a K / V^T fragment read from LDS feeds TWO MFMAs (both 32-row blocks): 16 ds_read_b128 per 32 MFMA instead of 32;
//   * one barrier per KV tile instead of two.
// Per KV tile and wave: 32 MFMA (1024 matrix-pipe cycles) and 64 exp2 + 64 row-sum adds + 32 v_cvt_pk + 16 fragment reads +
// 4 LDS-DMA pieces + the check = about six fillers per MFMA, each placed by hand.
// Software pipeline (iteration t, S = scores, P = exp2(S - m)):
//   segment 1: 16 MFMA  S(t+1) = K(t+1).Q^T  -> st[(t+1)&1]   | fillers: second half of P(t) (keys 32-63) -> pk[.][2,3],
//                                                               V^T(t) fragments, K(t+4) DMA
//   check    : row sums of tile t against 2^13 (deferred maximum, as attn_pp_item)
//   segment 2: 16 MFMA  O += V^T(t).P(t)                       | fillers: first half of P(t+1) (keys 0-31) -> pk[.][0,1] (written
//                                                               behind the MFMAs that read the old values), K(t+2) fragments,
//                                                               V^T(t+2) DMA, l += row sums of t; vmcnt(4) + barrier
// The raw scores of a tile stay in their st buffer until S(t+2) overwrites them, so the slow path (true row maximum, rescale,
// exp2 redone) has them at the check; it also shifts S(t+1), which segment 1 computed against the old maximum.
// Everything from "tiles staged, Q loaded" to "O and l complete" is ONE generated asm statement (csrc/gen_attn_q4.py ->
// attn_q4_body.inc; register map, phases and the reasons in its header): this function stages the first tiles, loads Q and, after
// the asm, normalises and stores O.  Hazards the assembler does not see: a VALU reads an MFMA result at least four MFMAs after its
// issue, the rare paths and the end of the body start with s_nop 15 x 2, fragment reads are awaited with one s_waitcnt lgkmcnt(0)
// per segment.
// LDS: ring of 4 slots x [K tile 8 KiB | V^T tile 8 KiB]; a wave stages pieces w and w + 4 (8 rows each) of every tile.
//   RAW  K(t+4), V(t+2) are issued in iteration t and awaited (vmcnt(4): everything but this iteration's four) at the end of
//        iteration t+1, one barrier before their first readers (K(t+4): segment 2 of t+2; V(t+2): segment 1 of t+2).
//   WAR  K(t+4) replaces K(t) (last read in segment 2 of t-2), V(t+2) replaces V(t-2) (last read in segment 1 of t-2).
#include "attn_q4_regs.h"
typedef __attribute__((ext_vector_type(32))) float f32x32;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(32))) unsigned int u32x32;
typedef __attribute__((ext_vector_type(8))) unsigned int u32x8;
#ifdef S2V_DIAG
__device__ unsigned long long g_qx_slow[2];
extern "C" __attribute__((visibility("default"))) int s2v_attn_slow_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qx_slow), 16) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long z[2] = {0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_qx_slow), z, 16) != hipSuccess) return -1;
    }
    return 0;
}
#endif
// JB = 32-row blocks per wave: 2 = attn_q4 (four waves x 64 rows, one per SIMD), 1 = attn_q8 (eight waves x 32 rows, two per SIMD, the
// same fine-grained stream in both: gen_attn_q4.py)
// F8 (JB = 2 only, attn_q4f): q and k are read as MX e4m3 images (AttnArgs::q8 / k8 + block scales, made by qk_quant_mx_k) and S^T = K.Q^T runs
// on v_mfma_scale_f32_32x32x64_f8f6f4 -- 4 MFMA of 64 cycles per KV tile instead of 16 of 32; V^T, P.V, softmax and the epilogue are the bf16
// kernel's.  No reference code for it (the reference has no fp8 path): parity unpinned, selected only by weight_format 2.
// P16 (JB = 2 only, attn_q4h / attn_q4fh): P and V^T in fp16, row sums by packed fp16 adds, deferred maximum 2^14 (gen_attn_q4.py, P16); the
// code around the body is the same -- the format lives in the V^T buffer (AttnArgs::p16) and in the generated instructions.
// H16 (JB = 2, with P16, not F8: attn_q4hh): the fp16 model dtype -- q, k and the output are fp16 as well; only the QK^T mnemonic of the body differs from
// attn_q4h's, and the code around it converts q and packs O as fp16.
template <int JB, bool F8 = false, bool P16 = false, bool H16 = false>
__device__ __forceinline__ void attn_qx_item(const AttnArgs& a, int nqb, int wg, char* smem, unsigned& slow_acc, unsigned& tile_acc) {
    static_assert((!F8 && !P16) || JB == 2, "the fp8 QK^T and fp16 P bodies exist for the four-wave form only");
    static_assert(!H16 || (P16 && !F8), "fp16 q / k come with fp16 P and bf16-free MFMAs only");
    constexpr int NW = 8 / JB;  // waves per work item
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, hi = lane >> 5;
    const int bh = wg / nqb, qb = wg - bh * nqb;
    const int b = bh / a.H, h = bh - b * a.H;
    const int D = a.H * 64;

    const bf16_t* qkv = (const bf16_t*)a.qkv + (size_t)b * a.Ntok * a.ld_qkv;
    const char* Kg = (const char*)(qkv + D + h * 64);  // (F8: the e4m3 image, below)
    const char* VTg = (const char*)((const bf16_t*)a.vt + (size_t)(b * a.H + h) * 64 * a.ntok_pad);
    const int nt = (a.Ntok + KV_TILE - 1) / KV_TILE;
    const unsigned lds0 = lds_base_u32(smem);

    // staging: a tile is eight pieces of 8 rows x 128 B; wave w stages piece w (and w + 4 when there are four waves), lane = (row,
    // 16-B chunk), chunk XOR on the SOURCE address ((row + 32) has the same XOR term: the second piece is the lane offset + 32 rows)
    const int srow = wave * 8 + (lane >> 3);
    const int sc = (lane & 7) ^ ((srow >> 1) & 7);
    u32x8 vin;  // [0..3] fragment address of k-step kk in slot 0 (the 32-row half and the slot are immediates), [4..7] staging offsets
    u32x4 kin = {0, 0, 0, 0};  // F8: [0 / 1] LDS address of chunks hi / 2 + hi of key fr in slot 0, [2] the lane's dword of a tile's block scales
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) vin[kk] = lds0 + fr * 128 + (((kk * 2 + hi) ^ ((fr >> 1) & 7)) << 4);
    if constexpr (F8) {
        // e4m3 K tile: 64 keys x 64 B; wave w stages keys 16 w .. 16 w + 15 (one 1-KiB piece), lane = (key, 16-byte position), the chunk XOR
        // ((key >> 2) & 3) on the SOURCE address; a lane's MFMA operand of a 32-key half = chunks hi and 2 + hi of key fr
        Kg = (const char*)a.k8 + (size_t)(b * a.H + h) * a.ntok_pad * 64;
        const int krow = wave * 16 + (lane >> 2);
        kin[0] = lds0 + fr * 64 + (((0 + hi) ^ ((fr >> 2) & 3)) << 4);
        kin[1] = lds0 + fr * 64 + (((2 + hi) ^ ((fr >> 2) & 3)) << 4);
        kin[2] = (unsigned)lane * 4;
        vin[4] = (unsigned)(krow * 64 + (((lane & 3) ^ ((krow >> 2) & 3)) << 4));
        vin[5] = 0;
    } else {
        vin[4] = (unsigned)(2 * (srow * a.ld_qkv + sc * 8));
        vin[5] = vin[4] + 64u * a.ld_qkv;
    }
    vin[6] = (unsigned)(2 * (srow * a.ntok_pad + sc * 8));
    vin[7] = vin[6] + 64u * a.ntok_pad;
    const unsigned k_tile_stride = F8 ? 4096u : (unsigned)KV_TILE * a.ld_qkv * 2;
    const unsigned m0w = lds0 + wave * 1024;  // LDS address of this wave's first piece in slot 0
    // tiles past the end are clamped (re-staged into a dead slot): every wave issues the same number of DMAs in every iteration
    auto k_src = [&](int t) __attribute__((always_inline)) { return Kg + (size_t)min(t, nt - 1) * k_tile_stride; };
    auto v_src = [&](int t) __attribute__((always_inline)) { return VTg + (size_t)min(t, nt - 1) * (KV_TILE * 2); };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < (F8 ? 1 : JB); ++p) glds16_saddr_m0(k_src(i), vin[4 + p], m0w + i * 16384 + p * 4096);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < JB; ++p) glds16_saddr_m0(v_src(i), vin[6 + p], m0w + i * 16384 + 8192 + p * 4096);

    // Q fragments of the wave's row blocks, pre-multiplied by scale * log2(e) and rounded to bf16 once (as attn_pp_item); word 16 j + 4 kk
    const float c0 = a.scale * 1.4426950408889634f;
    typedef __attribute__((ext_vector_type((F8 ? 8 : 16) * JB))) unsigned int qf_t;
    qf_t qf;
    int q_row[JB];
    u32x4 ks_ring = {0, 0, 0, 0};  // F8: block scales of K tiles 0 .. 2 (the body loads tile t + 3 in iteration t)
    u32x2 qs = {0, 0};             // F8: block scale of (query row of block j, 32-element block hi) in byte 0
    const char* ksb = nullptr;
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        q_row[j] = qb * 256 + wave * (32 * JB) + j * 32 + fr;
        const int q_ld = min(q_row[j], a.Ntok - 1);
        if constexpr (F8) {  // the image already carries scale * log2(e) (qk_quant_mx_k)
            const size_t qr = (size_t)(b * a.H + h) * a.Ntok + q_ld;
            const u32x4 lo = *(const u32x4*)(a.q8 + qr * 64 + hi * 16), hi4 = *(const u32x4*)(a.q8 + qr * 64 + 32 + hi * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { qf[8 * j + e] = lo[e]; qf[8 * j + 4 + e] = hi4[e]; }
            qs[j] = ((unsigned)a.q8s[qr] >> (8 * hi)) & 0xffu;
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 q = *(const bf16x8*)(qkv + (size_t)q_ld * a.ld_qkv + h * 64 + kk * 16 + hi * 8);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    if constexpr (H16) {
                        const f16x8 qh = __builtin_bit_cast(f16x8, q);
                        qf[16 * j + 4 * kk + (e >> 1)] = pack2h((float)qh[e] * c0, (float)qh[e + 1] * c0);
                    } else {
                        qf[16 * j + 4 * kk + (e >> 1)] = pack2bf((float)q[e] * c0, (float)q[e + 1] * c0);
                    }
                }
            }
        }
    }
    if constexpr (F8) {
        ksb = (const char*)(a.k8s + (size_t)(b * a.H + h) * (a.ntok_pad / 64) * 64);
#pragma unroll
        for (int i = 0; i < 3; ++i) ks_ring[i] = *(const unsigned*)(ksb + (size_t)min(i, nt - 1) * 256 + lane * 4);
    }
    u32x4 ptr, sin;  // sources of the next K / V^T tile to stage (64-bit each); nt, K tile stride, Ntok, LDS address of the wave's first piece
    {
        const unsigned long long kp = (unsigned long long)k_src(4), vp = (unsigned long long)v_src(2);
        ptr[0] = __builtin_amdgcn_readfirstlane((unsigned)kp); ptr[1] = __builtin_amdgcn_readfirstlane((unsigned)(kp >> 32));
        ptr[2] = __builtin_amdgcn_readfirstlane((unsigned)vp); ptr[3] = __builtin_amdgcn_readfirstlane((unsigned)(vp >> 32));
        sin[0] = (unsigned)nt; sin[1] = k_tile_stride; sin[2] = (unsigned)a.Ntok; sin[3] = m0w;
    }
    f32x32 OT[JB];  // ot(j, db)[e] = OT[j][16 db + e]
    f32x2 LR;
    unsigned slow_cnt;  // slow paths this wave took (diagnostics)

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (F8) {
        u32x2 ksbp;
        ksbp[0] = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)ksb);
        ksbp[1] = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)ksb >> 32));
        if constexpr (P16) {
            asm volatile(
#include "attn_q4fh_body.inc"
                : "=" Q4FH_OT0(OT[0]), "=" Q4FH_OT1(OT[JB - 1]), "=" Q4FH_LRUN(LR), "+" Q4FH_PTR(ptr), "=" Q4FH_CNT(slow_cnt), "+" Q4FH_KS(ks_ring)
                : Q4FH_QF(qf), Q4FH_VIN(vin), Q4FH_SIN(sin), Q4FH_QS(qs), Q4FH_KIN(kin), Q4FH_KSB(ksbp)
                : Q4FH_CLOBBERS);
        } else {
            asm volatile(
#include "attn_q4f_body.inc"
                : "=" Q4F_OT0(OT[0]), "=" Q4F_OT1(OT[JB - 1]), "=" Q4F_LRUN(LR), "+" Q4F_PTR(ptr), "=" Q4F_CNT(slow_cnt), "+" Q4F_KS(ks_ring)
                : Q4F_QF(qf), Q4F_VIN(vin), Q4F_SIN(sin), Q4F_QS(qs), Q4F_KIN(kin), Q4F_KSB(ksbp)
                : Q4F_CLOBBERS);
        }
    } else if constexpr (H16) {
        asm volatile(
#include "attn_q4hh_body.inc"
            : "=" Q4HH_OT0(OT[0]), "=" Q4HH_OT1(OT[JB - 1]), "=" Q4HH_LRUN(LR), "+" Q4HH_PTR(ptr), "=" Q4HH_CNT(slow_cnt)
            : Q4HH_QF(qf), Q4HH_VIN(vin), Q4HH_SIN(sin)
            : Q4HH_CLOBBERS);
    } else if constexpr (P16) {
        asm volatile(
#include "attn_q4h_body.inc"
            : "=" Q4H_OT0(OT[0]), "=" Q4H_OT1(OT[JB - 1]), "=" Q4H_LRUN(LR), "+" Q4H_PTR(ptr), "=" Q4H_CNT(slow_cnt)
            : Q4H_QF(qf), Q4H_VIN(vin), Q4H_SIN(sin)
            : Q4H_CLOBBERS);
    } else if constexpr (JB == 2) {
        asm volatile(
#include "attn_q4_body.inc"
            : "=" Q4_OT0(OT[0]), "=" Q4_OT1(OT[JB - 1]), "=" Q4_LRUN(LR), "+" Q4_PTR(ptr), "=" Q4_CNT(slow_cnt)
            : Q4_QF(qf), Q4_VIN(vin), Q4_SIN(sin)
            : Q4_CLOBBERS);
    } else {
        asm volatile(
#include "attn_q8_body.inc"
            : "=" Q8_OT0(OT[0]), "=" Q8_LRUN(LR), "+" Q8_PTR(ptr), "=" Q8_CNT(slow_cnt)
            : Q8_QF(qf), Q8_VIN(vin), Q8_SIN(sin)
            : Q8_CLOBBERS);
    }

    slow_acc += slow_cnt;   // census of this wave over its items (AttnArgs::stats), reported once per kernel
    tile_acc += (unsigned)nt;
#ifdef S2V_DIAG
    if (lane == 0) {  // slow-path census of tools/attn_harness: (wave, tile) pairs that took the slow path / that ran
        atomicAdd(&g_qx_slow[0], (unsigned long long)slow_cnt);
        atomicAdd(&g_qx_slow[1], (unsigned long long)nt);
    }
#endif
    // epilogue: as attn_pp_item, once per row block
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const float l_tot = LR[j] + __shfl_xor(LR[j], 32, 64);
        const float inv = 1.0f / l_tot;
        u32x2 og[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int db = g >> 2, rq = g & 3;
            if constexpr (H16) {
                og[g].x = pack2h(OT[j][16 * db + rq * 4 + 0] * inv, OT[j][16 * db + rq * 4 + 1] * inv);
                og[g].y = pack2h(OT[j][16 * db + rq * 4 + 2] * inv, OT[j][16 * db + rq * 4 + 3] * inv);
            } else {
                og[g].x = pack2bf(OT[j][16 * db + rq * 4 + 0] * inv, OT[j][16 * db + rq * 4 + 1] * inv);
                og[g].y = pack2bf(OT[j][16 * db + rq * 4 + 2] * inv, OT[j][16 * db + rq * 4 + 3] * inv);
            }
        }
        u32x4 vv[4];  // the lane's 8-column groups: columns 16 u + 8 hi .. + 7 of the head, u = 0 .. 3
#pragma unroll
        for (int g = 0; g < 8; g += 2) {
            const auto rx = __builtin_amdgcn_permlane32_swap(og[g].x, og[g + 1].x, false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(og[g].y, og[g + 1].y, false, false);
            vv[g >> 1] = u32x4{rx[0], ry[0], rx[1], ry[1]};
        }
        const size_t orow = (size_t)(b * a.Ntok + min(q_row[j], a.Ntok - 1));
        if (a.mx_q == nullptr) {
            bf16_t* o = (bf16_t*)a.out + orow * a.ld_out + h * 64 + hi * 8;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q_row[j] < a.Ntok) *(u32x4*)(o + 16 * u) = vv[u];
        } else {
            // fp8 engine: the out-projection's A operand as MX e4m3 (GemmArgs::mx_a_s) -- a head is two 32-column blocks, each held by
            // the row's two lanes (16 values each): block amax by v_pk_max_u16 + one cross-half exchange, scale = the smallest power
            // of two with amax / scale <= 448, conversion by v_cvt_scalef32_pk_fp8_bf16 (as the FF1 epilogue, gemm_epi.h)
            typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
            unsigned ebs = 0;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                u16x2_t pm = {0, 0};
#pragma unroll
                for (int u = 2 * blk; u < 2 * blk + 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) pm = __builtin_elementwise_max(pm, __builtin_bit_cast(u16x2_t, vv[u][e] & 0x7fff7fffu));
                unsigned amb = max((unsigned)pm[0], (unsigned)pm[1]) << 16;
                const auto sw = __builtin_amdgcn_permlane32_swap(amb, amb, false, false);
                amb = max(sw[0], sw[1]);
                unsigned eb = (__float_as_uint(__uint_as_float(amb) * (1.0f / 448.0f)) + 0x7fffffu) >> 23;
                eb = min(max(eb, 1u), 254u);
                ebs |= eb << (8 * blk);
                const float bscale = __uint_as_float(eb << 23);
#pragma unroll
                for (int u = 2 * blk; u < 2 * blk + 2; ++u) {
                    unsigned w0 = 0, w1 = 0;
                    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(w0) : "v"(vv[u][0]), "v"(bscale));
                    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(w0) : "v"(vv[u][1]), "v"(bscale));
                    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(w1) : "v"(vv[u][2]), "v"(bscale));
                    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(w1) : "v"(vv[u][3]), "v"(bscale));
                    if (q_row[j] < a.Ntok) *(u32x2*)(a.mx_q + orow * a.ld_out + h * 64 + 16 * u + hi * 8) = u32x2{w0, w1};
                }
            }
            if (hi == 0 && q_row[j] < a.Ntok) *(unsigned short*)(a.mx_s + ((size_t)(h >> 1) * a.mx_rows + mx_perm_row((int64_t)orow)) * 4 + (h & 1) * 2) = (unsigned short)ebs;  // K-tile major: head h = blocks 2 h, 2 h + 1 of K-tile h / 2
        }
    }
}
// one pair of atomics per wave and KERNEL (the persistent launch: 1024 per launch, on 256 distinct slots)
__device__ __forceinline__ void attn_report(const AttnArgs& a, unsigned slow, unsigned tiles) {
    if (a.stats != nullptr && (threadIdx.x & 63) == 0) {
        unsigned long long* s = a.stats + 2 * (blockIdx.x & 255);
        atomicAdd(s, (unsigned long long)slow);
        atomicAdd(s + 1, (unsigned long long)tiles);
    }
}
template <int JB, bool F8 = false, bool P16 = false, bool H16 = false>
__global__ __launch_bounds__(64 * (8 / JB), 1) void attn_qx_k(const AttnArgs a, int nqb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [4 slots][K tile | VT tile]
    int first, cnt;
    clk_stamp(a.clk, gridDim.x >> 1, 0);
    attn_xcd_range((int)gridDim.x, blockIdx.x & 7, first, cnt);
    unsigned slow = 0, tiles = 0;
    attn_qx_item<JB, F8, P16, H16>(a, nqb, first + (int)(blockIdx.x >> 3), smem, slow, tiles);
    attn_report(a, slow, tiles);
    clk_stamp(a.clk, gridDim.x >> 1, 1);
}
template <int JB, bool F8 = false, bool P16 = false, bool H16 = false>
__global__ __launch_bounds__(64 * (8 / JB), 1) void attn_qx_persist_k(const AttnArgs a, int nqb, int total, int* __restrict__ queue) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_item;
    const int xcd = blockIdx.x & 7;
    unsigned slow = 0, tiles = 0;
    clk_stamp(a.clk, 0, 0);  // persistent: workgroup 0 pulls work until the queues are empty
    // Start stagger (round 5, configs[4] locality experiment): the 32 workgroups of an XCD walk consecutive q-blocks of one head, i.e. the SAME
    // K / V^T tiles, and every item costs the same -- started together they request each tile within the latency of its first miss.  With
    // a.stagger > 0 slot s of the XCD starts s * stagger * 64 cycles late, once per launch: a convoy in which the leader misses in L2 and the
    // followers find the tile there.  0 = start together (the default; profiles/r05_attn_stagger.txt for what it measured: slower).  The
    // diagnostics build only: the product kernel carries neither this knob nor the a.order experiment below.
#ifdef S2V_DIAG
    if (a.stagger > 0) {
        const int slot = (int)(blockIdx.x >> 3);
        for (int i = 0; i < slot * a.stagger; i += 64) __builtin_amdgcn_s_sleep(64);
        if ((slot * a.stagger) & 63) __builtin_amdgcn_s_sleep(1);
    }
#endif
    for (;;) {
        if (threadIdx.x == 0) {
            int wg = -1;
            for (int k = 0; k < 8 && wg < 0; ++k) {
                const int y = (xcd + k) & 7;
                int first, cnt;
                // a.order 0: XCD y owns a contiguous range of items (a head's q-blocks stay on one XCD); 1 (experiment, round 5): a head's q-blocks
                // are dealt round-robin over the XCDs (q-block qb on XCD qb & 7), so all 32 workgroups of an XCD are on the SAME head at any time
#ifdef S2V_DIAG
                const bool dealt = a.order != 0;
#else
                constexpr bool dealt = false;
#endif
                const int per = dealt ? (nqb > y ? (nqb - y + 7) >> 3 : 0) : 0;
                if (dealt) { first = 0; cnt = (total / nqb) * per; }
                else attn_xcd_range(total, y, first, cnt);
                if (__hip_atomic_load(&queue[y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= cnt) continue;
                const int i = atomicAdd(&queue[y], 1);
                if (i < cnt) {
#ifdef S2V_DIAG
                    if (dealt) { const int bh = i / per, j = i - bh * per; wg = bh * nqb + y + 8 * j; }
                    else
#endif
                        wg = first + i;
                }
            }
            s_item = wg;
        }
        __syncthreads();
        const int wg = s_item;
        __syncthreads();
        if (wg < 0) break;
        attn_qx_item<JB, F8, P16, H16>(a, nqb, wg, smem, slow, tiles);
    }
    attn_report(a, slow, tiles);
    clk_stamp(a.clk, 0, 1);
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&queue[8], 1) == (int)gridDim.x - 1) {
            for (int i = 0; i < 9; ++i) __hip_atomic_store(&queue[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
        }
    }
}

template <int JB, bool F8 = false, bool P16 = false, bool H16 = false>
static int launch_attn_qx(const AttnArgs& a, bool persistent, hipStream_t st) {
    const int nqb = (a.Ntok + 255) / 256;  // 256 query rows per item in both forms
    const int total = nqb * a.B * a.H;
    const size_t lds = 65536;
    const dim3 blk(64 * (8 / JB));
    if (persistent && a.queue != nullptr && a.num_cus >= 8) {
        const void* fn = (const void*)attn_qx_persist_k<JB, F8, P16, H16>;
        S2V_TRY(ensure_lds_attr(fn, 65536));
        int* queue = a.queue;
        void* args[] = {(void*)&a, (void*)&nqb, (void*)&total, (void*)&queue};
        S2V_CHECK_HIP(hipLaunchKernel(fn, dim3((a.num_cus / 8) * 8), blk, args, lds, st));
        return 0;
    }
    const void* fn = (const void*)attn_qx_k<JB, F8, P16, H16>;
    S2V_TRY(ensure_lds_attr(fn, 65536));
    void* args[] = {(void*)&a, (void*)&nqb};
    S2V_CHECK_HIP(hipLaunchKernel(fn, dim3(total), blk, args, lds, st));
    return 0;
}
int launch_attn_q4(const AttnArgs& a, bool persistent, hipStream_t st) { return launch_attn_qx<2>(a, persistent, st); }
int launch_attn_q8(const AttnArgs& a, bool persistent, hipStream_t st) { return launch_attn_qx<1>(a, persistent, st); }
int launch_attn_q4f(const AttnArgs& a, bool persistent, hipStream_t st) {
    S2V_REQUIRE(a.q8 && a.k8 && a.q8s && a.k8s && a.vt, "attn_q4f: the MX images of q / k (launch_qk_quant_mx) and V^T are required");
    return a.p16 ? launch_attn_qx<2, true, true>(a, persistent, st) : launch_attn_qx<2, true>(a, persistent, st);
}
// fp16 model dtype: qkv / out fp16, a.vt = fp16 V^T (the transpose pass moves the bits); the four-wave kernel with fp16 q, k, P
int launch_attn_q4hh(const AttnArgs& a, bool persistent, hipStream_t st) {
    S2V_REQUIRE(a.vt && a.mx_q == nullptr, "attn_q4hh: the fp16 V^T is required (and no MX output)");
    return launch_attn_qx<2, false, true, true>(a, persistent, st);
}
int launch_attn_q4h(const AttnArgs& a, bool persistent, hipStream_t st) {
    S2V_REQUIRE(a.p16 && a.vt, "attn_q4h: the fp16 V^T (launch_v_transpose(..., to_f16)) is required");
    return launch_attn_qx<2, false, true>(a, persistent, st);
}
