// Epilogues shared by the MFMA GEMM kernels (gemm.hip, gemm_g4.hip): bias (+ GELU / gated residual / residual add / fused q-k LayerNorm
// + rotary embedding) on a wave's accumulator tile, staged through LDS and written as full 128-byte lines.
#pragma once
#include "common.h"
#include "kernels.h"
#include <type_traits>

// No fma contraction in the epilogues: they restate separately-rounded elementwise ops (LayerNorm, rotary embedding, gate, residual) and
// are compiled into more than one translation unit -- with -ffp-contract=fast hipcc fused `a * b + c` in gemm_g4.hip where it had not in
// gemm.hip, and the rows of a split GEMM (256-row tiles on one kernel, the row tail on another) then differed by a bf16 ulp.
// The translation unit's own setting is put back at the end of this header: hipcc's default `fast`, or `off` when the unit is built
// with -ffp-contract=off (elementwise.hip / vae.hip / t5.hip; build.py defines S2V_TU_FP_CONTRACT_OFF beside that flag -- clang has no
// predefined macro for it and `#pragma float_control(push / pop)` is not supported on amdgcn).
#pragma clang fp contract(off)

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
        if constexpr (std::is_same<T, bf16_t>::value) {
            u32x2 p;
            p.x = pack2bf(y[0], y[1]);
            p.y = pack2bf(y[2], y[3]);
            *(u32x2*)c = p;
        } else if constexpr (std::is_same<T, f16_t>::value) {
            u32x2 p;
            p.x = pack2h(y[0], y[1]);
            p.y = pack2h(y[2], y[3]);
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

// v + v[lane ^ 1], then + [lane ^ 2], then + [lane ^ 4]: the sum over the 8 lanes that hold one 64-column row, the same additions in the
// same order as three __shfl_xor steps -- but as DPP operands (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror: after the two
// quad steps every lane of a quad holds the quad's sum, and lane i's mirror 7 - i sits in the other quad) instead of three ds_bpermute
// round trips through the LDS crossbar, each behind its own s_waitcnt lgkmcnt(0): 192 of them per q / k wave tile were ~20 k of the
// ~37 k exposed cycles of a QKV tile's epilogue on a wave that has its SIMD to itself
__device__ __forceinline__ float oct_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
    return v;
}

// bias of a wave's 64 columns, 4 per lane and (i, rq).  The whole-tile case is ONE clause of eight independent loads (a per-load
// bounds branch serialised them: eight global latencies, ~6000 cycles of a 128 x 64 epilogue)
__device__ __forceinline__ void load_bias64(const GemmArgs& a, int nw, int hi, u32x2 (&bvec)[8]) {
    const bf16_t* bias = (const bf16_t*)a.bias;
    if (bias && nw + 64 <= a.N) {
#pragma unroll
        for (int q = 0; q < 8; ++q) bvec[q] = *(const u32x2*)(bias + nw + (q >> 2) * 32 + 8 * (q & 3) + 4 * hi);  // 8-byte aligned
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int nl = (q >> 2) * 32 + 8 * (q & 3) + 4 * hi;
            unsigned short t[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (bias && nw + nl + e < a.N) t[e] = ((const unsigned short*)bias)[nw + nl + e];
            bvec[q] = u32x2{(unsigned)t[0] | ((unsigned)t[1] << 16), (unsigned)t[2] | ((unsigned)t[3] << 16)};
        }
    }
}

// Epilogue of one wave's 64(m) x 64(n) accumulator tile (2x2 MFMA 32x32 blocks, D[i = n][j = m]).
// Accumulator layout -> +bias, (GELU), round to bf16 in registers -> the wave's private 8 KiB LDS patch (rows of
// 128 B, 16-B chunks XOR-swizzled by row&7) -> read back row-major, 16 B per lane, 8 lanes per 128-B line -> gate /
// residual in that layout -> full-line global stores.  A row-per-lane epilogue (8-B stores at a row stride) was
// store-issue bound: ~0.7 ms of a 3.4 ms FF1 launch.
// GRP = rows-of-8 groups whose patch reads and gate / residual loads are in flight together (one memory latency per GRP groups)
// one packed pair of 16-bit storage values <-> two floats: what the vector epilogue decodes and packs (round 5: the same epilogue serves the fp16
// model dtype; for bf16 these are exactly the shifts / masks / v_cvt_pk_bf16_f32 the epilogue had inline)
template <typename T16> struct H2;
template <> struct H2<bf16_t> {
    static __device__ __forceinline__ float lo(unsigned w) { return __uint_as_float(w << 16); }
    static __device__ __forceinline__ float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
    static __device__ __forceinline__ unsigned pack(float a, float b) { return pack2bf(a, b); }
    static __device__ __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }
};
template <> struct H2<f16_t> {
    static __device__ __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }
    static __device__ __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }
    static __device__ __forceinline__ unsigned pack(float a, float b) { return pack2h(a, b); }
    static __device__ __forceinline__ float rnd(float v) { return (float)(f16_t)v; }
};
template <int EPI, int MB, bool SC = false, int GRP = 4, typename T16 = bf16_t>
__device__ __forceinline__ void epilogue_wave_b(const GemmArgs& a, const f32x16 (&acc)[2][MB], int mw, int nw, char* patch, int lane,
                                                const u32x2 (&bvec)[8]) {
    // MB 32-row blocks: the patch holds MB*32 rows of 128 B (8 KiB for MB = 2, 16 KiB for MB = 4); all accumulator blocks are
    // written first, then read back, so the LDS round trip is paid once per wave tile
    const int fr = lane & 31, hi = lane >> 5;
    const int c16 = lane & 7;
    // EPI_BIAS_QKNORM: this wave's 64 columns are ONE head of q, k or v (nw % 64 == 0); 8 lanes hold a row of it.  The rotary
    // values of a row group (4 cos + 4 sin per lane: the table stores every duplicated pair once) are requested one group AHEAD of
    // their use -- fetched with the read-back they cost the epilogue a memory latency per group (0.16 ms per QKV launch at C3)
    const bool qk_head = EPI == EPI_BIAS_QKNORM && nw < 2 * a.qk_D;
    const bool qk_rot = qk_head && a.qk_cs != nullptr;
    float qw[8], qb[8];
    f32x4 rc[2][GRP], rs[2][GRP];
    bool rope[2][GRP];
    auto rot_load = [&](int it0, int buf) {
        const float inv_tok = 1.0f / (float)a.tok_per_batch;
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            const int m = min(mw + (it0 + u) * 8 + (lane >> 3), a.M - 1);
            int b = (int)((float)m * inv_tok);  // m / tok_per_batch without the integer division (corrected below)
            int r = m - b * a.tok_per_batch;
            if (r < 0) r += a.tok_per_batch;
            if (r >= a.tok_per_batch) r -= a.tok_per_batch;
            rope[buf][u] = qk_rot && r >= a.text_len;
            const float* tp = a.qk_cs + (size_t)(rope[buf][u] ? r - a.text_len : 0) * 64 + c16 * 4;
            if (qk_rot) {
                rc[buf][u] = *(const f32x4*)tp;
                rs[buf][u] = *(const f32x4*)(tp + 32);
            }
        }
    };
    if (EPI == EPI_BIAS_QKNORM) {
        const int which = nw >= a.qk_D;
        const u32x4 w4 = *(const u32x4*)((const bf16_t*)a.qk_w[which] + c16 * 8), b4 = *(const u32x4*)((const bf16_t*)a.qk_b[which] + c16 * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qw[2 * e] = H2<T16>::lo(w4[e]); qw[2 * e + 1] = H2<T16>::hi(w4[e]);
            qb[2 * e] = H2<T16>::lo(b4[e]); qb[2 * e + 1] = H2<T16>::hi(b4[e]);
        }
        rot_load(0, 0);
    }
    // EPI_BIAS_ADD on 64-row wave tiles (the VAE's residual convolutions on gemm_bf16_stag: K loops of 17 us, two dependent memory latencies per
    // epilogue were 0.65 ms of a 2.6 ms launch, tools/vae_conv_rates.py): the residual vectors of ALL row groups are requested here, before the
    // accumulators go through the patch, instead of one GRP at a time behind the read-back
    constexpr bool PRE = EPI == EPI_BIAS_ADD && MB == 2;
    u32x4 xpre[PRE ? MB * 4 : 1];
    if (PRE) {
        const int npre = nw + c16 * 8 < a.N ? nw + c16 * 8 : 0;
#pragma unroll
        for (int it = 0; it < MB * 4; ++it)
            xpre[it] = *(const u32x4*)((const bf16_t*)a.R + (size_t)min(mw + it * 8 + (lane >> 3), a.M - 1) * a.ldr + npre);
        __builtin_amdgcn_sched_barrier(0);
    }
    // SC (fp8 operands): acc * a_scale[row] * w_scale[column] first -- the dequantisation of the per-token / per-channel scales
    float sa[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) sa[j] = (SC && a.a_scale) ? a.a_scale[min(mw + j * 32 + fr, a.M - 1)] : 1.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int nl = i * 32 + 8 * rq + 4 * hi;  // local column of 4 consecutive outputs
            const u32x2 bq = bvec[i * 4 + rq];
            f32x4 sw = {1.f, 1.f, 1.f, 1.f};
            if (SC) sw = *(const f32x4*)(a.w_scale + nw + nl);  // the scale array is padded to the 256-column tile
            const float bv[4] = {H2<T16>::lo(bq.x), H2<T16>::hi(bq.x), H2<T16>::lo(bq.y),
                                 H2<T16>::hi(bq.y)};
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (SC ? acc[i][j][rq * 4 + e] * (sa[j] * sw[e]) : acc[i][j][rq * 4 + e]) + bv[e];
                const int row = j * 32 + fr;
                u32x2 p;
                p.x = H2<T16>::pack(y[0], y[1]);  // the linear's bf16 output (one rounding; a single wave per SIMD issues a VALU
                p.y = H2<T16>::pack(y[2], y[3]);  // instruction every ~5 cycles, so the instruction count of this loop is its time)
                if (EPI == EPI_BIAS_GELU) {  // GELU of the ROUNDED linear output, rounded again
                    p.x = H2<T16>::pack(gelu_tanh_fast(H2<T16>::lo(p.x)), gelu_tanh_fast(H2<T16>::hi(p.x)));
                    p.y = H2<T16>::pack(gelu_tanh_fast(H2<T16>::lo(p.y)), gelu_tanh_fast(H2<T16>::hi(p.y)));
                }
                *(u32x2*)(patch + row * 128 + ((((nl >> 3) ^ (row & 7))) << 4) + (nl & 4) * 2) = p;
            }
        }
    // same-wave LDS accesses are ordered; the compiler inserts the lgkmcnt wait for the dependent reads
    const int n = nw + c16 * 8;
    const bool n_ok = n < a.N;            // epi_vec_ok: N % 8 == 0, so a started 8-column group is whole
    const int nc = n_ok ? n : 0;          // loads stay in range (and unconditional: four rows' worth in flight at a time)
#pragma unroll
    for (int it0 = 0; it0 < MB * 4; it0 += GRP) {
        u32x4 v[GRP], g[GRP], xo[GRP];
        int mrow[GRP];
        const int cur = (it0 / GRP) & 1;
        if (EPI == EPI_BIAS_QKNORM && it0 + GRP < MB * 4) rot_load(it0 + GRP, cur ^ 1);
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            const int row = (it0 + u) * 8 + (lane >> 3);
            mrow[u] = mw + row;
            v[u] = *(const u32x4*)(patch + row * 128 + ((c16 ^ (row & 7)) << 4));
            const int m = min(mrow[u], a.M - 1);
            if (EPI == EPI_BIAS_GATE_RES) {
                const int b = m / a.tok_per_batch;
                const int r = m - b * a.tok_per_batch;
                const void* gsel = r < a.text_len ? a.gate_txt : (a.gate_ref != nullptr && r < a.text_len + a.ref_len) ? a.gate_ref : a.gate_vid;
                g[u] = *(const u32x4*)((const bf16_t*)gsel + (size_t)b * a.gate_stride + nc);
                xo[u] = *(const u32x4*)((const bf16_t*)a.X + (size_t)m * a.ldx + nc);
            } else if (EPI == EPI_BIAS_ADD) {
                if (PRE) xo[u] = xpre[PRE ? it0 + u : 0];
                else xo[u] = *(const u32x4*)((const bf16_t*)a.R + (size_t)m * a.ldr + nc);
            }
        }
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            u32x4 o = v[u];
            if (EPI == EPI_BIAS_QKNORM && qk_head) {  // qk_norm_rope_k's arithmetic on the rounded projection, lane = (row, octet)
                float x[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[2 * e] = H2<T16>::lo(v[u][e]); x[2 * e + 1] = H2<T16>::hi(v[u][e]); }
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) s += x[e];
                s = oct_sum(s);
                const float mean = s * (1.0f / 64.0f);
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = x[e] - mean; q += d * d; }
                q = oct_sum(q);
                const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + a.qk_eps);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = H2<T16>::rnd(((x[e] - mean) * rstd * qw[e] + qb[e]));
                if (rope[cur][u]) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const float x0 = x[e], x1 = x[e + 1];
                        const float cc = rc[cur][u][e >> 1], ss = rs[cur][u][e >> 1];  // cos / sin of the pair (e, e + 1)
                        x[e] = H2<T16>::rnd((x0 * cc + (-x1) * ss));
                        x[e + 1] = H2<T16>::rnd((x1 * cc + x0 * ss));
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = H2<T16>::pack(x[2 * e], x[2 * e + 1]);
            }
            if (EPI == EPI_BIAS_GATE_RES) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t0 = H2<T16>::rnd((H2<T16>::lo(g[u][e]) * H2<T16>::lo(v[u][e])));
                    const float t1 = H2<T16>::rnd((H2<T16>::hi(g[u][e]) * H2<T16>::hi(v[u][e])));
                    o[e] = H2<T16>::pack(H2<T16>::lo(xo[u][e]) + t0, H2<T16>::hi(xo[u][e]) + t1);
                }
            } else if (EPI == EPI_BIAS_ADD) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = H2<T16>::pack(H2<T16>::lo(v[u][e]) + H2<T16>::lo(xo[u][e]),
                                   H2<T16>::hi(v[u][e]) + H2<T16>::hi(xo[u][e]));
            }
            if (EPI == EPI_BIAS_GELU && a.mx_out_q) {
                // MX output (GemmArgs::mx_out_q): the lane's 8 values are a quarter of a 32-column block (lanes c16 & ~3 .. + 3 of
                // the row); block scale = smallest power of two s with amax / s <= 448, elements = rne_e4m3(x / s)
                // |x| of non-negative bf16 order like their bit patterns: the amax of the 8 values is a packed 16-bit integer max, the
                // quad's by two DPP exchanges; the conversion divides by the block scale itself (v_cvt_scalef32_pk_fp8_bf16)
                typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
                u16x2_t pm = __builtin_bit_cast(u16x2_t, o[0] & 0x7fff7fffu);
#pragma unroll
                for (int e = 1; e < 4; ++e) pm = __builtin_elementwise_max(pm, __builtin_bit_cast(u16x2_t, o[e] & 0x7fff7fffu));
                int amb = (int)max((unsigned)pm[0], (unsigned)pm[1]) << 16;                               // bits of amax as fp32
                amb = max(amb, __builtin_amdgcn_update_dpp(0, amb, 0xB1, 0xf, 0xf, true));                  // lane ^ 1
                amb = max(amb, __builtin_amdgcn_update_dpp(0, amb, 0x4E, 0xf, 0xf, true));                  // lane ^ 2
                unsigned eb = (__float_as_uint(__int_as_float(amb) * (1.0f / 448.0f)) + 0x7fffffu) >> 23;  // biased exponent, rounded up unless am / 448 is a power of two
                eb = min(max(eb, 1u), 254u);
                const float bscale = __uint_as_float(eb << 23);                                             // 2^(eb - 127)
                // v_cvt_scalef32_pk_fp8_bf16 d, s, scale = rne_e4m3(s / scale) on a packed bf16 pair (tools/probes/mx_quant.hip: bit-identical
                // to unpack + multiply + v_cvt_pk_fp8_f32).  Inline asm: the builtin form was mis-compiled by this hipcc (ROCm 7.2:
                // every call of a group read the first call's source register).
                unsigned w0 = 0, w1 = 0;
                asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(w0) : "v"(o[0]), "v"(bscale));
                asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(w0) : "v"(o[1]), "v"(bscale));
                asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(w1) : "v"(o[2]), "v"(bscale));
                asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(w1) : "v"(o[3]), "v"(bscale));
                const unsigned e2 = eb | ((unsigned)__builtin_amdgcn_update_dpp(0, (int)eb, 0x104, 0xf, 0xf, true) << 8);  // lane c16 = 0: + lane 4's block
                if (mrow[u] < a.M && n_ok) {
                    *(u32x2*)(a.mx_out_q + (size_t)mrow[u] * a.N + n) = u32x2{w0, w1};
                    if (c16 == 0) *(unsigned short*)(a.mx_out_s + ((size_t)(n >> 7) * a.mx_rows + mx_perm_row(mrow[u])) * 4 + ((n >> 5) & 3)) = (unsigned short)e2;  // K-tile major, rows permuted
                }
            } else if (mrow[u] < a.M && n_ok) {
                if (EPI == EPI_BIAS_GATE_RES) *(u32x4*)((bf16_t*)a.X + (size_t)mrow[u] * a.ldx + n) = o;
                else *(u32x4*)((bf16_t*)a.C + (size_t)mrow[u] * a.ldc + n) = o;
            }
        }
    }
}
template <int EPI, int MB, bool SC = false, typename T16 = bf16_t>
__device__ __forceinline__ void epilogue_wave(const GemmArgs& a, const f32x16 (&acc)[2][MB], int mw, int nw, char* patch, int lane) {
    u32x2 bvec[8];
    load_bias64(a, nw, lane >> 5, bvec);
    epilogue_wave_b<EPI, MB, SC, 4, T16>(a, acc, mw, nw, patch, lane, bvec);
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


#ifdef S2V_TU_FP_CONTRACT_OFF
#pragma clang fp contract(off)   // a unit built with -ffp-contract=off stays that way after the include
#else
#pragma clang fp contract(fast)  // hipcc's default
#endif
