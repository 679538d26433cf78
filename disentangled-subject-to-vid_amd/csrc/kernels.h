// Internal launcher declarations (host side). Every launcher enqueues on `st` and returns 0 / <0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { S2V_F32 = 0, S2V_BF16 = 1, S2V_F16 = 2 };
// T = the storage type of `dtype`
#define S2V_DT_DISPATCH(dtype, ...)                                        \
    switch (dtype) {                                                       \
        case S2V_BF16: { using T = bf16_t; __VA_ARGS__; } break;           \
        case S2V_F16: { using T = f16_t; __VA_ARGS__; } break;             \
        default: { using T = float; __VA_ARGS__; } break;                  \
    }
enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_GATE_RES = 2 };

// C[m][n] = sum_k A[m][k] * W[n][k] (+ bias[n]) followed by an epilogue.
//   EPI_BIAS          : C = rnd(acc + bias)
//   EPI_BIAS_GELU     : C = rnd(gelu_tanh(rnd(acc + bias)))
//   EPI_BIAS_GATE_RES : X[m][n] = rnd(X[m][n] + rnd(gate(m)[n] * rnd(acc + bias)))   (C unused)
// gate(m): row m belongs to sample b = m / tok_per_batch; rows with (m % tok_per_batch) < text_len use
// gate_txt[b], the others (reference-image + video tokens) gate_vid[b]
// (reference: cogvideox_transformer_3d.py:165-167,182-184; normalization.py:484).
struct GemmArgs {
    const void* A; int lda;
    const void* W; int ldw;   // W is [N_pad, K], nn.Linear layout
    const void* bias;         // [N] or null
    void* C; int ldc;
    int M, N, K;
    void* X; int ldx;
    const void* gate_vid; const void* gate_txt; int gate_stride;
    int tok_per_batch; int text_len;
    // optional third gate for the reference-image rows [text_len, text_len + ref_len) of every sample (null: they take gate_vid,
    // which is what the shipped reference computes; set under lora_adaln_scope = 1, normalization.py:468-478)
    const void* gate_ref; int ref_len;
    // EPI_BIAS_ADD: C = rnd(rnd(acc + bias) + R[m][n])  (resnet skip, autoencoder_kl_cogvideox.py:318)
    const void* R; int ldr;
    // Implicit-GEMM convolution addressing of A (conv != 0).  A is a zero-bordered channels-last tensor
    // [frames][Hp][Wp][cin]; output row m = (f, y, x) of an [F][oH][oW] grid; k = (tap, ci) with tap = (dt, dy, dx),
    // dt in [0,kt), dy,dx in [0,3): A[m][k] = in[f + dt][y + dy][x + dx][ci].  kt = 3: causal conv (frames 0,1 of the
    // buffer hold the conv cache); kt = 1: per-frame 3x3 conv.  (CogVideoXCausalConv3d, autoencoder_kl_cogvideox.py:120-137)
    int conv, cin, Hp, Wp, oH, oW, kt;
    int cstride;        // conv mode: spatial stride of the output grid (0 or 1 = dense; 2 = CogVideoXDownsample3D, downsampling.py:322-353)
    int gm;             // 256-row kernel: row tiles per group of the tile order (0: chosen by the launcher)
    int tile;           // plain bf16 GEMMs: 0 = the launcher's choice (256 x 256 tiles where the shape fits them); 1 = 256 x 128 tiles, 2 = 128 x 128
                        // (few-tile GEMMs: more, smaller workgroups fill the CUs that a single partial round of 256 x 256 tiles leaves idle)
    int ablate;         // diagnostics only
    long long* clk;     // profile pass only: shader-clock stamp slot (common.h clk_stamp; gemm_g4 / g4t / g4f), else null
    int f16;            // launch_gemm_bf16: the 16-bit operands are fp16, not bf16 (set by launch_gemm_f16)
    int valu_only;      // fp32 only: stay on the VALU kernel gemm_simple_k (cfg.force_simple; the A/B reference of gemm_f32m, which returns the same bits)
    int a_rows_padded;  // plain mode: rows physically present behind A (>= M); the 256-row kernel needs ceil256(M)
    int m_begin;        // first output row of this launch (row-tail launches of a split GEMM; 128-row kernel only)
    int w_rows_padded;  // rows physically present behind W (>= N, zero or don't-care beyond N); 0 = exactly N
    // fp8 (OCP e4m3) operands, launch_gemm_fp8 only: A [M_pad][K] and W [N_pad][K] are bytes, lda / ldw in elements (= bytes);
    // C = rnd(a_scale[m] * w_scale[n] * acc + bias) ... : per-token and per-output-channel fp32 scales (weights.py / api.hip)
    const float* a_scale;
    const float* w_scale;
    // EPI_BIAS_QKNORM only: LayerNorm(64) weights / biases of q and k ([64] each), the PAIRED rotary table [positions][32 cos | 32 sin]
    // fp32 (value k = cos / sin of dims 2k and 2k+1; null: no rotary embedding), width of the q (= k = v) range, LayerNorm epsilon;
    // tok_per_batch and text_len above give the position of a row
    const void* qk_w[2]; const void* qk_b[2];
    const float* qk_cs;
    int qk_D; float qk_eps;
    // gemm_g4 only, split K (few output tiles, long K: the FF2 of the short-sequence geometries): splitk > 1 workgroups share an output
    // tile, each reduces K / splitk; sk_ws holds their fp32 partial tiles (tiles * splitk * 256 KiB), sk_cnt one arrival counter per
    // tile (zero before the launch, zero again after it).  The last workgroup to arrive adds the partials IN SPLIT ORDER (so the sum
    // does not depend on who was last) and runs the epilogue.
    int splitk; float* sk_ws; unsigned* sk_cnt;
    // fp8 GEMMs only, MX (block-scaled) activations: v_mfma_scale_f32_32x32x64_f8f6f4 takes one E8M0 scale per lane = per (row, 32
    // consecutive K elements), so a producer can quantise its output tile in place -- no row amax across workgroups.
    //   mx_out_q / mx_out_s (EPI_BIAS_GELU): the epilogue writes e4m3 bytes [M][N] and scale bytes [M][N / 32] (2^(s - 127), the
    //   smallest power of two with amax / scale <= 448) INSTEAD of the bf16 C;
    //   mx_a_s: A is such an image with its block scales; a_scale may then be null (= 1).
    //   Layout of the scales (round 4): K-TILE MAJOR -- [K / 128][mx_rows] dwords, dword (kt, m) = the four block scales of row m inside
    //   the 128 elements of K-tile kt (byte b = block 4 kt + b).  A GEMM's K-tile then finds the dwords of its 256 rows in ONE contiguous
    //   KiB (an LDS-DMA dword per lane over [M][K / 32] touched 64 cache lines per wave and K-tile: the MX loops ran at 3.3 k cycles per
    //   K-tile against 2.15 k for the unit-scale loop).  mx_rows = rows of that array (>= the padded M, a multiple of 128).
    //   Inside every 128-row half the rows are PERMUTED: row m sits at mx_perm_row(m) = (m & ~127) | (m & 31) << 2 | (m >> 5) & 3, so that
    //   the rows j * 32 + fr (j = 0..3) of the four A fragments of a lane are four consecutive dwords -- one 16-byte load.
    unsigned char* mx_out_q; unsigned char* mx_out_s;
    const unsigned char* mx_a_s;
    int mx_rows;
};
// fp8 x fp8 -> bf16 GEMM on v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales; the per-row scales above in the epilogue): the
// 256 x 256 ping-pong schedule of gemm_bf16_pp64 on K-tiles of 128 bytes.  Plain mode only (no conv), K % 128 == 0, N_pad % 256 == 0.
int launch_gemm_fp8(const GemmArgs& a, int epi, hipStream_t st);
// rows [M][K] of bf16 (ld elements apart) -> e4m3 bytes [M][K] + scale[m] = amax(row) / 448 (dynamic per-row quantisation)
int launch_quant_rows_fp8(const void* src, int64_t ld, int64_t M, int K, void* dst, float* scale, hipStream_t st);
enum { EPI_BIAS_ADD = 3 };
// EPI_BIAS_QKNORM: the fused QKV projection.  C = rnd(acc + bias); then, on the 64-column heads of the q and k ranges (columns
// < 2 * qk_D), the per-head LayerNorm(64) + affine and the rotary embedding of attention_processor.py:2060-2080 /
// embeddings.py:759-778, with the same arithmetic and rounding points as qk_norm_rope_k (which it replaces on the MFMA path: one
// pass over 2/3 of the QKV buffer less per layer).  Rows are tokens: r = m % tok_per_batch, rotary for r >= text_len.
enum { EPI_BIAS_QKNORM = 4 };

__host__ __device__ __forceinline__ int64_t mx_perm_row(int64_t m) { return (m & ~(int64_t)127) | ((m & 31) << 2) | ((m >> 5) & 3); }
int launch_gemm_bf16(const GemmArgs& a, int epi, hipStream_t st);              // MFMA path, bf16 only
// any dtype, any shape.  fp32 calls that qualify (gemm_f32m_ok) run on the fp32 matrix pipe (gemm_f32m.hip) unless a.valu_only -- same bits
int launch_gemm_simple(const GemmArgs& a, int epi, int dtype, hipStream_t st);
// fp16 operands (the fp16 model dtype) on v_mfma_f32_32x32x16_f16: the 128 x 128 x 64 kernel of gemm.hip with its lane-local epilogue
bool gemm_f16_ok(const GemmArgs& a, int epi);
int launch_gemm_f16(const GemmArgs& a, int epi, hipStream_t st);
// gemm_f32m.hip: fp32 operands on v_mfma_f32_32x32x2_f32, bit-identical to gemm_simple_k<float>
bool gemm_f32m_ok(const GemmArgs& a);
int launch_gemm_f32m(const GemmArgs& a, int epi, hipStream_t st);
// gemm_g4.hip: 256 x 256 tiles, four waves, generated-asm K loop (plain bf16 operands; gemm_g4_ok says whether a call qualifies)
bool gemm_g4_ok(const GemmArgs& a, int epi);
int launch_gemm_g4(const GemmArgs& a, int epi, hipStream_t st);
// the same loop on fp16 operands (v_mfma_f32_32x32x16_f16) with the fp16 form of the vector epilogue: the fp16 model dtype's big linears
bool gemm_g4_f16_ok(const GemmArgs& a, int epi);
int launch_gemm_g4_f16(const GemmArgs& a, int epi, hipStream_t st);
// gemm_g4f.hip: the four-wave loop on e4m3 operands (launch_gemm_fp8 routes to it where gemm_g4f_ok)
bool gemm_g4f_ok(const GemmArgs& a, int epi);
int launch_gemm_g4f(const GemmArgs& a, int epi, hipStream_t st);
// gemm_g4t.hip: the same tiles as a persistent kernel whose epilogue is trickled through the next tile's K loop (gen_gemm_g4t.py);
// bias / bias + GELU epilogues on whole 256 x 256 tiles with at least two rounds of them on `ncu` CUs
bool gemm_g4t_ok(const GemmArgs& a, int epi, int ncu);
int launch_gemm_g4t(const GemmArgs& a, int epi, hipStream_t st);
// Few tiles and a long reduction (C1: the FF2 is 80 tiles of 120 K-tiles -- one K loop is 130 us however many CUs idle; the T5 encoder
// at M = 452: 32-96 tiles): split K over S workgroups per tile (GemmArgs::splitk), S the largest count <= 4 that still fits one round
// of `ncu` CUs and leaves an even number >= 16 of K-tiles per workgroup (below that the fp32 partial traffic costs what the shorter
// loop saves: measured on the C1 out-projection).  1 = do not split.  Workspace: S * tiles * 256 KiB of partials + tiles counters (zero).
int gemm_choose_splitk(int64_t tiles, int K, int64_t ncu);
// generic strided fp32 GEMM used at load time (LoRA merge): C[m,n] += alpha * sum_k A[m*sam+k*sak]*B[n*sbn+k*sbk]
int launch_gemm_strided_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk,
                            float* C, int64_t ldc, int M, int N, int K, float alpha, hipStream_t st);

// ---- attention ---------------------------------------------------------------
// qkv: [B*Ntok (+pad), 3*D] rows = tokens; q at col h*64, k at col D+h*64, v at col 2D+h*64.
// out: [B*Ntok, D] (col h*64+d).  softmax(q k^T / 8) v, no mask  (attention_processor.py:2083-2087)
struct AttnArgs {
    const void* qkv; int ld_qkv;
    const void* vt;            // bf16 path only: V^T [B][H][64][ntok_pad] (kv-permuted inside 16-groups)
    int ntok_pad;
    void* out; int ld_out;
    int B, H, Ntok;
    float scale;               // 1/sqrt(64)
    // bf16 path, optional: nine zero-initialised ints owned by the caller (one set per concurrently running launch) and the CU count
    // -> the persistent, work-pulling launch (attention.hip: attn_pp_persist_k); null -> one workgroup per q-block
    int* queue; int num_cus;
    // fp8 engine, attn_q4 only: write the output as MX e4m3 [B*Ntok][ld_out] bytes + block scales (K-tile major, GemmArgs::mx_a_s:
    // [ld_out / 128][mx_rows] dwords) INSTEAD of the bf16 `out` (the out-projection reads it through GemmArgs::mx_a_s)
    unsigned char* mx_q; unsigned char* mx_s; int mx_rows;
    // fp8 QK^T (weight_format 2, attn_q4f): MX e4m3 images of q (pre-multiplied by scale * log2 e) and k, written by launch_qk_quant_mx:
    //   q8 [B][H][Ntok][64] bytes, q8s [B][H][Ntok] (byte 0 / 1 = E8M0 scale of head-dim block 0 / 1),
    //   k8 [B][H][ntok_pad][64] bytes (rows past Ntok zero), k8s [B][H][ntok_pad / 64][64] dwords: dword (hi * 32 + r) of a 64-key tile holds in
    //   byte kb the E8M0 scale of (key 32 kb + r, block hi) -- the scale VGPR of the QK^T MFMA, loaded as is
    const unsigned char* q8; const unsigned short* q8s; const unsigned char* k8; const unsigned* k8s;
    // attn_p_format 1 (attn_q4h / attn_q4fh): `vt` holds fp16 (launch_v_transpose(..., to_f16 = true)), P is built as fp16 pairs and summed by
    // packed fp16 adds, P.V runs on v_mfma_f32_32x32x16_f16; deferred maximum 2^14 instead of 2^64 (gen_attn_q4.py, P16).  attn_q4 forms only.
    int p16;
    // optional census of the deferred-maximum slow path (attn_q4 forms): 256 slots of two counters, slot = workgroup & 255:
    // [2 s] += slow paths taken, [2 s + 1] += (wave, KV tile) pairs run.  The engine reads it to decide whether fp16 P pays on the data at hand.
    unsigned long long* stats;
    long long* clk;            // profile pass only: shader-clock stamp slot (common.h clk_stamp; the four-wave kernels), else null
    int order;                 // persistent four-wave launch: 0 = a head's q-blocks on one XCD (default), 1 = dealt over the XCDs (experiment, attn_qx_persist_k)
    int stagger;               // persistent four-wave launch: workgroup (blockIdx >> 3) = slot s of its XCD starts s * stagger * 64 cycles late (0 = together); see attn_qx_persist_k
    int valu_only;             // fp32 only: stay on attn_simple_k (cfg.force_simple) instead of the fp32-MFMA kernel attn_f32m_k
};
// sequences up to this length run attn_pp (launch_attn_bf16), longer ones attn_q4 (profiles/r03_attn_short_sequences.txt: attn_pp 9 % ahead at
// 4000 tokens, attn_q4 2 % ahead at 6000, 7 % at 8192); the fp8 engine's MX output is attn_q4's at any length
#ifndef ATTN_PP_MAX_TOKENS
#define ATTN_PP_MAX_TOKENS 4608
#endif
static inline bool attn_runs_q4(int Ntok, bool mx_out) { return mx_out || Ntok > ATTN_PP_MAX_TOKENS; }
// q / k of the (normalised, rotated) QKV buffer -> the MX e4m3 images above (elementwise.hip)
int launch_qk_quant_mx(const void* qkv, int ld_qkv, int B, int H, int Ntok, int ntok_pad, float q_prescale, unsigned char* q8, unsigned short* q8s,
                       unsigned char* k8, unsigned* k8s, hipStream_t st);
// the same images from the RAW q / k of the projection: per-head LayerNorm + rotary (qk_norm_rope_k's arithmetic) folded into the quantisation pass;
// a.vt is ignored (V^T stays launch_v_transpose's); bf16 only (elementwise.hip)
struct QkNormRopeArgs;
int launch_qk_norm_quant_mx(const QkNormRopeArgs& a, float q_prescale, unsigned char* q8, unsigned short* q8s, unsigned char* k8, unsigned* k8s, hipStream_t st);
// attn_q4 with QK^T on the scaled fp8 MFMA (attention_q4.hip); needs a.q8 / q8s / k8 / k8s and a.vt
int launch_attn_q4f(const AttnArgs& a, bool persistent, hipStream_t st);
// attn_q4 with P / V^T in fp16 (AttnArgs::p16 = 1)
int launch_attn_q4h(const AttnArgs& a, bool persistent, hipStream_t st);
int launch_attn_bf16(const AttnArgs& a, hipStream_t st);
// generic kernel, any dtype; fp32 and fp16 calls run attn_f32m (attention_f32m.hip: QK^T and P.V on v_mfma_f32_32x32x2_f32) unless a.valu_only
int launch_attn_simple(const AttnArgs& a, int dtype, hipStream_t st);
int launch_attn_f32m(const AttnArgs& a, int dtype, hipStream_t st);
// fp16 storage on v_mfma_f32_32x32x16_f16 (the lock-step kernel of attention.hip): what the fp16 engine runs; needs a.vt (fp16 V^T)
int launch_attn_f16(const AttnArgs& a, hipStream_t st);
// the four-wave asm kernel with fp16 q / k / V^T / P (attention_q4.hip, attn_q4hh); launch_attn_f16 routes long sequences to it
int launch_attn_q4hh(const AttnArgs& a, bool persistent, hipStream_t st);  // dtype S2V_F32 or S2V_F16 (storage; the arithmetic is fp32 either way)
// four-wave form of the bf16 kernel (attention_q4.hip); persistent needs a.queue / a.num_cus
int launch_attn_q4(const AttnArgs& a, bool persistent, hipStream_t st);
// eight waves x 32 rows running the same fine-grained stream, two waves per SIMD
int launch_attn_q8(const AttnArgs& a, bool persistent, hipStream_t st);

// ---- elementwise / normalisation ------------------------------------------------
// LayerNorm(eps, affine w,b) then (1+scale)*y+shift with per-row-range modulation sets.
// rows: [B*Ntok, D]; text rows use (shift_txt, scale_txt), other rows (shift_vid, scale_vid); all [B][mod_stride].
struct LnModArgs {
    const void* x; int ldx; void* y; int ldy;
    const void* w; const void* b; float eps;
    const void* shift_vid; const void* scale_vid; const void* shift_txt; const void* scale_txt; int mod_stride;
    int B, Ntok, text_len, D;
    const void* shift_ref; const void* scale_ref; int ref_len;  // optional set for rows [text_len, text_len + ref_len); null: vid
    // fp8 linears (bf16 storage only): when q8 is set the modulated row is NOT stored as bf16 but quantised on the spot exactly as
    // quant_rows_fp8_k would quantise its bf16 image -- e4m3 bytes [row][D] and scale[row] = amax / 448 -- for the projection that
    // consumes it (one read + one write pass over the activations less per LayerNorm)
    void* q8; float* q8_scale;
};
int launch_ln_modulate(const LnModArgs& a, int dtype, hipStream_t st);

// per-head LayerNorm(64, eps 1e-6, affine) on q,k (in place in qkv) + interleaved-pair RoPE on rows >= text_len,
// optional V^T production (bf16 path).  cos/sin: [Ntok - text_len, 64] fp32 (ref rows then video rows), null => no RoPE.
struct QkNormRopeArgs {
    void* qkv; int ld_qkv; int B, H, Ntok, text_len;
    const void* nq_w; const void* nq_b; const void* nk_w; const void* nk_b; float eps;
    const float* cos; const float* sin;
    void* vt; int ntok_pad;   // null => skip
    int vt_f16;               // V^T as fp16 (AttnArgs::p16)
};
int launch_qk_norm_rope(const QkNormRopeArgs& a, int dtype, hipStream_t st);
// bf16 V [B*Ntok, ld] (cols 2D + h*64 + d) -> V^T [B][H][64][ntok_pad] in the k-slot order attn_bf16 consumes
int launch_v_transpose(const void* qkv, int ld_qkv, int B, int H, int Ntok, void* vt, int ntok_pad, hipStream_t st, bool to_f16 = false);

// timestep sinusoid -> Linear -> SiLU -> Linear  (embeddings.py:27-78, 864-876)
int launch_time_embed(const float* t_dev, int B, int D, const void* w1, const void* b1, const void* w2, const void* b2,
                      int temb_dim, void* tmp /*[B,temb]*/, void* emb_out /*[B,temb]*/, int dtype, hipStream_t st);
// out[j][b][:] = W_j . silu(emb[b]) + bias_j for a batch of stacked linears: W [rows_total, temb], bias [rows_total]
int launch_mod_gemv(const void* emb, int B, int temb_dim, const void* W, const void* bias, int64_t rows_total,
                    void* out /*[B][rows_total]*/, int dtype, hipStream_t st, bool rowwise = false);

// latents [Bn, F, C, H, W] -> patches [Bn*F*(H/2)*(W/2), C*4] with feature order (c, py, px);
// lat_bstride = elements between samples (0 => every sample reads the same latent: the CFG pair)
int launch_patchify(const void* lat, int64_t lat_bstride, int Bn, int F, int C, int H, int W, void* out, int dtype,
                    hipStream_t st);
// y [B*V, C*4] (feature c*4+py*2+px) -> out [B, F, C, H, W]; rows of y start at y_row0 with batch stride y_bstride rows
int launch_unpatchify(const void* y, int ldy, int64_t y_bstride, void* out, int B, int F, int C, int H, int W, int dtype,
                      hipStream_t st);
// dst[r][:] = rnd(src[r][:] (+ add[r][:]))
int launch_copy_rows(const void* src, int lds_, const void* add, int ldadd, void* dst, int ldd, int rows, int D,
                     int dtype, hipStream_t st);
// final: y = LN2(LN1(x)) * (1+scale[b]) + shift[b]   on video rows  (cogvideox_transformer_3d.py:536-542)
struct TailNormArgs {
    const void* x; int ldx; void* y; int ldy;
    const void* w1; const void* b1; const void* w2; const void* b2; float eps;
    const void* shift; const void* scale; int mod_stride;
    int B, Ntok, row0, V, D;
};
int launch_tail_norm(const TailNormArgs& a, int dtype, hipStream_t st);

// CFG + scheduler step (custom_cogvideox_pipe.py:266-296; scheduling_{ddim,dpm}_cogvideox.py)
// Per-step scalars live in DEVICE memory so one captured hipGraph serves every step.
//   x0 = rnd(c_x0_x * x) - c_x0_v * v
//   kind 0 (DDIM)          : out = rnd(a_t * x) + b_t * x0
//   kind 1 (DPM first/last): out = (rnd(m1 * x) - m2 * x0) + rnd(mn * noise)
//   kind 2 (DPM multistep) : d = m3 * x0 - m4 * x0_old ; out = (rnd(m1 * x) - m2 * d) + rnd(mn * noise)
struct SchedCoef {
    int kind; float guidance;
    float c_x0_x, c_x0_v, a_t, b_t, m1, m2, m3, m4, mn;
    float pad;
};
struct SchedArgs {
    const void* noise_pred;   // [2, n] model dtype (uncond, cond) or [1, n] when !cfg
    const void* latents_in;   // [n] model dtype
    void* latents_out;        // [n] model dtype (may alias latents_in)
    float* x0_hist;           // [n] fp32: read as x0_old (kind 2), overwritten with this step's x0 (may be null for DDIM)
    const void* noise;        // [n] model dtype, DPM only
    int64_t n; int cfg;
    const SchedCoef* coef;    // device pointer, or null => use cval
    SchedCoef cval;
    int np_f32;               // noise_pred is fp32 (the scheduler-object seam: model_output after .float())
    int out_f32;              // latents_out is fp32 and un-rounded (scheduler.step's return value)
};
int launch_sched_step(const SchedArgs& a, int dtype, hipStream_t st);
// dst[i] = (Tdst) src[i]
int launch_convert(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, hipStream_t st);
// strided 2-D convert-copy: dst[r*ldd + c] = src[r*lds + c]
int launch_convert2d(const void* src, int src_dtype, int64_t lds_, void* dst, int dst_dtype, int64_t ldd, int64_t rows,
                     int64_t cols, hipStream_t st);
