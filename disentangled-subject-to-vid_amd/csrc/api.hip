// C ABI of libs2v_hip.so (see include/s2v_hip.h): context, weight ingest / re-packing, workspace, and the launch
// sequences of the transformer forward, the block / attention seams and the fused denoise step.
#define S2V_HOST
#include "common.h"
#include "kernels.h"
#include "../../include/s2v_hip.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

thread_local std::string g_s2v_err;
int s2v_fail(const char* file, int line, const char* msg, int code) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s:%d: %s", file, line, msg);
    g_s2v_err = buf;
    return code;
}

static inline int64_t rup(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

struct Slot {
    char* dst = nullptr;   // destination inside the arena
    int64_t rows = 0, cols = 0, ld = 0;
    bool loaded = false;
};

struct LayerW {
    char *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    char *wqkv, *bqkv, *nq_w, *nq_b, *nk_w, *nk_b, *wo, *bo;
    char *w1, *b1, *w2, *b2;
    // weight_format 1: e4m3 copies [N_pad][K] of the four big linears + per-output-channel scales (quantised at finalize)
    char *q_qkv = nullptr, *q_o = nullptr, *q_1 = nullptr, *q_2 = nullptr;
    float *s_qkv = nullptr, *s_o = nullptr, *s_1 = nullptr, *s_2 = nullptr;
};

struct GraphKey {
    void* latents; float* x0; const void* noise;
    int slot;   // -1: the whole step (s2v_denoise_step); 0 / 1: the forward-only graph of s2v_denoise_split_begin writing that half of the CFG pair
    bool operator==(const GraphKey& o) const { return latents == o.latents && x0 == o.x0 && noise == o.noise && slot == o.slot; }
};

struct s2v_ctx {
    s2v_model_config cfg;
    int D = 0, L = 0, dtype = 0, esz = 0, temb = 0;
    bool mfma = false;
    int attn_order = 0;          // AttnArgs::order (S2V_ATTN_ORDER at s2v_create of the DIAGNOSTICS build: an experiment knob, 0 in the product)
    int attn_stagger = 0;        // AttnArgs::stagger (S2V_ATTN_STAGGER, same: 0 in the product)
    bool h16 = false;            // fp16 model dtype: linears on v_mfma_f32_32x32x16_f16 (gemm_f16), attention on attn_f32m<f16_t>
    int mc = 6;                  // modulation chunks per norm{1,2}.linear in the stack: 6, or 9 under lora_adaln_scope = 1 (+ the
                                 // reference-image copy of chunks 0-2: cond_shift, cond_scale, cond_gate)
    float* lora_tmp = nullptr;   // fp32 scratch of s2v_merge_lora
    size_t lora_tmp_bytes = 0;
    bool fp8 = false;            // cfg.weight_format == 1 or 2
    bool attn_p16 = false;       // cfg.attn_p_format == 1: P / V^T of the four-wave attention kernels in fp16 (AttnArgs::p16)
    bool fp8_qk = false;         // cfg.weight_format == 2: additionally q / k as MX e4m3 and QK^T on the scaled fp8 MFMA (attn_q4f)
    unsigned char *q8 = nullptr, *k8 = nullptr; unsigned short* q8s = nullptr; unsigned* k8s = nullptr;  // workspace (fp8_qk): AttnArgs::q8 ... k8s
    char* aq = nullptr;          // workspace: e4m3 activations [Mpad][4D] of the GEMM being fed
    float* aq_scale = nullptr;   // workspace: their per-token scales [Mpad]
    unsigned char *hq = nullptr, *hs = nullptr;  // workspace (fp8): GELU(FF1) as MX e4m3 [Mpad][4D] + block scales [Mpad][4D / 32]
    bool finalized = false;
    // weights
    int num_cus = 256;
    int sk_tiles = 0; float* sk_ws = nullptr; unsigned* sk_cnt = nullptr;  // split-K workspace of the geometry (0: none)
    int* attn_queue = nullptr;               // nine counters of the persistent attention launch (zero between launches)
    unsigned long long* attn_stats = nullptr;  // AttnArgs::stats: 256 x [slow paths, (wave, KV tile) pairs] of the attn_q4 launches (s2v_attn_slow_stats)
    hipStream_t side = nullptr;              // fork/join stream for the row-tail launches of split GEMMs
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    char* arena = nullptr;
    int64_t arena_bytes = 0;
    std::unordered_map<std::string, Slot> slots;
    std::vector<LayerW> layers;
    char *patch_w, *patch_b, *text_w, *text_b, *te1_w, *te1_b, *te2_w, *te2_b;
    char *nf_w, *nf_b, *no_w, *no_b, *po_w, *po_b;
    char *mod_w, *mod_b;
    int64_t mod_rows = 0;
    // geometry + workspace
    int B = 0, T = 0, F = 0, H = 0, W = 0, R = 0, V = 0, Ntok = 0, ntok_pad = 0;
    int64_t M = 0, Mpad = 0;
    char* ws = nullptr;
    int64_t ws_bytes = 0;
    char *X, *Xn, *QKV, *Hb, *VT, *e0, *e1, *patches, *tailn, *proj, *mod, *tmp_te, *emb, *noise_pred;
    float *rope_cos, *rope_sin;
    float* rope_pk = nullptr;                // [positions][cos pair 0..31 | sin pair 0..31] when the tables repeat every value twice
    bool rope_paired = false;                // (get_3d_rotary_pos_embed: repeat_interleave(2)); the fused QKV epilogue needs it
    char* pos_tab;
    bool have_rope = false, have_pos = false, have_cond = false;
    float* t_dev = nullptr;
    SchedCoef* coef_dev = nullptr;
    // pinned staging ring for per-step scalars
    struct Stage { float t[4]; SchedCoef c; };
    Stage* ring = nullptr;
    int ring_pos = 0;
    // graph
    hipStream_t cap_stream = nullptr;
    hipGraphExec_t gexec = nullptr;
    GraphKey gkey{nullptr, nullptr, nullptr, -1};
    int split_kind = -1;         // scheduler kind of the s2v_denoise_split_begin that has not met its s2v_denoise_split_end yet (-1: none pending)
    // optional per-kernel-class timing with HIP events on the launch stream (bench.py's live roofline figure)
    // + shader-clock stamps (s_memtime / s_memrealtime pairs written by a one-lane kernel right before and right after every profiled launch)
    long long* clk_buf = nullptr;                    // device: CLK_SLOTS x [memtime0, realtime0, memtime1, realtime1]
    long long* clk_cur = nullptr;                    // the slot of the launch a ProfScope is open around (null outside the profile pass)
    std::vector<std::pair<int, int>> clk_rec;         // (class, slot) of every stamped launch since the last s2v_profile_read_clocks
    bool prof_on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev[8];
    size_t prof_used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

static const int CLK_SLOTS = 8192;
// shader clock of the profiled launches: ProfScope hands the launch a slot of clk_buf (s2v_ctx::clk_cur -> GemmArgs::clk / AttnArgs::clk); one
// designated workgroup of the kernel stamps s_memtime / s_memrealtime at its entry and exit (common.h clk_stamp).  Kernels without stamps (or
// launches whose designated workgroup left early) leave the slot zero and are skipped.
#define S2V_FP8_QK_AUTO_TOKENS 40000
enum { PK_QKV = 0, PK_ATTN = 1, PK_OUT = 2, PK_FF1 = 3, PK_FF2 = 4, PK_LNMOD = 5, PK_QKNORM = 6, PK_OTHER = 7, PK_NUM = 8 };

struct ProfScope {
    s2v_ctx* c; int k; hipStream_t st; bool on;
    ProfScope(s2v_ctx* c_, int k_, hipStream_t st_) : c(c_), k(k_), st(st_) {
        on = c->prof_on && st != c->cap_stream;
        if (!on) return;
        if (c->prof_used[k] == c->prof_ev[k].size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            c->prof_ev[k].push_back({a, b});
        }
        if (c->clk_buf && (int)c->clk_rec.size() < CLK_SLOTS) {
            slot = (int)c->clk_rec.size();
            c->clk_rec.push_back({k, slot});
            c->clk_cur = c->clk_buf + 4 * slot;
        }
        (void)hipEventRecord(c->prof_ev[k][c->prof_used[k]].first, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(c->prof_ev[k][c->prof_used[k]].second, st);
        c->clk_cur = nullptr;
        c->prof_used[k]++;
    }
    int slot = -1;
};

static const int RING = 256;

// ------------------------------------------------------------------------------------------------------
extern "C" const char* s2v_last_error(void) { return g_s2v_err.c_str(); }
extern "C" const char* s2v_version(void) { return "s2v_hip 0.1 (gfx950)"; }

static Slot* add_slot(s2v_ctx* c, const std::string& name, char* dst, int64_t rows, int64_t cols, int64_t ld) {
    Slot s;
    s.dst = dst; s.rows = rows; s.cols = cols; s.ld = ld;
    c->slots[name] = s;
    return &c->slots[name];
}

extern "C" int s2v_create(const s2v_model_config* cfg, s2v_ctx** out) {
    S2V_REQUIRE(cfg && out, "s2v_create: null argument");
    S2V_REQUIRE(cfg->dtype == S2V_DTYPE_F32 || cfg->dtype == S2V_DTYPE_BF16 || cfg->dtype == S2V_DTYPE_F16, "s2v_create: unsupported dtype");
    S2V_REQUIRE(cfg->patch_size == 2, "s2v_create: patch_size must be 2");
    S2V_REQUIRE(cfg->num_layers > 0 && cfg->num_heads > 0, "s2v_create: bad model size");
    S2V_REQUIRE(cfg->in_channels * 4 <= 4096 && cfg->out_channels > 0, "s2v_create: bad channel count");
    s2v_ctx* c = new s2v_ctx();
    c->cfg = *cfg;
    c->D = cfg->num_heads * 64;
    c->L = cfg->num_layers;
    c->dtype = cfg->dtype;
    c->esz = cfg->dtype == S2V_DTYPE_F32 ? 4 : 2;
    c->temb = cfg->time_embed_dim;
    {
        int dev = 0, ncu = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0)
            c->num_cus = ncu;
    }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        s2v_destroy(c);
        return s2v_fail(__FILE__, __LINE__, "s2v_create: stream / event creation failed", -2);
    }
    c->mfma = (cfg->dtype == S2V_DTYPE_BF16) && !cfg->force_simple;
    c->h16 = (cfg->dtype == S2V_DTYPE_F16) && !cfg->force_simple;
#ifdef S2V_DIAG  // experiment knobs of the diagnostics build only; clamped: slot * stagger * 64 cycles of s_sleep per launch
    if (const char* e = getenv("S2V_ATTN_STAGGER")) c->attn_stagger = std::min(std::max(atoi(e), 0), 64);
    if (const char* e = getenv("S2V_ATTN_ORDER")) c->attn_order = atoi(e) ? 1 : 0;
#endif
    if (hipMalloc((void**)&c->attn_stats, 4096) != hipSuccess || hipMemset(c->attn_stats, 0, 4096) != hipSuccess) {
        s2v_destroy(c);
        return s2v_fail(__FILE__, __LINE__, "s2v_create: attention census allocation failed", -2);
    }
    if (hipMalloc((void**)&c->attn_queue, 64) != hipSuccess || hipMemset(c->attn_queue, 0, 64) != hipSuccess) {
        s2v_destroy(c);
        return s2v_fail(__FILE__, __LINE__, "s2v_create: attention queue allocation failed", -2);
    }
    if (c->D > 4096) { s2v_destroy(c); return s2v_fail(__FILE__, __LINE__, "s2v_create: D > 4096 unsupported", -1); }
    if (c->temb % 8 != 0 || c->D % 8 != 0) { s2v_destroy(c); return s2v_fail(__FILE__, __LINE__, "bad dims", -1); }

    const int64_t D = c->D, E = c->esz, L = c->L, TE = c->temb;
    const int64_t Kp = cfg->in_channels * 4, Cout = cfg->out_channels * 4, TX = cfg->text_embed_dim;
    // two passes: size, then carve
    int64_t off = 0;
    auto carve = [&](int64_t elems) { int64_t o = off; off += rup(elems * E, 256); return o; };
    struct Offs { int64_t ln1_w, ln1_b, ln2_w, ln2_b, wqkv, bqkv, nq_w, nq_b, nk_w, nk_b, wo, bo, w1, b1, w2, b2; };
    std::vector<Offs> lo(L);
    const int64_t Dp = rup(D, 256);  // weight rows are padded to the 256-column GEMM tile (zero rows)
    for (int l = 0; l < L; ++l) {
        Offs& o = lo[l];
        o.ln1_w = carve(D); o.ln1_b = carve(D); o.ln2_w = carve(D); o.ln2_b = carve(D);
        o.wqkv = carve(rup(3 * D, 256) * D); o.bqkv = carve(3 * D);
        o.nq_w = carve(64); o.nq_b = carve(64); o.nk_w = carve(64); o.nk_b = carve(64);
        o.wo = carve(Dp * D); o.bo = carve(D);
        o.w1 = carve(rup(4 * D, 256) * D); o.b1 = carve(4 * D);
        o.w2 = carve(Dp * 4 * D); o.b2 = carve(D);
    }
    const int64_t o_patch_w = carve(Dp * Kp), o_patch_b = carve(D);
    const int64_t o_text_w = carve(Dp * TX), o_text_b = carve(D);
    const int64_t o_te1_w = carve(TE * D), o_te1_b = carve(TE), o_te2_w = carve(TE * TE), o_te2_b = carve(TE);
    const int64_t o_nf_w = carve(D), o_nf_b = carve(D), o_no_w = carve(D), o_no_b = carve(D);
    const int64_t o_po_w = carve(rup(Cout, 256) * D), o_po_b = carve(Cout);
    if (cfg->lora_adaln_scope != 0 && cfg->lora_adaln_scope != 1) { s2v_destroy(c); return s2v_fail(__FILE__, __LINE__, "s2v_create: lora_adaln_scope must be 0 or 1", -1); }
    c->mc = cfg->lora_adaln_scope ? 9 : 6;
    const int64_t MC = c->mc;
    c->mod_rows = 2 * L * MC * D + 2 * D;
    const int64_t o_mod_w = carve(c->mod_rows * TE), o_mod_b = carve(c->mod_rows);
    // fp8 copies live in the same arena (one broadcast replicates everything a replica needs)
    c->fp8 = cfg->weight_format >= 1 && cfg->weight_format <= 3;
    c->fp8_qk = cfg->weight_format == 2;
    c->attn_p16 = cfg->attn_p_format == 1;
    if (cfg->attn_p_format != 0 && cfg->attn_p_format != 1) { s2v_destroy(c); return s2v_fail(__FILE__, __LINE__, "s2v_create: attn_p_format must be 0 (bf16) or 1 (fp16)", -1); }
    if (cfg->weight_format < 0 || cfg->weight_format > 3) { s2v_destroy(c); return s2v_fail(__FILE__, __LINE__, "s2v_create: weight_format must be 0, 1, 2 or 3", -1); }
    if (c->fp8 && (!c->mfma || D % 128 != 0)) { s2v_destroy(c); return s2v_fail(__FILE__, __LINE__, "s2v_create: weight_format 1 / 2 (fp8) needs the bf16 MFMA path and inner_dim % 128 == 0", -1); }
    struct QOffs { int64_t q_qkv, q_o, q_1, q_2, s_qkv, s_o, s_1, s_2; };
    std::vector<QOffs> qo(c->fp8 ? L : 0);
    auto carve_b = [&](int64_t bytes) { int64_t o = off; off += rup(bytes, 256); return o; };
    for (auto& q : qo) {
        q.q_qkv = carve_b(rup(3 * D, 256) * D); q.q_o = carve_b(Dp * D); q.q_1 = carve_b(rup(4 * D, 256) * D); q.q_2 = carve_b(Dp * 4 * D);
        q.s_qkv = carve_b(rup(3 * D, 256) * 4); q.s_o = carve_b(Dp * 4); q.s_1 = carve_b(rup(4 * D, 256) * 4); q.s_2 = carve_b(Dp * 4);
    }
    c->arena_bytes = off;
    hipError_t e = hipMalloc((void**)&c->arena, c->arena_bytes);
    if (e != hipSuccess) { s2v_destroy(c); return s2v_fail(__FILE__, __LINE__, hipGetErrorString(e), -2); }
    e = hipMemset(c->arena, 0, c->arena_bytes);
    if (e != hipSuccess) { s2v_destroy(c); return s2v_fail(__FILE__, __LINE__, hipGetErrorString(e), -2); }
    char* A = c->arena;
    c->layers.resize(L);
    char nm[160];
    for (int l = 0; l < L; ++l) {
        const Offs& o = lo[l];
        LayerW& w = c->layers[l];
        w.ln1_w = A + o.ln1_w; w.ln1_b = A + o.ln1_b; w.ln2_w = A + o.ln2_w; w.ln2_b = A + o.ln2_b;
        w.wqkv = A + o.wqkv; w.bqkv = A + o.bqkv; w.nq_w = A + o.nq_w; w.nq_b = A + o.nq_b;
        w.nk_w = A + o.nk_w; w.nk_b = A + o.nk_b; w.wo = A + o.wo; w.bo = A + o.bo;
        w.w1 = A + o.w1; w.b1 = A + o.b1; w.w2 = A + o.w2; w.b2 = A + o.b2;
        if (c->fp8) {
            const QOffs& q = qo[l];
            w.q_qkv = A + q.q_qkv; w.q_o = A + q.q_o; w.q_1 = A + q.q_1; w.q_2 = A + q.q_2;
            w.s_qkv = (float*)(A + q.s_qkv); w.s_o = (float*)(A + q.s_o); w.s_1 = (float*)(A + q.s_1); w.s_2 = (float*)(A + q.s_2);
        }
#define NM(fmt) (snprintf(nm, sizeof(nm), "transformer_blocks.%d." fmt, l), std::string(nm))
        add_slot(c, NM("norm1.norm.weight"), w.ln1_w, 1, D, D);
        add_slot(c, NM("norm1.norm.bias"), w.ln1_b, 1, D, D);
        add_slot(c, NM("norm2.norm.weight"), w.ln2_w, 1, D, D);
        add_slot(c, NM("norm2.norm.bias"), w.ln2_b, 1, D, D);
        add_slot(c, NM("attn1.to_q.weight"), w.wqkv, D, D, D);
        add_slot(c, NM("attn1.to_k.weight"), w.wqkv + D * D * E, D, D, D);
        add_slot(c, NM("attn1.to_v.weight"), w.wqkv + 2 * D * D * E, D, D, D);
        add_slot(c, NM("attn1.to_q.bias"), w.bqkv, 1, D, D);
        add_slot(c, NM("attn1.to_k.bias"), w.bqkv + D * E, 1, D, D);
        add_slot(c, NM("attn1.to_v.bias"), w.bqkv + 2 * D * E, 1, D, D);
        add_slot(c, NM("attn1.norm_q.weight"), w.nq_w, 1, 64, 64);
        add_slot(c, NM("attn1.norm_q.bias"), w.nq_b, 1, 64, 64);
        add_slot(c, NM("attn1.norm_k.weight"), w.nk_w, 1, 64, 64);
        add_slot(c, NM("attn1.norm_k.bias"), w.nk_b, 1, 64, 64);
        add_slot(c, NM("attn1.to_out.0.weight"), w.wo, D, D, D);
        add_slot(c, NM("attn1.to_out.0.bias"), w.bo, 1, D, D);
        add_slot(c, NM("ff.net.0.proj.weight"), w.w1, 4 * D, D, D);
        add_slot(c, NM("ff.net.0.proj.bias"), w.b1, 1, 4 * D, 4 * D);
        add_slot(c, NM("ff.net.2.weight"), w.w2, D, 4 * D, 4 * D);
        add_slot(c, NM("ff.net.2.bias"), w.b2, 1, D, D);
        add_slot(c, NM("norm1.linear.weight"), A + o_mod_w + (int64_t)(2 * l) * MC * D * TE * E, 6 * D, TE, TE);
        add_slot(c, NM("norm1.linear.bias"), A + o_mod_b + (int64_t)(2 * l) * MC * D * E, 1, 6 * D, 6 * D);
        add_slot(c, NM("norm2.linear.weight"), A + o_mod_w + (int64_t)(2 * l + 1) * MC * D * TE * E, 6 * D, TE, TE);
        add_slot(c, NM("norm2.linear.bias"), A + o_mod_b + (int64_t)(2 * l + 1) * MC * D * E, 1, 6 * D, 6 * D);
#undef NM
    }
    c->patch_w = A + o_patch_w; c->patch_b = A + o_patch_b; c->text_w = A + o_text_w; c->text_b = A + o_text_b;
    c->te1_w = A + o_te1_w; c->te1_b = A + o_te1_b; c->te2_w = A + o_te2_w; c->te2_b = A + o_te2_b;
    c->nf_w = A + o_nf_w; c->nf_b = A + o_nf_b; c->no_w = A + o_no_w; c->no_b = A + o_no_b;
    c->po_w = A + o_po_w; c->po_b = A + o_po_b; c->mod_w = A + o_mod_w; c->mod_b = A + o_mod_b;
    add_slot(c, "patch_embed.proj.weight", c->patch_w, D, Kp, Kp);
    add_slot(c, "patch_embed.proj.bias", c->patch_b, 1, D, D);
    add_slot(c, "patch_embed.text_proj.weight", c->text_w, D, TX, TX);
    add_slot(c, "patch_embed.text_proj.bias", c->text_b, 1, D, D);
    add_slot(c, "time_embedding.linear_1.weight", c->te1_w, TE, D, D);
    add_slot(c, "time_embedding.linear_1.bias", c->te1_b, 1, TE, TE);
    add_slot(c, "time_embedding.linear_2.weight", c->te2_w, TE, TE, TE);
    add_slot(c, "time_embedding.linear_2.bias", c->te2_b, 1, TE, TE);
    add_slot(c, "norm_final.weight", c->nf_w, 1, D, D);
    add_slot(c, "norm_final.bias", c->nf_b, 1, D, D);
    add_slot(c, "norm_out.norm.weight", c->no_w, 1, D, D);
    add_slot(c, "norm_out.norm.bias", c->no_b, 1, D, D);
    add_slot(c, "norm_out.linear.weight", c->mod_w + (int64_t)2 * L * MC * D * TE * E, 2 * D, TE, TE);
    add_slot(c, "norm_out.linear.bias", c->mod_b + (int64_t)2 * L * MC * D * E, 1, 2 * D, 2 * D);
    add_slot(c, "proj_out.weight", c->po_w, Cout, D, D);
    add_slot(c, "proj_out.bias", c->po_b, 1, Cout, Cout);

    hipMalloc((void**)&c->t_dev, 4 * sizeof(float));
    hipMalloc((void**)&c->coef_dev, sizeof(SchedCoef));
    hipHostMalloc((void**)&c->ring, sizeof(s2v_ctx::Stage) * RING);
    hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking);
    if (!c->t_dev || !c->coef_dev || !c->ring || !c->cap_stream) {
        s2v_destroy(c);
        return s2v_fail(__FILE__, __LINE__, "s2v_create: allocation failed", -2);
    }
    *out = c;
    return 0;
}

extern "C" void s2v_destroy(s2v_ctx* c) {
    if (!c) return;
    if (c->gexec) hipGraphExecDestroy(c->gexec);
    if (c->cap_stream) hipStreamDestroy(c->cap_stream);
    if (c->side) hipStreamDestroy(c->side);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->ring) hipHostFree(c->ring);
    if (c->coef_dev) hipFree(c->coef_dev);
    if (c->t_dev) hipFree(c->t_dev);
    if (c->ws) hipFree(c->ws);
    if (c->attn_queue) hipFree(c->attn_queue);
    if (c->attn_stats) hipFree(c->attn_stats);
    if (c->clk_buf) hipFree(c->clk_buf);
    if (c->arena) hipFree(c->arena);
    if (c->lora_tmp) hipFree(c->lora_tmp);
    delete c;
}

// "transformer_blocks.<l>.norm{1,2}.linear.{weight,bias}"
static bool is_adaln_linear(const char* name) {
    const std::string n = name;
    return n.rfind("transformer_blocks.", 0) == 0 && (n.find(".norm1.linear.") != std::string::npos || n.find(".norm2.linear.") != std::string::npos);
}

extern "C" int s2v_load_weight(s2v_ctx* c, const char* name, const void* dev_ptr, const int64_t* shape, int32_t ndim,
                               int32_t src_dtype, s2v_stream stream) {
    S2V_REQUIRE(c && name && dev_ptr && shape, "s2v_load_weight: null argument");
    S2V_REQUIRE(!c->finalized, "s2v_load_weight: weights already finalized");
    S2V_REQUIRE(src_dtype == S2V_DTYPE_F32 || src_dtype == S2V_DTYPE_BF16 || src_dtype == S2V_DTYPE_F16, "s2v_load_weight: unsupported source dtype");
    auto it = c->slots.find(name);
    if (it == c->slots.end()) {
        std::string m = std::string("s2v_load_weight: unknown tensor name: ") + name;
        return s2v_fail(__FILE__, __LINE__, m.c_str(), -3);
    }
    Slot& s = it->second;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    if (n != s.rows * s.cols || (ndim >= 2 && shape[0] != s.rows)) {
        std::string m = std::string("s2v_load_weight: shape mismatch for ") + name;
        return s2v_fail(__FILE__, __LINE__, m.c_str(), -3);
    }
    S2V_TRY(launch_convert2d(dev_ptr, src_dtype, s.cols, s.dst, c->dtype, s.ld, s.rows, s.cols, (hipStream_t)stream));
    s.loaded = true;
    if (c->mc == 9 && is_adaln_linear(name)) {
        // lora_adaln_scope 1: the reference-image copy of chunks 0-2 (rows [0, 3D) of the weight, entries [0, 3D) of the bias)
        // sits right behind the six chunks; it starts as the base values and is what s2v_merge_lora then adds to
        const int64_t D3 = 3 * (int64_t)c->D;
        const bool bias = s.rows == 1;
        const int64_t bytes6 = (bias ? 6 * (int64_t)c->D : 6 * (int64_t)c->D * s.ld) * c->esz, bytes3 = bytes6 / 2;
        (void)D3;
        S2V_CHECK_HIP(hipMemcpyAsync(s.dst + bytes6, s.dst, (size_t)bytes3, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return 0;
}

extern "C" int s2v_merge_lora(s2v_ctx* c, const char* name, const float* A, const float* B, int32_t rank, float scale,
                              s2v_stream stream) {
    S2V_REQUIRE(c && name && A && B && rank > 0, "s2v_merge_lora: bad argument");
    S2V_REQUIRE(!c->finalized, "s2v_merge_lora: weights already finalized");
    auto it = c->slots.find(name);
    S2V_REQUIRE(it != c->slots.end(), "s2v_merge_lora: unknown tensor name");
    Slot& s = it->second;
    S2V_REQUIRE(s.loaded && s.rows > 1, "s2v_merge_lora: base weight must be a loaded matrix");
    hipStream_t st = (hipStream_t)stream;
    // lora_adaln_scope 1: on norm{1,2}.linear the LoRA reaches only the reference-image copy of chunks 0-2 (rows [0, 3D) of B)
    const bool scoped = c->mc == 9 && is_adaln_linear(name);
    char* dst = scoped ? s.dst + 6 * (int64_t)c->D * s.ld * c->esz : s.dst;
    const int64_t rows = scoped ? 3 * (int64_t)c->D : s.rows;
    // one fp32 scratch for all merges of a load (grown to the largest slot, freed by s2v_finalize_weights / s2v_destroy): no
    // allocation or synchronisation per key
    const size_t need = sizeof(float) * (size_t)rows * (size_t)s.cols;
    if (need > c->lora_tmp_bytes) {
        if (c->lora_tmp) { S2V_CHECK_HIP(hipStreamSynchronize(st)); (void)hipFree(c->lora_tmp); c->lora_tmp = nullptr; c->lora_tmp_bytes = 0; }
        S2V_CHECK_HIP(hipMalloc((void**)&c->lora_tmp, need));
        c->lora_tmp_bytes = need;
    }
    float* tmp = c->lora_tmp;
    int r = launch_convert2d(dst, c->dtype, s.ld, tmp, S2V_F32, s.cols, rows, s.cols, st);
    // tmp[out][in] += scale * sum_k B[out][k] * A[k][in]
    if (!r) r = launch_gemm_strided_f32(B, rank, 1, A, 1, s.cols, tmp, s.cols, (int)rows, (int)s.cols, rank, scale, st);
    if (!r) r = launch_convert2d(tmp, S2V_F32, s.cols, dst, c->dtype, s.ld, rows, s.cols, st);
    return r;  // stream-ordered: the caller keeps A and B alive until the stream has passed (as for s2v_load_weight)
}

extern "C" int s2v_finalize_weights(s2v_ctx* c, s2v_stream stream) {
    S2V_REQUIRE(c, "s2v_finalize_weights: null context");
    for (auto& kv : c->slots) {
        if (!kv.second.loaded) {
            std::string m = std::string("s2v_finalize_weights: tensor was never loaded: ") + kv.first;
            return s2v_fail(__FILE__, __LINE__, m.c_str(), -3);
        }
    }
    if (c->fp8) {
        // per-output-channel e4m3 quantisation of the (LoRA-merged) bf16 weights; the zero pad rows get scale 1, bytes 0
        const int64_t D = c->D, Dp = rup(D, 256);
        hipStream_t st = (hipStream_t)stream;
        for (auto& w : c->layers) {
            S2V_TRY(launch_quant_rows_fp8(w.wqkv, D, rup(3 * D, 256), (int)D, w.q_qkv, w.s_qkv, st));
            S2V_TRY(launch_quant_rows_fp8(w.wo, D, Dp, (int)D, w.q_o, w.s_o, st));
            S2V_TRY(launch_quant_rows_fp8(w.w1, D, rup(4 * D, 256), (int)D, w.q_1, w.s_1, st));
            S2V_TRY(launch_quant_rows_fp8(w.w2, 4 * D, Dp, (int)(4 * D), w.q_2, w.s_2, st));
        }
        S2V_CHECK_HIP(hipStreamSynchronize(st));
    }
    if (c->lora_tmp) {
        S2V_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
        (void)hipFree(c->lora_tmp);
        c->lora_tmp = nullptr;
        c->lora_tmp_bytes = 0;
    }
    c->finalized = true;
    return 0;
}

extern "C" int s2v_weight_slot(s2v_ctx* c, const char* name, int64_t* offset_bytes, int64_t* rows, int64_t* cols, int64_t* ld) {
    S2V_REQUIRE(c && name && offset_bytes && rows && cols && ld, "s2v_weight_slot: null argument");
    auto it = c->slots.find(name);
    if (it == c->slots.end()) {
        std::string m = std::string("s2v_weight_slot: unknown tensor name: ") + name;
        return s2v_fail(__FILE__, __LINE__, m.c_str(), -3);
    }
    const Slot& s = it->second;
    *offset_bytes = (int64_t)(s.dst - c->arena);
    *rows = s.rows; *cols = s.cols; *ld = s.ld;
    return 0;
}

extern "C" int s2v_weight_arena(s2v_ctx* c, void** dev_ptr, int64_t* bytes) {
    S2V_REQUIRE(c && dev_ptr && bytes, "s2v_weight_arena: null argument");
    *dev_ptr = c->arena;
    *bytes = c->arena_bytes;
    return 0;
}
// a replica that received the arena by broadcast marks itself loaded
extern "C" int s2v_mark_weights_loaded(s2v_ctx* c) {
    S2V_REQUIRE(c, "null context");
    for (auto& kv : c->slots) kv.second.loaded = true;
    c->finalized = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------------
extern "C" int s2v_set_geometry(s2v_ctx* c, int32_t B, int32_t T, int32_t F, int32_t H, int32_t W) {
    S2V_REQUIRE(c, "null context");
    S2V_REQUIRE(B >= 1 && B <= 4, "s2v_set_geometry: batch must be 1..4");
    S2V_REQUIRE(T >= 0 && F >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "s2v_set_geometry: bad geometry");
    if (c->ws && B == c->B && T == c->T && F == c->F && H == c->H && W == c->W) return 0;
    S2V_CHECK_HIP(hipDeviceSynchronize());
    if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
    if (c->ws) { hipFree(c->ws); c->ws = nullptr; }
    c->B = B; c->T = T; c->F = F; c->H = H; c->W = W;
    c->R = (H / 2) * (W / 2);
    c->V = F * c->R;
    c->Ntok = T + c->R + c->V;
    // weight_format 3 ("fp8-auto"): fp8 linears always, fp8 QK^T from S2V_FP8_QK_AUTO_TOKENS tokens on -- there the attention is > 80 % of the
    // step and the whole-run drift of fp8-qk is the fp8 engine's (profiles/r05_whole_run_c5_10steps.txt: 8.69e-3 against 8.67e-3)
    c->fp8_qk = c->cfg.weight_format == 2 || (c->cfg.weight_format == 3 && c->Ntok >= S2V_FP8_QK_AUTO_TOKENS);
    c->ntok_pad = (int)rup(c->Ntok, 64);
    c->M = (int64_t)B * c->Ntok;
    c->Mpad = rup(c->M, 256) + 256;
    c->have_rope = c->have_pos = c->have_cond = false;
    c->split_kind = -1;
    const int64_t D = c->D, E = c->esz;
    const int64_t Cin4 = c->cfg.in_channels * 4, Cout4 = c->cfg.out_channels * 4;
    const int64_t BVp = rup((int64_t)B * c->V, 256) + 256;
    int64_t off = 0;
    auto carve = [&](int64_t bytes) { int64_t o = off; off += rup(bytes, 256); return o; };
    const int64_t oX = carve(c->Mpad * D * E), oXn = carve(c->Mpad * D * E), oQKV = carve(c->Mpad * 3 * D * E);
    const int64_t oH = carve(c->Mpad * 4 * D * E);
    const int64_t oVT = carve((int64_t)B * c->cfg.num_heads * 64 * c->ntok_pad * 2);
    const int64_t oe0 = carve(rup((int64_t)B * T + 128, 128) * D * E), oe1 = carve(rup(c->R + 128, 128) * D * E);
    const int64_t opat = carve(BVp * Cin4 * E), otail = carve(BVp * D * E), oproj = carve(BVp * Cout4 * E);
    const int64_t omod = carve((int64_t)B * c->mod_rows * E);
    const int64_t ote = carve(((int64_t)B * D + (int64_t)B * c->temb) * E), oemb = carve((int64_t)B * c->temb * E);
    // room for the CFG pair even when the geometry holds ONE sample of it (CFG-parallel, s2v_denoise_split_*: the peer's half arrives here)
    const int64_t onp = carve((int64_t)std::max(B, 2) * F * c->cfg.out_channels * H * W * E);
    const int64_t ocos = carve((int64_t)(c->R + c->V) * 64 * 4), osin = carve((int64_t)(c->R + c->V) * 64 * 4);
    const int64_t opk = carve((int64_t)(c->R + c->V) * 64 * 4);
    const int64_t opos = carve((int64_t)c->V * D * E);
    const int64_t oaq = carve(c->fp8 ? c->Mpad * 4 * D : 0), oaqs = carve(c->fp8 ? c->Mpad * 4 : 0);
    // fp8: the FF1 epilogue leaves GELU(h) as MX e4m3 (bytes + one E8M0 scale per 32 columns), the FF2 reads it in place
    const int64_t ohq = carve(c->fp8 ? c->Mpad * 4 * D : 0), ohs = carve(c->fp8 ? c->Mpad * 4 * D / 32 : 0);
    // fp8_qk: MX e4m3 images of q and k for attn_q4f (AttnArgs::q8 ... k8s)
    const int64_t BH = (int64_t)B * c->cfg.num_heads;
    const int64_t oq8 = carve(c->fp8_qk ? BH * c->Ntok * 64 : 0), oq8s = carve(c->fp8_qk ? BH * c->Ntok * 2 : 0);
    const int64_t ok8 = carve(c->fp8_qk ? BH * c->ntok_pad * 64 : 0), ok8s = carve(c->fp8_qk ? BH * c->ntok_pad * 4 : 0);
    // split-K partial tiles + arrival counters (linear(): only geometries whose FF2 has at most half as many 256 x 256 tiles as CUs)
    c->sk_tiles = (c->mfma && ((c->M + 255) / 256) * ((D + 255) / 256) * 2 <= c->num_cus) ? c->num_cus : 0;
    const int64_t osk = carve((int64_t)c->sk_tiles * 262144), oskc = carve((int64_t)c->sk_tiles * 4);
    c->ws_bytes = off;
    S2V_CHECK_HIP(hipMalloc((void**)&c->ws, c->ws_bytes));
    S2V_CHECK_HIP(hipMemset(c->ws, 0, c->ws_bytes));
    char* w = c->ws;
    c->X = w + oX; c->Xn = w + oXn; c->QKV = w + oQKV; c->Hb = w + oH; c->VT = w + oVT; c->e0 = w + oe0; c->e1 = w + oe1;
    c->patches = w + opat; c->tailn = w + otail; c->proj = w + oproj; c->mod = w + omod; c->tmp_te = w + ote;
    c->emb = w + oemb; c->noise_pred = w + onp; c->rope_cos = (float*)(w + ocos); c->rope_sin = (float*)(w + osin);
    c->pos_tab = w + opos; c->rope_pk = (float*)(w + opk); c->rope_paired = false;
    c->aq = w + oaq; c->aq_scale = (float*)(w + oaqs);
    c->hq = (unsigned char*)(w + ohq); c->hs = (unsigned char*)(w + ohs);
    c->q8 = (unsigned char*)(w + oq8); c->q8s = (unsigned short*)(w + oq8s); c->k8 = (unsigned char*)(w + ok8); c->k8s = (unsigned*)(w + ok8s);
    c->sk_ws = (float*)(w + osk); c->sk_cnt = (unsigned*)(w + oskc);
    return 0;
}

extern "C" int s2v_fp8_qk_active(s2v_ctx* c, int32_t* active) {
    S2V_REQUIRE(c && active, "s2v_fp8_qk_active: null argument");
    *active = c->fp8_qk ? 1 : 0;
    return 0;
}

extern "C" int s2v_set_rope(s2v_ctx* c, const float* cos_dev, const float* sin_dev, s2v_stream stream) {
    S2V_REQUIRE(c && c->ws, "s2v_set_rope: call s2v_set_geometry first");
    // the captured step bakes in have_rope and the fused-epilogue decision (rope_paired): a change of either drops the graph
    auto drop_graph = [&]() { if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; } };
    if (!cos_dev || !sin_dev) {
        if (c->have_rope) drop_graph();
        c->have_rope = false;
        return 0;
    }
    const bool had_rope = c->have_rope, was_paired = c->rope_paired;
    const size_t bytes = (size_t)(c->R + c->V) * 64 * 4;
    S2V_CHECK_HIP(hipMemcpyAsync(c->rope_cos, cos_dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    S2V_CHECK_HIP(hipMemcpyAsync(c->rope_sin, sin_dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    c->have_rope = true;
    // The reference builds its tables with repeat_interleave(2) (embeddings.py:get_3d_rotary_pos_embed): cos[2k] == cos[2k+1].  When
    // that holds (checked, once per geometry, on the host) a packed copy [position][32 cos | 32 sin] halves what the fused QKV
    // epilogue has to fetch per row; tables without that structure take the stand-alone qk_norm_rope_k pass instead.
    const size_t n = (size_t)(c->R + c->V) * 64;
    std::vector<float> hc(n), hs(n), pk(n);
    S2V_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    S2V_CHECK_HIP(hipMemcpy(hc.data(), c->rope_cos, bytes, hipMemcpyDeviceToHost));
    S2V_CHECK_HIP(hipMemcpy(hs.data(), c->rope_sin, bytes, hipMemcpyDeviceToHost));
    bool paired = true;
    for (size_t i = 0; i < n && paired; i += 2) paired = hc[i] == hc[i + 1] && hs[i] == hs[i + 1];
    c->rope_paired = paired;
    if (!had_rope || was_paired != paired) drop_graph();
    if (paired) {
        for (size_t p = 0; p < n / 64; ++p)
            for (int k = 0; k < 32; ++k) {
                pk[p * 64 + k] = hc[p * 64 + 2 * k];
                pk[p * 64 + 32 + k] = hs[p * 64 + 2 * k];
            }
        S2V_CHECK_HIP(hipMemcpy(c->rope_pk, pk.data(), bytes, hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" int s2v_set_pos_embed(s2v_ctx* c, const void* table_dev, s2v_stream stream) {
    S2V_REQUIRE(c && c->ws, "s2v_set_pos_embed: call s2v_set_geometry first");
    if (!table_dev) { c->have_pos = false; return 0; }
    S2V_CHECK_HIP(hipMemcpyAsync(c->pos_tab, table_dev, (size_t)c->V * c->D * c->esz, hipMemcpyDeviceToDevice,
                                 (hipStream_t)stream));
    c->have_pos = true;
    return 0;
}

static int linear(s2v_ctx* c, const GemmArgs& g0, int epi, hipStream_t st) {
    GemmArgs g = g0;
    g.clk = c->clk_cur;
    // every GEMM operand of the transformer lives in a workspace buffer with >= 256 rows of slack behind it
    g.a_rows_padded = (int)(rup(g.M, 256));
    g.w_rows_padded = (int)(rup(g.N, 256));  // every weight of the arena is carved with its rows padded to 256
    if ((c->mfma || c->h16) && g.K % 64 == 0 && g.lda % 8 == 0 && g.ldw % 8 == 0) {
        g.f16 = c->h16 ? 1 : 0;  // the fp16 model dtype takes the same dispatch on the kernels' fp16 instantiations (launch_gemm_bf16)
        // Tile-count quantisation: the 256 x 256 kernel runs one tile per CU, so a grid that spills a few tiles into an extra
        // round pays a whole round (C3: 150 x 12 = 1800 tiles = 7.03 rounds of 256 CUs for the out-proj / FF2).  When the
        // last row tile is partial and dropping it saves a round, the full row tiles run on the big kernel and the row tail
        // (108 rows at C3) on the 128 x 128 kernel.
        const int64_t tn = (g.N + 255) / 256, tm = (g.M + 255) / 256, ncu = c->num_cus;
        if (c->sk_tiles > 0 && g.splitk == 0) {
            const int S = gemm_choose_splitk(tm * tn, g.K, ncu);
            if (S > 1 && (int64_t)S * tm * tn <= c->sk_tiles) { g.splitk = S; g.sk_ws = c->sk_ws; g.sk_cnt = c->sk_cnt; }
        }
        // Few tiles (C1: M = 2500): half a round of 256 x 256 tiles or less and no split K (the out-projection: 80 tiles of 30 K-tiles)
        // -> 256 x 128 tiles put twice the workgroups on the part: 36 us against 59 (tools/microbench.py gemm_c1, profiles/r03_c1_*).
        // Measured and dropped: peeling the sparse second round of the FF1 (300 tiles on 256 CUs) off as 128 x 128 tiles -- 62 + 35 us
        // against 100 for the two rounds on the same stream, slower (117) as a fork on the side stream.
        if (g.splitk == 0 && epi != EPI_BIAS_QKNORM && g.tile == 0 && tm * tn * 2 <= ncu) g.tile = 1;
        const int rem = (int)(g.M % 256);
        bool split_tail = rem > 0 && tm > 1 && g.N >= 256 && (tm * tn + ncu - 1) / ncu > ((tm - 1) * tn + ncu - 1) / ncu;
        // Round 6 (the B = 1 geometry of a CFG-parallel rank: M = 19 126): the persistent kernel with the trickled epilogue (gemm_g4t) takes whole
        // 256-row tiles only, so a QKV projection whose row tail does NOT cost a round (75 x 36 = 2700 tiles = 10.5 rounds either way) used to run on
        // gemm_g4 with the exposed q/k-norm epilogue: 0.90 ms where half the B = 2 launch is 0.80.  Split the tail off whenever that lets the whole
        // tiles take gemm_g4t; the tail's 128 x 128 kernel returns the same bits (the B = 2 launch has mixed the two since round 5).
        if (!split_tail && rem > 0 && tm > 1 && g.N >= 256 && c->mfma && g.splitk == 0 && g.tile == 0) {
            GemmArgs gw = g;
            gw.M = (int)((tm - 1) * 256);
            split_tail = gemm_g4t_ok(gw, epi, (int)ncu);
        }
        if (split_tail) {
            GemmArgs gm = g, gt = g;
            gt.clk = nullptr;  // the row tail runs beside the main launch on the side stream: one stamp per profiled launch
            gm.M = (int)((tm - 1) * 256);
            gt.m_begin = gm.M;
            // fork: the tail runs on the side stream beside the main launch (event fork/join, valid under stream capture)
            S2V_CHECK_HIP(hipEventRecord(c->ev_fork, st));
            S2V_CHECK_HIP(hipStreamWaitEvent(c->side, c->ev_fork, 0));
            S2V_TRY(launch_gemm_bf16(gt, epi, c->side));
            S2V_CHECK_HIP(hipEventRecord(c->ev_join, c->side));
            S2V_TRY(launch_gemm_bf16(gm, epi, st));
            S2V_CHECK_HIP(hipStreamWaitEvent(st, c->ev_join, 0));
            return 0;
        }
        return launch_gemm_bf16(g, epi, st);
    }
    g.valu_only = c->cfg.force_simple;  // fp32: the matrix-pipe kernel (gemm_f32m, same bits) unless the cross-check path is asked for
    return launch_gemm_simple(g, epi, c->dtype, st);
}

// weight_format 1: quantise the activation rows (per token, dynamic) and run the fp8 GEMM against the e4m3 weight copy
// prequant: the producer (ln_modulate_k) already left the e4m3 image and the row scales of A in c->aq / c->aq_scale
// g0.mx_a_s set: A is already an MX image (g0.A bytes, block scales g0.mx_a_s) left by the producing GEMM's epilogue
static int linear_fp8(s2v_ctx* c, const GemmArgs& g0, int epi, const char* wq, const float* wscale, hipStream_t st, bool prequant = false) {
    GemmArgs g = g0;
    g.clk = c->clk_cur;
    if (g.mx_a_s) {
        g.lda = g.K; g.W = wq; g.ldw = g.K; g.a_scale = nullptr; g.w_scale = wscale;
        g.a_rows_padded = (int)rup(g.M, 256);
        g.w_rows_padded = (int)rup(g.N, 256);
        return launch_gemm_fp8(g, epi, st);
    }
    if (!prequant) S2V_TRY(launch_quant_rows_fp8(g.A, g.lda, g.M, g.K, c->aq, c->aq_scale, st));
    g.A = c->aq; g.lda = g.K; g.W = wq; g.ldw = g.K;
    g.a_scale = c->aq_scale; g.w_scale = wscale;
    g.a_rows_padded = (int)rup(g.M, 256);
    g.w_rows_padded = (int)rup(g.N, 256);
    return launch_gemm_fp8(g, epi, st);
}

extern "C" int s2v_set_conditioning(s2v_ctx* c, const void* text_dev, const void* ref_latent_dev, s2v_stream stream) {
    S2V_REQUIRE(c && c->ws && c->finalized, "s2v_set_conditioning: geometry and weights must be set first");
    S2V_REQUIRE(text_dev && ref_latent_dev, "s2v_set_conditioning: null input");
    hipStream_t st = (hipStream_t)stream;
    const int D = c->D;
    // text_proj (cogvideox_transformer_3d.py:494).  The MFMA kernel stages 128-row tiles, so the operand is first
    // copied into the (padded, zero-initialised) Hb scratch.
    const int TX = c->cfg.text_embed_dim;
    if (c->T > 0) {
        S2V_TRY(launch_convert2d(text_dev, c->dtype, TX, c->Hb, c->dtype, TX, (int64_t)c->B * c->T, TX, st));
        GemmArgs g{};
        g.A = c->Hb; g.lda = TX; g.W = c->text_w; g.ldw = TX; g.bias = c->text_b;
        g.C = c->e0; g.ldc = D; g.M = c->B * c->T; g.N = D; g.K = TX;
        S2V_TRY(linear(c, g, EPI_BIAS, st));
    }
    // reference image latent -> patch tokens (cogvideox_transformer_3d.py:496-501); identical for every sample
    const int K = c->cfg.in_channels * 4;
    S2V_TRY(launch_patchify(ref_latent_dev, 0, 1, 1, c->cfg.in_channels, c->H, c->W, c->patches, c->dtype, st));
    GemmArgs g{};
    g.A = c->patches; g.lda = K; g.W = c->patch_w; g.ldw = K; g.bias = c->patch_b;
    g.C = c->e1; g.ldc = D; g.M = c->R; g.N = D; g.K = K;
    S2V_TRY(linear(c, g, EPI_BIAS, st));
    c->have_cond = true;
    return 0;
}

// ---- one transformer block on the packed residual buffer X ------------------------------------------------
#ifdef S2V_DIAG
static int g_fp8_mx = 1;
extern "C" __attribute__((visibility("default"))) int s2v_set_fp8_mx(int on) { g_fp8_mx = on; return 0; }
static int g_fused_q8 = 1;
extern "C" __attribute__((visibility("default"))) int s2v_set_fused_q8(int on) { g_fused_q8 = on; return 0; }
static int g_fused_qk = 1;
extern "C" __attribute__((visibility("default"))) int s2v_set_fused_qk(int on) { g_fused_qk = on; return 0; }
// the fused QKV projection (EPI_BIAS_QKNORM) as an operator of the diagnostics build: A [M (rows allocated up to a multiple of 256)][K],
// W [3 D][K], C [M][3 D], LayerNorm(64) parameters of q and k, paired rotary table [positions][32 cos | 32 sin] fp32 -- what the block
// forward launches, so that the kernels that implement it (gemm_g4t's trickle, gemm_g4's / the eight-wave kernels' C++ epilogue) can be
// compared bit for bit on real shapes (tests/test_gpu_gemm_schedules.py) and timed (tools/qkv_trickle_bench.py)
extern "C" __attribute__((visibility("default"))) int s2v_diag_qkv_qknorm(const void* A, const void* W, const void* bias, const void* nq_w, const void* nq_b,
                                                                           const void* nk_w, const void* nk_b, const float* cs, void* C, int32_t M, int32_t D,
                                                                           int32_t K, int32_t tok_per_batch, int32_t text_len, float eps, s2v_stream stream) {
    S2V_REQUIRE(A && W && bias && C && nq_w && nq_b && nk_w && nk_b, "s2v_diag_qkv_qknorm: null argument");
    GemmArgs g{};
    g.A = A; g.lda = K; g.W = W; g.ldw = K; g.bias = bias; g.C = C; g.ldc = 3 * D; g.M = M; g.N = 3 * D; g.K = K;
    g.a_rows_padded = (int)rup(M, 256); g.w_rows_padded = (int)rup(3 * D, 256);
    g.tok_per_batch = tok_per_batch; g.text_len = text_len;
    g.qk_w[0] = nq_w; g.qk_b[0] = nq_b; g.qk_w[1] = nk_w; g.qk_b[1] = nk_b; g.qk_cs = cs; g.qk_D = D; g.qk_eps = eps;
    return launch_gemm_bf16(g, EPI_BIAS_QKNORM, (hipStream_t)stream);
}
#endif
// mx_out (fp8 engine, MFMA path): the attention kernel leaves its output as MX e4m3 in c->aq / c->hs for the out-projection instead of bf16 in Xn
static bool attn_mx_out(const s2v_ctx* c) {
    bool mx = c->fp8 && c->mfma && c->D % 128 == 0;
#ifdef S2V_DIAG
    mx = mx && g_fp8_mx;
#endif
    return mx;
}
static int run_attention(s2v_ctx* c, int l, hipStream_t st, bool prequant = false) {
    // Xn -> QKV -> (qk-norm, rope, V^T) -> attention -> Xn (reused as the attention output buffer)
    const LayerW& w = c->layers[l];
    const int D = c->D;
    GemmArgs g{};
    g.A = c->Xn; g.lda = D; g.W = w.wqkv; g.ldw = D; g.bias = w.bqkv;
    g.C = c->QKV; g.ldc = 3 * D; g.M = (int)c->M; g.N = 3 * D; g.K = D;
    // MFMA path: the per-head LayerNorm + rotary embedding of q and k run in the projection's epilogue (EPI_BIAS_QKNORM), on the
    // rounded projection as the stand-alone kernel does; only the V^T production remains a pass of its own
    // fp8 QK^T (weight_format 2 / 3 beyond its token threshold): q and k reach the attention kernel only as MX e4m3 images, so their LayerNorm + rotary
    // embedding ride in the pass that makes the images (qk_norm_quant_mx_k: the same bytes moved, the same bits produced) and the projection keeps the plain
    // bias epilogue -- its exposed q/k-norm epilogue is 20-25 % of an fp8 tile, whose K loop is half as long as the bf16 one
#ifdef S2V_DIAG
    const bool fuse_any = (c->mfma || c->h16) && D % 64 == 0 && (!c->have_rope || c->rope_paired) && g_fused_qk != 0;  // A/B switch of the diagnostics build:
    const bool norm_in_quant = fuse_any && c->fp8_qk && g_fused_qk != 3;                                               // 0 stand-alone kernel, 3 epilogue even for fp8 QK^T
#else
    const bool fuse_any = (c->mfma || c->h16) && D % 64 == 0 && (!c->have_rope || c->rope_paired);
    const bool norm_in_quant = fuse_any && c->fp8_qk;
#endif
    const bool fused_qk = fuse_any && !norm_in_quant;
    if (fused_qk) {
        g.tok_per_batch = c->Ntok; g.text_len = c->T;
        g.qk_w[0] = w.nq_w; g.qk_b[0] = w.nq_b; g.qk_w[1] = w.nk_w; g.qk_b[1] = w.nk_b;
        g.qk_cs = c->have_rope ? c->rope_pk : nullptr;
        g.qk_D = D; g.qk_eps = 1e-6f;
    }
    {
        ProfScope ps(c, PK_QKV, st);
        const int epi = fused_qk ? EPI_BIAS_QKNORM : EPI_BIAS;
        if (c->fp8) S2V_TRY(linear_fp8(c, g, epi, w.q_qkv, w.s_qkv, st, prequant));
        else S2V_TRY(linear(c, g, epi, st));
    }
    // fp16 P / V^T is the four-wave kernels' (attn_q4h / attn_q4fh); short sequences run attn_pp on bf16 V^T
    const bool p16 = c->attn_p16 && c->mfma && (c->fp8_qk || attn_runs_q4(c->Ntok, attn_mx_out(c)));
    if (fused_qk || norm_in_quant) {
        ProfScope ps(c, PK_QKNORM, st);
        S2V_TRY(launch_v_transpose(c->QKV, 3 * D, c->B, c->cfg.num_heads, c->Ntok, c->VT, c->ntok_pad, st, p16));
    } else {
        QkNormRopeArgs q{};
        q.qkv = c->QKV; q.ld_qkv = 3 * D; q.B = c->B; q.H = c->cfg.num_heads; q.Ntok = c->Ntok; q.text_len = c->T;
        q.nq_w = w.nq_w; q.nq_b = w.nq_b; q.nk_w = w.nk_w; q.nk_b = w.nk_b; q.eps = 1e-6f;
        q.cos = c->have_rope ? c->rope_cos : nullptr; q.sin = c->have_rope ? c->rope_sin : nullptr;
        q.vt = (c->mfma || c->h16) ? c->VT : nullptr; q.ntok_pad = c->ntok_pad; q.vt_f16 = p16 ? 1 : 0;  // fp16 dtype: the pass moves fp16 bits
        ProfScope ps(c, PK_QKNORM, st);
        S2V_TRY(launch_qk_norm_rope(q, c->dtype, st));
    }
    AttnArgs a{};
    a.qkv = c->QKV; a.ld_qkv = 3 * D; a.vt = c->VT; a.ntok_pad = c->ntok_pad; a.out = c->Xn; a.ld_out = D;
    a.B = c->B; a.H = c->cfg.num_heads; a.Ntok = c->Ntok; a.scale = 0.125f;
    a.queue = c->attn_queue; a.num_cus = c->num_cus;  // launches of one context are ordered on its stream: one queue suffices
    a.p16 = p16 ? 1 : 0;
    a.stats = c->attn_stats;
    a.valu_only = c->cfg.force_simple;
    a.stagger = c->attn_stagger;
    a.order = c->attn_order;
    if (attn_mx_out(c)) { a.mx_q = (unsigned char*)c->aq; a.mx_s = c->hs; a.mx_rows = (int)c->Mpad; }
    if (c->fp8_qk) {  // weight_format 2: q (times scale * log2 e) and k as MX e4m3, QK^T on the scaled fp8 MFMA (the pass is timed with the V^T pass)
        {
            ProfScope ps(c, PK_QKNORM, st);
            if (norm_in_quant) {
                QkNormRopeArgs q{};
                q.qkv = c->QKV; q.ld_qkv = 3 * D; q.B = c->B; q.H = c->cfg.num_heads; q.Ntok = c->Ntok; q.text_len = c->T; q.ntok_pad = c->ntok_pad;
                q.nq_w = w.nq_w; q.nq_b = w.nq_b; q.nk_w = w.nk_w; q.nk_b = w.nk_b; q.eps = 1e-6f;
                q.cos = c->have_rope ? c->rope_cos : nullptr; q.sin = c->have_rope ? c->rope_sin : nullptr;
                S2V_TRY(launch_qk_norm_quant_mx(q, a.scale * 1.4426950408889634f, c->q8, c->q8s, c->k8, c->k8s, st));
            } else {
                S2V_TRY(launch_qk_quant_mx(c->QKV, 3 * D, c->B, c->cfg.num_heads, c->Ntok, c->ntok_pad, a.scale * 1.4426950408889634f, c->q8, c->q8s, c->k8,
                                           c->k8s, st));
            }
        }
        a.q8 = c->q8; a.q8s = c->q8s; a.k8 = c->k8; a.k8s = c->k8s;
        ProfScope ps(c, PK_ATTN, st);
        a.clk = c->clk_cur;
        return launch_attn_q4f(a, true, st);
    }
    ProfScope ps(c, PK_ATTN, st);
    a.clk = c->clk_cur;
    if (c->mfma) S2V_TRY(launch_attn_bf16(a, st));
    else if (c->h16) S2V_TRY(launch_attn_f16(a, st));   // fp16 dtype: q, k, V^T, P in fp16 on v_mfma_f32_32x32x16_f16
    else S2V_TRY(launch_attn_simple(a, c->dtype, st));
    return 0;
}

static int run_block(s2v_ctx* c, int l, const char* mod_base /* [B][mod_stride] rows of this layer's norm1 */,
                     int64_t mod_stride, hipStream_t st) {
    const LayerW& w = c->layers[l];
    const int D = c->D;
    const int64_t E = c->esz;
    for (int half = 0; half < 2; ++half) {
        const char* mb = mod_base + (int64_t)half * c->mc * D * E;
        LnModArgs n{};
        n.x = c->X; n.ldx = D; n.y = c->Xn; n.ldy = D;
        n.w = half ? w.ln2_w : w.ln1_w; n.b = half ? w.ln2_b : w.ln1_b; n.eps = c->cfg.norm_eps;
        n.shift_vid = mb; n.scale_vid = mb + D * E; n.shift_txt = mb + 3 * D * E; n.scale_txt = mb + 4 * D * E;
        n.mod_stride = (int)mod_stride; n.B = c->B; n.Ntok = c->Ntok; n.text_len = c->T; n.D = D;
        if (c->mc == 9) { n.shift_ref = mb + 6 * D * E; n.scale_ref = mb + 7 * D * E; n.ref_len = c->R; }
        // fp8 linears: the LayerNorm output feeds exactly one projection (QKV / FF1), so it is quantised where it is produced
        bool prequant = c->fp8;
#ifdef S2V_DIAG
        prequant = prequant && g_fused_q8;
#endif
        if (prequant) { n.q8 = c->aq; n.q8_scale = c->aq_scale; }
        { ProfScope ps(c, PK_LNMOD, st); S2V_TRY(launch_ln_modulate(n, c->dtype, st)); }
        GemmArgs g{};
        g.X = c->X; g.ldx = D; g.gate_vid = mb + 2 * D * E; g.gate_txt = mb + 5 * D * E; g.gate_stride = (int)mod_stride;
        g.tok_per_batch = c->Ntok; g.text_len = c->T; g.M = (int)c->M; g.N = D;
        if (c->mc == 9) { g.gate_ref = mb + 8 * D * E; g.ref_len = c->R; }
        if (half == 0) {
            S2V_TRY(run_attention(c, l, st, prequant));
            g.A = c->Xn; g.lda = D; g.W = w.wo; g.ldw = D; g.bias = w.bo; g.K = D;
            if (attn_mx_out(c)) { g.A = c->aq; g.mx_a_s = c->hs; g.mx_rows = (int)c->Mpad; }
            ProfScope ps(c, PK_OUT, st);
            if (c->fp8) S2V_TRY(linear_fp8(c, g, EPI_BIAS_GATE_RES, w.q_o, w.s_o, st));
            else S2V_TRY(linear(c, g, EPI_BIAS_GATE_RES, st));
        } else {
            GemmArgs f{};
            f.A = c->Xn; f.lda = D; f.W = w.w1; f.ldw = D; f.bias = w.b1; f.C = c->Hb; f.ldc = 4 * D;
            f.M = (int)c->M; f.N = 4 * D; f.K = D;
            bool mx = c->fp8 && (4 * D) % 128 == 0;
#ifdef S2V_DIAG
            mx = mx && g_fp8_mx;
#endif
            if (mx) { f.mx_out_q = c->hq; f.mx_out_s = c->hs; f.mx_rows = (int)c->Mpad; }
            {
                ProfScope ps(c, PK_FF1, st);
                if (c->fp8) S2V_TRY(linear_fp8(c, f, EPI_BIAS_GELU, w.q_1, w.s_1, st, prequant));
                else S2V_TRY(linear(c, f, EPI_BIAS_GELU, st));
            }
            g.A = c->Hb; g.lda = 4 * D; g.W = w.w2; g.ldw = 4 * D; g.bias = w.b2; g.K = 4 * D;
            if (mx) { g.A = c->hq; g.mx_a_s = c->hs; g.mx_rows = (int)c->Mpad; }
            ProfScope ps(c, PK_FF2, st);
            if (c->fp8) S2V_TRY(linear_fp8(c, g, EPI_BIAS_GATE_RES, w.q_2, w.s_2, st));
            else S2V_TRY(linear(c, g, EPI_BIAS_GATE_RES, st));
        }
    }
    return 0;
}

static int forward_impl(s2v_ctx* c, const void* latents, int64_t lat_bstride, const float* t_dev, void* out,
                        hipStream_t st) {
    S2V_REQUIRE(c->ws && c->finalized && c->have_cond, "transformer_forward: geometry, weights and conditioning required");
    S2V_REQUIRE(!c->cfg.use_rope || c->have_rope, "transformer_forward: RoPE tables missing (s2v_set_rope)");
    S2V_REQUIRE(c->cfg.use_rope || c->have_pos, "transformer_forward: sincos table missing (s2v_set_pos_embed)");
    const int D = c->D, B = c->B;
    const int64_t E = c->esz;
    // 1. timestep embedding + every AdaLN modulation of the step in one batched GEMV (temb is block-invariant)
    S2V_TRY(launch_time_embed(t_dev, B, D, c->te1_w, c->te1_b, c->te2_w, c->te2_b, c->temb, c->tmp_te, c->emb, c->dtype, st));
    {
        ProfScope ps(c, PK_OTHER, st);  // the step's modulation GEMV: every norm1 / norm2 / norm_out linear in one launch
        S2V_TRY(launch_mod_gemv(c->emb, B, c->temb, c->mod_w, c->mod_b, c->mod_rows, c->mod, c->dtype, st));
    }
    // 2. residual streams: [text | ref | video] per sample
    const int K = c->cfg.in_channels * 4;
    S2V_TRY(launch_patchify(latents, lat_bstride, B, c->F, c->cfg.in_channels, c->H, c->W, c->patches, c->dtype, st));
    for (int b = 0; b < B; ++b) {
        char* xb = c->X + (int64_t)b * c->Ntok * D * E;
        S2V_TRY(launch_copy_rows(c->e0 + (int64_t)b * c->T * D * E, D, nullptr, 0, xb, D, c->T, D, c->dtype, st));
        S2V_TRY(launch_copy_rows(c->e1, D, nullptr, 0, xb + (int64_t)c->T * D * E, D, c->R, D, c->dtype, st));
        char* xv = xb + (int64_t)(c->T + c->R) * D * E;
        GemmArgs g{};
        g.A = c->patches + (int64_t)b * c->V * K * E; g.lda = K; g.W = c->patch_w; g.ldw = K; g.bias = c->patch_b;
        g.C = xv; g.ldc = D; g.M = c->V; g.N = D; g.K = K;
        S2V_TRY(linear(c, g, EPI_BIAS, st));
        if (!c->cfg.use_rope) S2V_TRY(launch_copy_rows(xv, D, c->pos_tab, D, xv, D, c->V, D, c->dtype, st));
    }
    // 3. blocks
    for (int l = 0; l < c->L; ++l)
        S2V_TRY(run_block(c, l, c->mod + (int64_t)(2 * l) * c->mc * D * E, c->mod_rows, st));
    // 4. tail
    const char* mo = c->mod + (int64_t)2 * c->L * c->mc * D * E;
    TailNormArgs t{};
    t.x = c->X; t.ldx = D; t.y = c->tailn; t.ldy = D; t.w1 = c->nf_w; t.b1 = c->nf_b; t.w2 = c->no_w; t.b2 = c->no_b;
    t.eps = c->cfg.norm_eps; t.shift = mo; t.scale = mo + D * E; t.mod_stride = (int)c->mod_rows;
    t.B = B; t.Ntok = c->Ntok; t.row0 = c->T + c->R; t.V = c->V; t.D = D;
    S2V_TRY(launch_tail_norm(t, c->dtype, st));
    const int Co = c->cfg.out_channels * 4;
    GemmArgs g{};
    g.A = c->tailn; g.lda = D; g.W = c->po_w; g.ldw = D; g.bias = c->po_b; g.C = c->proj; g.ldc = Co;
    g.M = B * c->V; g.N = Co; g.K = D;
    S2V_TRY(linear(c, g, EPI_BIAS, st));
    S2V_TRY(launch_unpatchify(c->proj, Co, c->V, out, B, c->F, c->cfg.out_channels, c->H, c->W, c->dtype, st));
    return 0;
}

extern "C" int s2v_transformer_forward(s2v_ctx* c, const void* latents, int64_t lat_bstride, const float* timesteps_dev,
                                       void* out, s2v_stream stream) {
    S2V_REQUIRE(c && latents && timesteps_dev && out, "s2v_transformer_forward: null argument");
    return forward_impl(c, latents, lat_bstride, timesteps_dev, out, (hipStream_t)stream);
}

extern "C" int s2v_block_forward(s2v_ctx* c, int32_t layer, const void* hidden, const void* enc0, const void* enc1,
                                 const void* temb, void* out_hidden, void* out_enc0, void* out_enc1, s2v_stream stream) {
    S2V_REQUIRE(c && c->ws && c->finalized, "s2v_block_forward: geometry and weights required");
    S2V_REQUIRE(layer >= 0 && layer < c->L, "s2v_block_forward: bad layer");
    S2V_REQUIRE(hidden && enc1 && temb && out_hidden && out_enc1, "s2v_block_forward: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int D = c->D, B = c->B;
    const int64_t E = c->esz;
    // this layer's two modulation linears are contiguous in the stack: rows [(2l)*6D, (2l+2)*6D)
    const int64_t r0 = (int64_t)(2 * layer) * c->mc * D;
    char* modl = c->mod;  // [B][12D]
    S2V_TRY(launch_mod_gemv(temb, B, c->temb, c->mod_w + r0 * c->temb * E, c->mod_b + r0 * E, 2 * c->mc * D, modl, c->dtype, st));
    for (int b = 0; b < B; ++b) {
        char* xb = c->X + (int64_t)b * c->Ntok * D * E;
        if (c->T) S2V_TRY(launch_copy_rows((const char*)enc0 + (int64_t)b * c->T * D * E, D, nullptr, 0, xb, D, c->T, D, c->dtype, st));
        S2V_TRY(launch_copy_rows((const char*)enc1 + (int64_t)b * c->R * D * E, D, nullptr, 0, xb + (int64_t)c->T * D * E, D, c->R, D, c->dtype, st));
        S2V_TRY(launch_copy_rows((const char*)hidden + (int64_t)b * c->V * D * E, D, nullptr, 0, xb + (int64_t)(c->T + c->R) * D * E, D, c->V, D, c->dtype, st));
    }
    S2V_TRY(run_block(c, layer, modl, 2 * c->mc * D, st));
    for (int b = 0; b < B; ++b) {
        const char* xb = c->X + (int64_t)b * c->Ntok * D * E;
        if (c->T) S2V_TRY(launch_copy_rows(xb, D, nullptr, 0, (char*)out_enc0 + (int64_t)b * c->T * D * E, D, c->T, D, c->dtype, st));
        S2V_TRY(launch_copy_rows(xb + (int64_t)c->T * D * E, D, nullptr, 0, (char*)out_enc1 + (int64_t)b * c->R * D * E, D, c->R, D, c->dtype, st));
        S2V_TRY(launch_copy_rows(xb + (int64_t)(c->T + c->R) * D * E, D, nullptr, 0, (char*)out_hidden + (int64_t)b * c->V * D * E, D, c->V, D, c->dtype, st));
    }
    return 0;
}

extern "C" int s2v_attn_forward(s2v_ctx* c, int32_t layer, const void* hidden, const void* encoder, void* out_hidden,
                                void* out_encoder, s2v_stream stream) {
    S2V_REQUIRE(c && c->ws && c->finalized, "s2v_attn_forward: geometry and weights required");
    S2V_REQUIRE(layer >= 0 && layer < c->L, "s2v_attn_forward: bad layer");
    S2V_REQUIRE(hidden && encoder && out_hidden && out_encoder, "s2v_attn_forward: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int D = c->D, B = c->B, TR = c->T + c->R;
    const int64_t E = c->esz;
    for (int b = 0; b < B; ++b) {
        char* xb = c->Xn + (int64_t)b * c->Ntok * D * E;
        S2V_TRY(launch_copy_rows((const char*)encoder + (int64_t)b * TR * D * E, D, nullptr, 0, xb, D, TR, D, c->dtype, st));
        S2V_TRY(launch_copy_rows((const char*)hidden + (int64_t)b * c->V * D * E, D, nullptr, 0, xb + (int64_t)TR * D * E, D, c->V, D, c->dtype, st));
    }
    S2V_TRY(run_attention(c, layer, st));
    const LayerW& w = c->layers[layer];
    GemmArgs g{};
    g.A = c->Xn; g.lda = D; g.W = w.wo; g.ldw = D; g.bias = w.bo; g.C = c->Hb; g.ldc = D; g.M = (int)c->M; g.N = D; g.K = D;
    if (attn_mx_out(c)) { g.A = c->aq; g.mx_a_s = c->hs; g.mx_rows = (int)c->Mpad; }
    if (c->fp8) S2V_TRY(linear_fp8(c, g, EPI_BIAS, w.q_o, w.s_o, st));  // the same operands run_block feeds its out-projection
    else S2V_TRY(linear(c, g, EPI_BIAS, st));
    for (int b = 0; b < B; ++b) {
        const char* xb = c->Hb + (int64_t)b * c->Ntok * D * E;
        S2V_TRY(launch_copy_rows(xb, D, nullptr, 0, (char*)out_encoder + (int64_t)b * TR * D * E, D, TR, D, c->dtype, st));
        S2V_TRY(launch_copy_rows(xb + (int64_t)TR * D * E, D, nullptr, 0, (char*)out_hidden + (int64_t)b * c->V * D * E, D, c->V, D, c->dtype, st));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------
static void fill_coef(SchedCoef& d, const s2v_sched_coef& s) {
    d.kind = s.kind; d.guidance = s.guidance; d.c_x0_x = s.c_x0_x; d.c_x0_v = s.c_x0_v; d.a_t = s.a_t; d.b_t = s.b_t;
    d.m1 = s.m1; d.m2 = s.m2; d.m3 = s.m3; d.m4 = s.m4; d.mn = s.mn; d.pad = 0.f;
}

extern "C" int s2v_sched_step(s2v_ctx* c, const s2v_sched_coef* coef_host, const void* noise_pred, int32_t flags,
                              const void* latents_in, void* latents_out, float* x0_hist, const void* noise, int64_t n,
                              int32_t dtype, s2v_stream stream) {
    S2V_REQUIRE(coef_host && noise_pred && latents_in && latents_out, "s2v_sched_step: null argument");
    S2V_REQUIRE(coef_host->kind == 0 || noise, "s2v_sched_step: DPM step needs a noise tensor");
    S2V_REQUIRE(coef_host->kind != 2 || x0_hist, "s2v_sched_step: DPM multistep needs x0_hist");
    S2V_REQUIRE(dtype == S2V_DTYPE_F32 || dtype == S2V_DTYPE_BF16 || dtype == S2V_DTYPE_F16, "s2v_sched_step: unsupported dtype");
    (void)c;
    SchedArgs a{};
    a.noise_pred = noise_pred; a.latents_in = latents_in; a.latents_out = latents_out; a.x0_hist = x0_hist;
    a.noise = noise; a.n = n; a.cfg = flags & 1; a.np_f32 = (flags >> 1) & 1; a.out_f32 = (flags >> 2) & 1;
    a.coef = nullptr;
    fill_coef(a.cval, *coef_host);
    return launch_sched_step(a, dtype, (hipStream_t)stream);
}

static int step_launches(s2v_ctx* c, void* latents, float* x0_hist, const void* noise, hipStream_t st) {
    S2V_TRY(forward_impl(c, latents, 0, c->t_dev, c->noise_pred, st));
    SchedArgs a{};
    a.noise_pred = c->noise_pred; a.latents_in = latents; a.latents_out = latents; a.x0_hist = x0_hist; a.noise = noise;
    a.n = (int64_t)c->F * c->cfg.out_channels * c->H * c->W; a.cfg = c->B == 2 ? 1 : 0; a.coef = c->coef_dev;
    return launch_sched_step(a, c->dtype, st);
}

extern "C" int s2v_denoise_step(s2v_ctx* c, void* latents, float timestep, const s2v_sched_coef* coef_host,
                                float* x0_hist, const void* noise, int32_t use_graph, s2v_stream stream) {
    S2V_REQUIRE(c && latents && coef_host, "s2v_denoise_step: null argument");
    S2V_REQUIRE(c->ws && (c->B == 1 || c->B == 2), "s2v_denoise_step: geometry with B = 1 or 2 (CFG pair) required");
    S2V_REQUIRE(c->cfg.in_channels == c->cfg.out_channels, "s2v_denoise_step: in/out channels must match");
    S2V_REQUIRE(coef_host->kind == 0 || (noise && x0_hist), "s2v_denoise_step: DPM needs noise and x0_hist");
    hipStream_t st = (hipStream_t)stream;
    s2v_ctx::Stage& sg = c->ring[c->ring_pos];
    c->ring_pos = (c->ring_pos + 1) % RING;
    for (int i = 0; i < 4; ++i) sg.t[i] = timestep;
    fill_coef(sg.c, *coef_host);
    S2V_CHECK_HIP(hipMemcpyAsync(c->t_dev, sg.t, sizeof(float) * 4, hipMemcpyHostToDevice, st));
    S2V_CHECK_HIP(hipMemcpyAsync(c->coef_dev, &sg.c, sizeof(SchedCoef), hipMemcpyHostToDevice, st));
    if (!use_graph) return step_launches(c, latents, x0_hist, noise, st);
    GraphKey key{latents, x0_hist, noise, -1};
    if (!c->gexec || !(c->gkey == key)) {
        if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
        hipGraph_t graph = nullptr;
        S2V_CHECK_HIP(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal));
        int r = step_launches(c, latents, x0_hist, noise, c->cap_stream);
        hipError_t e = hipStreamEndCapture(c->cap_stream, &graph);
        if (r != 0) { if (graph) hipGraphDestroy(graph); return r; }
        S2V_CHECK_HIP(e);
        e = hipGraphInstantiate(&c->gexec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        S2V_CHECK_HIP(e);
        c->gkey = key;
    }
    S2V_CHECK_HIP(hipGraphLaunch(c->gexec, st));
    return 0;
}

// ---- CFG-parallel (round 6): ONE video on TWO GPUs ---------------------------------------------------------------------------------
// The CFG pair of custom_cogvideox_pipe.py:255-279 is two independent forwards that meet only in `noise_pred_uncond + g * (noise_pred_text -
// noise_pred_uncond)` (:266-279).  Each rank of a pair holds a B = 1 geometry with ITS half of the prompt embeddings (slot 0 = negative /
// unconditional, slot 1 = positive: the order of :196) and the un-duplicated reference tokens (cogvideox_transformer_3d.py:503-504 duplicates them
// only to fill the pair).  begin: the rank's forward into half `slot` of the context-owned pair buffer (a forward-only hipGraph);  the caller
// exchanges the halves (s2v_rccl_allgather in place, or any transport -- the collective stays OUTSIDE the captured graph);  end: fp32 CFG +
// scheduler step + round on the pair, run REDUNDANTLY by both ranks (:266-296): both hold bit-identical latents without a second collective.
int s2v_rccl_pair_check(s2v_rccl_comm* comm, int slot);  // rccl.hip (not exported)
static int64_t pair_half_bytes(const s2v_ctx* c) { return (int64_t)c->F * c->cfg.out_channels * c->H * c->W * c->esz; }

extern "C" int s2v_denoise_split_begin(s2v_ctx* c, const void* latents, float timestep, const s2v_sched_coef* coef_host, int32_t slot,
                                       int32_t use_graph, s2v_stream stream) {
    S2V_REQUIRE(c && latents && coef_host, "s2v_denoise_split_begin: null argument");
    S2V_REQUIRE(c->ws && c->B == 1, "s2v_denoise_split_begin: a geometry with B = 1 (one sample of the CFG pair per rank) is required");
    S2V_REQUIRE(slot == 0 || slot == 1, "s2v_denoise_split_begin: slot must be 0 (unconditional) or 1 (conditional)");
    S2V_REQUIRE(c->cfg.in_channels == c->cfg.out_channels, "s2v_denoise_split_begin: in/out channels must match");
    hipStream_t st = (hipStream_t)stream;
    s2v_ctx::Stage& sg = c->ring[c->ring_pos];
    c->ring_pos = (c->ring_pos + 1) % RING;
    for (int i = 0; i < 4; ++i) sg.t[i] = timestep;
    fill_coef(sg.c, *coef_host);
    S2V_CHECK_HIP(hipMemcpyAsync(c->t_dev, sg.t, sizeof(float) * 4, hipMemcpyHostToDevice, st));
    S2V_CHECK_HIP(hipMemcpyAsync(c->coef_dev, &sg.c, sizeof(SchedCoef), hipMemcpyHostToDevice, st));
    char* out = c->noise_pred + (int64_t)slot * pair_half_bytes(c);
    c->split_kind = coef_host->kind;
    if (!use_graph) return forward_impl(c, latents, 0, c->t_dev, out, st);
    GraphKey key{(void*)latents, nullptr, nullptr, slot};
    if (!c->gexec || !(c->gkey == key)) {
        if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
        hipGraph_t graph = nullptr;
        S2V_CHECK_HIP(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal));
        int r = forward_impl(c, latents, 0, c->t_dev, out, c->cap_stream);
        hipError_t e = hipStreamEndCapture(c->cap_stream, &graph);
        if (r != 0) { if (graph) hipGraphDestroy(graph); return r; }
        S2V_CHECK_HIP(e);
        e = hipGraphInstantiate(&c->gexec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        S2V_CHECK_HIP(e);
        c->gkey = key;
    }
    S2V_CHECK_HIP(hipGraphLaunch(c->gexec, st));
    return 0;
}

extern "C" int s2v_cfg_pair(s2v_ctx* c, void** dev_ptr, int64_t* bytes_per_half) {
    S2V_REQUIRE(c && c->ws && dev_ptr && bytes_per_half, "s2v_cfg_pair: no workspace");
    *dev_ptr = c->noise_pred;
    *bytes_per_half = pair_half_bytes(c);
    return 0;
}

extern "C" int s2v_denoise_split_end(s2v_ctx* c, void* latents, float* x0_hist, const void* noise, s2v_stream stream) {
    S2V_REQUIRE(c && latents, "s2v_denoise_split_end: null argument");
    S2V_REQUIRE(c->ws && c->B == 1, "s2v_denoise_split_end: a geometry with B = 1 is required (s2v_denoise_split_begin ran before)");
    // the step's coefficients live on the device since s2v_denoise_split_begin: one end per begin, and a DPM step brings its noise and x0 history
    S2V_REQUIRE(c->split_kind >= 0, "s2v_denoise_split_end: no s2v_denoise_split_begin is pending (its coefficients are the step's)");
    S2V_REQUIRE(c->split_kind == 0 || (noise && x0_hist), "s2v_denoise_split_end: DPM needs noise and x0_hist");
    c->split_kind = -1;
    SchedArgs a{};
    a.noise_pred = c->noise_pred; a.latents_in = latents; a.latents_out = latents; a.x0_hist = x0_hist; a.noise = noise;
    a.n = (int64_t)c->F * c->cfg.out_channels * c->H * c->W; a.cfg = 1; a.coef = c->coef_dev;  // the coefficients s2v_denoise_split_begin uploaded
    return launch_sched_step(a, c->dtype, (hipStream_t)stream);
}

extern "C" int s2v_denoise_step_cfg_parallel(s2v_ctx* c, s2v_rccl_comm* comm, int32_t slot, void* latents, float timestep,
                                             const s2v_sched_coef* coef_host, float* x0_hist, const void* noise, int32_t use_graph,
                                             s2v_stream stream) {
    S2V_REQUIRE(c && comm && latents && coef_host, "s2v_denoise_step_cfg_parallel: null argument");
    S2V_REQUIRE(coef_host->kind == 0 || (noise && x0_hist), "s2v_denoise_step_cfg_parallel: DPM needs noise and x0_hist");
    S2V_TRY(s2v_rccl_pair_check(comm, slot));  // a two-rank communicator whose rank IS the slot: the in-place all-gather puts rank r's bytes into half r
    S2V_TRY(s2v_denoise_split_begin(c, latents, timestep, coef_host, slot, use_graph, stream));
    const int64_t half = pair_half_bytes(c);
    S2V_TRY(s2v_rccl_allgather(comm, c->noise_pred + (int64_t)slot * half, c->noise_pred, half, stream));  // in place: rank r owns half r
    return s2v_denoise_split_end(c, latents, x0_hist, noise, stream);
}

// Per-kernel-class timing (HIP events recorded on the launch stream around every launch of the class).
// classes: 0 qkv GEMM, 1 attention, 2 out-proj GEMM, 3 FF1 GEMM, 4 FF2 GEMM, 5 LN-modulate, 6 qk-norm/rope/V^T
extern "C" int s2v_profile_enable(s2v_ctx* c, int32_t on) {
    S2V_REQUIRE(c, "null context");
    c->prof_on = on != 0;
    for (int k = 0; k < PK_NUM; ++k) c->prof_used[k] = 0;
    c->clk_rec.clear();
    if (on && !c->clk_buf) S2V_CHECK_HIP(hipMalloc((void**)&c->clk_buf, sizeof(long long) * 4 * CLK_SLOTS));
    if (on) S2V_CHECK_HIP(hipMemset(c->clk_buf, 0, sizeof(long long) * 4 * CLK_SLOTS));
    return 0;
}
// Average shader clock (MHz) under the launches of every class since s2v_profile_enable(1): sum of s_memtime spans / sum of s_memrealtime spans
// (100 MHz ticks) of the stamp pairs around them; 0 for a class without stamped launches.  Synchronises; clears the records.
extern "C" int s2v_profile_read_clocks(s2v_ctx* c, float* mhz_by_class, int32_t nclass) {
    S2V_REQUIRE(c && mhz_by_class, "s2v_profile_read_clocks: null argument");
    S2V_CHECK_HIP(hipDeviceSynchronize());
    std::vector<long long> h((size_t)4 * c->clk_rec.size());
    if (!h.empty()) S2V_CHECK_HIP(hipMemcpy(h.data(), c->clk_buf, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
    double cyc[PK_NUM] = {0}, ticks[PK_NUM] = {0};
    for (auto& r : c->clk_rec) {
        const long long* s = h.data() + 4 * r.second;
        if (s[2] > s[0] && s[3] > s[1]) { cyc[r.first] += (double)(s[2] - s[0]); ticks[r.first] += (double)(s[3] - s[1]); }
    }
    for (int k = 0; k < nclass && k < PK_NUM; ++k) mhz_by_class[k] = ticks[k] > 0 ? (float)(cyc[k] / ticks[k] * 100.0) : 0.f;
    c->clk_rec.clear();
    // the slots are handed out again from 0: a later pass that did not go through s2v_profile_enable(1) must not read these stamps
    if (c->clk_buf) S2V_CHECK_HIP(hipMemset(c->clk_buf, 0, sizeof(long long) * 4 * CLK_SLOTS));
    return 0;
}
// Synchronises, then returns total milliseconds and launch counts per class since the last read; resets.
extern "C" int s2v_profile_read(s2v_ctx* c, float* ms_by_class, int32_t* launches_by_class, int32_t nclass) {
    S2V_REQUIRE(c && ms_by_class && launches_by_class, "s2v_profile_read: null argument");
    S2V_CHECK_HIP(hipDeviceSynchronize());
    for (int k = 0; k < nclass && k < PK_NUM; ++k) {
        float tot = 0.f;
        for (size_t i = 0; i < c->prof_used[k]; ++i) {
            float ms = 0.f;
            S2V_CHECK_HIP(hipEventElapsedTime(&ms, c->prof_ev[k][i].first, c->prof_ev[k][i].second));
            tot += ms;
        }
        ms_by_class[k] = tot;
        launches_by_class[k] = (int32_t)c->prof_used[k];
        c->prof_used[k] = 0;
    }
    return 0;
}

extern "C" int s2v_last_noise_pred(s2v_ctx* c, void** dev_ptr) {
    S2V_REQUIRE(c && c->ws && dev_ptr, "s2v_last_noise_pred: no workspace");
    *dev_ptr = c->noise_pred;
    return 0;
}

// ------------------------------------------------------------------------------------------------------
extern "C" int s2v_op_linear(const void* A, const void* W, const void* bias, void* C, int32_t M, int32_t N, int32_t K,
                             int32_t epilogue, int32_t dtype, int32_t impl, s2v_stream stream) {
    S2V_REQUIRE(A && W && C, "s2v_op_linear: null argument");
    S2V_REQUIRE(epilogue == EPI_BIAS || epilogue == EPI_BIAS_GELU, "s2v_op_linear: epilogue must be 0 or 1");
    GemmArgs g{};
    g.A = A; g.lda = K; g.W = W; g.ldw = K; g.bias = bias; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K;
    if (impl == 0) {
        S2V_REQUIRE(dtype == S2V_DTYPE_BF16, "s2v_op_linear: the MFMA kernel is bf16 only");
        S2V_REQUIRE(M % 128 == 0 && N % 128 == 0, "s2v_op_linear: impl 0 needs M and N padded to 128 by the caller");
        g.a_rows_padded = M;  // the 256-row ring kernel is used when M is a multiple of 256
        return launch_gemm_bf16(g, epilogue, (hipStream_t)stream);
    }
    if (impl == 2) {  // the engine's split-K form of a few-tile GEMM (linear()), with a workspace of its own: synchronous
        S2V_REQUIRE(dtype == S2V_DTYPE_BF16 && M % 256 == 0 && N % 256 == 0, "s2v_op_linear: impl 2 is bf16 with M and N multiples of 256");
        int dev = 0, ncu = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        const int64_t tiles = (int64_t)(M / 256) * (N / 256);
        const int S = gemm_choose_splitk(tiles, K, ncu);
        S2V_REQUIRE(S > 1, "s2v_op_linear: impl 2: this shape does not split (tiles * 2 <= CUs, K / S a multiple of 128 and >= 1024)");
        char* ws = nullptr;
        const size_t pb = (size_t)S * tiles * 262144;
        S2V_CHECK_HIP(hipMalloc((void**)&ws, pb + tiles * 4));
        g.a_rows_padded = M; g.w_rows_padded = N;
        g.splitk = S; g.sk_ws = (float*)ws; g.sk_cnt = (unsigned*)(ws + pb);
        int rc = hipMemsetAsync(ws + pb, 0, tiles * 4, (hipStream_t)stream) == hipSuccess ? 0 : -2;
        if (rc == 0) rc = launch_gemm_bf16(g, epilogue, (hipStream_t)stream);
        if (rc == 0) rc = launch_gemm_bf16(g, epilogue, (hipStream_t)stream);  // twice: the counters must be back at zero
        hipStreamSynchronize((hipStream_t)stream);
        hipFree(ws);
        return rc;
    }
    if (impl == 4) {  // fp16 operands on v_mfma_f32_32x32x16_f16 (what the fp16 engine runs); M, N padded to 128 by the caller
        S2V_REQUIRE(dtype == S2V_DTYPE_F16, "s2v_op_linear: impl 4 is fp16 only");
        S2V_REQUIRE(M % 128 == 0 && N % 128 == 0, "s2v_op_linear: impl 4 needs M and N padded to 128 by the caller");
        g.a_rows_padded = M; g.w_rows_padded = N;
        return launch_gemm_f16(g, epilogue, (hipStream_t)stream);
    }
    if (impl == 3 || (impl >= 30 && impl <= 33)) {  // fp32 operands on the fp32 matrix pipe (what the fp32 engine runs; bit-identical to impl 1)
        S2V_REQUIRE(dtype == S2V_DTYPE_F32, "s2v_op_linear: impl 3 / 30 .. 33 are fp32 only");
        if (impl >= 30) g.tile = impl;  // 30 .. 33: force the 128 x 128 / 128 x 64 / 64 x 128 / 64 x 64 tile (3: the launcher's own choice)
        return launch_gemm_f32m(g, epilogue, (hipStream_t)stream);
    }
    g.valu_only = 1;
    return launch_gemm_simple(g, epilogue, dtype, (hipStream_t)stream);
}

// C = epilogue(dequant(quant_rows(A) . quant_rows(W)^T) + bias): both bf16 operands are quantised per row to e4m3 (dynamic
// per-token / per-output-channel scales) into `scratch` and multiplied on the fp8 matrix cores.  M, N multiples of 256, K of 128.
extern "C" int s2v_op_linear_fp8(const void* A, const void* W, const void* bias, void* C, int32_t M, int32_t N, int32_t K,
                                 int32_t epilogue, void* scratch, int64_t scratch_bytes, s2v_stream stream) {
    S2V_REQUIRE(A && W && C && scratch, "s2v_op_linear_fp8: null argument");
    S2V_REQUIRE(epilogue == EPI_BIAS || epilogue == EPI_BIAS_GELU, "s2v_op_linear_fp8: epilogue must be 0 or 1");
    S2V_REQUIRE(M % 256 == 0 && N % 256 == 0 && K % 128 == 0, "s2v_op_linear_fp8: M, N multiples of 256 and K of 128");
    const int64_t need = (int64_t)M * K + (int64_t)N * K + 4 * ((int64_t)M + N);
    S2V_REQUIRE(scratch_bytes >= need, "s2v_op_linear_fp8: scratch too small (M*K + N*K + 4*(M+N) bytes)");
    hipStream_t st = (hipStream_t)stream;
    char* aq = (char*)scratch;
    char* wq = aq + (int64_t)M * K;
    float* as = (float*)(wq + (int64_t)N * K);
    float* ws = as + M;
    S2V_TRY(launch_quant_rows_fp8(A, K, M, K, aq, as, st));
    S2V_TRY(launch_quant_rows_fp8(W, K, N, K, wq, ws, st));
    GemmArgs g{};
    g.A = aq; g.lda = K; g.W = wq; g.ldw = K; g.bias = bias; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K;
    g.a_rows_padded = M; g.w_rows_padded = N; g.a_scale = as; g.w_scale = ws;
    return launch_gemm_fp8(g, epilogue, st);
}

// FeedForward (attention.py:1237-1243) on the fp8 matrix cores as the fp8 engine runs it: x, W1, W2 quantised per row (e4m3, amax / 448);
// h = GELU(x W1^T + b1) either (mx = 1) left by the FF1 epilogue as MX e4m3 -- one power-of-two scale per 32 columns -- and read in
// place by the FF2, or (mx = 0) written in bf16 and quantised per row by a separate pass; out = h W2^T + b2, bf16.  Allocates its
// scratch and synchronises: a test / micro-benchmark entry point.
extern "C" int s2v_op_ff_fp8(const void* x, const void* w1, const void* b1, const void* w2, const void* b2, void* out, int32_t M, int32_t D,
                             int32_t F, int32_t mx, s2v_stream stream) {
    S2V_REQUIRE(x && w1 && w2 && out, "s2v_op_ff_fp8: null argument");
    S2V_REQUIRE(M % 256 == 0 && D % 256 == 0 && F % 256 == 0, "s2v_op_ff_fp8: M, D, F must be multiples of 256");
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = (size_t)M * D + (size_t)F * D + (size_t)D * F + 4 * ((size_t)2 * M + F + D) + (size_t)M * F * 3 + (size_t)M * F / 32 + 4096;
    char* p = nullptr;
    S2V_CHECK_HIP(hipMalloc((void**)&p, bytes));
    char* xq = p; char* w1q = xq + (size_t)M * D; char* w2q = w1q + (size_t)F * D;
    float* xs = (float*)(w2q + (size_t)D * F); float* hs_row = xs + M; float* w1s = hs_row + M; float* w2s = w1s + F;
    char* hb = (char*)(w2s + D);                 // bf16 h (mx = 0)
    char* hq = hb + (size_t)M * F * 2;           // e4m3 h
    unsigned char* hsc = (unsigned char*)hq + (size_t)M * F;  // MX block scales
    int rc = 0;
    auto run = [&]() -> int {
        S2V_TRY(launch_quant_rows_fp8(x, D, M, D, xq, xs, st));
        S2V_TRY(launch_quant_rows_fp8(w1, D, F, D, w1q, w1s, st));
        S2V_TRY(launch_quant_rows_fp8(w2, F, D, F, w2q, w2s, st));
        GemmArgs f{};
        f.A = xq; f.lda = D; f.W = w1q; f.ldw = D; f.bias = b1; f.C = hb; f.ldc = F; f.M = M; f.N = F; f.K = D;
        f.a_rows_padded = M; f.w_rows_padded = F; f.a_scale = xs; f.w_scale = w1s;
        if (mx) { f.mx_out_q = (unsigned char*)hq; f.mx_out_s = hsc; f.mx_rows = M; }
        S2V_TRY(launch_gemm_fp8(f, EPI_BIAS_GELU, st));
        GemmArgs g{};
        g.A = hq; g.lda = F; g.W = w2q; g.ldw = F; g.bias = b2; g.C = out; g.ldc = D; g.M = M; g.N = D; g.K = F;
        g.a_rows_padded = M; g.w_rows_padded = D; g.w_scale = w2s;
        if (mx) { g.mx_a_s = hsc; g.mx_rows = M; }
        else {
            S2V_TRY(launch_quant_rows_fp8(hb, F, M, F, hq, hs_row, st));
            g.a_scale = hs_row;
        }
        return launch_gemm_fp8(g, EPI_BIAS, st);
    };
    rc = run();
    hipStreamSynchronize(st);
    hipFree(p);
    return rc;
}

// Census of the attention kernel's deferred-maximum slow path since the last reset: slow = slow paths taken, total = (wave, KV tile) pairs run by
// the four-wave kernels (0 / 0 when only attn_pp ran).  Synchronises the device.  With attn_p_format 1 the threshold is 2^14 instead of 2^64: a
// caller that sees more than a fraction of a percent of slow paths on its data switches back with s2v_set_attn_p_format.
extern "C" int s2v_attn_slow_stats(s2v_ctx* c, uint64_t* slow, uint64_t* total, int32_t reset) {
    S2V_REQUIRE(c && slow && total, "s2v_attn_slow_stats: null argument");
    unsigned long long h[512];
    S2V_CHECK_HIP(hipDeviceSynchronize());
    S2V_CHECK_HIP(hipMemcpy(h, c->attn_stats, 4096, hipMemcpyDeviceToHost));
    uint64_t s = 0, t = 0;
    for (int i = 0; i < 256; ++i) { s += h[2 * i]; t += h[2 * i + 1]; }
    *slow = s; *total = t;
    if (reset) S2V_CHECK_HIP(hipMemset(c->attn_stats, 0, 4096));
    return 0;
}
// Change attn_p_format (include/s2v_hip.h, s2v_model_config) of a live context; a captured step is dropped and re-captured at its next use.
extern "C" int s2v_set_attn_p_format(s2v_ctx* c, int32_t fmt) {
    S2V_REQUIRE(c, "s2v_set_attn_p_format: null context");
    S2V_REQUIRE(fmt == 0 || fmt == 1, "s2v_set_attn_p_format: 0 (bf16) or 1 (fp16)");
    if (c->attn_p16 != (fmt == 1)) {
        S2V_CHECK_HIP(hipDeviceSynchronize());
        c->attn_p16 = fmt == 1;
        if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
    }
    return 0;
}

extern "C" int s2v_op_mod_gemv(const void* emb, const void* W, const void* bias, void* out, int32_t B, int32_t temb_dim,
                               int64_t rows, int32_t dtype, int32_t impl, s2v_stream stream) {
    S2V_REQUIRE(emb && W && out, "s2v_op_mod_gemv: null argument");
    S2V_REQUIRE(dtype == S2V_DTYPE_BF16 || dtype == S2V_DTYPE_F32 || dtype == S2V_DTYPE_F16, "s2v_op_mod_gemv: dtype must be f32, bf16 or f16");
    return launch_mod_gemv(emb, B, temb_dim, W, bias, rows, out, dtype, (hipStream_t)stream, impl == 1);
}

// attention with QK^T on the scaled fp8 MFMA as weight_format 2 runs it: q (times scale * log2 e) and k of the bf16 qkv rows become MX e4m3
// images in `scratch` (also returned to the caller for inspection: layout in kernels.h AttnArgs::q8 ... k8s, offsets below), V^T and P.V stay bf16.
extern "C" int s2v_op_attention_fp8qk(const void* qkv, void* vt_scratch, void* scratch, int64_t scratch_bytes, void* out, int32_t B, int32_t H,
                                      int32_t Ntok, s2v_stream stream) {
    S2V_REQUIRE(qkv && vt_scratch && scratch && out, "s2v_op_attention_fp8qk: null argument");
    const int D = H * 64;
    const int64_t ntok_pad = rup(Ntok, 64), BH = (int64_t)B * H;
    const int64_t oq8 = 0, oq8s = oq8 + rup(BH * Ntok * 64, 256), ok8 = oq8s + rup(BH * Ntok * 2, 256), ok8s = ok8 + rup(BH * ntok_pad * 64, 256);
    S2V_REQUIRE(scratch_bytes >= ok8s + BH * ntok_pad * 4, "s2v_op_attention_fp8qk: scratch too small (B*H*(66*Ntok + 68*ntok_pad) + 1024 bytes)");
    AttnArgs a{};
    a.qkv = qkv; a.ld_qkv = 3 * D; a.out = out; a.ld_out = D; a.B = B; a.H = H; a.Ntok = Ntok; a.scale = 0.125f;
    a.ntok_pad = (int)ntok_pad; a.vt = vt_scratch;
    char* w = (char*)scratch;
    a.q8 = (unsigned char*)(w + oq8); a.q8s = (unsigned short*)(w + oq8s); a.k8 = (unsigned char*)(w + ok8); a.k8s = (unsigned*)(w + ok8s);
    hipStream_t st = (hipStream_t)stream;
    S2V_TRY(launch_v_transpose(qkv, 3 * D, B, H, Ntok, vt_scratch, a.ntok_pad, st));
    S2V_TRY(launch_qk_quant_mx(qkv, 3 * D, B, H, Ntok, a.ntok_pad, a.scale * 1.4426950408889634f, (unsigned char*)a.q8, (unsigned short*)a.q8s,
                               (unsigned char*)a.k8, (unsigned*)a.k8s, st));
    return launch_attn_q4f(a, false, st);
}

extern "C" int s2v_op_attention(const void* qkv, void* vt_scratch, void* out, int32_t B, int32_t H, int32_t Ntok,
                                int32_t dtype, int32_t impl, s2v_stream stream) {
    S2V_REQUIRE(qkv && out, "s2v_op_attention: null argument");
    const int D = H * 64;
    AttnArgs a{};
    a.qkv = qkv; a.ld_qkv = 3 * D; a.out = out; a.ld_out = D; a.B = B; a.H = H; a.Ntok = Ntok; a.scale = 0.125f;
    a.ntok_pad = (int)rup(Ntok, 64);
    hipStream_t st = (hipStream_t)stream;
    if (impl == 0 || impl == 3 || impl == 4) {  // 3: as 0 with P / V^T in fp16 where the four-wave kernel runs (attn_p_format 1); 4: attn_q4h at any length
        S2V_REQUIRE(dtype == S2V_DTYPE_BF16 && vt_scratch, "s2v_op_attention: impl 0 / 3 / 4 are bf16 and need vt_scratch");
        a.vt = vt_scratch;
        a.p16 = (impl == 4 || (impl == 3 && attn_runs_q4(Ntok, false))) ? 1 : 0;
        S2V_TRY(launch_v_transpose(qkv, 3 * D, B, H, Ntok, vt_scratch, a.ntok_pad, st, a.p16 != 0));
        if (impl == 4) return launch_attn_q4h(a, false, st);
        return launch_attn_bf16(a, st);
    }
    if (impl == 6) {  // fp16 on v_mfma_f32_32x32x16_f16 (what the fp16 engine runs); vt_scratch as for impl 0
        S2V_REQUIRE(dtype == S2V_DTYPE_F16 && vt_scratch, "s2v_op_attention: impl 6 is fp16 and needs vt_scratch");
        a.vt = vt_scratch;
        S2V_TRY(launch_v_transpose(qkv, 3 * D, B, H, Ntok, vt_scratch, a.ntok_pad, st, false));
        return launch_attn_f16(a, st);
    }
    if (impl == 5) {  // fp32 on the fp32 matrix pipe (what the fp32 engine runs)
        S2V_REQUIRE(dtype == S2V_DTYPE_F32 || dtype == S2V_DTYPE_F16, "s2v_op_attention: impl 5 is fp32 / fp16 only");
        return launch_attn_f32m(a, dtype, st);
    }
    a.valu_only = 1;
    return launch_attn_simple(a, dtype, st);
}
