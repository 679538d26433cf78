// T5 v1.1 ENCODER stack: the prompt embeddings of the pipeline.
// Replaces `self.text_encoder(text_input_ids.to(device))[0]` of pipelines/cogvideo/pipeline_cogvideox.py:227 for the model
// src/inference.py:183-187 loads (T5EncoderModel, T5-v1.1-XXL: 24 blocks, d_model 4096, 64 heads x 64, gated-GELU d_ff 10240).
// The arithmetic is transformers' (third party, not in the reference tree): models/t5/modeling_t5.py T5LayerNorm, T5Attention
// (no 1/sqrt(d), additive relative-position bias shared from block 0, softmax in fp32), T5DenseGatedActDense (gelu_new), T5Block.
// No attention mask (the pipeline passes none).  Rounding points follow the bf16 tensor ops of that implementation.
//
// GEMMs run on the kernels of gemm.hip (weights fused: [q|k|v] and [wi_0|wi_1]); new here: embedding gather, RMS norm,
// gated GELU, and a small attention kernel with the additive bias (226 tokens: one wave per query row is enough).
#define S2V_HOST
#include "common.h"
#include "kernels.h"
#include "../../include/s2v_hip.h"

#include <string>
#include <unordered_map>
#include <vector>

template <typename T>
__global__ void t5_embed_k(const long long* ids, const T* table, int M, int d, int vocab, T* out) {
    const int64_t total = (int64_t)M * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / d), c = (int)(i - (int64_t)m * d);
        long long id = ids[m];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        out[i] = table[(size_t)id * d + c];
    }
}

// T5LayerNorm: y = w * cast(x * rsqrt(mean(x^2) + eps)); one wave per row
template <typename T>
__global__ __launch_bounds__(256) void t5_rms_norm_k(const T* x, const T* w, int M, int d, float eps, T* out) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const T* xr = x + (size_t)m * d;
    T* o = out + (size_t)m * d;
    constexpr int VN = Vec16<T>::N;
    float ss = 0.f;
    if (d % (64 * VN) == 0 && d <= 64 * VN * 8) {  // the row in registers, 16-byte accesses (d_model 4096 in bf16: 8 rounds)
        float v[8][VN];
        const int nr = d / (64 * VN);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nr) Vec16<T>::ld(xr + (i * 64 + lane) * VN, v[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nr)
#pragma unroll
                for (int e = 0; e < VN; ++e) ss += v[i][e] * v[i][e];
        ss = wave_sum(ss);
        const float r = 1.0f / sqrtf(ss / (float)d + eps);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nr) {
                float wv[VN], y[VN];
                Vec16<T>::ld(w + (i * 64 + lane) * VN, wv);
#pragma unroll
                for (int e = 0; e < VN; ++e) y[e] = ET<T>::rnd(wv[e] * ET<T>::rnd(v[i][e] * r));
                Vec16<T>::st(o + (i * 64 + lane) * VN, y);
            }
        return;
    }
    for (int c = lane; c < d; c += 64) {
        const float v = ET<T>::ld(xr + c);
        ss += v * v;
    }
    ss = wave_sum(ss);
    const float r = 1.0f / sqrtf(ss / (float)d + eps);
    for (int c = lane; c < d; c += 64) ET<T>::st(o + c, ET<T>::rnd(ET<T>::ld(w + c) * ET<T>::rnd(ET<T>::ld(xr + c) * r)));
}

// T5DenseGatedActDense: out = gelu_new(a[:, :F]) * a[:, F:]   (both factors and the product rounded to T)
template <typename T>
__global__ void t5_gate_k(const T* a, int M, int F, T* out) {
    const int64_t total = (int64_t)M * F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / F), c = (int)(i - (int64_t)m * F);
        // NewGELUActivation is a chain of tensor ops, each rounding to T:
        //   0.5 * x * (1 + tanh(sqrt(2/pi) * (x + 0.044715 * pow(x, 3))))
        const float x = ET<T>::ld(a + (size_t)m * 2 * F + c);
        const float x3 = ET<T>::rnd(x * x * x);
        const float t1 = ET<T>::rnd(0.044715f * x3);
        const float t2 = ET<T>::rnd(x + t1);
        const float t3 = ET<T>::rnd(0.7978845608028654f * t2);
        const float t4 = ET<T>::rnd(tanhf(t3));
        const float t5 = ET<T>::rnd(1.0f + t4);
        const float t6 = ET<T>::rnd(0.5f * x);
        const float g = ET<T>::rnd(t6 * t5);
        const float u = ET<T>::ld(a + (size_t)m * 2 * F + F + c);
        ET<T>::st(out + i, ET<T>::rnd(g * u));
    }
}

// one wave per (b, h, query): scores = round(q . k) (no scale), + bias[h][q][k] (rounded), softmax in fp32, weights rounded
// to T, o = sum w_k v_k.  qkv [B*T][3*H*64] (q | k | v), head_dim 64.
template <typename T>
__global__ __launch_bounds__(256) void t5_attn_k(const T* qkv, const T* bias, int B, int H, int Tn, T* out) {
    __shared__ float sq[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
    const int D = H * 64, ld = 3 * D;
    const bool active = q < Tn;
    const int ql = active ? q : Tn - 1;
    const T* base = qkv + (size_t)b * Tn * ld;
    sq[wave][lane] = ET<T>::ld(base + (size_t)ql * ld + h * 64 + lane);
    __syncthreads();
    const T* brow = bias + ((size_t)h * Tn + ql) * Tn;
    // pass 1: row maximum of the rounded, biased scores
    float m = -INFINITY;
    for (int kv0 = 0; kv0 < Tn; kv0 += 64) {
        const int kv = kv0 + lane;
        float s = -INFINITY;
        if (kv < Tn) {
            const T* kp = base + (size_t)kv * ld + D + h * 64;
            float acc = 0.f;
#pragma unroll 8
            for (int d = 0; d < 64; ++d) acc = fmaf(sq[wave][d], ET<T>::ld(kp + d), acc);
            s = ET<T>::rnd(ET<T>::rnd(acc) + ET<T>::ld(brow + kv));
        }
        m = fmaxf(m, wave_max(s));
    }
    // pass 2: exp, sum, weighted values (scores recomputed: T is small)
    float l = 0.f;
    for (int kv0 = 0; kv0 < Tn; kv0 += 64) {
        const int kv = kv0 + lane;
        float p = 0.f;
        if (kv < Tn) {
            const T* kp = base + (size_t)kv * ld + D + h * 64;
            float acc = 0.f;
#pragma unroll 8
            for (int d = 0; d < 64; ++d) acc = fmaf(sq[wave][d], ET<T>::ld(kp + d), acc);
            p = expf(ET<T>::rnd(ET<T>::rnd(acc) + ET<T>::ld(brow + kv)) - m);
        }
        l += wave_sum(p);
    }
    const float inv = 1.0f / l;
    float o = 0.f;
    for (int kv0 = 0; kv0 < Tn; kv0 += 64) {
        const int kv = kv0 + lane;
        float p = 0.f;
        if (kv < Tn) {
            const T* kp = base + (size_t)kv * ld + D + h * 64;
            float acc = 0.f;
#pragma unroll 8
            for (int d = 0; d < 64; ++d) acc = fmaf(sq[wave][d], ET<T>::ld(kp + d), acc);
            p = ET<T>::rnd(expf(ET<T>::rnd(ET<T>::rnd(acc) + ET<T>::ld(brow + kv)) - m) * inv);
        }
        const int nv = min(64, Tn - kv0);
        for (int j = 0; j < nv; ++j) {
            const float pj = __shfl(p, j, 64);
            o = fmaf(pj, ET<T>::ld(base + (size_t)(kv0 + j) * ld + 2 * D + h * 64 + lane), o);
        }
    }
    if (active) ET<T>::st(out + (size_t)(b * Tn + q) * D + h * 64 + lane, o);
}

// The same arithmetic (same rounding points, same summation order: bit-identical to t5_attn_k) with the head's K and V staged ONCE per
// workgroup in LDS as fp32 (K rows padded to 65 words: lane = key reads are conflict-free) and four queries per wave sharing every K
// read: t5_attn_k fetched each key row from global memory three times per query with 2-byte strided loads -- 451 us per layer at
// T5-XXL, 2 x 226 tokens, 54 % of the encode.  Tn <= T5_ATTN_MAX_T (LDS: Tn * 129 words + the queries).
constexpr int T5_ATTN_MAX_T = 280, T5_ATTN_WAVES = 16, T5_ATTN_QB = 4 * T5_ATTN_WAVES, T5_ATTN_NC = (T5_ATTN_MAX_T + 63) / 64;
template <typename T>
__global__ __launch_bounds__(64 * T5_ATTN_WAVES) void t5_attn_lds_k(const T* qkv, const T* bias, int B, int H, int Tn, T* out) {
    extern __shared__ float t5s[];
    float* Ks = t5s;                         // [Tn][65]
    float* Vs = Ks + (size_t)Tn * 65;        // [Tn][64]
    float* Qs = Vs + (size_t)Tn * 64;        // [waves][4 queries][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const int D = H * 64, ld = 3 * D;
    const T* base = qkv + (size_t)b * Tn * ld;
    {   // staging: 16-byte chunks, every load of a thread in flight before the first LDS store (a load-store loop took one memory
        // round trip per element pair: 110 us per workgroup)
        constexpr int VN = Vec16<T>::N, CPR = 64 / VN;                         // chunks per 64-element row
        constexpr int NTH = 64 * T5_ATTN_WAVES, NCH = (T5_ATTN_MAX_T * CPR + NTH - 1) / NTH;  // chunks per thread and operand
        const int total = Tn * CPR;
        float kv_[2][NCH][VN];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = min((int)threadIdx.x + i * NTH, total - 1), kv = c / CPR, d = (c % CPR) * VN;
            Vec16<T>::ld(base + (size_t)kv * ld + D + h * 64 + d, kv_[0][i]);
            Vec16<T>::ld(base + (size_t)kv * ld + 2 * D + h * 64 + d, kv_[1][i]);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = threadIdx.x + i * NTH, kv = c / CPR, d = (c % CPR) * VN;
            if (c < total)
#pragma unroll
                for (int e = 0; e < VN; ++e) {
                    Ks[kv * 65 + d + e] = kv_[0][i][e];
                    Vs[kv * 64 + d + e] = kv_[1][i][e];
                }
        }
    }
    __syncthreads();
    float* qs = Qs + wave * 256;
    // sixteen waves (four per SIMD: the loops below are LDS-latency bound with one), four queries each
    const int q0 = blockIdx.x * T5_ATTN_QB + wave * 4;
    if (q0 < Tn) {
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) qs[qi * 64 + lane] = ET<T>::ld(base + (size_t)min(q0 + qi, Tn - 1) * ld + h * 64 + lane);
        // (a wave reads back only what it wrote: no barrier)
        float acc[4][T5_ATTN_NC];
#pragma unroll
        for (int qi = 0; qi < 4; ++qi)
#pragma unroll
            for (int c = 0; c < T5_ATTN_NC; ++c) acc[qi][c] = 0.f;
        int krow[T5_ATTN_NC];
#pragma unroll
        for (int c = 0; c < T5_ATTN_NC; ++c) krow[c] = min(c * 64 + lane, Tn - 1) * 65;
#pragma unroll 4
        for (int d = 0; d < 64; ++d) {
            float kd[T5_ATTN_NC];
#pragma unroll
            for (int c = 0; c < T5_ATTN_NC; ++c) kd[c] = Ks[krow[c] + d];
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const float qv = qs[qi * 64 + d];
#pragma unroll
                for (int c = 0; c < T5_ATTN_NC; ++c) acc[qi][c] = fmaf(qv, kd[c], acc[qi][c]);
            }
        }
        float pw[4][T5_ATTN_NC];  // normalised, rounded weights of the four queries
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const int q = min(q0 + qi, Tn - 1);
            const T* brow = bias + ((size_t)h * Tn + q) * Tn;
            float sc[T5_ATTN_NC], m = -INFINITY;
#pragma unroll
            for (int c = 0; c < T5_ATTN_NC; ++c) {
                const int kv = c * 64 + lane;
                sc[c] = kv < Tn ? ET<T>::rnd(ET<T>::rnd(acc[qi][c]) + ET<T>::ld(brow + min(kv, Tn - 1))) : -INFINITY;
                m = fmaxf(m, sc[c]);
            }
            m = wave_max(m);
            float l = 0.f;
#pragma unroll
            for (int c = 0; c < T5_ATTN_NC; ++c)
                if (c * 64 < Tn) l += wave_sum(c * 64 + lane < Tn ? expf(sc[c] - m) : 0.f);
            const float inv = 1.0f / l;
#pragma unroll
            for (int c = 0; c < T5_ATTN_NC; ++c) pw[qi][c] = c * 64 + lane < Tn ? ET<T>::rnd(expf(sc[c] - m) * inv) : 0.f;
        }
        // o[qi][lane = d] = sum over keys in order: one V read feeds the four queries, the weight of key j comes by v_readlane
        // (__shfl is a ds_bpermute per key and query: 110 us of LDS traffic per workgroup)
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < T5_ATTN_NC; ++c) {
            if (c * 64 >= Tn) break;
            const int nv = min(64, Tn - c * 64);
            for (int j = 0; j < nv; ++j) {
                const float v = Vs[(c * 64 + j) * 64 + lane];
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) o[qi] = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(pw[qi][c]), j)), v, o[qi]);
            }
        }
#pragma unroll
        for (int qi = 0; qi < 4; ++qi)
            if (q0 + qi < Tn) ET<T>::st(out + (size_t)(b * Tn + q0 + qi) * D + h * 64 + lane, o[qi]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct T5Layer { char *ln0, *wqkv, *wo, *ln1, *wi, *wff; };
struct T5Slot { char* dst; int64_t rows, cols; bool loaded; };

struct s2v_t5 {
    s2v_t5_config cfg;
    int dtype = 0, esz = 0, inner = 0;
    bool mfma = false, h16 = false, finalized = false, have_bias = false;
    int* inf_flag = nullptr;  // fp16 only: "any inf in the residual stream" of the clamp transformers applies after every sub-layer
    std::vector<T5Layer> layers;
    char *shared = nullptr, *final_ln = nullptr, *rel_table = nullptr;
    std::unordered_map<std::string, T5Slot> slots;
    std::vector<void*> allocs;
    // workspace for (B, T)
    int B = 0, T = 0;
    int64_t Mpad = 0;
    char *X = nullptr, *Xn = nullptr, *QKV = nullptr, *AO = nullptr, *FF = nullptr, *G = nullptr, *bias = nullptr;
    char* sk = nullptr;  // split-K partial tiles (num_cus x 256 KiB) followed by num_cus arrival counters
    int num_cus = 256;
    std::vector<void*> ws_allocs;
    // every weight lives in ONE arena (s2v_t5_weight_arena: the replica broadcast); sized by a first pass of the plan
    char* arena = nullptr;
    int64_t arena_bytes = 0, arena_off = 0;
    bool sizing = false;
};

static int64_t rup_(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

static int t5_alloc(s2v_t5* t, char** p, int64_t bytes, bool ws = false) {
    if (!ws) {  // weight carve-out, 256-byte aligned; the arena is zero-initialised
        *p = t->sizing ? nullptr : t->arena + t->arena_off;
        t->arena_off += rup_(bytes > 0 ? bytes : 16, 256);
        return 0;
    }
    void* q = nullptr;
    S2V_CHECK_HIP(hipMalloc(&q, (size_t)(bytes > 0 ? bytes : 16)));
    S2V_CHECK_HIP(hipMemset(q, 0, (size_t)(bytes > 0 ? bytes : 16)));
    (ws ? t->ws_allocs : t->allocs).push_back(q);
    *p = (char*)q;
    return 0;
}

extern "C" void s2v_t5_destroy(s2v_t5* t) {
    if (!t) return;
    (void)hipDeviceSynchronize();
    for (void* p : t->allocs) (void)hipFree(p);
    for (void* p : t->ws_allocs) (void)hipFree(p);
    if (t->arena) (void)hipFree(t->arena);
    delete t;
}

extern "C" int s2v_t5_create(const s2v_t5_config* cfg, s2v_t5** out) {
    S2V_REQUIRE(cfg && out, "s2v_t5_create: null argument");
    S2V_REQUIRE(cfg->dtype == S2V_DTYPE_F32 || cfg->dtype == S2V_DTYPE_BF16 || cfg->dtype == S2V_DTYPE_F16, "s2v_t5_create: unsupported dtype");
    S2V_REQUIRE(cfg->d_kv == 64, "s2v_t5_create: head dimension must be 64 (T5 v1.1 XXL)");
    S2V_REQUIRE(cfg->num_layers > 0 && cfg->num_heads > 0 && cfg->d_model > 0 && cfg->d_ff > 0 && cfg->vocab_size > 0,
                "s2v_t5_create: bad model size");
    S2V_REQUIRE(cfg->d_model % 8 == 0 && cfg->d_ff % 8 == 0, "s2v_t5_create: d_model and d_ff must be multiples of 8");
    s2v_t5* t = new s2v_t5();
    t->cfg = *cfg;
    t->dtype = cfg->dtype;
    t->esz = cfg->dtype == S2V_DTYPE_F32 ? 4 : 2;
    t->inner = cfg->num_heads * cfg->d_kv;
    t->mfma = cfg->dtype == S2V_DTYPE_BF16 && !cfg->force_simple && cfg->d_model % 64 == 0 && cfg->d_ff % 64 == 0;
    t->h16 = cfg->dtype == S2V_DTYPE_F16 && !cfg->force_simple;  // fp16 (src/inference.py:209,214: text_encoder.to(device, dtype=weight_dtype)): linears on gemm_f16
    auto build = [&]() -> int {
        const int64_t d = cfg->d_model, in = t->inner, F = cfg->d_ff, E = t->esz;
        int r = t5_alloc(t, &t->shared, (int64_t)cfg->vocab_size * d * E);
        if (!r) r = t5_alloc(t, &t->final_ln, d * E);
        if (!r) r = t5_alloc(t, &t->rel_table, (int64_t)cfg->relative_attention_num_buckets * cfg->num_heads * E);
        t->slots["shared.weight"] = T5Slot{t->shared, cfg->vocab_size, d, false};
        t->slots["encoder.final_layer_norm.weight"] = T5Slot{t->final_ln, d, 1, false};
        t->slots["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"] =
            T5Slot{t->rel_table, cfg->relative_attention_num_buckets, cfg->num_heads, false};
        t->layers.resize(cfg->num_layers);
        char nm[160];
        for (int i = 0; i < cfg->num_layers && !r; ++i) {
            T5Layer& L = t->layers[i];
            // weight rows padded to the 256-column GEMM tile (zero rows)
            if (!r) r = t5_alloc(t, &L.ln0, d * E);
            if (!r) r = t5_alloc(t, &L.wqkv, rup_(3 * in, 256) * d * E);
            if (!r) r = t5_alloc(t, &L.wo, rup_(d, 256) * in * E);
            if (!r) r = t5_alloc(t, &L.ln1, d * E);
            if (!r) r = t5_alloc(t, &L.wi, rup_(2 * F, 256) * d * E);
            if (!r) r = t5_alloc(t, &L.wff, rup_(d, 256) * F * E);
            if (r) break;
            const char* qkvn[3] = {"q", "k", "v"};
            for (int j = 0; j < 3; ++j) {
                snprintf(nm, sizeof(nm), "encoder.block.%d.layer.0.SelfAttention.%s.weight", i, qkvn[j]);
                t->slots[nm] = T5Slot{L.wqkv + (int64_t)j * in * d * E, in, d, false};
            }
            snprintf(nm, sizeof(nm), "encoder.block.%d.layer.0.SelfAttention.o.weight", i);
            t->slots[nm] = T5Slot{L.wo, d, in, false};
            snprintf(nm, sizeof(nm), "encoder.block.%d.layer.0.layer_norm.weight", i);
            t->slots[nm] = T5Slot{L.ln0, d, 1, false};
            snprintf(nm, sizeof(nm), "encoder.block.%d.layer.1.layer_norm.weight", i);
            t->slots[nm] = T5Slot{L.ln1, d, 1, false};
            snprintf(nm, sizeof(nm), "encoder.block.%d.layer.1.DenseReluDense.wi_0.weight", i);
            t->slots[nm] = T5Slot{L.wi, F, d, false};
            snprintf(nm, sizeof(nm), "encoder.block.%d.layer.1.DenseReluDense.wi_1.weight", i);
            t->slots[nm] = T5Slot{L.wi + F * d * E, F, d, false};
            snprintf(nm, sizeof(nm), "encoder.block.%d.layer.1.DenseReluDense.wo.weight", i);
            t->slots[nm] = T5Slot{L.wff, d, F, false};
        }
        return r;
    };
    t->sizing = true;
    int r = build();
    if (!r) {
        t->arena_bytes = t->arena_off;
        if (hipMalloc((void**)&t->arena, (size_t)t->arena_bytes) != hipSuccess || hipMemset(t->arena, 0, (size_t)t->arena_bytes) != hipSuccess)
            r = s2v_fail(__FILE__, __LINE__, "s2v_t5_create: weight arena allocation failed", -2);
    }
    if (!r) {
        t->sizing = false;
        t->arena_off = 0;
        t->slots.clear();
        r = build();
    }
    if (r) { s2v_t5_destroy(t); return r; }
    *out = t;
    return 0;
}

/* One device range holding every weight, for the replica broadcast; the receiver calls s2v_t5_mark_weights_loaded. */
extern "C" int s2v_t5_weight_arena(s2v_t5* t, void** dev_ptr, int64_t* bytes) {
    S2V_REQUIRE(t && dev_ptr && bytes, "s2v_t5_weight_arena: null argument");
    *dev_ptr = t->arena;
    *bytes = t->arena_bytes;
    return 0;
}
extern "C" int s2v_t5_mark_weights_loaded(s2v_t5* t) {
    S2V_REQUIRE(t, "s2v_t5_mark_weights_loaded: null argument");
    for (auto& kv : t->slots) kv.second.loaded = true;
    t->finalized = true;
    return 0;
}

extern "C" int s2v_t5_load_weight(s2v_t5* t, const char* name, const void* dev_ptr, const int64_t* shape, int32_t ndim,
                                  int32_t src_dtype, s2v_stream stream) {
    S2V_REQUIRE(t && name && dev_ptr && shape, "s2v_t5_load_weight: null argument");
    S2V_REQUIRE(src_dtype == S2V_DTYPE_F32 || src_dtype == S2V_DTYPE_BF16 || src_dtype == S2V_DTYPE_F16, "s2v_t5_load_weight: unsupported dtype");
    std::string key = name;
    if (key == "encoder.embed_tokens.weight") key = "shared.weight";  // tied
    auto it = t->slots.find(key);
    if (it == t->slots.end()) {
        std::string m = std::string("s2v_t5_load_weight: unknown tensor name: ") + name;
        return s2v_fail(__FILE__, __LINE__, m.c_str(), -3);
    }
    T5Slot& s = it->second;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    if (n != s.rows * s.cols || shape[0] != s.rows) {
        std::string m = std::string("s2v_t5_load_weight: shape mismatch for ") + name;
        return s2v_fail(__FILE__, __LINE__, m.c_str(), -3);
    }
    S2V_TRY(launch_convert(dev_ptr, src_dtype, s.dst, t->dtype, n, (hipStream_t)stream));
    s.loaded = true;
    return 0;
}

extern "C" int s2v_t5_finalize(s2v_t5* t) {
    S2V_REQUIRE(t, "null t5");
    for (auto& kv : t->slots)
        if (!kv.second.loaded) {
            std::string m = std::string("s2v_t5_finalize: tensor was never loaded: ") + kv.first;
            return s2v_fail(__FILE__, __LINE__, m.c_str(), -3);
        }
    t->finalized = true;
    return 0;
}

static int t5_workspace(s2v_t5* t, int B, int T) {
    if (t->B == B && t->T == T) return 0;
    S2V_CHECK_HIP(hipDeviceSynchronize());
    for (void* p : t->ws_allocs) (void)hipFree(p);
    t->ws_allocs.clear();
    t->have_bias = false;
    const int64_t M = (int64_t)B * T, E = t->esz, d = t->cfg.d_model, in = t->inner, F = t->cfg.d_ff;
    t->Mpad = rup_(M, 256) + 256;
    S2V_TRY(t5_alloc(t, &t->X, t->Mpad * d * E, true));
    S2V_TRY(t5_alloc(t, &t->Xn, t->Mpad * d * E, true));
    S2V_TRY(t5_alloc(t, &t->QKV, t->Mpad * 3 * in * E, true));
    S2V_TRY(t5_alloc(t, &t->AO, t->Mpad * in * E, true));
    S2V_TRY(t5_alloc(t, &t->FF, t->Mpad * 2 * F * E, true));
    S2V_TRY(t5_alloc(t, &t->G, t->Mpad * F * E, true));
    S2V_TRY(t5_alloc(t, &t->bias, (int64_t)t->cfg.num_heads * T * T * E, true));
    t->sk = nullptr;
    if (t->mfma) {  // a prompt is a few hundred rows: every GEMM is 32-160 tiles, most of them split K (gemm_choose_splitk)
        int dev = 0, ncu = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0)
            t->num_cus = ncu;
        S2V_TRY(t5_alloc(t, &t->sk, (int64_t)t->num_cus * (262144 + 4), true));
        S2V_CHECK_HIP(hipMemset(t->sk + (int64_t)t->num_cus * 262144, 0, (size_t)t->num_cus * 4));
    }
    t->B = B; t->T = T;
    return 0;
}

// position_bias [H][T][T] in the model dtype (compute_bias of block 0, built on the host with the implementation's own
// torch ops: tables.t5_position_bias); valid until the token count changes
extern "C" int s2v_t5_set_position_bias(s2v_t5* t, const void* bias_dev, int32_t B, int32_t T, s2v_stream stream) {
    S2V_REQUIRE(t && bias_dev && B > 0 && T > 0, "s2v_t5_set_position_bias: bad argument");
    S2V_TRY(t5_workspace(t, B, T));
    S2V_CHECK_HIP(hipMemcpyAsync(t->bias, bias_dev, (size_t)t->cfg.num_heads * T * T * t->esz, hipMemcpyDeviceToDevice,
                                 (hipStream_t)stream));
    t->have_bias = true;
    return 0;
}

static int t5_linear(s2v_t5* t, const void* A, int lda, const void* W, void* C, int M, int N, int K, int epi, const void* R,
                     hipStream_t st) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.bias = nullptr; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K;
    g.R = R; g.ldr = N;
    g.a_rows_padded = (int)rup_(M, 256);
    g.w_rows_padded = (int)rup_(N, 256);
    if (t->mfma && K % 64 == 0) {
        const int64_t tiles = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
        const int S = t->sk ? gemm_choose_splitk(tiles, K, t->num_cus) : 1;
        if (S > 1 && S * tiles <= t->num_cus) {
            g.splitk = S; g.sk_ws = (float*)t->sk; g.sk_cnt = (unsigned*)(t->sk + (int64_t)t->num_cus * 262144);
        }
        return launch_gemm_bf16(g, epi, st);
    }
    if (t->h16 && gemm_f16_ok(g, epi)) return launch_gemm_f16(g, epi, st);
    g.valu_only = t->cfg.force_simple;
    return launch_gemm_simple(g, epi, t->dtype, st);
}

// transformers' T5Block (modeling_t5.py, T5Block.forward) after the self-attention and after the feed-forward sub-layer, fp16 only:
//   clamp_value = finfo(fp16).max - 1000 if isinf(hidden_states).any() else finfo(fp16).max;  hidden_states = clamp(hidden_states, -c, c)
// -- an overflowed residual stream (T5-XXL's feed-forward outputs exceed 65504) is pulled back to 64504 (stored as fp16: 64512) instead
// of carrying inf through the remaining blocks.  Two launches: the "any inf" flag over the whole tensor, then the clamp.
__global__ void t5_inf_flag_k(const f16_t* x, int64_t n, int* flag) {
    bool inf = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = (float)x[i];
        inf = inf || v == INFINITY || v == -INFINITY;
    }
    if (__any(inf) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
__global__ void t5_clamp_k(f16_t* x, int64_t n, const int* flag) {
    const float c = *flag ? 65504.0f - 1000.0f : 65504.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = (float)x[i];
        x[i] = (f16_t)fminf(fmaxf(v, -c), c);  // NaN: fmaxf / fminf return the other operand -- torch.clamp keeps NaN; keep it
        if (v != v) x[i] = (f16_t)v;
    }
}
static int t5_fp16_clamp(s2v_t5* t, int64_t n, hipStream_t st) {
    if (t->dtype != S2V_DTYPE_F16) return 0;
    if (!t->inf_flag) {
        void* q = nullptr;
        S2V_CHECK_HIP(hipMalloc(&q, 256));
        t->allocs.push_back(q);  // freed by s2v_t5_destroy
        t->inf_flag = (int*)q;
    }
    S2V_CHECK_HIP(hipMemsetAsync(t->inf_flag, 0, sizeof(int), st));
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(t5_inf_flag_k, dim3(grid), dim3(256), 0, st, (const f16_t*)t->X, n, t->inf_flag);
    hipLaunchKernelGGL(t5_clamp_k, dim3(grid), dim3(256), 0, st, (f16_t*)t->X, n, t->inf_flag);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

static inline unsigned grid1d(int64_t n) { return (unsigned)std::min<int64_t>((n + 255) / 256, 65535 * 16); }

extern "C" int s2v_t5_encode(s2v_t5* t, const int64_t* input_ids_dev, int32_t B, int32_t T, void* out, s2v_stream stream) {
    S2V_REQUIRE(t && input_ids_dev && out, "s2v_t5_encode: null argument");
    S2V_REQUIRE(t->finalized, "s2v_t5_encode: weights not finalized");
    S2V_REQUIRE(t->have_bias && t->B == B && t->T == T, "s2v_t5_encode: call s2v_t5_set_position_bias for this (B, T) first");
    hipStream_t st = (hipStream_t)stream;
    const int M = B * T, d = t->cfg.d_model, in = t->inner, F = t->cfg.d_ff, H = t->cfg.num_heads;
    const float eps = t->cfg.layer_norm_epsilon;
    const int dtype = t->dtype, Tn = T;  // Tn: the token count inside S2V_DT_DISPATCH bodies, where `T` names the storage type
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(t5_embed_k<T>, dim3(grid1d((int64_t)M * d)), dim3(256), 0, st, (const long long*)input_ids_dev,
                                              (const T*)t->shared, M, d, t->cfg.vocab_size, (T*)t->X))
    S2V_CHECK_HIP(hipGetLastError());
    const dim3 rows((M + 3) / 4), attn_grid((T + 3) / 4, H, B);
    for (auto& L : t->layers) {
        S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(t5_rms_norm_k<T>, rows, dim3(256), 0, st, (const T*)t->X, (const T*)L.ln0, M, d, eps, (T*)t->Xn))
        S2V_CHECK_HIP(hipGetLastError());
        S2V_TRY(t5_linear(t, t->Xn, d, L.wqkv, t->QKV, M, 3 * in, d, EPI_BIAS, nullptr, st));
        if (T <= T5_ATTN_MAX_T) {
            const dim3 g((T + T5_ATTN_QB - 1) / T5_ATTN_QB, H, B);
            const size_t lds = ((size_t)T * 129 + 256 * T5_ATTN_WAVES) * sizeof(float);
            S2V_DT_DISPATCH(dtype, {
                S2V_TRY(ensure_lds_attr((const void*)t5_attn_lds_k<T>, (int)lds));
                hipLaunchKernelGGL(t5_attn_lds_k<T>, g, dim3(64 * T5_ATTN_WAVES), lds, st, (const T*)t->QKV, (const T*)t->bias, B, H, Tn, (T*)t->AO);
            })
        } else {
            S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(t5_attn_k<T>, attn_grid, dim3(256), 0, st, (const T*)t->QKV, (const T*)t->bias, B, H, Tn, (T*)t->AO))
        }
        S2V_CHECK_HIP(hipGetLastError());
        S2V_TRY(t5_linear(t, t->AO, in, L.wo, t->X, M, d, in, EPI_BIAS_ADD, t->X, st));  // x = x + o(...)
        S2V_TRY(t5_fp16_clamp(t, (int64_t)M * d, st));
        S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(t5_rms_norm_k<T>, rows, dim3(256), 0, st, (const T*)t->X, (const T*)L.ln1, M, d, eps, (T*)t->Xn))
        S2V_CHECK_HIP(hipGetLastError());
        S2V_TRY(t5_linear(t, t->Xn, d, L.wi, t->FF, M, 2 * F, d, EPI_BIAS, nullptr, st));
        S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(t5_gate_k<T>, dim3(grid1d((int64_t)M * F)), dim3(256), 0, st, (const T*)t->FF, M, F, (T*)t->G))
        S2V_CHECK_HIP(hipGetLastError());
        S2V_TRY(t5_linear(t, t->G, F, L.wff, t->X, M, d, F, EPI_BIAS_ADD, t->X, st));  // x = x + wo(g * u)
        S2V_TRY(t5_fp16_clamp(t, (int64_t)M * d, st));
    }
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(t5_rms_norm_k<T>, rows, dim3(256), 0, st, (const T*)t->X, (const T*)t->final_ln, M, d, eps, (T*)out))
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// the block-0 relative attention bias table [num_buckets][H] in the model dtype (the host gathers it into [H][T][T])
extern "C" int s2v_t5_rel_table(s2v_t5* t, void** dev_ptr) {
    S2V_REQUIRE(t && dev_ptr, "s2v_t5_rel_table: null argument");
    *dev_ptr = t->rel_table;
    return 0;
}
