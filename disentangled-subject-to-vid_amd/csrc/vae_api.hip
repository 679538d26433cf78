// C ABI of the CogVideoX 3-D causal VAE decode (include/s2v_hip.h, s2v_vae_*): weight ingest / re-packing, buffer plan,
// and the launch sequence of AutoencoderKLCogVideoX.decode (autoencoder_kl_cogvideox.py:1231-1282, 1374-1455).
#define S2V_HOST
#include "common.h"
#include "kernels.h"
#include "vae_kernels.h"
#include "../../include/s2v_hip.h"

#include <cmath>
#include <cstdio>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>
#include <algorithm>
#include <cstdlib>

static inline int64_t rup64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

struct ConvL {
    int cin = 0, cout = 0, kt = 3;      // kt: 3 causal 3x3x3, 1 per-frame 3x3, 0 pointwise 1x1x1
    char* w = nullptr; char* b = nullptr;  // repacked [cout_pad][taps*cin], [cout]
    char* pad = nullptr; int64_t pad_bytes = 0;  // own zero-bordered operand buffer (conv cache in frames 0,1 when kt==3)
    int level = 0;
};
struct SNormL {
    int C = 0;
    char* gn_w = nullptr; char* gn_b = nullptr;
    float *wy = nullptr, *by = nullptr, *wb = nullptr, *bb = nullptr;
};
struct ResnetL { SNormL n1, n2; ConvL c1, c2, sc; bool has_sc = false; int cin = 0, cout = 0; };
struct StageL { std::vector<ResnetL> res; bool has_up = false; ConvL up; int compress_time = 0; };

struct VSlot { int kind; void* dst; int64_t d0, d1, taps; bool loaded; };  // kind 0 vector(model dtype) 1 conv repack 2 snorm W^T fp32 3 fp32 vector

struct s2v_vae {
    s2v_vae_config cfg;
    int dtype = 0, esz = 0, G = 0, Cz = 0;
    bool mfma = false, h16 = false, finalized = false;
    bool encoder = false;  // plan built by s2v_vae_enc_create (reference-image encode) instead of the decoder
    ConvL conv_in, conv_out;
    SNormL norm_out;
    std::vector<StageL> stages;
    std::unordered_map<std::string, VSlot> slots;
    std::vector<void*> allocs;
    // geometry-dependent
    int th = 0, tw = 0;        // allocated tile capacity (latent)
    int cur_h = 0, cur_w = 0;  // layout the zero borders are currently valid for
    int fmax[8] = {0};
    char* dense[3] = {nullptr, nullptr, nullptr};
    int64_t dense_bytes = 0;
    char* zq = nullptr; char* yt = nullptr; char* bt = nullptr;
    double* sums = nullptr; double* gn_part = nullptr;
    std::vector<void*> geo_allocs;
    std::vector<char*> tiles; std::vector<int> tile_h, tile_w;
    int tiles_F = 0;
    // Tiled decode: the tiles are independent until the blends, and at the latent levels a tile's GEMMs are a handful of workgroups
    // (30 x 45 x 2 rows = 11 row tiles on 256 CUs).  NWS workspace sets -- every geometry-dependent buffer above, once per set -- let
    // tiles k, k + 1, ... run concurrently on side streams; the members above always alias the ACTIVE set (ws_activate), so the launch
    // sequences are written once.
    struct WS {
        std::vector<char*> pads;  // in for_each_conv order
        char* dense[3] = {nullptr, nullptr, nullptr};
        char *zq = nullptr, *yt = nullptr, *bt = nullptr;
        double *sums = nullptr, *gn_part = nullptr;
        int cur_h = 0, cur_w = 0;
    };
    std::vector<WS> ws;
    int ws_active = 0, ws_req = 0;  // ws_req: sets asked for when ws was built (memory may have granted fewer)
    int fault_geo_at = 0, fault_geo_seen = 0;  // S2V_VAE_FAULT_GEO_ALLOC (dmalloc)
    int64_t set_bytes = 0;          // bytes of one set at the current capacity
    std::vector<hipStream_t> side;  // one per workspace set beyond the first
    std::vector<hipEvent_t> ev_side;
    hipEvent_t ev_fork = nullptr;
    // every weight lives in ONE arena (replicas receive it by one broadcast, s2v_vae_weight_arena): the plan is built twice,
    // a sizing pass (arena == nullptr) that only adds up the carve-outs, then the real pass that bumps through the arena
    char* arena = nullptr;
    int64_t arena_bytes = 0, arena_off = 0;
    bool sizing = false;
};

static int vfail(const char* m) { return s2v_fail(__FILE__, __LINE__, m, -1); }

template <typename T>
static int wmalloc(s2v_vae* v, T** p, int64_t bytes) {  // weight carve-out (256-byte aligned, zero-initialised with the arena)
    const int64_t sz = rup64(bytes > 0 ? bytes : 16, 256);
    *p = v->sizing ? nullptr : (T*)(v->arena + v->arena_off);
    v->arena_off += sz;
    return 0;
}
template <typename T>
static int dmalloc(s2v_vae* v, T** p, int64_t bytes, bool geo = false) {
    void* q = nullptr;
    // fault injection for tests/test_gpu_vae.py: S2V_VAE_FAULT_GEO_ALLOC=n makes the n-th workspace ("geo") allocation of a
    // prepare_tile_capacity call fail as an out-of-memory hipMalloc would; read per call (a getenv on an allocation path, never on a launch path)
    if (geo && v->fault_geo_at > 0 && ++v->fault_geo_seen == v->fault_geo_at)
        return s2v_fail(__FILE__, __LINE__, "dmalloc: injected allocation failure (S2V_VAE_FAULT_GEO_ALLOC)", -2);
    S2V_CHECK_HIP(hipMalloc(&q, (size_t)(bytes > 0 ? bytes : 16)));
    S2V_CHECK_HIP(hipMemset(q, 0, (size_t)(bytes > 0 ? bytes : 16)));
    (geo ? v->geo_allocs : v->allocs).push_back(q);
    *p = (T*)q;
    return 0;
}

static int make_conv(s2v_vae* v, ConvL& c, const std::string& name, int cin, int cout, int kt, int level) {
    c.cin = cin; c.cout = cout; c.kt = kt; c.level = level;
    const int taps = kt == 3 ? 27 : (kt == 1 ? 9 : 1);
    S2V_TRY(wmalloc(v, &c.w, rup64(cout, 256) * taps * cin * v->esz));
    S2V_TRY(wmalloc(v, &c.b, (int64_t)cout * v->esz));
    v->slots[name + ".weight"] = VSlot{1, c.w, cout, cin, taps, false};
    v->slots[name + ".bias"] = VSlot{0, c.b, cout, 1, 1, false};
    return 0;
}
static int make_snorm(s2v_vae* v, SNormL& n, const std::string& name, int C) {
    n.C = C;
    const int Cz = v->Cz;
    S2V_TRY(wmalloc(v, &n.gn_w, (int64_t)C * v->esz));
    S2V_TRY(wmalloc(v, &n.gn_b, (int64_t)C * v->esz));
    S2V_TRY(wmalloc(v, &n.wy, (int64_t)C * Cz * 4));
    S2V_TRY(wmalloc(v, &n.wb, (int64_t)C * Cz * 4));
    S2V_TRY(wmalloc(v, &n.by, (int64_t)C * 4));
    S2V_TRY(wmalloc(v, &n.bb, (int64_t)C * 4));
    v->slots[name + ".norm_layer.weight"] = VSlot{0, n.gn_w, C, 1, 1, false};
    v->slots[name + ".norm_layer.bias"] = VSlot{0, n.gn_b, C, 1, 1, false};
    v->slots[name + ".conv_y.conv.weight"] = VSlot{2, n.wy, C, Cz, 1, false};
    v->slots[name + ".conv_b.conv.weight"] = VSlot{2, n.wb, C, Cz, 1, false};
    v->slots[name + ".conv_y.conv.bias"] = VSlot{3, n.by, C, 1, 1, false};
    v->slots[name + ".conv_b.conv.bias"] = VSlot{3, n.bb, C, 1, 1, false};
    return 0;
}
static int make_resnet(s2v_vae* v, ResnetL& r, const std::string& name, int cin, int cout, int level) {
    r.cin = cin; r.cout = cout; r.has_sc = cin != cout;
    S2V_TRY(make_snorm(v, r.n1, name + ".norm1", cin));
    S2V_TRY(make_conv(v, r.c1, name + ".conv1.conv", cin, cout, 3, level));
    S2V_TRY(make_snorm(v, r.n2, name + ".norm2", cout));
    S2V_TRY(make_conv(v, r.c2, name + ".conv2.conv", cout, cout, 3, level));
    if (r.has_sc) S2V_TRY(make_conv(v, r.sc, name + ".conv_shortcut", cin, cout, 0, level));
    return 0;
}

extern "C" void s2v_vae_destroy(s2v_vae* v) {
    if (v) {
        for (auto st_ : v->side) (void)hipStreamDestroy(st_);
        for (auto e : v->ev_side) (void)hipEventDestroy(e);
        if (v->ev_fork) (void)hipEventDestroy(v->ev_fork);
    }
    if (!v) return;
    (void)hipDeviceSynchronize();
    for (void* p : v->allocs) (void)hipFree(p);
    for (void* p : v->geo_allocs) (void)hipFree(p);
    for (char* p : v->tiles) (void)hipFree(p);
    if (v->arena) (void)hipFree(v->arena);
    delete v;
}

static int build_two_pass(s2v_vae* v, int (*build)(s2v_vae*)) {
    v->sizing = true;
    v->arena_off = 0;
    S2V_TRY(build(v));
    v->arena_bytes = v->arena_off;
    S2V_CHECK_HIP(hipMalloc((void**)&v->arena, (size_t)v->arena_bytes));
    S2V_CHECK_HIP(hipMemset(v->arena, 0, (size_t)v->arena_bytes));
    v->sizing = false;
    v->arena_off = 0;
    v->stages.clear();
    v->slots.clear();
    S2V_TRY(build(v));
    return dmalloc(v, &v->sums, sizeof(double) * 2 * v->G);
}

static int build_decoder(s2v_vae* v) {
    const s2v_vae_config* cfg = &v->cfg;
    const int nb = cfg->num_blocks;
    // decoder channel plan = reversed block_out_channels (autoencoder_kl_cogvideox.py:871-915)
    std::vector<int> rc(nb);
    for (int i = 0; i < nb; ++i) rc[i] = cfg->block_out_channels[nb - 1 - i];
    const int tlevel = (int)std::lround(std::log2((double)cfg->temporal_compression_ratio));
    int r = make_conv(v, v->conv_in, "decoder.conv_in.conv", v->Cz, rc[0], 3, 0);
    v->stages.resize(nb + 1);
    char nm[128];
    if (!r) {
        v->stages[0].res.resize(2);
        for (int i = 0; i < 2 && !r; ++i) {
            snprintf(nm, sizeof(nm), "decoder.mid_block.resnets.%d", i);
            r = make_resnet(v, v->stages[0].res[i], nm, rc[0], rc[0], 0);
        }
    }
    int prev = rc[0], level = 0;
    for (int b = 0; b < nb && !r; ++b) {
        StageL& s = v->stages[b + 1];
        s.res.resize(cfg->layers_per_block + 1);
        for (int i = 0; i <= cfg->layers_per_block && !r; ++i) {
            snprintf(nm, sizeof(nm), "decoder.up_blocks.%d.resnets.%d", b, i);
            r = make_resnet(v, s.res[i], nm, i == 0 ? prev : rc[b], rc[b], level);
        }
        prev = rc[b];
        s.has_up = b != nb - 1;
        s.compress_time = b < tlevel;
        if (s.has_up && !r) {
            snprintf(nm, sizeof(nm), "decoder.up_blocks.%d.upsamplers.0.conv", b);
            r = make_conv(v, s.up, nm, rc[b], rc[b], 1, level + 1);
            level++;
        }
    }
    if (!r) r = make_snorm(v, v->norm_out, "decoder.norm_out", rc[nb - 1]);
    if (!r) r = make_conv(v, v->conv_out, "decoder.conv_out.conv", rc[nb - 1], cfg->out_channels, 3, level);
    return r;
}

extern "C" int s2v_vae_create(const s2v_vae_config* cfg, s2v_vae** out) {
    S2V_REQUIRE(cfg && out, "s2v_vae_create: null argument");
    S2V_REQUIRE(cfg->dtype == S2V_DTYPE_F32 || cfg->dtype == S2V_DTYPE_BF16 || cfg->dtype == S2V_DTYPE_F16, "s2v_vae_create: unsupported dtype");
    S2V_REQUIRE(cfg->num_blocks >= 1 && cfg->num_blocks <= 6, "s2v_vae_create: 1..6 blocks");
    s2v_vae* v = new s2v_vae();
    v->cfg = *cfg;
    v->dtype = cfg->dtype;
    v->esz = cfg->dtype == S2V_DTYPE_F32 ? 4 : 2;
    v->G = cfg->norm_num_groups;
    v->Cz = cfg->latent_channels;
    v->mfma = cfg->dtype == S2V_DTYPE_BF16 && !cfg->force_simple;
    v->h16 = cfg->dtype == S2V_DTYPE_F16 && !cfg->force_simple;  // fp16: convolutions / shortcuts with cin % 64 == 0 on v_mfma_f32_32x32x16_f16 (gemm_f16)
    const int r = build_two_pass(v, build_decoder);
    if (r) { s2v_vae_destroy(v); return r; }
    *out = v;
    return 0;
}

/* One device range holding every weight of the handle (decoder or encoder), for the replica broadcast; a replica that
 * received it calls s2v_vae_mark_weights_loaded instead of s2v_vae_load_weight + s2v_vae_finalize. */
extern "C" int s2v_vae_weight_arena(s2v_vae* v, void** dev_ptr, int64_t* bytes) {
    S2V_REQUIRE(v && dev_ptr && bytes, "s2v_vae_weight_arena: null argument");
    *dev_ptr = v->arena;
    *bytes = v->arena_bytes;
    return 0;
}
extern "C" int s2v_vae_mark_weights_loaded(s2v_vae* v) {
    S2V_REQUIRE(v, "s2v_vae_mark_weights_loaded: null argument");
    for (auto& kv : v->slots) kv.second.loaded = true;
    v->finalized = true;
    return 0;
}

template <typename TS>
__global__ void to_f32_transposed_k(const TS* src, int rows, int cols, float* dst) {  // dst[c][r] = src[r][c]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int r = i / cols, c = i - r * cols;
    dst[(size_t)c * rows + r] = ET<TS>::ld(src + i);
}

extern "C" int s2v_vae_load_weight(s2v_vae* v, const char* name, const void* dev_ptr, const int64_t* shape, int32_t ndim,
                                   int32_t src_dtype, s2v_stream stream) {
    S2V_REQUIRE(v && name && dev_ptr && shape, "s2v_vae_load_weight: null argument");
    S2V_REQUIRE(src_dtype == S2V_DTYPE_F32 || src_dtype == S2V_DTYPE_BF16 || src_dtype == S2V_DTYPE_F16, "s2v_vae_load_weight: unsupported dtype");
    auto it = v->slots.find(name);
    if (it == v->slots.end()) {
        std::string m = std::string("s2v_vae_load_weight: unknown tensor name: ") + name;
        return s2v_fail(__FILE__, __LINE__, m.c_str(), -3);
    }
    VSlot& s = it->second;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    if (n != s.d0 * s.d1 * s.taps || shape[0] != s.d0) {
        std::string m = std::string("s2v_vae_load_weight: shape mismatch for ") + name;
        return s2v_fail(__FILE__, __LINE__, m.c_str(), -3);
    }
    hipStream_t st = (hipStream_t)stream;
    if (s.kind == 0) S2V_TRY(launch_convert(dev_ptr, src_dtype, s.dst, v->dtype, n, st));
    else if (s.kind == 3) S2V_TRY(launch_convert(dev_ptr, src_dtype, s.dst, S2V_F32, n, st));
    else if (s.kind == 1) S2V_TRY(launch_conv_w_repack(dev_ptr, src_dtype, (int)s.d0, (int)s.d1, (int)s.taps, s.dst, v->dtype, st));
    else {
        // the reference holds conv_y / conv_b in the model dtype: round first (through a scratch of that dtype unless the source already is
        // of it or the model is fp32), then widen + transpose
        const int tot = (int)n;
        const void* srcp = dev_ptr;
        int sdt = src_dtype;
        void* tmp = nullptr;
        if (v->dtype != S2V_DTYPE_F32 && src_dtype != v->dtype) {
            S2V_CHECK_HIP(hipMalloc(&tmp, 2 * (size_t)n));
            const int r = launch_convert(dev_ptr, src_dtype, tmp, v->dtype, n, st);
            if (r) { (void)hipFree(tmp); return r; }
            srcp = tmp; sdt = v->dtype;
        }
        S2V_DT_DISPATCH(sdt, hipLaunchKernelGGL(to_f32_transposed_k<T>, dim3((tot + 255) / 256), dim3(256), 0, st, (const T*)srcp, (int)s.d0, (int)s.d1, (float*)s.dst))
        if (tmp) { (void)hipStreamSynchronize(st); (void)hipFree(tmp); }
        S2V_CHECK_HIP(hipGetLastError());
    }
    if (s.kind == 3 && v->dtype != S2V_DTYPE_F32 && src_dtype != v->dtype) {
        // bias held in the model dtype by the reference: round through it
        void* tmp = nullptr;
        S2V_CHECK_HIP(hipMalloc(&tmp, 2 * (size_t)n));
        int r = launch_convert(dev_ptr, src_dtype, tmp, v->dtype, n, st);
        if (!r) r = launch_convert(tmp, v->dtype, s.dst, S2V_F32, n, st);
        (void)hipStreamSynchronize(st);
        (void)hipFree(tmp);
        if (r) return r;
    }
    s.loaded = true;
    return 0;
}

extern "C" int s2v_vae_finalize(s2v_vae* v) {
    S2V_REQUIRE(v, "null vae");
    for (auto& kv : v->slots)
        if (!kv.second.loaded) {
            std::string m = std::string("s2v_vae_finalize: tensor was never loaded: ") + kv.first;
            return s2v_fail(__FILE__, __LINE__, m.c_str(), -3);
        }
    v->finalized = true;
    return 0;
}

// ---- geometry ------------------------------------------------------------------------------------------------
static int up_frames(int F, int ct) { return (!ct || F == 1) ? F : ((F & 1) ? 1 + 2 * (F - 1) : 2 * F); }

static void for_each_conv(s2v_vae* v, const std::function<void(ConvL&)>& fn) {
    fn(v->conv_in);
    for (auto& s : v->stages) {
        for (auto& r : s.res) { fn(r.c1); fn(r.c2); }
        if (s.has_up) fn(s.up);
    }
    fn(v->conv_out);
}

// the live members <-> workspace set k
static void ws_save(s2v_vae* v, int k) {
    s2v_vae::WS& w = v->ws[k];
    w.pads.clear();
    for_each_conv(v, [&](ConvL& c) { w.pads.push_back(c.pad); });
    for (int i = 0; i < 3; ++i) w.dense[i] = v->dense[i];
    w.zq = v->zq; w.yt = v->yt; w.bt = v->bt; w.sums = v->sums; w.gn_part = v->gn_part;
    w.cur_h = v->cur_h; w.cur_w = v->cur_w;
}
// set k -> the live members, unconditionally
static void ws_load(s2v_vae* v, int k) {
    const s2v_vae::WS& w = v->ws[k];
    size_t i = 0;
    for_each_conv(v, [&](ConvL& c) { c.pad = w.pads[i++]; });
    for (int j = 0; j < 3; ++j) v->dense[j] = w.dense[j];
    v->zq = w.zq; v->yt = w.yt; v->bt = w.bt; v->sums = w.sums; v->gn_part = w.gn_part;
    v->cur_h = w.cur_h; v->cur_w = w.cur_w;
    v->ws_active = k;
}
static void ws_activate(s2v_vae* v, int k) {
    if (v->ws.empty() || k == v->ws_active) return;
    v->ws[v->ws_active].cur_h = v->cur_h; v->ws[v->ws_active].cur_w = v->cur_w;
    ws_load(v, k);
}

// one workspace set (every operand / cache / scratch buffer of a decode at window th x tw) into the live members
static int alloc_workspace_set(s2v_vae* v, int th, int tw, int fz_max) {
    int rc = 0;
    int64_t dmax = 0;
    for_each_conv(v, [&](ConvL& c) {
        const int64_t H = (int64_t)th << c.level, W = (int64_t)tw << c.level;
        const int F = v->fmax[c.level] + (c.kt == 3 ? 2 : 0);
        c.pad_bytes = (int64_t)F * (H + 2) * (W + 2) * c.cin * v->esz + 1024;
        if (!rc) rc = dmalloc(v, &c.pad, c.pad_bytes, true);
        const int64_t d = (int64_t)v->fmax[c.level] * H * W * (c.cin > c.cout ? c.cin : c.cout) * v->esz;
        dmax = d > dmax ? d : dmax;
    });
    if (rc) return rc;
    v->dense_bytes = dmax + (int64_t)256 * 1024 * v->esz;  // + one 128-row MFMA tile of slack
    for (int i = 0; i < 3; ++i) S2V_TRY(dmalloc(v, &v->dense[i], v->dense_bytes, true));
    S2V_TRY(dmalloc(v, &v->zq, (int64_t)fz_max * th * tw * v->Cz * v->esz + 64, true));
    int cmax = 0;
    for_each_conv(v, [&](ConvL& c) { cmax = c.cin > cmax ? c.cin : cmax; });
    S2V_TRY(dmalloc(v, &v->yt, (int64_t)fz_max * th * tw * cmax * v->esz + 64, true));
    S2V_TRY(dmalloc(v, &v->bt, (int64_t)fz_max * th * tw * cmax * v->esz + 64, true));
    int lv = 0;
    for (size_t s = 1; s < v->stages.size(); ++s) if (v->stages[s].has_up) lv++;
    const int64_t pmax = (int64_t)v->fmax[lv] * ((int64_t)th << lv) * ((int64_t)tw << lv);
    S2V_TRY(dmalloc(v, &v->gn_part, gn_stats_scratch_bytes(pmax, v->G) + 64, true));
    S2V_TRY(dmalloc(v, &v->sums, sizeof(double) * 2 * v->G, true));
    return 0;
}

// Workspace for windows up to th x tw latents, fz_max latent frames per batch, `nws` sets (tiles in flight).
// The capacity fields (th / tw / fmax / ws_req) and v->ws are committed only AFTER the sets exist: a failed allocation leaves the
// context at "no capacity" (the next decode allocates again instead of running on empty sets -- ADVICE r3), a failed set k > 0 is
// freed and the decode proceeds with k sets.  The number of sets is bounded by 70 % of the free memory at this moment (other ranks
// on the device and torch's caching allocator may move that figure: hence the retry-with-fewer path) and by a byte cap (a quarter of the
// device's memory unless S2V_VAE_WORKSPACE_MAX_GB says otherwise).
static int prepare_tile_capacity(s2v_vae* v, int th, int tw, int fz_max, int nws = 1) {
    // Sets sized for a LARGER window than this request leave fewer tiles in flight than the byte cap allows: an untiled decode of 90 x 160 latents
    // (one 57.7 GB set) followed by the tiled decode of the same latents (30 x 45-latent tiles: 5.5 GB per set, six fit) used to keep the one big
    // set and run the twenty tiles one after the other -- 1.9 s instead of 1.4 (bench.py times untiled, then tiled; at 480 x 720
    // three 21.7 GB sets instead of six of 5.5 GB).  A request for several sets that the present capacity cannot serve is rebuilt at the REQUESTED
    // size; the next untiled decode grows it again (a decoder that alternates the two modes re-allocates each time: the pipeline only tiles).
    const bool oversized = !v->ws.empty() && (v->th > th || v->tw > tw) && nws > 1 && (int)v->ws.size() < nws;
    if (!oversized && v->th >= th && v->tw >= tw && v->fmax[0] >= fz_max && v->ws_req >= nws && !v->ws.empty()) return 0;
    S2V_CHECK_HIP(hipDeviceSynchronize());
    for (void* p : v->geo_allocs) (void)hipFree(p);
    v->geo_allocs.clear();
    if (!oversized) {  // growing: never below what an earlier decode needed
        th = th > v->th ? th : v->th;
        tw = tw > v->tw ? tw : v->tw;
        nws = std::max(nws, v->ws_req);
    }
    fz_max = fz_max > v->fmax[0] ? fz_max : v->fmax[0];
    const int nws_req = nws;
    // from here until the commit below the context has NO capacity
    v->th = v->tw = 0; v->ws_req = 0; v->ws.clear(); v->ws_active = 0; v->cur_h = v->cur_w = 0;
    for (int i = 0; i < 8; ++i) v->fmax[i] = 0;
    // frames per level
    int f = fz_max, lvl = 0;
    v->fmax[0] = f;
    for (size_t s = 1; s < v->stages.size(); ++s)
        if (v->stages[s].has_up) { f = up_frames(f, v->stages[s].compress_time); v->fmax[++lvl] = f; }
    {  // as many sets as 70 % of the free memory (and the optional byte cap) hold, never fewer than one
        int64_t set_bytes = 0, dmax = 0;
        for_each_conv(v, [&](ConvL& c) {
            const int64_t H = (int64_t)th << c.level, W = (int64_t)tw << c.level;
            set_bytes += (int64_t)(v->fmax[c.level] + (c.kt == 3 ? 2 : 0)) * (H + 2) * (W + 2) * c.cin * v->esz;
            dmax = std::max(dmax, (int64_t)v->fmax[c.level] * H * W * std::max(c.cin, c.cout) * v->esz);
        });
        set_bytes += 3 * dmax;
        v->set_bytes = set_bytes;
        size_t free_b = 0, total_b = 0;
        double budget = -1.0;
        // default byte cap: a quarter of the device's memory (MI355X: 72 GB = three ~21.7 GB sets at the real widths, 602 ms per 49 x 480 x 720
        // decode against 554 with six sets and 130 GB -- profiles/r03_vae_tiles_in_flight.txt; VERDICT r4: 130 GB of workspaces to save 0.15 s
        // is not a default).  S2V_VAE_WORKSPACE_MAX_GB = n caps at n GB instead, <= 0 lifts the byte cap (70 % of the free memory remains).
        double cap = -1.0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) { budget = 0.7 * (double)free_b; cap = 0.25 * (double)total_b; }
        if (const char* e = getenv("S2V_VAE_WORKSPACE_MAX_GB")) cap = atof(e) * 1e9;
        if (cap > 0 && (budget < 0 || cap < budget)) budget = cap;
        if (budget >= 0 && set_bytes > 0)
            nws = (int)std::max<int64_t>(1, std::min<int64_t>(nws, (int64_t)budget / set_bytes));
    }
    std::vector<s2v_vae::WS> sets;
    int rc = 0;
    v->fault_geo_seen = 0;
    {
        const char* e = getenv("S2V_VAE_FAULT_GEO_ALLOC");
        v->fault_geo_at = e ? atoi(e) : 0;
    }
    for (int k = 0; k < nws; ++k) {
        const size_t mark = v->geo_allocs.size();
        rc = alloc_workspace_set(v, th, tw, fz_max);
        if (rc) {  // give the partial set back; with k sets built the decode runs k tiles in flight
            (void)hipGetLastError();
            for (size_t i = mark; i < v->geo_allocs.size(); ++i) (void)hipFree(v->geo_allocs[i]);
            v->geo_allocs.resize(mark);
            break;
        }
        v->ws.push_back(s2v_vae::WS());
        v->cur_h = v->cur_w = 0;
        ws_save(v, (int)v->ws.size() - 1);
    }
    if (v->ws.empty()) {  // not even one set: stay at "no capacity" and report the allocation failure
        for (int i = 0; i < 8; ++i) v->fmax[i] = 0;
        for_each_conv(v, [&](ConvL& c) { c.pad = nullptr; });
        return rc ? rc : vfail("prepare_tile_capacity: no workspace set could be allocated");
    }
    nws = (int)v->ws.size();
    // The live members hold whatever alloc_workspace_set wrote LAST -- the last set built, or the freed pointers of a set that failed part-way
    // (with one surviving set ws_activate(0) returned early and the next decode wrote through them).  Reload set 0 unconditionally.
    ws_load(v, 0);
    while ((int)v->side.size() < nws - 1) {
        hipStream_t sst = nullptr; hipEvent_t e = nullptr;
        S2V_CHECK_HIP(hipStreamCreateWithFlags(&sst, hipStreamNonBlocking));
        S2V_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        v->side.push_back(sst); v->ev_side.push_back(e);
    }
    if (!v->ev_fork) S2V_CHECK_HIP(hipEventCreateWithFlags(&v->ev_fork, hipEventDisableTiming));
    // commit: the capacity this context now really has (ws_req keeps the REQUEST so that a memory-bounded result is not retried on
    // every decode; a later, larger request allocates again)
    v->th = th; v->tw = tw; v->ws_req = nws_req;
    return 0;
}

// A new window size: the zero ring of every padded operand moves.  Decoder (ring_only): only the ring of the new size is cleared
// (launch_zero_border: interiors are rewritten by their producers before they are read; the buffers were zeroed when they were
// allocated) -- with nine tiles of four sizes dealt to six workspace sets the whole-buffer memsets (20 GB per set) were 35-40 ms of a
// 555 ms tiled decode.  The encoder's one-frame buffers are small and keep the plain memset.
static int set_layout(s2v_vae* v, int h, int w, hipStream_t st, bool ring_only = false) {
    if (v->cur_h == h && v->cur_w == w) return 0;
    int rc = 0;
    for_each_conv(v, [&](ConvL& c) {
        if (rc) return;
        if (ring_only) rc = launch_zero_border(c.pad, v->fmax[c.level] + (c.kt == 3 ? 2 : 0), h << c.level, w << c.level, c.cin, v->esz, st);
        else if (hipMemsetAsync(c.pad, 0, (size_t)c.pad_bytes, st) != hipSuccess) rc = -2;
    });
    if (rc) return vfail("clearing the operand borders failed");
    v->cur_h = h; v->cur_w = w;
    return 0;
}

// ---- launch sequences --------------------------------------------------------------------------------------------
#ifdef S2V_DIAG
// tools/vae_conv_rates.py: EVERY GEMM-shaped launch of the decode in launch order -- convolutions AND the 1 x 1 shortcut GEMMs between them --
// so that the log zips one-to-one with the GEMM kernels of a kernel trace (round 4 logged the convolutions only and matched the rest by grid
// size: a shortcut GEMM with a convolution's grid then stood in for it, and rows above the 2.5 PF peak appeared)
static void vae_gemm_log(const GemmArgs& g, int epi, bool mfma, const char* kind) {
    if (const char* lg = getenv("S2V_VAE_CONV_LOG"))
        if (FILE* f = fopen(lg, "a")) { fprintf(f, "%d %d %d %d %d %s\n", g.M, g.N, g.K, epi, (int)mfma, kind); fclose(f); }
}
#endif
// direct_dst (conv_out only): the [C][Ftot][H][W] tile output -- when the direct small-N kernel qualifies (launch_conv_out_direct) it writes there
// and *did_direct is set; otherwise the implicit GEMM writes `out` and the caller converts the layout
static int run_conv(s2v_vae* v, ConvL& c, int F, int H, int W, bool first, int epi, const void* resid, void* out,
                    hipStream_t st, char* direct_dst = nullptr, int Ftot = 0, int f0 = 0, bool* did_direct = nullptr) {
    const int64_t fb = (int64_t)(H + 2) * (W + 2) * c.cin * v->esz;
    if (c.kt == 3 && first) {
        S2V_CHECK_HIP(hipMemcpyAsync(c.pad, c.pad + 2 * fb, fb, hipMemcpyDeviceToDevice, st));
        S2V_CHECK_HIP(hipMemcpyAsync(c.pad + fb, c.pad + 2 * fb, fb, hipMemcpyDeviceToDevice, st));
    }
    const int taps = c.kt == 3 ? 27 : 9;
    bool direct = false;
    bool want_direct = direct_dst && c.kt == 3 && epi == EPI_BIAS && !v->cfg.force_simple && (v->mfma || v->h16);
#ifdef S2V_DIAG
    if (getenv("S2V_VAE_NO_DIRECT_CONV_OUT")) want_direct = false;  // same-box A/B against the implicit-GEMM conv_out (diagnostics build only)
#endif
    if (want_direct) {
        const int rc = launch_conv_out_direct(c.pad, c.w, taps * c.cin, c.b, F, H, W, c.cin, c.cout, direct_dst, Ftot, f0, v->dtype, st);
        if (rc < 0) return rc;
        direct = rc == 1;
    }
    if (did_direct) *did_direct = direct;
    GemmArgs g{};
    g.A = c.pad; g.W = c.w; g.ldw = taps * c.cin; g.bias = c.b; g.C = out; g.ldc = c.cout;
    g.M = F * H * W; g.N = c.cout; g.K = taps * c.cin;
    g.R = resid; g.ldr = c.cout;
    g.conv = 1; g.cin = c.cin; g.Hp = H + 2; g.Wp = W + 2; g.oH = H; g.oW = W; g.kt = c.kt;
    g.w_rows_padded = (int)rup64(c.cout, 256);
#ifdef S2V_DIAG
    if (const char* e = getenv("S2V_VAE_ADD_EXPERIMENT")) {  // timing experiments on the residual-add epilogue (results are wrong): 1 = no add, 2 = residual from another buffer
        if (epi == EPI_BIAS_ADD && atoi(e) == 1) { epi = EPI_BIAS; g.R = nullptr; }
        if (epi == EPI_BIAS_ADD && atoi(e) == 2) g.R = (const char*)c.pad;  // some other resident buffer of at least M x cout elements
    }
    if (!direct) vae_gemm_log(g, epi, v->mfma && c.cin % 64 == 0, "conv");
#endif
    if (direct) {}  // conv_out ran as the direct kernel
    else if (v->mfma && c.cin % 64 == 0) S2V_TRY(launch_gemm_bf16(g, epi, st));  // cout = 3 (conv_out) without the direct kernel: one padded 128-column tile
    else if (v->h16 && gemm_f16_ok(g, epi)) S2V_TRY(launch_gemm_f16(g, epi, st));
    else { g.valu_only = v->cfg.force_simple; S2V_TRY(launch_gemm_simple(g, epi, v->dtype, st)); }
    if (c.kt == 3) {  // conv cache: the last two frames of the operand become frames 0, 1 of the next batch (adjacent on both sides: one copy unless they overlap)
        if (F >= 2) S2V_CHECK_HIP(hipMemcpyAsync(c.pad, c.pad + (int64_t)F * fb, 2 * fb, hipMemcpyDeviceToDevice, st));
        else {
            S2V_CHECK_HIP(hipMemcpyAsync(c.pad, c.pad + (int64_t)F * fb, fb, hipMemcpyDeviceToDevice, st));
            S2V_CHECK_HIP(hipMemcpyAsync(c.pad + fb, c.pad + (int64_t)(F + 1) * fb, fb, hipMemcpyDeviceToDevice, st));
        }
    }
    return 0;
}

static int run_snorm(s2v_vae* v, const SNormL& n, const void* x, int F, int H, int W, int Fz, int hz, int wz, void* out_pad,
                     int f_off, hipStream_t st) {
    S2V_TRY(launch_gn_stats(x, (int64_t)F * H * W, n.C, v->G, v->sums, v->gn_part, v->dtype, st));
    SNormArgs a{};
    a.x = x; a.F = F; a.H = H; a.W = W; a.C = n.C; a.G = v->G; a.sums = v->sums; a.eps = v->cfg.norm_eps;
    a.gn_w = n.gn_w; a.gn_b = n.gn_b; a.wy = n.wy; a.by = n.by; a.wb = n.wb; a.bb = n.bb;
    a.zq = v->zq; a.Fz = Fz; a.hz = hz; a.wz = wz; a.Cz = v->Cz; a.out = out_pad; a.f_off = f_off; a.silu = 1; a.yt = v->yt; a.bt = v->bt;
    return launch_snorm_apply(a, v->dtype, st);
}

// one decoder pass over a frame batch of `Fz` latent frames of an (h x w) latent window; writes its output frames
static int decode_batch(s2v_vae* v, int Fz, int h, int w, bool first, char* dst, int Ftot, int f0, int* frames_out,
                        hipStream_t st) {
    int cur = 0, t1 = 1, t2 = 2;
    int F = Fz, H = h, W = w;
    S2V_TRY(launch_dense_to_padded(v->zq, Fz, h, w, v->Cz, v->conv_in.pad, 2, v->dtype, st));
    S2V_TRY(run_conv(v, v->conv_in, F, H, W, first, EPI_BIAS, nullptr, v->dense[cur], st));
    for (auto& s : v->stages) {
        for (auto& r : s.res) {
            S2V_TRY(run_snorm(v, r.n1, v->dense[cur], F, H, W, Fz, h, w, r.c1.pad, 2, st));
            S2V_TRY(run_conv(v, r.c1, F, H, W, first, EPI_BIAS, nullptr, v->dense[t1], st));
            S2V_TRY(run_snorm(v, r.n2, v->dense[t1], F, H, W, Fz, h, w, r.c2.pad, 2, st));
            if (r.has_sc) {
                GemmArgs g{};
                g.A = v->dense[cur]; g.lda = r.cin; g.W = r.sc.w; g.ldw = r.cin; g.bias = r.sc.b;
                g.C = v->dense[t2]; g.ldc = r.cout; g.M = F * H * W; g.N = r.cout; g.K = r.cin;
                g.a_rows_padded = (int)rup64(g.M, 256);  // dense buffers carry a 256-row slack
#ifdef S2V_DIAG
                vae_gemm_log(g, EPI_BIAS, v->mfma && r.cin % 64 == 0, "shortcut");
#endif
                if (v->mfma && r.cin % 64 == 0) S2V_TRY(launch_gemm_bf16(g, EPI_BIAS, st));
                else if (v->h16 && gemm_f16_ok(g, EPI_BIAS)) S2V_TRY(launch_gemm_f16(g, EPI_BIAS, st));
                else { g.valu_only = v->cfg.force_simple; S2V_TRY(launch_gemm_simple(g, EPI_BIAS, v->dtype, st)); }
                S2V_TRY(run_conv(v, r.c2, F, H, W, first, EPI_BIAS_ADD, v->dense[t2], v->dense[t2], st));
                std::swap(cur, t2);
            } else {
                S2V_TRY(run_conv(v, r.c2, F, H, W, first, EPI_BIAS_ADD, v->dense[cur], v->dense[cur], st));
            }
        }
        if (s.has_up) {
            S2V_TRY(launch_upsample(v->dense[cur], F, H, W, s.up.cin, s.compress_time, s.up.pad, v->dtype, st));
            F = up_frames(F, s.compress_time); H *= 2; W *= 2;
            S2V_TRY(run_conv(v, s.up, F, H, W, first, EPI_BIAS, nullptr, v->dense[t1], st));
            std::swap(cur, t1);
        }
    }
    S2V_TRY(run_snorm(v, v->norm_out, v->dense[cur], F, H, W, Fz, h, w, v->conv_out.pad, 2, st));
    bool direct = false;
    S2V_TRY(run_conv(v, v->conv_out, F, H, W, first, EPI_BIAS, nullptr, v->dense[t1], st, dst, Ftot, f0, &direct));
    if (!direct) S2V_TRY(launch_to_ncfhw(v->dense[t1], F, H, W, v->cfg.out_channels, dst, Ftot, f0, v->dtype, st));
    *frames_out = F;
    return 0;
}

struct Batch { int s, e; };
static std::vector<Batch> frame_batches(int F) {  // autoencoder_kl_cogvideox.py:1237-1245 (num_latent_frames_batch_size = 2)
    const int fbs = 2;
    const int nb = F / fbs > 1 ? F / fbs : 1, rem = F % fbs;
    std::vector<Batch> b;
    for (int i = 0; i < nb; ++i) b.push_back({fbs * i + (i == 0 ? 0 : rem), std::min(fbs * (i + 1) + rem, F)});
    return b;
}
static int out_frames_of(s2v_vae* v, int F) {
    int tot = 0;
    for (auto b : frame_batches(F)) {
        int f = b.e - b.s;
        for (size_t s = 1; s < v->stages.size(); ++s)
            if (v->stages[s].has_up) f = up_frames(f, v->stages[s].compress_time);
        tot += f;
    }
    return tot;
}
static int spatial_scale(s2v_vae* v) { return 1 << (v->cfg.num_blocks - 1); }

struct TileGeo { int tl_h, tl_w, ov_h, ov_w, bl_h, bl_w, lim_h, lim_w; };
static TileGeo tile_geo(s2v_vae* v) {  // :1102-1114, 1400-1406 incl. the int() truncations
    TileGeo t;
    const int ts_h = v->cfg.sample_height / 2, ts_w = v->cfg.sample_width / 2, sc = spatial_scale(v);
    t.tl_h = (int)((double)ts_h / sc); t.tl_w = (int)((double)ts_w / sc);
    t.ov_h = (int)(t.tl_h * (1.0 - 1.0 / 6.0)); t.ov_w = (int)(t.tl_w * (1.0 - 1.0 / 5.0));
    t.bl_h = (int)(ts_h * (1.0 / 6.0)); t.bl_w = (int)(ts_w * (1.0 / 5.0));
    t.lim_h = ts_h - t.bl_h; t.lim_w = ts_w - t.bl_w;
    return t;
}
static bool use_tiles(s2v_vae* v, int h, int w, int tiling) {
    TileGeo t = tile_geo(v);
    return tiling && (w > t.tl_w || h > t.tl_h);
}

extern "C" int s2v_vae_workspace_info(s2v_vae* v, int32_t* sets, int64_t* bytes_per_set) {
    S2V_REQUIRE(v && sets && bytes_per_set, "s2v_vae_workspace_info: null argument");
    *sets = (int32_t)v->ws.size();
    *bytes_per_set = v->set_bytes;
    return 0;
}

extern "C" int s2v_vae_out_shape(s2v_vae* v, int32_t F, int32_t h, int32_t w, int32_t tiling, int32_t* Fo, int32_t* Ho,
                                 int32_t* Wo) {
    S2V_REQUIRE(v && Fo && Ho && Wo && F >= 1 && h >= 1 && w >= 1, "s2v_vae_out_shape: bad argument");
    *Fo = out_frames_of(v, F);
    const int sc = spatial_scale(v);
    if (!use_tiles(v, h, w, tiling)) { *Ho = h * sc; *Wo = w * sc; return 0; }
    TileGeo t = tile_geo(v);
    S2V_REQUIRE(t.ov_h > 0 && t.ov_w > 0, "s2v_vae_out_shape: degenerate tile overlap");
    int H = 0, W = 0;
    for (int i = 0; i < h; i += t.ov_h) H += std::min(std::min(t.tl_h, h - i) * sc, t.lim_h);
    for (int j = 0; j < w; j += t.ov_w) W += std::min(std::min(t.tl_w, w - j) * sc, t.lim_w);
    *Ho = H; *Wo = W;
    return 0;
}

static int decode_window(s2v_vae* v, const char* lat, int F, int h, int w, int y0, int x0, int th, int tw, char* dst,
                         int Ftot, int scaled, hipStream_t st) {
    S2V_TRY(set_layout(v, th, tw, st, true));
    const float inv_sf = scaled ? (float)(1.0 / (double)v->cfg.scaling_factor) : 1.0f;
    int f0 = 0;
    bool first = true;
    for (auto b : frame_batches(F)) {
        const int Fz = b.e - b.s;
        S2V_TRY(launch_latent_to_zq(lat + (int64_t)b.s * v->Cz * h * w * v->esz, Fz, v->Cz, h, w, inv_sf, v->zq, y0, x0, th,
                                    tw, v->dtype, st));
        int fo = 0;
        S2V_TRY(decode_batch(v, Fz, th, tw, first, dst, Ftot, f0, &fo, st));
        f0 += fo;
        first = false;
    }
    return 0;
}

extern "C" int s2v_vae_decode(s2v_vae* v, const void* latents, int32_t F, int32_t h, int32_t w, int32_t tiling,
                              int32_t scaled, void* out, s2v_stream stream) {
    S2V_REQUIRE(v && latents && out, "s2v_vae_decode: null argument");
    S2V_REQUIRE(!v->encoder, "s2v_vae_decode: this handle holds the encoder");
    S2V_REQUIRE(v->finalized, "s2v_vae_decode: weights not finalized");
    S2V_REQUIRE(F >= 1 && h >= 1 && w >= 1, "s2v_vae_decode: bad geometry");
    hipStream_t st = (hipStream_t)stream;
    const int sc = spatial_scale(v);
    const int Ftot = out_frames_of(v, F);
    int fz_max = 0;
    for (auto b : frame_batches(F)) fz_max = std::max(fz_max, b.e - b.s);
    if (!use_tiles(v, h, w, tiling)) {
        S2V_TRY(prepare_tile_capacity(v, h, w, fz_max));
        ws_activate(v, 0);
        return decode_window(v, (const char*)latents, F, h, w, 0, 0, h, w, (char*)out, Ftot, scaled, st);
    }
    TileGeo t = tile_geo(v);
    S2V_REQUIRE(t.ov_h > 0 && t.ov_w > 0, "s2v_vae_decode: degenerate tile overlap");
    std::vector<int> is, js;
    for (int i = 0; i < h; i += t.ov_h) is.push_back(i);
    for (int j = 0; j < w; j += t.ov_w) js.push_back(j);
    const size_t nt = is.size() * js.size();
    // tiles in flight: 49 x 480 x 720 at the real widths measured 928 / 684 / 602 / 605 / 554 / 590 ms with 1 / 2 / 3 / 4 / 6 / 9 sets
    // (profiles/r03_vae_tiles_in_flight.txt; round 6: 505-510 ms with six); up to six, as many as the byte cap of prepare_tile_capacity holds (a set
    // sized for a 30 x 45-latent tile is 5.5 GB: all six)
    int nws_cap = 6;
    if (const char* e = getenv("S2V_VAE_TILES_IN_FLIGHT")) nws_cap = std::max(1, atoi(e));
    int nws = (int)std::min<size_t>(nt, (size_t)nws_cap);
    S2V_TRY(prepare_tile_capacity(v, std::min(t.tl_h, h), std::min(t.tl_w, w), fz_max, nws));
    nws = std::min(nws, (int)v->ws.size());
    const int C = v->cfg.out_channels;
    // per-tile outputs [C][Ftot][8 th][8 tw] (re-allocated only when the tiling changes)
    bool realloc_tiles = v->tiles.size() != nt || v->tiles_F != Ftot;
    for (size_t k = 0; !realloc_tiles && k < nt; ++k) {
        const int th = std::min(t.tl_h, h - is[k / js.size()]), tw = std::min(t.tl_w, w - js[k % js.size()]);
        if (v->tile_h[k] != th * sc || v->tile_w[k] != tw * sc) realloc_tiles = true;
    }
    if (realloc_tiles) {
        S2V_CHECK_HIP(hipDeviceSynchronize());
        for (char* p : v->tiles) (void)hipFree(p);
        v->tiles.assign(nt, nullptr); v->tile_h.assign(nt, 0); v->tile_w.assign(nt, 0);
        for (size_t k = 0; k < nt; ++k) {
            const int th = std::min(t.tl_h, h - is[k / js.size()]), tw = std::min(t.tl_w, w - js[k % js.size()]);
            v->tile_h[k] = th * sc; v->tile_w[k] = tw * sc;
            S2V_CHECK_HIP(hipMalloc((void**)&v->tiles[k], (size_t)C * Ftot * th * sc * tw * sc * v->esz));
        }
        v->tiles_F = Ftot;
    }
    // tile k runs on workspace set k mod nws: set 0 on the caller's stream, the others on side streams forked from it and joined
    // before the blends (tiles of one set serialise on its stream: they share its buffers)
    if (nws > 1) S2V_CHECK_HIP(hipEventRecord(v->ev_fork, st));
    for (int q = 1; q < nws; ++q) S2V_CHECK_HIP(hipStreamWaitEvent(v->side[q - 1], v->ev_fork, 0));
    // raster order, round-robin over the sets (largest-tile-first onto the least loaded set measured the same at 3-4 sets and 18 % worse
    // at 6: what overlaps well is a big tile's chip-filling GEMMs with a small tile's latency-bound ones, which raster order pairs up)
    for (size_t k = 0; k < nt; ++k) {
        const int i = is[k / js.size()], j = js[k % js.size()];
        const int q = (int)(k % nws);
        ws_activate(v, q);
        S2V_TRY(decode_window(v, (const char*)latents, F, h, w, i, j, v->tile_h[k] / sc, v->tile_w[k] / sc, v->tiles[k], Ftot, scaled,
                              q == 0 ? st : v->side[q - 1]));
    }
    ws_activate(v, 0);
    for (int q = 1; q < nws; ++q) {
        S2V_CHECK_HIP(hipEventRecord(v->ev_side[q - 1], v->side[q - 1]));
        S2V_CHECK_HIP(hipStreamWaitEvent(st, v->ev_side[q - 1], 0));
    }
    // raster-order in-place blends, then crop + concatenate (:1437-1450)
    int32_t Fo, Ho, Wo;
    S2V_TRY(s2v_vae_out_shape(v, F, h, w, 1, &Fo, &Ho, &Wo));
    int y0 = 0;
    for (size_t r = 0; r < is.size(); ++r) {
        int x0 = 0, ch = 0;
        for (size_t c = 0; c < js.size(); ++c) {
            const size_t k = r * js.size() + c;
            const int Ht = v->tile_h[k], Wt = v->tile_w[k];
            if (r > 0) {
                const size_t ka = (r - 1) * js.size() + c;
                const int E = std::min(std::min(v->tile_h[ka], Ht), t.bl_h);
                S2V_TRY(launch_blend(v->tiles[ka], v->tile_h[ka], v->tile_w[ka], v->tiles[k], Ht, Wt, C * Ftot, E, 1, v->dtype, st));
            }
            if (c > 0) {
                const size_t ka = k - 1;
                const int E = std::min(std::min(v->tile_w[ka], Wt), t.bl_w);
                S2V_TRY(launch_blend(v->tiles[ka], v->tile_h[ka], v->tile_w[ka], v->tiles[k], Ht, Wt, C * Ftot, E, 0, v->dtype, st));
            }
            ch = std::min(Ht, t.lim_h);
            const int cw = std::min(Wt, t.lim_w);
            S2V_TRY(launch_paste(v->tiles[k], Ht, Wt, ch, cw, out, Ho, Wo, y0, x0, C * Ftot, v->dtype, st));
            x0 += cw;
        }
        y0 += ch;
    }
    return 0;
}

extern "C" int s2v_vae_postprocess_u8(const void* video, int32_t C, int32_t F, int32_t H, int32_t W, uint8_t* out, int32_t dtype,
                                      s2v_stream stream) {
    S2V_REQUIRE(video && out, "s2v_vae_postprocess_u8: null argument");
    return launch_postprocess_u8(video, C, F, H, W, out, dtype, (hipStream_t)stream);
}

extern "C" int s2v_vae_postprocess(const void* video, int32_t C, int32_t F, int32_t H, int32_t W, float* out, int32_t dtype,
                                   s2v_stream stream) {
    S2V_REQUIRE(video && out, "s2v_vae_postprocess: null argument");
    return launch_postprocess(video, C, F, H, W, out, dtype, (hipStream_t)stream);
}

// =====================================================================================================================
// Reference-image ENCODE (src/video_generate.py:26-38): one frame through CogVideoXEncoder3D
// (autoencoder_kl_cogvideox.py:755-814: conv_in, DownBlock3D x4 of plain-GroupNorm resnets + CogVideoXDownsample3D,
// MidBlock3D, norm_out, conv_out), untiled or tiled (:1177-1203, 1300-1372), moments out; the posterior sample is
// s2v_vae_gaussian_sample.  The handle is an s2v_vae whose plan was built by s2v_vae_enc_create: weights load through
// s2v_vae_load_weight under the reference's "encoder.*" names and it is destroyed by s2v_vae_destroy.
// Plan layout: stages[0 .. nb-1] = down blocks (StageL::up is the stride-2 downsampler conv), stages[nb] = mid block;
// ConvL::level l works at (tile height >> l) x (tile width >> l).
static int make_gn(s2v_vae* v, SNormL& n, const std::string& name, int C) {
    n.C = C;
    S2V_TRY(wmalloc(v, &n.gn_w, (int64_t)C * v->esz));
    S2V_TRY(wmalloc(v, &n.gn_b, (int64_t)C * v->esz));
    v->slots[name + ".weight"] = VSlot{0, n.gn_w, C, 1, 1, false};
    v->slots[name + ".bias"] = VSlot{0, n.gn_b, C, 1, 1, false};
    return 0;
}
static int make_resnet_gn(s2v_vae* v, ResnetL& r, const std::string& name, int cin, int cout, int level) {
    r.cin = cin; r.cout = cout; r.has_sc = cin != cout;
    S2V_TRY(make_gn(v, r.n1, name + ".norm1", cin));
    S2V_TRY(make_conv(v, r.c1, name + ".conv1.conv", cin, cout, 3, level));
    S2V_TRY(make_gn(v, r.n2, name + ".norm2", cout));
    S2V_TRY(make_conv(v, r.c2, name + ".conv2.conv", cout, cout, 3, level));
    if (r.has_sc) S2V_TRY(make_conv(v, r.sc, name + ".conv_shortcut", cin, cout, 0, level));
    return 0;
}

static int build_encoder(s2v_vae* v) {
    const s2v_vae_config* cfg = &v->cfg;
    const int nb = cfg->num_blocks;
    const int tlevel = (int)std::lround(std::log2((double)cfg->temporal_compression_ratio));
    int r = make_conv(v, v->conv_in, "encoder.conv_in.conv", cfg->out_channels, cfg->block_out_channels[0], 3, 0);
    v->stages.resize(nb + 1);
    char nm[128];
    int prev = cfg->block_out_channels[0], level = 0;
    for (int b = 0; b < nb && !r; ++b) {
        StageL& s = v->stages[b];
        const int co = cfg->block_out_channels[b];
        s.res.resize(cfg->layers_per_block);
        for (int i = 0; i < cfg->layers_per_block && !r; ++i) {
            snprintf(nm, sizeof(nm), "encoder.down_blocks.%d.resnets.%d", b, i);
            r = make_resnet_gn(v, s.res[i], nm, i == 0 ? prev : co, co, level);
        }
        prev = co;
        s.has_up = b != nb - 1;  // here: has a downsampler
        s.compress_time = b < tlevel;
        if (s.has_up && !r) {
            snprintf(nm, sizeof(nm), "encoder.down_blocks.%d.downsamplers.0.conv", b);
            r = make_conv(v, s.up, nm, co, co, 1, level);  // operand at the level it reads; output one level down
            level++;
        }
    }
    if (!r) {
        v->stages[nb].res.resize(2);
        for (int i = 0; i < 2 && !r; ++i) {
            snprintf(nm, sizeof(nm), "encoder.mid_block.resnets.%d", i);
            r = make_resnet_gn(v, v->stages[nb].res[i], nm, prev, prev, level);
        }
    }
    if (!r) r = make_gn(v, v->norm_out, "encoder.norm_out", prev);
    if (!r) r = make_conv(v, v->conv_out, "encoder.conv_out.conv", prev, 2 * v->Cz, 3, level);
    return r;
}

extern "C" int s2v_vae_enc_create(const s2v_vae_config* cfg, s2v_vae** out) {
    S2V_REQUIRE(cfg && out, "s2v_vae_enc_create: null argument");
    S2V_REQUIRE(cfg->dtype == S2V_DTYPE_F32 || cfg->dtype == S2V_DTYPE_BF16 || cfg->dtype == S2V_DTYPE_F16, "s2v_vae_enc_create: unsupported dtype");
    S2V_REQUIRE(cfg->num_blocks >= 1 && cfg->num_blocks <= 6, "s2v_vae_enc_create: 1..6 blocks");
    s2v_vae* v = new s2v_vae();
    v->cfg = *cfg;
    v->dtype = cfg->dtype;
    v->esz = cfg->dtype == S2V_DTYPE_F32 ? 4 : 2;
    v->G = cfg->norm_num_groups;
    v->Cz = cfg->latent_channels;
    v->mfma = cfg->dtype == S2V_DTYPE_BF16 && !cfg->force_simple;
    v->h16 = cfg->dtype == S2V_DTYPE_F16 && !cfg->force_simple;  // fp16: convolutions / shortcuts with cin % 64 == 0 on v_mfma_f32_32x32x16_f16 (gemm_f16)
    v->encoder = true;
    const int r = build_two_pass(v, build_encoder);
    if (r) { s2v_vae_destroy(v); return r; }
    *out = v;
    return 0;
}

static int enc_prepare(s2v_vae* v, int TH, int TW) {
    if (v->th >= TH && v->tw >= TW) return 0;
    S2V_CHECK_HIP(hipDeviceSynchronize());
    for (void* p : v->geo_allocs) (void)hipFree(p);
    v->geo_allocs.clear();
    TH = std::max(TH, v->th); TW = std::max(TW, v->tw);
    v->th = TH; v->tw = TW;
    int rc = 0;
    int64_t dmax = 0;
    for_each_conv(v, [&](ConvL& c) {
        const int64_t H = TH >> c.level, W = TW >> c.level;
        const int F = c.kt == 3 ? 3 : 1;
        c.pad_bytes = (int64_t)F * (H + 2) * (W + 2) * c.cin * v->esz + 1024;
        if (!rc) rc = dmalloc(v, &c.pad, c.pad_bytes, true);
        dmax = std::max(dmax, H * W * std::max(c.cin, c.cout) * v->esz);
    });
    if (rc) return rc;
    v->dense_bytes = dmax + (int64_t)256 * 1024 * v->esz;
    for (int i = 0; i < 3; ++i) S2V_TRY(dmalloc(v, &v->dense[i], v->dense_bytes, true));
    S2V_TRY(dmalloc(v, &v->gn_part, gn_stats_scratch_bytes((int64_t)TH * TW, v->G) + 64, true));
    v->cur_h = v->cur_w = 0;
    return 0;
}

static int run_gn_plain(s2v_vae* v, const SNormL& n, const void* x, int H, int W, void* out_pad, int f_off, hipStream_t st) {
    S2V_TRY(launch_gn_stats(x, (int64_t)H * W, n.C, v->G, v->sums, v->gn_part, v->dtype, st));
    SNormArgs a{};
    a.x = x; a.F = 1; a.H = H; a.W = W; a.C = n.C; a.G = v->G; a.sums = v->sums; a.eps = v->cfg.norm_eps;
    a.gn_w = n.gn_w; a.gn_b = n.gn_b; a.Fz = 1; a.hz = 1; a.wz = 1; a.Cz = v->Cz;
    a.out = out_pad; a.f_off = f_off; a.silu = 1;  // yt == nullptr: plain GroupNorm
    return launch_snorm_apply(a, v->dtype, st);
}

// CogVideoXDownsample3D for one frame: F.pad(x, (0,1,0,1)) then Conv2d(3x3, stride 2): the operand sits at (1,1) of its
// zero-bordered buffer, so reading from (1,1) with stride 2 sees the zero column / row on the right / bottom only
static int run_conv_down(s2v_vae* v, ConvL& c, int H, int W, void* out, hipStream_t st) {
    GemmArgs g{};
    g.A = c.pad + ((int64_t)(W + 2) + 1) * c.cin * v->esz; g.W = c.w; g.ldw = 9 * c.cin; g.bias = c.b; g.C = out; g.ldc = c.cout;
    g.M = (H / 2) * (W / 2); g.N = c.cout; g.K = 9 * c.cin;
    g.conv = 1; g.cin = c.cin; g.Hp = H + 2; g.Wp = W + 2; g.oH = H / 2; g.oW = W / 2; g.kt = 1; g.cstride = 2;
    g.w_rows_padded = (int)rup64(c.cout, 256);
    if (v->mfma && c.cin % 64 == 0) return launch_gemm_bf16(g, EPI_BIAS, st);
    if (v->h16 && gemm_f16_ok(g, EPI_BIAS)) return launch_gemm_f16(g, EPI_BIAS, st);
    g.valu_only = v->cfg.force_simple;
    return launch_gemm_simple(g, EPI_BIAS, v->dtype, st);
}

// encoder pass over the (th x tw) pixel window at (y0, x0) of image [3][1][H][W]; dst = moments [2Cz][1][th/s][tw/s]
static int encode_window(s2v_vae* v, const void* image, int Himg, int Wimg, int y0, int x0, int th, int tw, char* dst,
                         hipStream_t st) {
    S2V_TRY(set_layout(v, th, tw, st));
    int cur = 0, t1 = 1, t2 = 2;
    int H = th, W = tw;
    S2V_TRY(launch_image_to_padded(image, v->cfg.out_channels, 1, Himg, Wimg, y0, x0, th, tw, v->conv_in.pad, 2, v->dtype, st));
    S2V_TRY(run_conv(v, v->conv_in, 1, H, W, true, EPI_BIAS, nullptr, v->dense[cur], st));
    for (auto& s : v->stages) {
        for (auto& r : s.res) {
            S2V_TRY(run_gn_plain(v, r.n1, v->dense[cur], H, W, r.c1.pad, 2, st));
            S2V_TRY(run_conv(v, r.c1, 1, H, W, true, EPI_BIAS, nullptr, v->dense[t1], st));
            S2V_TRY(run_gn_plain(v, r.n2, v->dense[t1], H, W, r.c2.pad, 2, st));
            if (r.has_sc) {
                GemmArgs g{};
                g.A = v->dense[cur]; g.lda = r.cin; g.W = r.sc.w; g.ldw = r.cin; g.bias = r.sc.b;
                g.C = v->dense[t2]; g.ldc = r.cout; g.M = H * W; g.N = r.cout; g.K = r.cin;
                g.a_rows_padded = (int)rup64(g.M, 256);
                if (v->mfma && r.cin % 64 == 0) S2V_TRY(launch_gemm_bf16(g, EPI_BIAS, st));
                else if (v->h16 && gemm_f16_ok(g, EPI_BIAS)) S2V_TRY(launch_gemm_f16(g, EPI_BIAS, st));
                else { g.valu_only = v->cfg.force_simple; S2V_TRY(launch_gemm_simple(g, EPI_BIAS, v->dtype, st)); }
                S2V_TRY(run_conv(v, r.c2, 1, H, W, true, EPI_BIAS_ADD, v->dense[t2], v->dense[t2], st));
                std::swap(cur, t2);
            } else {
                S2V_TRY(run_conv(v, r.c2, 1, H, W, true, EPI_BIAS_ADD, v->dense[cur], v->dense[cur], st));
            }
        }
        if (s.has_up) {  // downsampler (one frame: the temporal average pool of compress_time keeps the frame as it is)
            S2V_TRY(launch_dense_to_padded(v->dense[cur], 1, H, W, s.up.cin, s.up.pad, 0, v->dtype, st));
            S2V_TRY(run_conv_down(v, s.up, H, W, v->dense[t1], st));
            std::swap(cur, t1);
            H /= 2; W /= 2;
        }
    }
    S2V_TRY(run_gn_plain(v, v->norm_out, v->dense[cur], H, W, v->conv_out.pad, 2, st));
    S2V_TRY(run_conv(v, v->conv_out, 1, H, W, true, EPI_BIAS, nullptr, v->dense[t1], st));
    return launch_to_ncfhw(v->dense[t1], 1, H, W, 2 * v->Cz, dst, 1, 0, v->dtype, st);
}

struct EncTileGeo { int ts_h, ts_w, ov_h, ov_w, bl_h, bl_w, lim_h, lim_w; };
static EncTileGeo enc_tile_geo(s2v_vae* v) {  // :1102-1114, 1317-1323 incl. the int() truncations
    EncTileGeo t;
    const int sc = spatial_scale(v);
    t.ts_h = v->cfg.sample_height / 2; t.ts_w = v->cfg.sample_width / 2;
    const int tl_h = (int)((double)t.ts_h / sc), tl_w = (int)((double)t.ts_w / sc);
    t.ov_h = (int)(t.ts_h * (1.0 - 1.0 / 6.0)); t.ov_w = (int)(t.ts_w * (1.0 - 1.0 / 5.0));
    t.bl_h = (int)(tl_h * (1.0 / 6.0)); t.bl_w = (int)(tl_w * (1.0 / 5.0));
    t.lim_h = tl_h - t.bl_h; t.lim_w = tl_w - t.bl_w;
    return t;
}

extern "C" int s2v_vae_encode_shape(s2v_vae* v, int32_t H, int32_t W, int32_t tiling, int32_t* ho, int32_t* wo) {
    S2V_REQUIRE(v && v->encoder && ho && wo && H >= 1 && W >= 1, "s2v_vae_encode_shape: bad argument");
    const int sc = spatial_scale(v);
    EncTileGeo t = enc_tile_geo(v);
    if (!(tiling && (W > t.ts_w || H > t.ts_h))) { *ho = H / sc; *wo = W / sc; return 0; }
    S2V_REQUIRE(t.ov_h > 0 && t.ov_w > 0, "s2v_vae_encode_shape: degenerate tile overlap");
    int h = 0, w = 0;
    for (int i = 0; i < H; i += t.ov_h) h += std::min(std::min(t.ts_h, H - i) / sc, t.lim_h);
    for (int j = 0; j < W; j += t.ov_w) w += std::min(std::min(t.ts_w, W - j) / sc, t.lim_w);
    *ho = h; *wo = w;
    return 0;
}

extern "C" int s2v_vae_encode(s2v_vae* v, const void* image, int32_t H, int32_t W, int32_t tiling, void* moments,
                              s2v_stream stream) {
    S2V_REQUIRE(v && image && moments, "s2v_vae_encode: null argument");
    S2V_REQUIRE(v->encoder, "s2v_vae_encode: this handle holds the decoder (use s2v_vae_enc_create)");
    S2V_REQUIRE(v->finalized, "s2v_vae_encode: weights not finalized");
    const int sc = spatial_scale(v);
    hipStream_t st = (hipStream_t)stream;
    EncTileGeo t = enc_tile_geo(v);
    const bool tiled = tiling && (W > t.ts_w || H > t.ts_h);
    if (!tiled) {
        S2V_REQUIRE(H % sc == 0 && W % sc == 0, "s2v_vae_encode: image sides must be multiples of the spatial compression");
        S2V_TRY(enc_prepare(v, H, W));
        return encode_window(v, image, H, W, 0, 0, H, W, (char*)moments, st);
    }
    S2V_REQUIRE(t.ov_h > 0 && t.ov_w > 0, "s2v_vae_encode: degenerate tile overlap");
    std::vector<int> is, js;
    for (int i = 0; i < H; i += t.ov_h) is.push_back(i);
    for (int j = 0; j < W; j += t.ov_w) js.push_back(j);
    for (int i : is) S2V_REQUIRE(std::min(t.ts_h, H - i) % sc == 0, "s2v_vae_encode: tile height not a multiple of the spatial compression");
    for (int j : js) S2V_REQUIRE(std::min(t.ts_w, W - j) % sc == 0, "s2v_vae_encode: tile width not a multiple of the spatial compression");
    S2V_TRY(enc_prepare(v, std::min(t.ts_h, H), std::min(t.ts_w, W)));
    const size_t nt = is.size() * js.size();
    const int C = 2 * v->Cz;
    bool realloc_tiles = v->tiles.size() != nt;
    for (size_t k = 0; !realloc_tiles && k < nt; ++k) {
        const int th = std::min(t.ts_h, H - is[k / js.size()]) / sc, tw = std::min(t.ts_w, W - js[k % js.size()]) / sc;
        if (v->tile_h[k] != th || v->tile_w[k] != tw) realloc_tiles = true;
    }
    if (realloc_tiles) {
        S2V_CHECK_HIP(hipDeviceSynchronize());
        for (char* p : v->tiles) (void)hipFree(p);
        v->tiles.assign(nt, nullptr); v->tile_h.assign(nt, 0); v->tile_w.assign(nt, 0);
        for (size_t k = 0; k < nt; ++k) {
            const int th = std::min(t.ts_h, H - is[k / js.size()]) / sc, tw = std::min(t.ts_w, W - js[k % js.size()]) / sc;
            v->tile_h[k] = th; v->tile_w[k] = tw;
            S2V_CHECK_HIP(hipMalloc((void**)&v->tiles[k], (size_t)C * th * tw * v->esz + 16));
        }
    }
    for (size_t k = 0; k < nt; ++k)
        S2V_TRY(encode_window(v, image, H, W, is[k / js.size()], js[k % js.size()], v->tile_h[k] * sc, v->tile_w[k] * sc, v->tiles[k], st));
    int32_t ho, wo;
    S2V_TRY(s2v_vae_encode_shape(v, H, W, 1, &ho, &wo));
    int y0 = 0;
    for (size_t r = 0; r < is.size(); ++r) {
        int x0 = 0, ch = 0;
        for (size_t c = 0; c < js.size(); ++c) {
            const size_t k = r * js.size() + c;
            const int Ht = v->tile_h[k], Wt = v->tile_w[k];
            if (r > 0) {
                const size_t ka = (r - 1) * js.size() + c;
                const int E = std::min(std::min(v->tile_h[ka], Ht), t.bl_h);
                S2V_TRY(launch_blend(v->tiles[ka], v->tile_h[ka], v->tile_w[ka], v->tiles[k], Ht, Wt, C, E, 1, v->dtype, st));
            }
            if (c > 0) {
                const size_t ka = k - 1;
                const int E = std::min(std::min(v->tile_w[ka], Wt), t.bl_w);
                S2V_TRY(launch_blend(v->tiles[ka], v->tile_h[ka], v->tile_w[ka], v->tiles[k], Ht, Wt, C, E, 0, v->dtype, st));
            }
            ch = std::min(Ht, t.lim_h);
            const int cw = std::min(Wt, t.lim_w);
            S2V_TRY(launch_paste(v->tiles[k], Ht, Wt, ch, cw, moments, ho, wo, y0, x0, C, v->dtype, st));
            x0 += cw;
        }
        y0 += ch;
    }
    return 0;
}

extern "C" int s2v_vae_gaussian_sample(const void* moments, const void* noise, int32_t latent_channels, int64_t n_spatial, void* out,
                                       int32_t dtype, s2v_stream stream) {
    S2V_REQUIRE(moments && noise && out && latent_channels > 0 && n_spatial > 0, "s2v_vae_gaussian_sample: bad argument");
    S2V_REQUIRE(dtype == S2V_DTYPE_F32 || dtype == S2V_DTYPE_BF16 || dtype == S2V_DTYPE_F16, "s2v_vae_gaussian_sample: unsupported dtype");
    return launch_gaussian_sample(moments, noise, (int64_t)latent_channels * n_spatial, out, dtype, (hipStream_t)stream);
}
