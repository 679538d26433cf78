// HBM-bound kernels of the denoise step: AdaLN-Zero modulate, per-head QK LayerNorm + RoPE, V^T production,
// timestep / modulation GEMVs, patchify / unpatchify, tail norms, CFG + scheduler step.
// Every kernel is templated on the storage type T (float for the fp32 CPU-parity mode, bf16 otherwise); math is
// fp32 and results pass through ET<T>::rnd at the points where the reference materialises a tensor in the model
// dtype, so the bf16 path rounds where the reference's bf16 CPU run rounds.
#define S2V_HOST
#include "common.h"
#include "kernels.h"
#include <cstdlib>
#include <type_traits>

// ---------------------------------------------------------------------------------------------------
// Row LayerNorm helpers: one wave per row, row cached in registers (D <= 4096, D % (16/sizeof(T)) == 0)
// NR = rounds of 64 lanes x 16 bytes that cover the row (compile time: 6 for D = 3072 in bf16).  Every load is UNCONDITIONAL on a
// clamped address and the lane mask is applied to the value: with a per-chunk bounds branch the compiler closed each load with its
// own s_waitcnt vmcnt(0), i.e. a wave fetched its row in NR dependent memory round trips (3.1 TB/s on the C3 step).
#define LN_MAX_FLOATS 64
template <typename T, int NR>
__device__ __forceinline__ void row_load(const T* x, int D, int lane, float* v) {
    constexpr int VN = Vec16<T>::N;
    const int chunks = D / VN;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = lane + 64 * i;
        const bool ok = c < chunks;
        Vec16<T>::ld(x + (ok ? c : 0) * VN, v + i * VN);
#pragma unroll
        for (int e = 0; e < VN; ++e) v[i * VN + e] = ok ? v[i * VN + e] : 0.f;
    }
}
// normalise the register-resident row in place: v = rnd((v-mean)*rstd*w + b)
template <typename T, int NR>
__device__ __forceinline__ void row_layernorm(float* v, int D, int lane, const T* w, const T* b, float eps) {
    constexpr int VN = Vec16<T>::N;
    const int chunks = D / VN;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NR * VN; ++i) s += v[i];
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const bool ok = lane + 64 * i < chunks;
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const float d = v[i * VN + e] - mean;
            q += ok ? d * d : 0.f;
        }
    }
    const float var = wave_sum(q) / (float)D;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = lane + 64 * i;
        const bool ok = c < chunks;
        const int cc = ok ? c : 0;
        float wv[VN], bv[VN];
        Vec16<T>::ld(w + cc * VN, wv);
        Vec16<T>::ld(b + cc * VN, bv);
#pragma unroll
        for (int e = 0; e < VN; ++e) v[i * VN + e] = ok ? ET<T>::rnd((v[i * VN + e] - mean) * rstd * wv[e] + bv[e]) : 0.f;  // lanes past the row stay 0 (tail_norm normalises twice)
    }
}
// v = rnd(rnd(v * rnd(1+scale)) + shift), then store
template <typename T, int NR>
__device__ __forceinline__ void row_modulate_store(float* v, int D, int lane, const T* shift, const T* scale, T* y) {
    constexpr int VN = Vec16<T>::N;
    const int chunks = D / VN;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = lane + 64 * i;
        const bool ok = c < chunks;
        const int cc = ok ? c : 0;
        float sc[VN], sh[VN], o[VN];
        Vec16<T>::ld(scale + cc * VN, sc);
        Vec16<T>::ld(shift + cc * VN, sh);
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const float s1 = ET<T>::rnd(1.0f + sc[e]);
            const float p = ET<T>::rnd(v[i * VN + e] * s1);
            o[e] = p + sh[e];
        }
        if (ok) Vec16<T>::st(y + c * VN, o);
    }
}
// the same modulation, then per-row e4m3 quantisation of the bf16-rounded result instead of the bf16 store (bf16 rows only): the
// arithmetic of quant_rows_fp8_k on values that never leave the registers
template <int NR>
__device__ __forceinline__ void row_modulate_quant_store(float* v, int D, int lane, const bf16_t* shift, const bf16_t* scale, unsigned char* q,
                                                         float* qscale) {
    const int chunks = D / 8;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = lane + 64 * i;
        const bool ok = c < chunks;
        const int cc = ok ? c : 0;
        float sc[8], sh[8];
        Vec16<bf16_t>::ld(scale + cc * 8, sc);
        Vec16<bf16_t>::ld(shift + cc * 8, sh);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float s1 = ET<bf16_t>::rnd(1.0f + sc[e]);
            const float p = ET<bf16_t>::rnd(v[i * 8 + e] * s1);
            const float o = ok ? ET<bf16_t>::rnd(p + sh[e]) : 0.f;  // the value the bf16 store would have held
            v[i * 8 + e] = o;
            amax = fmaxf(amax, fabsf(o));
        }
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) *qscale = sc;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = lane + 64 * i;
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i * 8 + 0] * inv, v[i * 8 + 1] * inv, w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i * 8 + 2] * inv, v[i * 8 + 3] * inv, w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i * 8 + 4] * inv, v[i * 8 + 5] * inv, w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i * 8 + 6] * inv, v[i * 8 + 7] * inv, w1, true);
        if (c < chunks) *(u32x2*)(q + c * 8) = u32x2{(unsigned)w0, (unsigned)w1};
    }
}
// The same three steps with the parameter vectors (LayerNorm weight / bias, modulation scale / shift) REQUESTED WITH THE ROW: as written
// above the compiler issues each parameter chunk where it is used -- behind the two wave reductions, one chunk at a time -- so a wave
// paid up to 2 NR dependent L2 round trips after its row had arrived (C1: 13.5 us for a 9.6 MB pass).  row_params issues all 4 NR loads
// behind the row's own (clamped addresses, values of lanes past the row are never used); a scheduling barrier keeps them there.
template <typename T, int NR>
struct RowParams { typename Vec16<T>::raw_t w[NR], b[NR], sc[NR], sh[NR]; };
template <typename T, int NR>
__device__ __forceinline__ void row_params(RowParams<T, NR>& p, int D, int lane, const T* w, const T* b, const T* scale, const T* shift) {
    constexpr int VN = Vec16<T>::N;
    const int chunks = D / VN;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = lane + 64 * i;
        const int cc = c < chunks ? c : 0;
        p.w[i] = Vec16<T>::ldraw(w + cc * VN);
        p.b[i] = Vec16<T>::ldraw(b + cc * VN);
        p.sc[i] = Vec16<T>::ldraw(scale + cc * VN);
        p.sh[i] = Vec16<T>::ldraw(shift + cc * VN);
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <typename T, int NR>
__device__ __forceinline__ void row_layernorm_p(float* v, int D, int lane, const RowParams<T, NR>& p, float eps) {
    constexpr int VN = Vec16<T>::N;
    const int chunks = D / VN;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NR * VN; ++i) s += v[i];
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const bool ok = lane + 64 * i < chunks;
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const float d = v[i * VN + e] - mean;
            q += ok ? d * d : 0.f;
        }
    }
    const float var = wave_sum(q) / (float)D;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const bool ok = lane + 64 * i < chunks;
        float wv[VN], bv[VN];
        Vec16<T>::dec(p.w[i], wv);
        Vec16<T>::dec(p.b[i], bv);
#pragma unroll
        for (int e = 0; e < VN; ++e) v[i * VN + e] = ok ? ET<T>::rnd((v[i * VN + e] - mean) * rstd * wv[e] + bv[e]) : 0.f;
    }
}
template <typename T, int NR>
__device__ __forceinline__ void row_modulate_store_p(float* v, int D, int lane, const RowParams<T, NR>& p, T* y) {
    constexpr int VN = Vec16<T>::N;
    const int chunks = D / VN;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = lane + 64 * i;
        float sc[VN], sh[VN], o[VN];
        Vec16<T>::dec(p.sc[i], sc);
        Vec16<T>::dec(p.sh[i], sh);
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const float s1 = ET<T>::rnd(1.0f + sc[e]);
            const float pr = ET<T>::rnd(v[i * VN + e] * s1);
            o[e] = pr + sh[e];
        }
        if (c < chunks) Vec16<T>::st(y + c * VN, o);
    }
}

// rounds needed for D elements of T, rounded up to an instantiated count
template <typename T> static int ln_rounds(int D) {
    const int need = (D / Vec16<T>::N + 63) / 64;
    for (int r : {1, 2, 3, 4, 6, 8, 12, 16})
        if (r >= need) return r;
    return -1;
}
#define S2V_LN_DISPATCH(T, D, ...)                                                   \
    switch (ln_rounds<T>(D)) {                                                        \
        case 1: { constexpr int NR = 1; __VA_ARGS__; } break;                                \
        case 2: { constexpr int NR = 2; __VA_ARGS__; } break;                                \
        case 3: { constexpr int NR = 3; __VA_ARGS__; } break;                                \
        case 4: { constexpr int NR = 4; __VA_ARGS__; } break;                                \
        case 6: { constexpr int NR = 6; __VA_ARGS__; } break;                                \
        case 8: { constexpr int NR = 8; __VA_ARGS__; } break;                                \
        case 12: if constexpr (sizeof(T) == 4) { constexpr int NR = 12; __VA_ARGS__; } break; \
        case 16: if constexpr (sizeof(T) == 4) { constexpr int NR = 16; __VA_ARGS__; } break; \
        default: return s2v_fail(__FILE__, __LINE__, "layer norm: row too long", -1); \
    }

// CogVideoXLayerNormZero.forward (normalization.py:467-484): LN(x)*(1+scale)+shift, three token ranges
template <typename T, int NR>
__global__ __launch_bounds__(256) void ln_modulate_k(const LnModArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.B * a.Ntok) return;
    const int b = row / a.Ntok, r = row - b * a.Ntok;
    const bool txt = r < a.text_len;
    const bool ref = a.shift_ref != nullptr && !txt && r < a.text_len + a.ref_len;
    const T* shift = (const T*)(txt ? a.shift_txt : ref ? a.shift_ref : a.shift_vid) + (size_t)b * a.mod_stride;
    const T* scale = (const T*)(txt ? a.scale_txt : ref ? a.scale_ref : a.scale_vid) + (size_t)b * a.mod_stride;
    float v[NR * Vec16<T>::N];
    if constexpr (std::is_same<T, bf16_t>::value) {
        if (a.q8 != nullptr) {  // fp8 engine: per-row e4m3 image instead of the bf16 store
            row_load<T, NR>((const T*)a.x + (size_t)row * a.ldx, a.D, lane, v);
            row_layernorm<T, NR>(v, a.D, lane, (const T*)a.w, (const T*)a.b, a.eps);
            row_modulate_quant_store<NR>(v, a.D, lane, shift, scale, (unsigned char*)a.q8 + (size_t)row * a.D, a.q8_scale + row);
            return;
        }
    }
    // the row and all four parameter vectors in flight together (row_params), then the same arithmetic as row_layernorm / row_modulate_store
    typename Vec16<T>::raw_t xr[NR];
    {
        constexpr int VN = Vec16<T>::N;
        const int chunks = a.D / VN;
        const T* x = (const T*)a.x + (size_t)row * a.ldx;
#pragma unroll
        for (int i = 0; i < NR; ++i) xr[i] = Vec16<T>::ldraw(x + (lane + 64 * i < chunks ? lane + 64 * i : 0) * VN);
        RowParams<T, NR> prm;
        row_params<T, NR>(prm, a.D, lane, (const T*)a.w, (const T*)a.b, scale, shift);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            Vec16<T>::dec(xr[i], v + i * VN);
            const bool ok = lane + 64 * i < chunks;
#pragma unroll
            for (int e = 0; e < VN; ++e) v[i * VN + e] = ok ? v[i * VN + e] : 0.f;
        }
        row_layernorm_p<T, NR>(v, a.D, lane, prm, a.eps);
        row_modulate_store_p<T, NR>(v, a.D, lane, prm, (T*)a.y + (size_t)row * a.ldy);
    }
}
// The same rows with the four parameter vectors (LayerNorm weight / bias, modulation scale / shift) staged ONCE per workgroup in LDS: ln_modulate_k
// keeps them in registers per row (96 of its 179 VGPRs at D = 3072: two waves per SIMD, 24 KiB of L1 / L2 reads per 6-KiB row); here a workgroup of
// four waves handles 4 x RPW consecutive rows, reads the parameters from LDS where they are used and keeps ~100 registers.  The rows of a workgroup
// share their parameter set unless a text / reference / video or sample boundary falls inside it: those workgroups read the parameters from global
// memory per row, as before.  Arithmetic is row_layernorm / row_modulate_store's, i.e. ln_modulate_k's.
template <typename T, int NR, int RPW>
__global__ __launch_bounds__(256) void ln_modulate_lds_k(const LnModArgs a) {
    extern __shared__ __attribute__((aligned(16))) char ln_smem[];
    constexpr int VN = Vec16<T>::N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows = a.B * a.Ntok, r0 = blockIdx.x * (4 * RPW), rl = min(r0 + 4 * RPW, rows) - 1;
    auto params_of = [&](int row, const T*& shift, const T*& scale) {
        const int b = row / a.Ntok, r = row - b * a.Ntok;
        const bool txt = r < a.text_len;
        const bool ref = a.shift_ref != nullptr && !txt && r < a.text_len + a.ref_len;
        shift = (const T*)(txt ? a.shift_txt : ref ? a.shift_ref : a.shift_vid) + (size_t)b * a.mod_stride;
        scale = (const T*)(txt ? a.scale_txt : ref ? a.scale_ref : a.scale_vid) + (size_t)b * a.mod_stride;
    };
    const T *sh0, *sc0, *sh1, *sc1;
    params_of(r0, sh0, sc0);
    params_of(rl, sh1, sc1);
    const bool uniform = sh0 == sh1 && sc0 == sc1;   // parameter sets are contiguous row ranges: equal ends = one set
    T* lw = (T*)ln_smem;
    T *lb = lw + a.D, *lsc = lb + a.D, *lsh = lsc + a.D;
    if (uniform) {
        const int chunks = a.D / VN;
        for (int c = threadIdx.x; c < chunks; c += 256) {
            *(typename Vec16<T>::raw_t*)(lw + c * VN) = Vec16<T>::ldraw((const T*)a.w + c * VN);
            *(typename Vec16<T>::raw_t*)(lb + c * VN) = Vec16<T>::ldraw((const T*)a.b + c * VN);
            *(typename Vec16<T>::raw_t*)(lsc + c * VN) = Vec16<T>::ldraw(sc0 + c * VN);
            *(typename Vec16<T>::raw_t*)(lsh + c * VN) = Vec16<T>::ldraw(sh0 + c * VN);
        }
        __syncthreads();
    }
#pragma unroll 1
    for (int k = 0; k < RPW; ++k) {
        const int row = r0 + k * 4 + wave;
        if (row >= rows) break;
        const T *pw = lw, *pb = lb, *psc = lsc, *psh = lsh;
        if (!uniform) { pw = (const T*)a.w; pb = (const T*)a.b; params_of(row, psh, psc); }
        float v[NR * VN];
        row_load<T, NR>((const T*)a.x + (size_t)row * a.ldx, a.D, lane, v);
        row_layernorm<T, NR>(v, a.D, lane, pw, pb, a.eps);
        if constexpr (std::is_same<T, bf16_t>::value) {
            if (a.q8 != nullptr) {
                row_modulate_quant_store<NR>(v, a.D, lane, psh, psc, (unsigned char*)a.q8 + (size_t)row * a.D, a.q8_scale + row);
                continue;
            }
        }
        row_modulate_store<T, NR>(v, a.D, lane, psh, psc, (T*)a.y + (size_t)row * a.ldy);
    }
}
int launch_ln_modulate(const LnModArgs& a, int dtype, hipStream_t st) {
    S2V_REQUIRE(a.D <= 4096 && a.D % 8 == 0, "ln_modulate: D must be <= 4096 and a multiple of 8");
    S2V_REQUIRE(a.q8 == nullptr || (dtype == S2V_BF16 && a.q8_scale != nullptr), "ln_modulate: the fp8 output needs bf16 rows and a scale vector");
    const int rows = a.B * a.Ntok;
    dim3 grid((rows + 3) / 4);
    // many rows (at least four workgroups of sixteen rows per CU): parameters staged in LDS per workgroup
    int lds_rows = 16384;
#ifdef S2V_DIAG
    if (const char* e = getenv("S2V_LN_LDS_ROWS")) lds_rows = atoi(e);  // tools/ln_lds_probe.py (diagnostics library): 0 = never
#endif
    if (dtype == S2V_BF16 && lds_rows > 0 && rows >= lds_rows) {
        constexpr int RPW = 4;
        dim3 g16((rows + 4 * RPW - 1) / (4 * RPW));
        const size_t lds = (size_t)4 * a.D * 2;
        S2V_LN_DISPATCH(bf16_t, a.D, hipLaunchKernelGGL((ln_modulate_lds_k<bf16_t, NR, RPW>), g16, dim3(256), lds, st, a))
        S2V_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (dtype == S2V_BF16) {
        S2V_LN_DISPATCH(bf16_t, a.D, hipLaunchKernelGGL((ln_modulate_k<bf16_t, NR>), grid, dim3(256), 0, st, a))
    } else if (dtype == S2V_F16) {
        S2V_LN_DISPATCH(f16_t, a.D, hipLaunchKernelGGL((ln_modulate_k<f16_t, NR>), grid, dim3(256), 0, st, a))
    } else {
        S2V_LN_DISPATCH(float, a.D, hipLaunchKernelGGL((ln_modulate_k<float, NR>), grid, dim3(256), 0, st, a))
    }
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// norm_final -> norm_out (AdaLayerNorm, shift first) on the video rows (cogvideox_transformer_3d.py:536-542)
template <typename T, int NR>
__global__ __launch_bounds__(256) void tail_norm_k(const TailNormArgs a) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= a.B * a.V) return;
    const int b = idx / a.V, vr = idx - b * a.V;
    const size_t xrow = (size_t)b * a.Ntok + a.row0 + vr;
    float v[NR * Vec16<T>::N];
    row_load<T, NR>((const T*)a.x + xrow * a.ldx, a.D, lane, v);
    row_layernorm<T, NR>(v, a.D, lane, (const T*)a.w1, (const T*)a.b1, a.eps);
    row_layernorm<T, NR>(v, a.D, lane, (const T*)a.w2, (const T*)a.b2, a.eps);
    const T* shift = (const T*)a.shift + (size_t)b * a.mod_stride;
    const T* scale = (const T*)a.scale + (size_t)b * a.mod_stride;
    row_modulate_store<T, NR>(v, a.D, lane, shift, scale, (T*)a.y + (size_t)idx * a.ldy);
}
int launch_tail_norm(const TailNormArgs& a, int dtype, hipStream_t st) {
    S2V_REQUIRE(a.D <= 4096 && a.D % 8 == 0, "tail_norm: D must be <= 4096 and a multiple of 8");
    dim3 grid((a.B * a.V + 3) / 4);
    if (dtype == S2V_BF16) {
        S2V_LN_DISPATCH(bf16_t, a.D, hipLaunchKernelGGL((tail_norm_k<bf16_t, NR>), grid, dim3(256), 0, st, a))
    } else if (dtype == S2V_F16) {
        S2V_LN_DISPATCH(f16_t, a.D, hipLaunchKernelGGL((tail_norm_k<f16_t, NR>), grid, dim3(256), 0, st, a))
    } else {
        S2V_LN_DISPATCH(float, a.D, hipLaunchKernelGGL((tail_norm_k<float, NR>), grid, dim3(256), 0, st, a))
    }
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// per-head LN(64) on q and k + RoPE (attention_processor.py:2060-2080, embeddings.py:759-778).
// 8 lanes per 64-wide head vector (8 contiguous elements per lane); one thread-task = (row, q|k, head, octet)
// Grid: x = row, y = 256-task slices of the row's 2 * H * 8 tasks (no division by the task count); the rotary tables of the row are
// requested together with its q / k chunk (clamped address for text rows), so a thread pays ONE memory latency, not two.
template <typename T>
__global__ __launch_bounds__(256) void qk_norm_rope_k(const QkNormRopeArgs a) {
    const int D = a.H * 64;
    const int row = blockIdx.x;
    const int t = blockIdx.y * 256 + threadIdx.x;
    if (t >= 2 * a.H * 8) return;  // whole 8-lane groups exit together
    const int which = t >= a.H * 8;  // 0 = q, 1 = k
    const int hh = (t >> 3) - which * a.H;
    const int oct = t & 7;
    T* p = (T*)a.qkv + (size_t)row * a.ld_qkv + which * D + hh * 64 + oct * 8;
    const int r = row % a.Ntok;
    const bool rope = a.cos != nullptr && r >= a.text_len;
    const size_t tab = (size_t)(rope ? r - a.text_len : 0) * 64 + oct * 8;
    float v[8], cs[8], sn[8], wv[8], bv[8];
    if constexpr (sizeof(T) == 2) {
        Vec16<T>::ld(p, v);
    } else {
        Vec16<T>::ld(p, v);
        Vec16<T>::ld(p + 4, v + 4);
    }
    if (a.cos != nullptr) {
        Vec16<float>::ld(a.cos + tab, cs);
        Vec16<float>::ld(a.cos + tab + 4, cs + 4);
        Vec16<float>::ld(a.sin + tab, sn);
        Vec16<float>::ld(a.sin + tab + 4, sn + 4);
    }
    const T* w = (const T*)(which ? a.nk_w : a.nq_w) + oct * 8;
    const T* bb = (const T*)(which ? a.nk_b : a.nq_b) + oct * 8;
    if constexpr (sizeof(T) == 2) {
        Vec16<T>::ld(w, wv);
        Vec16<T>::ld(bb, bv);
    } else {
        Vec16<T>::ld(w, wv);
        Vec16<T>::ld(w + 4, wv + 4);
        Vec16<T>::ld(bb, bv);
        Vec16<T>::ld(bb + 4, bv + 4);
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    const float mean = s * (1.0f / 64.0f);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; q += d * d; }
    q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + a.eps);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ET<T>::rnd((v[e] - mean) * rstd * wv[e] + bv[e]);
    if (rope) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const float x0 = v[e], x1 = v[e + 1];
            v[e] = ET<T>::rnd(x0 * cs[e] + (-x1) * sn[e]);
            v[e + 1] = ET<T>::rnd(x1 * cs[e + 1] + x0 * sn[e + 1]);
        }
    }
    if constexpr (sizeof(T) == 2) {
        Vec16<T>::st(p, v);
    } else {
        Vec16<T>::st(p, v);
        Vec16<T>::st(p + 4, v + 4);
    }
}

// V [B*Ntok, .] (cols 2D + h*64 + d) -> V^T [B][H][64][ntok_pad]; inside every aligned 16-token group the
// tokens are stored in the order [0-3, 8-11, 4-7, 12-15] (bits 2 and 3 of the token index swapped) which is the
// k-slot order the PV MFMA of attn_bf16_k consumes.  64 tokens x 64 dims per block through LDS.
// to_f16: the values are written as fp16 (attn_q4h: P.V on the fp16 MFMA).  A bf16 value converts exactly when its magnitude lies in
// [2^-14, 65504]; below, it rounds into the fp16 subnormals; ABOVE, it is SATURATED to +-65504 -- an infinity in V^T would turn the whole
// head-dim column into NaN through 0 * inf in P.V, which no slow-path census would notice.  Values of that size do not occur
// in V of this model (LayerNorm-ed inputs through one projection); bf16 P (attn_p_format 0, the default) has no such bound.
__global__ __launch_bounds__(256) void v_transpose_k(const bf16_t* qkv, int ld_qkv, int B, int H, int Ntok, bf16_t* vt,
                                                     int ntok_pad, int to_f16) {
    __shared__ bf16_t tile[64][64 + 2];
    const int n0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int D = H * 64;
    const int tid = threadIdx.x;
    // load: 64 rows x 128 B, 8 lanes x 16 B per row
    for (int i = tid; i < 64 * 8; i += 256) {
        const int rr = i >> 3, cc = (i & 7) * 8;
        const int n = n0 + rr;
        u32x4 val = {0, 0, 0, 0};
        if (n < Ntok) val = *(const u32x4*)(qkv + (size_t)(b * Ntok + n) * ld_qkv + 2 * D + h * 64 + cc);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tile[rr][cc + 2 * e] = (bf16_t)(val[e] & 0xffff);
            tile[rr][cc + 2 * e + 1] = (bf16_t)(val[e] >> 16);
        }
    }
    __syncthreads();
    bf16_t* dst = vt + (size_t)(b * H + h) * 64 * ntok_pad;
    for (int i = tid; i < 64 * 64; i += 256) {
        const int d = i >> 6, pos = i & 63;
        const int tok = (pos & ~12) | ((pos & 4) << 1) | ((pos & 8) >> 1);  // token stored at this position
        if (n0 + tok < Ntok) {
            bf16_t v = tile[tok][d];
            if (to_f16) {
                float f = __uint_as_float((unsigned)v << 16);
                f = f != f ? f : fminf(fmaxf(f, -65504.0f), 65504.0f);  // NaN stays NaN (it is one in bf16 too); +-inf and > 65504 saturate
                v = __builtin_bit_cast(unsigned short, (_Float16)f);
            }
            dst[(size_t)d * ntok_pad + n0 + pos] = v;
        }
        else if (n0 + pos < ntok_pad) dst[(size_t)d * ntok_pad + n0 + pos] = 0;
    }
}

// q / k of the normalised, rotated QKV buffer -> MX e4m3 images for the fp8 QK^T of attn_q4f (AttnArgs::q8 ... k8s; weight_format 2; no
// reference code: the reference has no fp8 path).  A block = 64 tokens of one head of q or k; a thread = 16 head-dim elements, two threads = one
// 32-element MX block: x = bf16 value (q: times scale * log2 e, in fp32), amax over the block, E8M0 scale = the smallest power of two with
// amax / scale <= 448, byte = rne_e4m3(x / scale).  Rows [Ntok, ntok_pad) of k are written as zeros with unit scales.
__global__ __launch_bounds__(256) void qk_quant_mx_k(const bf16_t* qkv, int ld_qkv, int H, int Ntok, int ntok_pad, float q_prescale,
                                                      unsigned char* q8, unsigned short* q8s, unsigned char* k8, unsigned* k8s) {
    const int tile = blockIdx.x, h = blockIdx.y, b = blockIdx.z >> 1, is_k = blockIdx.z & 1;
    const int tid = threadIdx.x, kr = tid >> 2, qt = tid & 3, blk = qt >> 1;
    const int n = tile * 64 + kr;
    if (!is_k && n >= Ntok) return;
    const int D = H * 64;
    float x[16];
    if (n < Ntok) {
        const bf16_t* src = qkv + (size_t)(b * Ntok + n) * ld_qkv + is_k * D + h * 64 + qt * 16;
        const u32x4 v0 = *(const u32x4*)src, v1 = *(const u32x4*)(src + 8);
        const float c = is_k ? 1.0f : q_prescale;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[2 * e] = __uint_as_float(v0[e] << 16) * c; x[2 * e + 1] = __uint_as_float(v0[e] & 0xffff0000u) * c;
            x[8 + 2 * e] = __uint_as_float(v1[e] << 16) * c; x[8 + 2 * e + 1] = __uint_as_float(v1[e] & 0xffff0000u) * c;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = 0.f;
    }
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) amax = fmaxf(amax, fabsf(x[e]));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    unsigned eb = (__float_as_uint(amax * (1.0f / 448.0f)) + 0x7fffffu) >> 23;
    eb = n < Ntok ? min(max(eb, 1u), 253u) : 127u;
    const float inv = __uint_as_float((254u - eb) << 23);
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        int r = 0;
        r = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * w] * inv, x[4 * w + 1] * inv, r, false);
        r = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * w + 2] * inv, x[4 * w + 3] * inv, r, true);
        o[w] = (unsigned)r;
    }
    const size_t bh = (size_t)b * H + h;
    if (is_k) {
        *(u32x4*)(k8 + (bh * ntok_pad + n) * 64 + qt * 16) = o;
        if ((qt & 1) == 0)
            ((unsigned char*)k8s)[((bh * (ntok_pad / 64) + tile) * 64 + blk * 32 + (kr & 31)) * 4 + (kr >> 5)] = (unsigned char)eb;
    } else {
        *(u32x4*)(q8 + (bh * Ntok + n) * 64 + qt * 16) = o;
        if ((qt & 1) == 0) ((unsigned char*)q8s)[(bh * Ntok + n) * 2 + blk] = (unsigned char)eb;
    }
}
int launch_qk_quant_mx(const void* qkv, int ld_qkv, int B, int H, int Ntok, int ntok_pad, float q_prescale, unsigned char* q8, unsigned short* q8s,
                       unsigned char* k8, unsigned* k8s, hipStream_t st) {
    S2V_REQUIRE(ntok_pad % 64 == 0 && ntok_pad >= Ntok, "qk_quant_mx: ntok_pad must be a multiple of 64 covering Ntok");
    hipLaunchKernelGGL(qk_quant_mx_k, dim3(ntok_pad / 64, H, 2 * B), dim3(256), 0, st, (const bf16_t*)qkv, ld_qkv, H, Ntok, ntok_pad, q_prescale, q8,
                       q8s, k8, k8s);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// The fp8-QK^T engine's q / k pass with the per-head LayerNorm + rotary embedding folded in (round 5): the RAW q / k of the QKV projection (plain
// bias epilogue) -> qk_norm_rope_k's arithmetic with its rounding points (the same bits as the stand-alone kernel and as the projection's fused
// EPI_BIAS_QKNORM epilogue: octet sums in element order, the 8-lane butterfly -- here one in-lane add of the thread's two octets and two shuffles --,
// correctly rounded sqrt / reciprocal, (d * rstd) * w + b rounded to bf16, the rotary pair on the rounded values) -> qk_quant_mx_k's MX e4m3 images.
// The attention kernel reads q and k ONLY through these images (attn_q4f), so the normalised bf16 q / k are never written: the pass moves the bytes
// the quantisation pass moved anyway, and the projection's epilogue -- exposed VALU work behind a K loop half as long as the bf16 one -- goes back
// to bias + rounding.  Thread = 16 head-dim elements (octets 2 qt, 2 qt + 1) of a (token, head); grid as qk_quant_mx_k.
__global__ __launch_bounds__(256) void qk_norm_quant_mx_k(const QkNormRopeArgs a, float q_prescale, unsigned char* q8, unsigned short* q8s, unsigned char* k8,
                                                           unsigned* k8s, int hg) {
    // a block = 64 tokens x hg heads of q or k: the rotary values and the LayerNorm parameters of a thread (its token, its 16 head-dim elements) are the
    // same for every head, so they are fetched once per hg heads (one head per block read four times the q / k bytes in table values)
    const int tile = blockIdx.x, h0 = blockIdx.y * hg, b = blockIdx.z >> 1, is_k = blockIdx.z & 1;
    const int tid = threadIdx.x, kr = tid >> 2, qt = tid & 3, blk = qt >> 1;
    const int n = tile * 64 + kr;
    const int H = a.H, Ntok = a.Ntok, ntok_pad = a.ntok_pad;
    if (!is_k && n >= Ntok) return;  // the four threads of a row leave together
    const int D = H * 64;
    const bool live = n < Ntok;
    const bool rope = live && a.cos != nullptr && n >= a.text_len;
    float cs[16], sn[16], wv[16], bv[16];
    if (live) {
        const size_t tab = (size_t)(rope ? n - a.text_len : 0) * 64 + qt * 16;
        if (a.cos != nullptr) {
#pragma unroll
            for (int e = 0; e < 16; e += 4) {
                Vec16<float>::ld(a.cos + tab + e, cs + e);
                Vec16<float>::ld(a.sin + tab + e, sn + e);
            }
        }
        const bf16_t* w = (const bf16_t*)(is_k ? a.nk_w : a.nq_w) + qt * 16;
        const bf16_t* bb = (const bf16_t*)(is_k ? a.nk_b : a.nq_b) + qt * 16;
        Vec16<bf16_t>::ld(w, wv); Vec16<bf16_t>::ld(w + 8, wv + 8);
        Vec16<bf16_t>::ld(bb, bv); Vec16<bf16_t>::ld(bb + 8, bv + 8);
    }
    const float c = is_k ? 1.0f : q_prescale;
#pragma unroll 1
    for (int h = h0; h < h0 + hg; ++h) {
        float x[16];
        if (live) {
            const bf16_t* src = (const bf16_t*)a.qkv + (size_t)(b * Ntok + n) * a.ld_qkv + is_k * D + h * 64 + qt * 16;
            Vec16<bf16_t>::ld(src, x);
            Vec16<bf16_t>::ld(src + 8, x + 8);
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { s0 += x[e]; s1 += x[8 + e]; }
            float s = s0 + s1;
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
            const float mean = s * (1.0f / 64.0f);
            float q0 = 0.f, q1 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d0 = x[e] - mean, d1 = x[8 + e] - mean;
                q0 += d0 * d0; q1 += d1 * d1;
            }
            float q = q0 + q1;
            q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64);
            const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + a.eps);
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e] = ET<bf16_t>::rnd((x[e] - mean) * rstd * wv[e] + bv[e]);
            if (rope) {
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const float x0 = x[e], x1 = x[e + 1];
                    x[e] = ET<bf16_t>::rnd(x0 * cs[e] + (-x1) * sn[e]);
                    x[e + 1] = ET<bf16_t>::rnd(x1 * cs[e + 1] + x0 * sn[e + 1]);
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e] *= c;
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e] = 0.f;
        }
        float amax = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) amax = fmaxf(amax, fabsf(x[e]));
        amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
        unsigned eb = (__float_as_uint(amax * (1.0f / 448.0f)) + 0x7fffffu) >> 23;
        eb = live ? min(max(eb, 1u), 253u) : 127u;
        const float inv = __uint_as_float((254u - eb) << 23);
        u32x4 o;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            int r = 0;
            r = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * w] * inv, x[4 * w + 1] * inv, r, false);
            r = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * w + 2] * inv, x[4 * w + 3] * inv, r, true);
            o[w] = (unsigned)r;
        }
        const size_t bh = (size_t)b * H + h;
        if (is_k) {
            *(u32x4*)(k8 + (bh * ntok_pad + n) * 64 + qt * 16) = o;
            if ((qt & 1) == 0)
                ((unsigned char*)k8s)[((bh * (ntok_pad / 64) + tile) * 64 + blk * 32 + (kr & 31)) * 4 + (kr >> 5)] = (unsigned char)eb;
        } else {
            *(u32x4*)(q8 + (bh * Ntok + n) * 64 + qt * 16) = o;
            if ((qt & 1) == 0) ((unsigned char*)q8s)[(bh * Ntok + n) * 2 + blk] = (unsigned char)eb;
        }
    }
}
int launch_qk_norm_quant_mx(const QkNormRopeArgs& a, float q_prescale, unsigned char* q8, unsigned short* q8s, unsigned char* k8, unsigned* k8s, hipStream_t st) {
    S2V_REQUIRE(a.ntok_pad % 64 == 0 && a.ntok_pad >= a.Ntok, "qk_norm_quant_mx: ntok_pad must be a multiple of 64 covering Ntok");
    S2V_REQUIRE(a.nq_w && a.nq_b && a.nk_w && a.nk_b && (a.cos == nullptr) == (a.sin == nullptr), "qk_norm_quant_mx: LayerNorm parameters / rotary tables");
    const int hg = a.H % 4 == 0 ? 4 : a.H % 2 == 0 ? 2 : 1;
    hipLaunchKernelGGL(qk_norm_quant_mx_k, dim3(a.ntok_pad / 64, a.H / hg, 2 * a.B), dim3(256), 0, st, a, q_prescale, q8, q8s, k8, k8s, hg);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_qk_norm_rope(const QkNormRopeArgs& a, int dtype, hipStream_t st) {
    dim3 grid((unsigned)(a.B * a.Ntok), (unsigned)((2 * a.H * 8 + 255) / 256));
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(qk_norm_rope_k<T>, grid, dim3(256), 0, st, a))
    S2V_CHECK_HIP(hipGetLastError());
    if (a.vt != nullptr) {
        S2V_REQUIRE(dtype == S2V_BF16 || (dtype == S2V_F16 && !a.vt_f16), "V^T is produced for 16-bit storage only (fp16 rows are moved as they are)");
        S2V_TRY(launch_v_transpose(a.qkv, a.ld_qkv, a.B, a.H, a.Ntok, a.vt, a.ntok_pad, st, a.vt_f16 != 0));
    }
    return 0;
}
int launch_v_transpose(const void* qkv, int ld_qkv, int B, int H, int Ntok, void* vt, int ntok_pad, hipStream_t st, bool to_f16) {
    dim3 g2((Ntok + 63) / 64, H, B);
    hipLaunchKernelGGL(v_transpose_k, g2, dim3(256), 0, st, (const bf16_t*)qkv, ld_qkv, B, H, Ntok, (bf16_t*)vt, ntok_pad, to_f16 ? 1 : 0);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// timestep sinusoid [cos | sin] (flip_sin_to_cos=True, shift 0), fp32 math then cast (embeddings.py:56-73)
template <typename T>
__global__ void timestep_sincos_k(const float* t, int B, int D, T* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, j = i - b * D;
    const int half = D / 2;
    const int f = j < half ? j : j - half;
    const float expo = -logf(10000.0f) * (float)f / (float)half;
    const float arg = t[b] * expf(expo);
    ET<T>::st(out + i, j < half ? cosf(arg) : sinf(arg));
}

// out[b][row] = rnd(sum_k f(in[b][k]) * W[row][k] + bias[row]); f = rnd(silu(.)) when PRE_SILU. One wave per row.
template <typename T, bool PRE_SILU>
__global__ __launch_bounds__(256) void gemv_rows_k(const T* in, int B, int K, const T* W, const T* bias, int64_t rows,
                                                   T* out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int VN = Vec16<T>::N;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const T* w = W + row * K;
    for (int k = lane * VN; k < K; k += 64 * VN) {
        float wv[VN];
        Vec16<T>::ld(w + k, wv);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b < B) {
                float xv[VN];
                Vec16<T>::ld(in + (size_t)b * K + k, xv);
#pragma unroll
                for (int e = 0; e < VN; ++e) {
                    float x = xv[e];
                    if (PRE_SILU) x = ET<T>::rnd(silu_f(x));
                    acc[b] = fmaf(x, wv[e], acc[b]);
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (b < B) {
            const float s = wave_sum(acc[b]);
            if (lane == 0) ET<T>::st(out + (size_t)b * rows + row, s + (bias ? ET<T>::ld(bias + row) : 0.f));
        }
    }
}
// The same sums in the same order (bit-identical), shaped for the weight stream: K = KC * 64 * (16-byte vector) exactly, so a lane
// owns KC fixed chunks of every row; it applies f to its slice of the B input vectors ONCE (the one-wave-per-row kernel re-evaluated
// silu for every row: 16 exp per 16 bytes of weights, which held the step's modulation GEMV at 1.1 TB/s) and then streams GEMV_R rows
// with all their loads in flight.
constexpr int GEMV_R = 8;
template <typename T, bool PRE_SILU, int KC, int BN>
__global__ __launch_bounds__(256) void gemv_rows_reg_k(const T* in, const T* W, const T* bias, int64_t rows, T* out) {
    constexpr int VN = Vec16<T>::N, K = KC * 64 * VN;
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GEMV_R;
    if (row0 >= rows) return;
    float x[BN][KC][VN];
#pragma unroll
    for (int b = 0; b < BN; ++b)
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            Vec16<T>::ld(in + (size_t)b * K + (c * 64 + lane) * VN, x[b][c]);
            if (PRE_SILU)
#pragma unroll
                for (int e = 0; e < VN; ++e) x[b][c][e] = ET<T>::rnd(silu_f(x[b][c][e]));
        }
    const int nr = (int)min((int64_t)GEMV_R, rows - row0);
    float wv[GEMV_R][KC][VN];
#pragma unroll
    for (int r = 0; r < GEMV_R; ++r) {
        const T* w = W + (row0 + min(r, nr - 1)) * K;
#pragma unroll
        for (int c = 0; c < KC; ++c) Vec16<T>::ld(w + (c * 64 + lane) * VN, wv[r][c]);
    }
#pragma unroll
    for (int r = 0; r < GEMV_R; ++r) {
        float acc[BN];
#pragma unroll
        for (int b = 0; b < BN; ++b) {
            acc[b] = 0.f;
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[b] = fmaf(x[b][c][e], wv[r][c][e], acc[b]);
            acc[b] = wave_sum(acc[b]);
        }
        if (lane == 0 && r < nr) {
            const float bv = bias ? ET<T>::ld(bias + row0 + r) : 0.f;
#pragma unroll
            for (int b = 0; b < BN; ++b) ET<T>::st(out + (size_t)b * rows + row0 + r, acc[b] + bv);
        }
    }
}
template <typename T, bool PRE_SILU, int KC>
static int gemv_rows_reg(const void* in, int B, const void* W, const void* bias, int64_t rows, void* out, hipStream_t st) {
    dim3 grid((unsigned)((rows + 4 * GEMV_R - 1) / (4 * GEMV_R)));
#define S2V_GEMV_B(BN)                                                                                                     \
    case BN:                                                                                                               \
        hipLaunchKernelGGL((gemv_rows_reg_k<T, PRE_SILU, KC, BN>), grid, dim3(256), 0, st, (const T*)in, (const T*)W,      \
                           (const T*)bias, rows, (T*)out);                                                                 \
        break;
    switch (B) {
        S2V_GEMV_B(1) S2V_GEMV_B(2) S2V_GEMV_B(3) S2V_GEMV_B(4)
        default: return s2v_fail(__FILE__, __LINE__, "gemv_rows: batch must be <= 4", -1);
    }
#undef S2V_GEMV_B
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

template <typename T>
static int gemv_rows(const void* in, int B, int K, const void* W, const void* bias, int64_t rows, void* out,
                     bool pre_silu, hipStream_t st, bool rowwise = false) {
    S2V_REQUIRE(B <= 4, "gemv_rows: batch must be <= 4");
    S2V_REQUIRE(K % Vec16<T>::N == 0, "gemv_rows: K must be a multiple of the 16-byte vector width");
    if (pre_silu && rows >= 4096 && !rowwise) {  // the modulation stack (K = time_embed_dim): one or two chunks per lane
        if (K == 64 * Vec16<T>::N) return gemv_rows_reg<T, true, 1>(in, B, W, bias, rows, out, st);
        if (K == 128 * Vec16<T>::N) return gemv_rows_reg<T, true, 2>(in, B, W, bias, rows, out, st);
    }
    dim3 grid((unsigned)((rows + 3) / 4));
    if (pre_silu)
        hipLaunchKernelGGL((gemv_rows_k<T, true>), grid, dim3(256), 0, st, (const T*)in, B, K, (const T*)W,
                           (const T*)bias, rows, (T*)out);
    else
        hipLaunchKernelGGL((gemv_rows_k<T, false>), grid, dim3(256), 0, st, (const T*)in, B, K, (const T*)W,
                           (const T*)bias, rows, (T*)out);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_time_embed(const float* t_dev, int B, int D, const void* w1, const void* b1, const void* w2, const void* b2,
                      int temb_dim, void* tmp, void* emb_out, int dtype, hipStream_t st) {
    // tmp holds [B, D] sinusoid followed by [B, temb] hidden
    const int n = B * D;
    S2V_DT_DISPATCH(dtype, {
        T* sc = (T*)tmp;
        T* h1 = sc + n;
        hipLaunchKernelGGL(timestep_sincos_k<T>, dim3((n + 255) / 256), dim3(256), 0, st, t_dev, B, D, sc);
        S2V_CHECK_HIP(hipGetLastError());
        S2V_TRY(gemv_rows<T>(sc, B, D, w1, b1, temb_dim, h1, false, st));
        S2V_TRY(gemv_rows<T>(h1, B, temb_dim, w2, b2, temb_dim, emb_out, true, st));
    })
    return 0;
}

int launch_mod_gemv(const void* emb, int B, int temb_dim, const void* W, const void* bias, int64_t rows_total,
                    void* out, int dtype, hipStream_t st, bool rowwise) {
    S2V_DT_DISPATCH(dtype, return gemv_rows<T>(emb, B, temb_dim, W, bias, rows_total, out, true, st, rowwise))
    return 0;
}

// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void patchify_k(const T* lat, int64_t lat_bstride, int Bn, int F, int C, int H, int W, T* out) {
    // out[(((b*F + f)*hp + y)*wp + x)][c*4 + py*2 + px] = lat[b][f][c][2y+py][2x+px]
    // (the im2col operand of the 2x2 stride-2 patch conv, embeddings.py:414-419)
    const int hp = H / 2, wp = W / 2, K = C * 4;
    const int64_t total = (int64_t)Bn * F * hp * wp * K;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % K);
    const int64_t tok = i / K;
    const int x = (int)(tok % wp);
    const int y = (int)((tok / wp) % hp);
    const int f = (int)((tok / ((int64_t)wp * hp)) % F);
    const int64_t b = tok / ((int64_t)wp * hp * F);
    const int c = k >> 2, py = (k >> 1) & 1, px = k & 1;
    out[i] = lat[b * lat_bstride + (((int64_t)f * C + c) * H + 2 * y + py) * W + 2 * x + px];
}
int launch_patchify(const void* lat, int64_t lat_bstride, int Bn, int F, int C, int H, int W, void* out, int dtype,
                    hipStream_t st) {
    const int64_t total = (int64_t)Bn * F * (H / 2) * (W / 2) * C * 4;
    dim3 grid((unsigned)((total + 255) / 256));
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(patchify_k<T>, grid, dim3(256), 0, st, (const T*)lat, lat_bstride, Bn, F, C, H, W, (T*)out))
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

template <typename T>
__global__ void unpatchify_k(const T* y, int ldy, int64_t y_bstride, T* out, int B, int F, int C, int H, int W) {
    // out[b][f][c][yy][xx] = y[b*y_bstride + (f*hp + yy/2)*wp + xx/2][c*4 + (yy&1)*2 + (xx&1)]
    // (cogvideox_transformer_3d.py:549-551)
    const int hp = H / 2, wp = W / 2;
    const int64_t total = (int64_t)B * F * C * H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int xx = (int)(i % W);
    const int yy = (int)((i / W) % H);
    const int c = (int)((i / ((int64_t)W * H)) % C);
    const int f = (int)((i / ((int64_t)W * H * C)) % F);
    const int b = (int)(i / ((int64_t)W * H * C * F));
    const int64_t row = (int64_t)b * y_bstride + ((int64_t)f * hp + yy / 2) * wp + xx / 2;
    out[i] = y[row * ldy + c * 4 + (yy & 1) * 2 + (xx & 1)];
}
int launch_unpatchify(const void* y, int ldy, int64_t y_bstride, void* out, int B, int F, int C, int H, int W, int dtype,
                      hipStream_t st) {
    const int64_t total = (int64_t)B * F * C * H * W;
    dim3 grid((unsigned)((total + 255) / 256));
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(unpatchify_k<T>, grid, dim3(256), 0, st, (const T*)y, ldy, y_bstride, (T*)out, B, F, C, H, W))
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

template <typename T>
__global__ void copy_rows_k(const T* src, int lds_, const T* add, int ldadd, T* dst, int ldd, int rows, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * D) return;
    const int r = (int)(i / D), c = (int)(i - (int64_t)r * D);
    float v = ET<T>::ld(src + (size_t)r * lds_ + c);
    if (add) v += ET<T>::ld(add + (size_t)r * ldadd + c);
    ET<T>::st(dst + (size_t)r * ldd + c, v);
}
int launch_copy_rows(const void* src, int lds_, const void* add, int ldadd, void* dst, int ldd, int rows, int D,
                     int dtype, hipStream_t st) {
    const int64_t total = (int64_t)rows * D;
    if (total == 0) return 0;
    dim3 grid((unsigned)((total + 255) / 256));
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(copy_rows_k<T>, grid, dim3(256), 0, st, (const T*)src, lds_, (const T*)add, ldadd, (T*)dst, ldd, rows, D))
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// CFG + scheduler step (custom_cogvideox_pipe.py:266-296; scheduling_ddim_cogvideox.py:364-394;
// scheduling_dpm_cogvideox.py:391-434).  Coefficients that multiply a model-dtype tensor (c_x0_x, a_t, m1, mn)
// arrive already rounded to the model dtype by the host: torch rounds a 0-dim fp64 scalar to the tensor dtype
// before the multiply (pinned by tests/golden/sched_*.npz).  __f*_rn keep the reference's separate roundings
// (no fma contraction), so the fp32 arithmetic is bit-identical to the CPU reference.
template <typename T>
__global__ void sched_step_k(const SchedArgs a) {
#pragma clang fp contract(off)  // hipcc defaults to -ffp-contract=fast and __fmul_rn is a plain multiply in HIP
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const SchedCoef k = a.coef ? *a.coef : a.cval;
    float v;
    if (a.np_f32) {
        const float* np = (const float*)a.noise_pred;
        v = a.cfg ? __fadd_rn(np[i], __fmul_rn(k.guidance, __fsub_rn(np[a.n + i], np[i]))) : np[i];
    } else {
        const T* np = (const T*)a.noise_pred;
        if (a.cfg) {
            const float u = ET<T>::ld(np + i), c = ET<T>::ld(np + a.n + i);
            v = __fadd_rn(u, __fmul_rn(k.guidance, __fsub_rn(c, u)));
        } else {
            v = ET<T>::ld(np + i);
        }
    }
    const float x = ET<T>::ld((const T*)a.latents_in + i);
    const float x0 = __fsub_rn(ET<T>::rnd(__fmul_rn(x, k.c_x0_x)), __fmul_rn(k.c_x0_v, v));
    float prev;
    if (k.kind == 0) {
        prev = __fadd_rn(ET<T>::rnd(__fmul_rn(k.a_t, x)), __fmul_rn(k.b_t, x0));
    } else {
        float d = x0;
        if (k.kind == 2) d = __fsub_rn(__fmul_rn(k.m3, x0), __fmul_rn(k.m4, a.x0_hist[i]));
        const float nz = ET<T>::ld((const T*)a.noise + i);
        prev = __fadd_rn(__fsub_rn(ET<T>::rnd(__fmul_rn(k.m1, x)), __fmul_rn(k.m2, d)), ET<T>::rnd(__fmul_rn(k.mn, nz)));
    }
    if (a.x0_hist) a.x0_hist[i] = x0;
    if (a.out_f32) ((float*)a.latents_out)[i] = prev;
    else ET<T>::st((T*)a.latents_out + i, prev);
}
int launch_sched_step(const SchedArgs& a, int dtype, hipStream_t st) {
    dim3 grid((unsigned)((a.n + 255) / 256));
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(sched_step_k<T>, grid, dim3(256), 0, st, a))
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ void convert2d_k(const TS* src, int64_t lds_, TD* dst, int64_t ldd, int64_t rows, int64_t cols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t r = i / cols, c = i - r * cols;
    ET<TD>::st(dst + r * ldd + c, ET<TS>::ld(src + r * lds_ + c));
}
int launch_convert2d(const void* src, int sdt, int64_t lds_, void* dst, int ddt, int64_t ldd, int64_t rows, int64_t cols,
                     hipStream_t st) {
    const int64_t total = rows * cols;
    if (total == 0) return 0;
    dim3 grid((unsigned)((total + 255) / 256));
#define CV(TS, TD) hipLaunchKernelGGL((convert2d_k<TS, TD>), grid, dim3(256), 0, st, (const TS*)src, lds_, (TD*)dst, ldd, rows, cols)
    S2V_REQUIRE(sdt >= 0 && sdt <= 2 && ddt >= 0 && ddt <= 2, "convert: unknown dtype");
    switch (sdt * 3 + ddt) {
        case S2V_F32 * 3 + S2V_F32: CV(float, float); break;
        case S2V_F32 * 3 + S2V_BF16: CV(float, bf16_t); break;
        case S2V_F32 * 3 + S2V_F16: CV(float, f16_t); break;
        case S2V_BF16 * 3 + S2V_F32: CV(bf16_t, float); break;
        case S2V_BF16 * 3 + S2V_BF16: CV(bf16_t, bf16_t); break;
        case S2V_BF16 * 3 + S2V_F16: CV(bf16_t, f16_t); break;
        case S2V_F16 * 3 + S2V_F32: CV(f16_t, float); break;
        case S2V_F16 * 3 + S2V_BF16: CV(f16_t, bf16_t); break;
        default: CV(f16_t, f16_t); break;
    }
#undef CV
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_convert(const void* src, int sdt, void* dst, int ddt, int64_t n, hipStream_t st) {
    return launch_convert2d(src, sdt, n, dst, ddt, n, 1, n, st);
}

// ---------------------------------------------------------------------------------------------------
// Dynamic per-row fp8 quantisation (W8A8, BASELINE configs[4]): one wave per row, two passes over a row that stays in L2 / registers:
// amax -> scale = amax / 448 (the largest e4m3 magnitude) -> q = rne_e4m3(x / scale).  Also quantises the weights at load time (a
// weight row = an output channel).  Rows are padded to the caller's tile by zero scales / zero bytes outside [0, M).
__global__ __launch_bounds__(256) void quant_rows_fp8_k(const bf16_t* src, int64_t ld, int64_t M, int K, unsigned char* dst, float* scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bf16_t* x = src + row * ld;
    float amax = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        const u32x4 v = *(const u32x4*)(x + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            amax = fmaxf(amax, fabsf(__uint_as_float(v[e] << 16)));
            amax = fmaxf(amax, fabsf(__uint_as_float(v[e] & 0xffff0000u)));
        }
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
    unsigned char* q = dst + row * (int64_t)K;
    for (int k = lane * 8; k < K; k += 512) {
        const u32x4 v = *(const u32x4*)(x + k);
        u32x2 o;
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(v[0] << 16) * inv, __uint_as_float(v[0] & 0xffff0000u) * inv, w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(v[1] << 16) * inv, __uint_as_float(v[1] & 0xffff0000u) * inv, w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(v[2] << 16) * inv, __uint_as_float(v[2] & 0xffff0000u) * inv, w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(v[3] << 16) * inv, __uint_as_float(v[3] & 0xffff0000u) * inv, w1, true);
        o.x = (unsigned)w0; o.y = (unsigned)w1;
        *(u32x2*)(q + k) = o;
    }
}
int launch_quant_rows_fp8(const void* src, int64_t ld, int64_t M, int K, void* dst, float* scale, hipStream_t st) {
    S2V_REQUIRE(K % 8 == 0 && ld % 8 == 0, "quant_rows_fp8: K and the row stride must be multiples of 8");
    if (M <= 0) return 0;
    hipLaunchKernelGGL(quant_rows_fp8_k, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, (const bf16_t*)src, ld, M, K, (unsigned char*)dst, scale);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}
