// Shared device/host helpers for the s2v HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define S2V_WAVE 64

// ---- scalar conversions -------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned int)h) << 16); }
// round-to-nearest-even, lowers to v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two floats -> one packed pair, ONE v_cvt_pk_bf16_f32 (the shift/or form costs three more VALU slots per pair)
__device__ __forceinline__ unsigned int pack2bf(float lo, float hi) {
    bf16x2_t v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned int, v);
}

// Element-type traits: T is the model's storage dtype (float or bf16_t).
template <typename T> struct ET;
template <> struct ET<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ float rnd(float v) { return v; }
};
template <> struct ET<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    // value after a round trip through the storage type (mimics the reference's per-op bf16 rounding)
    static __device__ __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }
};

// fp16 storage (round 5): the model dtype the reference selects for every non-5B checkpoint (src/inference.py:191,209).  A distinct C++ type
// (bf16_t is an integer typedef), converted by v_cvt_f16_f32 / v_cvt_f32_f16: round-to-nearest-even, overflow to infinity past 65504 as
// torch's conversion does.
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <> struct ET<f16_t> {
    static __device__ __forceinline__ float ld(const f16_t* p) { return (float)*p; }
    static __device__ __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)v; }
    static __device__ __forceinline__ float rnd(float v) { return (float)(f16_t)v; }
};
__device__ __forceinline__ unsigned int pack2h(float lo, float hi) {
    f16x2_t v;
    v[0] = (_Float16)lo;
    v[1] = (_Float16)hi;
    return __builtin_bit_cast(unsigned int, v);
}

// Wave-wide butterfly v (op) v[lane ^ 32], ^ 16, ^ 8, ^ 4, ^ 2, ^ 1 -- the operand order of the __shfl_xor loop this replaces, so sums
// keep their bits -- without the LDS crossbar: __shfl_xor lowers to ds_bpermute_b32 behind five VALU of index arithmetic and an
// s_waitcnt lgkmcnt(0) per step.  ^ 32 / ^ 16: v_permlane32_swap / v_permlane16_swap of two copies (gfx950); ^ 8: DPP row_ror:8;
// ^ 4: row_shl:4 into the banks whose lanes have bit 2 clear + row_shr:4 into the others; ^ 2 / ^ 1: quad_perm.
template <typename Op>
__device__ __forceinline__ float wave_butterfly(float v, Op op) {
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = op(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = op(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true)));  // row_ror:8
    {
        int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xf, 0x5, true);                  // row_shl:4 -> banks 0, 2
        t = __builtin_amdgcn_update_dpp(t, __float_as_int(v), 0x114, 0xf, 0xa, false);                     // row_shr:4 -> banks 1, 3
        v = op(v, __int_as_float(t));
    }
    v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true)));   // quad_perm [2,3,0,1]
    v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true)));   // quad_perm [1,0,3,2]
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    return wave_butterfly(v, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ float wave_max(float v) {
    return wave_butterfly(v, [](float a, float b) { return fmaxf(a, b); });
}

// 16-byte vector of T
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    typedef f32x4 raw_t;  // the 16 bytes as loaded; dec() turns them into N floats (loads issued early, decoded at their use)
    static __device__ __forceinline__ raw_t ldraw(const float* p) { return *(const f32x4*)p; }
    static __device__ __forceinline__ void dec(const raw_t& t, float* v) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ void ld(const float* p, float* v) {
        f32x4 t = *(const f32x4*)p;
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void st(float* p, const float* v) { *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]}; }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    typedef u32x4 raw_t;
    static __device__ __forceinline__ raw_t ldraw(const bf16_t* p) { return *(const u32x4*)p; }
    static __device__ __forceinline__ void dec(const raw_t& t, float* v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(t[i] << 16);
            v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void ld(const bf16_t* p, float* v) {
        u32x4 t = *(const u32x4*)p;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(t[i] << 16);
            v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float* v) {
        u32x4 t;
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = pack2bf(v[2 * i], v[2 * i + 1]);
        *(u32x4*)p = t;
    }
};

template <> struct Vec16<f16_t> {
    static constexpr int N = 8;
    typedef u32x4 raw_t;
    static __device__ __forceinline__ raw_t ldraw(const f16_t* p) { return *(const u32x4*)p; }
    static __device__ __forceinline__ void dec(const raw_t& t, float* v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned int w = t[i];  // a copy: __builtin_bit_cast of a vector ELEMENT expression reads element 0 whatever the index (clang)
            const f16x2_t h = __builtin_bit_cast(f16x2_t, w);
            v[2 * i] = (float)h[0];
            v[2 * i + 1] = (float)h[1];
        }
    }
    static __device__ __forceinline__ void ld(const f16_t* p, float* v) { dec(*(const u32x4*)p, v); }
    static __device__ __forceinline__ void st(f16_t* p, const float* v) {
        u32x4 t;
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = pack2h(v[2 * i], v[2 * i + 1]);
        *(u32x4*)p = t;
    }
};

__device__ __forceinline__ float gelu_tanh_f(float x) {
    // 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715 x^3)))  -- torch F.gelu(approximate="tanh")
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float u = k0 * (x + k1 * x * x * x);
    return 0.5f * x * (1.0f + tanhf(u));
}
// same function as x * sigmoid(2u): one v_exp_f32 + one v_rcp_f32 (relative error ~1e-6, far below a bf16 ulp);
// used by the bf16 MFMA epilogue where tanhf() cost ~10 % of the FF1 GEMM
// seven VALU: the exp2 argument -log2(e) * 2u = x * (k0 + k1 x^2) with the constants folded (the four-wave GEMM's epilogue is VALU-issue
// bound: one instruction per ~5 cycles and wave, 256 outputs per lane)
__device__ __forceinline__ float gelu_tanh_fast(float x) {
    constexpr float k0 = -1.4426950408889634f * 1.5957691216057308f, k1 = k0 * 0.044715f;
    const float p = fmaf(x * x, k1, k0);
    const float e = __builtin_amdgcn_exp2f(x * p);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
// x * sigmoid(x) on the hardware exp2 / rcp (relative error ~2e-7): for values that are rounded to bf16 next
__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
template <typename T> __device__ __forceinline__ float silu_t(float x) { return sizeof(T) == 2 ? silu_fast(x) : silu_f(x); }

// LDS-DMA of 16 B per lane in the saddr + 32-bit voffset form: global address = sbase (wave-uniform, SGPR pair) + voff
// (per-lane byte offset), LDS address = M0 + 16 * lane.  hipcc picks the 64-bit vaddr form inside loops; this pins the
// cheaper one (one address VGPR per lane instead of two).
__device__ __forceinline__ void glds16_saddr(const char* sbase, unsigned voff, char* lds_dst) {
    const unsigned m0v = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_dst;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0"
                 :
                 : "s"(sbase), "v"(voff), "s"(__builtin_amdgcn_readfirstlane(m0v))
                 : "memory", "m0");
}

// the same with the LDS destination given as a byte offset into the block's dynamic LDS (lds_off from lds_base_u32()): an
// address-space cast of a generic pointer costs a null check (s_cmp_lg_u64 + s_cselect) per call, integers do not
__device__ __forceinline__ unsigned lds_base_u32(char* smem) { return (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem; }
__device__ __forceinline__ void glds16_saddr_m0(const char* sbase, unsigned voff, unsigned m0val) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0"
                 :
                 : "s"(sbase), "v"(voff), "s"(m0val)
                 : "memory", "m0");
}

// one dword per lane: LDS address = M0 + 4 * lane (the MX block scales of 64 rows)
__device__ __forceinline__ void glds4_saddr_m0(const char* sbase, unsigned voff, unsigned m0val) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0"
                 :
                 : "s"(sbase), "v"(voff), "s"(m0val)
                 : "memory", "m0");
}

// the same with compile-time displacements: M0 = m0base + m0add (one SALU), global address + off (instruction immediate, 13-bit
// signed).  m0add / off must fold to constants after inlining and unrolling ("i" constraints)
__device__ __forceinline__ void glds16_saddr_m0_imm(const char* sbase, unsigned voff, unsigned m0base, int m0add, int off) {
    // the instruction offset displaces the LDS address as well as the global one (LDS = M0 + offset + lane * 16): taken back out of M0
    asm volatile("s_add_i32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 offset:%4"
                 :
                 : "s"(sbase), "v"(voff), "s"(m0base), "i"(m0add - off), "i"(off)
                 : "memory", "m0", "scc");
}

// Shader-clock stamp of a launch (bench.py's roofline.shader_clock_mhz): ONE designated workgroup writes s_memtime (shader-clock cycles)
// and s_memrealtime (constant 100 MHz) at its entry (which = 0) and exit (which = 1) into four int64 of a caller-owned slot -- both pairs
// from the same CU (the counters are per CU / XCD, not chip-synchronous: stamps from neighbouring launches disagree by millions of cycles).
// clk == nullptr (every launch but the profile pass): one scalar compare.
__device__ __forceinline__ void clk_stamp(long long* clk, unsigned designated_wg, int which) {
#ifdef S2V_NO_CLK_STAMP   // tools: a build without the stamps, for the same-box A/B that shows they cost nothing
    return;
#endif
    if (clk != nullptr && blockIdx.x == designated_wg && threadIdx.x == 0) {
        clk[2 * which] = (long long)__builtin_amdgcn_s_memtime();
        clk[2 * which + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    }
}

// ---- host side ------------------------------------------------------------
#ifdef S2V_HOST
#include <string>
extern thread_local std::string g_s2v_err;
int s2v_fail(const char* file, int line, const char* msg, int code);
#define S2V_CHECK_HIP(expr)                                                          \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) return s2v_fail(__FILE__, __LINE__, hipGetErrorString(_e), -2); \
    } while (0)
#define S2V_REQUIRE(cond, msg)                                                       \
    do {                                                                             \
        if (!(cond)) return s2v_fail(__FILE__, __LINE__, msg, -1);                   \
    } while (0)
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): the attribute is per device, and a process may
// drive several GPUs (transformer on one, VAE / T5 on another)
#include <mutex>
#include <set>
#include <utility>
static inline int ensure_lds_attr(const void* fn, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    S2V_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({fn, dev})) return 0;
    S2V_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({fn, dev});
    return 0;
}
#define S2V_TRY(expr)                                                                \
    do {                                                                             \
        int _r = (expr);                                                             \
        if (_r != 0) return _r;                                                      \
    } while (0)
#endif
