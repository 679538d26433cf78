// fp32 attention on the matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32, an fmaf chain in k order; 64 cycles per instruction and
// SIMD).  The fp32 model dtype is the CPU-reference-parity mode (<= 1e-3 on the latents); attn_simple_k
// (attention.hip) computes it on the VALU with one wave per query row -- minutes per step at 19 126 tokens.  This kernel keeps its
// arithmetic (q pre-multiplied by the scale, scores as an fmaf chain over the head dimension in ascending order, online softmax
// with expf, fp32 P and V) and runs both products as MFMAs, so a C3 step takes seconds and an fp32 run can serve as the on-GPU
// reference of a whole denoise loop (tools/whole_run_parity.py).  softmax(q k^T / 8) v, no mask: attention_processor.py:2083-2087.
//
// One workgroup = 4 waves x 32 query rows of one (sample, head); KV tiles of 32 keys, double-buffered in LDS:
//   S^T[key][q] = sum_d K[key][d] Qs[q][d]   32 MFMAs: A = K^T image [d][key] (lanes 0-31: d = 2s, lanes 32-63: d = 2s + 1),
//                                             B = the wave's Q rows, held in 32 registers for the whole launch
//   a lane owns ONE query column (q = lane & 31) and 16 of the tile's 32 keys (the other 16 sit in lane ^ 32): the tile maximum is
//   15 max + one permlane32 swap; P^T stays in the accumulator registers
//   O^T[d][q] += sum_key V[key][d] P^T[key][q]   2 x 16 MFMAs: B = P^T register e AS IT IS (its two lane halves hold the keys
//                                             k(e) and k(e) + 4, k(e) = (e & 3) + 8 (e >> 2)), A = rows k(e), k(e) + 4 of the V tile
// so no value ever crosses lanes except the maximum and the final row sum.
#define S2V_HOST
#include "common.h"
#include "kernels.h"
#include <type_traits>

#define AQ_ROWS 128   // query rows per workgroup
#define AK_TILE 32    // keys per KV tile
#define AKT_PITCH 40  // floats per d-row of the K^T image (32 keys + 8: the two d-rows of a fragment read differ by 40 = 8 mod 32 banks)
#define AV_PITCH 72   // floats per key row of the V image (rows k and k + 4 of a fragment read: 288 = 32 mod 64 banks)

// T = float, or f16_t: fp16 storage (the fp16 ENGINE runs launch_attn_f16, the fp16-MFMA kernel of attention.hip; this form stays as the
// higher-accuracy operator-level reference, s2v_op_attention impl 5).  fp16 values convert exactly to fp32 and so do their products, so
// QK^T is what an fp16 MFMA with fp32 accumulation would return; the probabilities are rounded to fp16 before P.V (the row sum is taken
// before that rounding) -- the points at which torch's CPU flash kernel rounds for a reduced-precision dtype -- and the output is stored
// as fp16.
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_f32m_k(const AttnArgs a) {
    __shared__ float sKT[2][64][AKT_PITCH];
    __shared__ float sV[2][AK_TILE][AV_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int D = a.H * 64;
    const T* base = (const T*)a.qkv + (size_t)b * a.Ntok * a.ld_qkv;
    const int q = blockIdx.x * AQ_ROWS + wave * 32 + fr;
    const int ql = min(q, a.Ntok - 1);
    // B operand of the QK^T MFMA number s: Qs[q][2 s + hi]
    float qreg[32];
    {
        const T* qp = base + (size_t)ql * a.ld_qkv + h * 64;
#pragma unroll
        for (int s = 0; s < 32; ++s) qreg[s] = ET<T>::ld(qp + 2 * s + hi) * a.scale;
    }
    // staging of a KV tile: 32 keys x 64 floats of K and of V = 512 + 512 chunks of 16 bytes, two of each per thread
    const int skey = tid >> 3, sc = tid & 7;  // key 0..31, chunks sc and sc + 8 of its 16 (fp32: 4 floats = 16 bytes; fp16: 4 halves = 8 bytes)
    f32x4 rk[2], rv[2];
    auto ld4 = [](const T* p) -> f32x4 {
        if constexpr (std::is_same<T, float>::value) {
            return *(const f32x4*)p;
        } else {
            const u32x2 t = *(const u32x2*)p;
            const unsigned w0 = t.x, w1 = t.y;  // copies: __builtin_bit_cast of a vector element expression reads element 0 (clang)
            const f16x2_t lo = __builtin_bit_cast(f16x2_t, w0), hi2 = __builtin_bit_cast(f16x2_t, w1);
            return f32x4{(float)lo[0], (float)lo[1], (float)hi2[0], (float)hi2[1]};
        }
    };
    auto gload = [&](int kv0) {
        const int key = min(kv0 + skey, a.Ntok - 1);
        const T* kp = base + (size_t)key * a.ld_qkv + D + h * 64;
        const T* vp = kp + D;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            rk[j] = ld4(kp + (sc + 8 * j) * 4);
            rv[j] = ld4(vp + (sc + 8 * j) * 4);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int d0 = (sc + 8 * j) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) sKT[buf][d0 + e][skey] = rk[j][e];
            *(f32x4*)&sV[buf][skey][d0] = rv[j];
        }
    };
    f32x16 o0, o1;  // O^T blocks d = 0..31 / 32..63: register e of a lane = head dim (e & 3) + 8 (e >> 2) + 4 hi, column q = fr
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
    float m = -INFINITY, l = 0.f;  // running maximum of the query row; this lane's share of the row sum (its 16 keys per tile)

    const int ntile = (a.Ntok + AK_TILE - 1) / AK_TILE;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1, kv0 = t * AK_TILE;
        if (t + 1 < ntile) gload(kv0 + AK_TILE);
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s = __builtin_amdgcn_mfma_f32_32x32x2f32(sKT[buf][2 * i + hi][fr], qreg[i], s, 0, 0, 0);
        // register e: key kv0 + (e & 3) + 8 (e >> 2) + 4 hi
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = kv0 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            if (key >= a.Ntok) s[e] = -INFINITY;
            mx = fmaxf(mx, s[e]);
        }
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float mn = fmaxf(m, mx);
        const float alpha = expf(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = expf(s[e] - mn);
            ps += s[e];
            s[e] = ET<T>::rnd(s[e]);  // fp16 dtype: P.V takes fp16 probabilities
        }
        l = l * alpha + ps;
#pragma unroll
        for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int kr = (e & 3) + 8 * (e >> 2) + 4 * hi;
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[buf][kr][fr], s[e], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[buf][kr][32 + fr], s[e], o1, 0, 0, 0);
        }
        if (t + 1 < ntile) lstore(buf ^ 1);
        __syncthreads();
    }
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
        l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    if (q < a.Ntok) {
        T* op = (T*)a.out + (size_t)(b * a.Ntok + q) * a.ld_out + h * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 8 * g + 4 * hi;
            if constexpr (std::is_same<T, float>::value) {
                *(f32x4*)(op + d) = f32x4{o0[4 * g] / l, o0[4 * g + 1] / l, o0[4 * g + 2] / l, o0[4 * g + 3] / l};
                *(f32x4*)(op + 32 + d) = f32x4{o1[4 * g] / l, o1[4 * g + 1] / l, o1[4 * g + 2] / l, o1[4 * g + 3] / l};
            } else {
                *(u32x2*)(op + d) = u32x2{pack2h(o0[4 * g] / l, o0[4 * g + 1] / l), pack2h(o0[4 * g + 2] / l, o0[4 * g + 3] / l)};
                *(u32x2*)(op + 32 + d) = u32x2{pack2h(o1[4 * g] / l, o1[4 * g + 1] / l), pack2h(o1[4 * g + 2] / l, o1[4 * g + 3] / l)};
            }
        }
    }
}

int launch_attn_f32m(const AttnArgs& a, int dtype, hipStream_t st) {
    S2V_REQUIRE(dtype == S2V_F32 || dtype == S2V_F16, "attn_f32m: fp32 or fp16 storage");
    S2V_REQUIRE(a.ld_qkv % 4 == 0 && a.ld_out % 4 == 0 && ((uintptr_t)a.qkv & 15) == 0 && ((uintptr_t)a.out & 15) == 0,
                "attn_f32m: qkv / out must be 16-byte aligned with leading dimensions that are multiples of 4 elements");
    S2V_REQUIRE(a.Ntok > 0 && a.H <= 65535 && a.B <= 65535, "attn_f32m: bad shape");
    const dim3 grid((a.Ntok + AQ_ROWS - 1) / AQ_ROWS, a.H, a.B);
    if (dtype == S2V_F16) hipLaunchKernelGGL(attn_f32m_k<f16_t>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(attn_f32m_k<float>, grid, dim3(256), 0, st, a);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}
