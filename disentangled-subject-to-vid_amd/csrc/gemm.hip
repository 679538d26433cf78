// GEMM kernels for the CogVideoX denoise path (replaces the nn.Linear call sites
// attention_processor.py:2049-2051,2090; attention.py:1241-1243; activations.py:87-90).
//
//  gemm_bf16_128   : bf16 MFMA (v_mfma_f32_32x32x16_bf16), 128x128x64 block tile, 4 waves (2x2, 64x64 each),
//                    operands staged HBM -> LDS with global_load_lds (16 B/lane), LDS image XOR-swizzled on the
//                    SOURCE address (chunk ^= (row>>1)&7) so every ds_read_b128 lane-group hits 16 distinct
//                    16-B slots; double buffered; XCD-aware + grouped tile order for L2 reuse.
//                    MFMA is issued "swapped" (first operand = weight rows) so each lane owns 4 consecutive
//                    output columns of one token row -> 8-byte stores and lane-local bias/gate epilogue.
//  gemm_simple<T>  : LDS-tiled VALU kernel for fp32 (the CPU-reference-parity mode) and odd shapes.
#define S2V_HOST
#include "common.h"
#include "kernels.h"

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
        const T* gate = (const T*)(r < a.text_len ? a.gate_txt : a.gate_vid) + (size_t)b * a.gate_stride;
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
        if (sizeof(T) == 2) {
            u32x2 p;
            p.x = pack2bf(y[0], y[1]);
            p.y = pack2bf(y[2], y[3]);
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

// ---------------------------------------------------------------------------------------------------
// bf16 MFMA kernel
#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (BM * BK * 2)  // 16 KiB per operand per stage

// element offset of A[m][0] (plain: m*lda; conv: the top-left-front input pixel of output pixel m)
__device__ __forceinline__ int64_t a_row_base(const GemmArgs& a, int m) {
    if (!a.conv) return (int64_t)m * a.lda;
    m = min(m, a.M - 1);
    const int hw = a.oH * a.oW;
    const int f = m / hw, rem = m - f * hw;
    const int y = rem / a.oW, x = rem - y * a.oW;
    return (((int64_t)f * a.Hp + y) * a.Wp + x) * a.cin;
}
// element offset added for column k (plain: k; conv: tap displacement + channel)
__device__ __forceinline__ int64_t a_k_off(const GemmArgs& a, int k) {
    if (!a.conv) return k;
    const int tap = k / a.cin, ci = k - tap * a.cin;
    const int dt = a.kt == 3 ? tap / 9 : 0;
    const int r9 = tap - dt * 9;
    const int dy = r9 / 3, dx = r9 - dy * 3;
    return (((int64_t)dt * a.Hp + dy) * a.Wp + dx) * a.cin + ci;
}

// A operand: per-thread row bases (fixed for the whole K loop) + a per-k-step displacement
__device__ __forceinline__ void stage_tile_a(const bf16_t* __restrict__ g, const int64_t rb[4], int64_t koff, char* lds,
                                             int tid) {
    const int wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gi = i * 256 + tid;
        const int row = gi >> 3;
        const int cp = gi & 7;
        const int c = cp ^ ((row >> 1) & 7);
        const bf16_t* src = g + rb[i] + koff + c * 8;
        char* dst = lds + (i * 256 + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, int ld, int row0, int k0, char* lds, int tid) {
    // 128 rows x 128 B; thread t of round i owns LDS chunk (i*256+t): row = g>>3, physical 16-B chunk = g&7.
    const int wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gi = i * 256 + tid;
        const int row = gi >> 3;
        const int cp = gi & 7;
        const int c = cp ^ ((row >> 1) & 7);
        const bf16_t* src = g + (size_t)(row0 + row) * ld + k0 + c * 8;
        char* dst = lds + (i * 256 + wave * 64) * 16;  // wave-uniform base; HW adds lane*16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 lds_frag(const char* tile, int row, int cl) {
    const int off = row * 128 + ((cl ^ ((row >> 1) & 7)) << 4);
    return *(const bf16x8*)(tile + off);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_128(const GemmArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap (block b runs on XCD b%8), then grouped (8 tile rows) order
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int GM = 8;
    const int per_group = GM * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int in_g = wg - group * per_group;
    const int tile_m = first_m + in_g % gsz;
    const int tile_n = in_g / gsz;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const bf16_t* A = (const bf16_t*)a.A;
    const bf16_t* W = (const bf16_t*)a.W;
    // stage s: A tile at s*2*TILE_BYTES, W tile right behind it

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nt = a.K / BK;
    int64_t rb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = a_row_base(a, m0 + ((i * 256 + tid) >> 3));
    stage_tile_a(A, rb, a_k_off(a, 0), smem, tid);
    stage_tile(W, a.ldw, n0, 0, smem + TILE_BYTES, tid);
    __syncthreads();

    const int fr = lane & 31, hi = lane >> 5;
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            stage_tile_a(A, rb, a_k_off(a, (t + 1) * BK), smem + (cur ^ 1) * 2 * TILE_BYTES, tid);
            stage_tile(W, a.ldw, n0, (t + 1) * BK, smem + (cur ^ 1) * 2 * TILE_BYTES + TILE_BYTES, tid);
        }
        const char* tA = smem + cur * 2 * TILE_BYTES;
        const char* tW = tA + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 wf[2], af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) wf[i] = lds_frag(tW, wn * 64 + i * 32 + fr, kk * 2 + hi);
#pragma unroll
            for (int j = 0; j < 2; ++j) af[j] = lds_frag(tA, wm * 64 + j * 32 + fr, kk * 2 + hi);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: D[i = n][j = m]; lane: m = fr, n = (reg&3) + 8*(reg>>2) + 4*hi
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + wm * 64 + j * 32 + fr;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int n = n0 + wn * 64 + i * 32 + 8 * rq + 4 * hi;
                float v[4] = {acc[i][j][rq * 4 + 0], acc[i][j][rq * 4 + 1], acc[i][j][rq * 4 + 2], acc[i][j][rq * 4 + 3]};
                if (n < a.N) epilogue4<bf16_t, EPI>(a, m, n, v);
            }
        }
}

int launch_gemm_bf16(const GemmArgs& a, int epi, hipStream_t st) {
    S2V_REQUIRE(a.K % BK == 0, "gemm_bf16: K must be a multiple of 64");
    S2V_REQUIRE((a.conv ? a.cin % 64 == 0 : a.lda % 8 == 0) && a.ldw % 8 == 0, "gemm_bf16: bad leading dims");
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int grid = tiles_m * tiles_n;
    const size_t shmem = 4 * TILE_BYTES;
    switch (epi) {
        case EPI_BIAS:
            hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS>, dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            break;
        case EPI_BIAS_GELU:
            hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS_GELU>, dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            break;
        case EPI_BIAS_GATE_RES:
            hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS_GATE_RES>, dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            break;
        case EPI_BIAS_ADD:
            hipLaunchKernelGGL(gemm_bf16_128<EPI_BIAS_ADD>, dim3(grid), dim3(256), shmem, st, a, tiles_m, tiles_n);
            break;
        default:
            return s2v_fail(__FILE__, __LINE__, "gemm_bf16: bad epilogue", -1);
    }
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// simple tiled kernel: 64x64 tile, BK 16, 256 threads, 4x4 outputs / thread (n contiguous)
template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm_simple_k(const GemmArgs a) {
    __shared__ float sA[16][64 + 4];
    __shared__ float sW[16][64 + 4];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tm = (tid >> 4) * 4, tn = (tid & 15) * 4;
    const T* A = (const T*)a.A;
    const T* W = (const T*)a.W;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < a.K; k0 += 16) {
        // 64 rows x 16 k per operand = 1024 elements, 4 per thread
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * 256 + tid;
            const int row = e >> 4, k = e & 15;
            float va = 0.f, vw = 0.f;
            if (k0 + k < a.K) {
                if (m0 + row < a.M) va = ET<T>::ld(A + a_row_base(a, m0 + row) + a_k_off(a, k0 + k));
                if (n0 + row < a.N) vw = ET<T>::ld(W + (size_t)(n0 + row) * a.ldw + k0 + k);
            }
            sA[k][row] = va;
            sW[k][row] = vw;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float av[4], wv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = sA[k][tm + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = sW[k][tn + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], wv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (n0 + tn < a.N) epilogue4<T, EPI>(a, m0 + tm + i, n0 + tn, acc[i]);
    }
}

template <typename T>
static int launch_simple_t(const GemmArgs& a, int epi, hipStream_t st) {
    dim3 grid((a.N + 63) / 64, (a.M + 63) / 64);
    switch (epi) {
        case EPI_BIAS: hipLaunchKernelGGL((gemm_simple_k<T, EPI_BIAS>), grid, dim3(256), 0, st, a); break;
        case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm_simple_k<T, EPI_BIAS_GELU>), grid, dim3(256), 0, st, a); break;
        case EPI_BIAS_GATE_RES:
            hipLaunchKernelGGL((gemm_simple_k<T, EPI_BIAS_GATE_RES>), grid, dim3(256), 0, st, a);
            break;
        case EPI_BIAS_ADD: hipLaunchKernelGGL((gemm_simple_k<T, EPI_BIAS_ADD>), grid, dim3(256), 0, st, a); break;
        default: return s2v_fail(__FILE__, __LINE__, "gemm_simple: bad epilogue", -1);
    }
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_simple(const GemmArgs& a, int epi, int dtype, hipStream_t st) {
    S2V_REQUIRE((a.M + 63) / 64 <= 65535, "gemm_simple: M too large for the generic kernel");
    return dtype == S2V_BF16 ? launch_simple_t<bf16_t>(a, epi, st) : launch_simple_t<float>(a, epi, st);
}

// generic strided fp32 GEMM-accumulate (load-time only; LoRA merge W += alpha * B.A)
__global__ void gemm_strided_f32_k(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk,
                                   float* C, int64_t ldc, int M, int N, int K, float alpha) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = blockIdx.y;
    if (n >= N || m >= M) return;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(A[m * sam + k * sak], B[n * sbn + k * sbk], acc);
    C[m * ldc + n] += alpha * acc;
}
int launch_gemm_strided_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk,
                            float* C, int64_t ldc, int M, int N, int K, float alpha, hipStream_t st) {
    dim3 grid((N + 255) / 256, M);
    hipLaunchKernelGGL(gemm_strided_f32_k, grid, dim3(256), 0, st, A, sam, sak, B, sbn, sbk, C, ldc, M, N, K, alpha);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}
