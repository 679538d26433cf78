// HBM-bound kernels of the CogVideoX 3-D causal VAE decoder (autoencoder_kl_cogvideox.py).  Activations are
// channels-last: dense [F][H][W][C] between layers, and zero-bordered [frames][H+2][W+2][C] ("padded") as the
// operand of every 3x3(x3) convolution, so the implicit-GEMM conv (gemm.hip, conv addressing) needs no bounds
// checks and a 64-channel k-step is one contiguous 128-byte line.  Two leading frames of a causal conv's padded
// buffer hold its conv_cache (CogVideoXCausalConv3d.forward :128-137).
//   latent_to_zq      decode_latents' 1/scaling_factor * z, [1,F,C,h,w] -> dense [F][th][tw][C] (tile window)
//   dense_to_padded   interior copy (conv_in operand)
//   gn_stats          GroupNorm statistics over (frames, pixels, channels of the group) of one frame batch
//   snorm_apply       SpatialNorm3D (:167-188): GN(f) * conv_y(zq^) + conv_b(zq^), optional SiLU, written padded
//   upsample          CogVideoXUpsample3D's nearest x2 (+ time, first-frame rule) (upsampling.py:384-404), padded
//   to_ncfhw / blend  output layout + the tile cross-fades of tiled_decode (:1284-1298,1437-1447)
//   postprocess       VideoProcessor.postprocess_video 'np' (video_processor.py:99-113): clamp(x/2+.5) -> [F,H,W,3]
#define S2V_HOST
#include "common.h"
#include "kernels.h"
#include <algorithm>
#include <type_traits>
#include "vae_kernels.h"

template <typename T> struct V16;
template <> struct V16<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void ld(const float* p, float* v) { f32x4 t = *(const f32x4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ void st(float* p, const float* v) { *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]}; }
};
template <> struct V16<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void ld(const bf16_t* p, float* v) {
        u32x4 t = *(const u32x4*)p;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(t[i] << 16); v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float* v) {
        u32x4 t;
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = pack2bf(v[2 * i], v[2 * i + 1]);
        *(u32x4*)p = t;
    }
};

template <> struct V16<f16_t> {  // the fp16 model dtype (round 5): the same 16-byte vector of eight values
    static constexpr int N = 8;
    static __device__ __forceinline__ void ld(const f16_t* p, float* v) { Vec16<f16_t>::ld(p, v); }
    static __device__ __forceinline__ void st(f16_t* p, const float* v) { Vec16<f16_t>::st(p, v); }
};

static inline unsigned grid_for(int64_t n, int cap = 1 << 20) {
    int64_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void latent_to_zq_k(const T* lat, int F, int C, int h, int w, float inv_sf, T* out, int y0, int x0, int th,
                               int tw) {
    const int64_t total = (int64_t)F * th * tw * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % tw);
        const int y = (int)((i / ((int64_t)C * tw)) % th);
        const int f = (int)(i / ((int64_t)C * tw * th));
        const float v = ET<T>::ld(lat + (((int64_t)f * C + c) * h + y0 + y) * w + x0 + x);
        ET<T>::st(out + i, inv_sf * v);
    }
}
int launch_latent_to_zq(const void* lat, int F, int C, int h, int w, float inv_sf, void* out, int y0, int x0, int th,
                        int tw, int dtype, hipStream_t st) {
    const int64_t total = (int64_t)F * th * tw * C;
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(latent_to_zq_k<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)lat, F, C, h, w,
                       inv_sf, (T*)out, y0, x0, th, tw))
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

template <typename T>
__global__ void dense_to_padded_k(const T* in, int F, int H, int W, int C, T* out, int f_off) {
    const int64_t total = (int64_t)F * H * W * C;
    const int Hp = H + 2, Wp = W + 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % W);
        const int y = (int)((i / ((int64_t)C * W)) % H);
        const int f = (int)(i / ((int64_t)C * W * H));
        out[((((int64_t)(f + f_off)) * Hp + y + 1) * Wp + x + 1) * C + c] = in[i];
    }
}
int launch_dense_to_padded(const void* in, int F, int H, int W, int C, void* out, int f_off, int dtype, hipStream_t st) {
    const int64_t total = (int64_t)F * H * W * C;
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(dense_to_padded_k<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)in, F, H, W,
                       C, (T*)out, f_off))
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// the zero ring of a padded operand [F][Hp][Wp][C] (the implicit-GEMM convolution's padding = 1): a new window size moves the ring into
// what was interior, and only the ring has to be cleared -- every interior cell is rewritten by the operand's producer before a
// convolution reads it.  (Clearing whole buffers on every change of the tile size cost the tiled decode 35-40 ms of memsets.)
// Units of two bytes (U per cell) so that any channel count of either dtype is covered.
__global__ void zero_border_k(unsigned short* pad, int F, int Hp, int Wp, int U) {
    const int ring = 2 * Wp + 2 * (Hp - 2);
    const int64_t total = (int64_t)F * ring * U;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int u = (int)(i % U);
        const int r = (int)((i / U) % ring);
        const int f = (int)(i / ((int64_t)U * ring));
        int y, x;
        if (r < Wp) { y = 0; x = r; }
        else if (r < 2 * Wp) { y = Hp - 1; x = r - Wp; }
        else { const int k = r - 2 * Wp; y = 1 + (k >> 1); x = (k & 1) ? Wp - 1 : 0; }
        pad[(((int64_t)f * Hp + y) * Wp + x) * U + u] = 0;
    }
}
int launch_zero_border(void* pad, int F, int H, int W, int C, int esz, hipStream_t st) {
    const int Hp = H + 2, Wp = W + 2, U = C * esz / 2;
    const int64_t total = (int64_t)F * (2 * Wp + 2 * (Hp - 2)) * U;
    hipLaunchKernelGGL(zero_border_k, dim3(grid_for(total)), dim3(256), 0, st, (unsigned short*)pad, F, Hp, Wp, U);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// window (y0, x0, th, tw) of an image [C][F][H][W] -> zero-bordered channels-last operand [f_off + F][th+2][tw+2][C]
template <typename T>
__global__ void image_to_padded_k(const T* img, int C, int F, int H, int W, int y0, int x0, int th, int tw, T* out, int f_off) {
    const int64_t total = (int64_t)F * th * tw * C;
    const int Hp = th + 2, Wp = tw + 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % tw);
        const int y = (int)((i / ((int64_t)C * tw)) % th);
        const int f = (int)(i / ((int64_t)C * tw * th));
        out[((((int64_t)(f + f_off)) * Hp + y + 1) * Wp + x + 1) * C + c] = img[(((int64_t)c * F + f) * H + y0 + y) * W + x0 + x];
    }
}
int launch_image_to_padded(const void* img, int C, int F, int H, int W, int y0, int x0, int th, int tw, void* out, int f_off,
                           int dtype, hipStream_t st) {
    const int64_t total = (int64_t)F * th * tw * C;
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(image_to_padded_k<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)img, C, F, H, W, y0, x0,
                       th, tw, (T*)out, f_off))
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// DiagonalGaussianDistribution(moments).sample() with the caller's noise (autoencoders/vae.py:767-790): moments [2*Cz][n],
// mean = first half, logvar = clamp(second half, -30, 20), out = mean + exp(0.5 * logvar) * noise; every op rounds to T
template <typename T>
__global__ void gaussian_sample_k(const T* mom, const T* noise, int64_t n, T* out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float mean = ET<T>::ld(mom + i);
        const float lv = fminf(fmaxf(ET<T>::ld(mom + n + i), -30.0f), 20.0f);
        const float sd = ET<T>::rnd(expf(ET<T>::rnd(0.5f * lv)));
        ET<T>::st(out + i, ET<T>::rnd(mean + ET<T>::rnd(sd * ET<T>::ld(noise + i))));
    }
}
int launch_gaussian_sample(const void* mom, const void* noise, int64_t n, void* out, int dtype, hipStream_t st) {
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(gaussian_sample_k<T>, dim3(grid_for(n)), dim3(256), 0, st, (const T*)mom, (const T*)noise, n, (T*)out))
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm statistics: sums[g] = {sum x, sum x^2} in fp64 over P pixels x (C/G) channels
#define GN_PIX_PER_BLOCK 512
// Deterministic (no atomics): per-thread partials -> LDS [pixel lane][channel] -> per-channel sums in lane order ->
// per-group fp64 block partials part[block][g][2]; gn_finalize_k adds them with a fixed 256-way split + tree (no atomics).
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_k(const T* x, int64_t P, int C, int G, double* part) {
    constexpr int VN = V16<T>::N;
    __shared__ float s_sum[256 * VN], s_sq[256 * VN];  // [pl][c] with ppi * C == 256 * VN
    __shared__ float c_sum[1024], c_sq[1024];
    const int tid = threadIdx.x;
    const int vpp = C / VN;              // 16-byte vectors per pixel
    const int ppi = 256 / vpp;           // pixels per iteration
    const int vi = tid % vpp, pl = tid / vpp;
    float a[VN], q[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) { a[e] = 0.f; q[e] = 0.f; }
    const int64_t p0 = (int64_t)blockIdx.x * GN_PIX_PER_BLOCK;
    const int64_t p1 = min(p0 + GN_PIX_PER_BLOCK, P);
#pragma unroll 4
    for (int64_t p = p0 + pl; p < p1; p += ppi) {  // four loads in flight, sums in the same order
        float v[VN];
        V16<T>::ld(x + p * C + vi * VN, v);
#pragma unroll
        for (int e = 0; e < VN; ++e) { a[e] += v[e]; q[e] += v[e] * v[e]; }
    }
#pragma unroll
    for (int e = 0; e < VN; ++e) { s_sum[pl * C + vi * VN + e] = a[e]; s_sq[pl * C + vi * VN + e] = q[e]; }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = 0.f, s2 = 0.f;
        for (int l = 0; l < ppi; ++l) { s += s_sum[l * C + c]; s2 += s_sq[l * C + c]; }
        c_sum[c] = s; c_sq[c] = s2;
    }
    __syncthreads();
    const int cpg = C / G;
    for (int g = tid; g < G; g += 256) {
        double s = 0.0, s2 = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { s += (double)c_sum[c]; s2 += (double)c_sq[c]; }
        part[((int64_t)blockIdx.x * G + g) * 2] = s;
        part[((int64_t)blockIdx.x * G + g) * 2 + 1] = s2;
    }
}
__global__ __launch_bounds__(256) void gn_finalize_k(const double* part, int nblocks, int G, double* sums) {
    // one block per (group, sum | sum of squares): 256 threads add strided subsets of the block partials in index order,
    // then a fixed-shape tree in LDS -- deterministic (no atomics) and 256-wide instead of one thread walking every partial
    __shared__ double red[256];
    const int i = blockIdx.x, t = threadIdx.x;
    double s = 0.0;
    for (int b = t; b < nblocks; b += 256) s += part[(size_t)b * 2 * G + i];
    red[t] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
    }
    if (t == 0) sums[i] = red[0];
}
int64_t gn_stats_scratch_bytes(int64_t P, int G) {
    return ((P + GN_PIX_PER_BLOCK - 1) / GN_PIX_PER_BLOCK) * G * 2 * (int64_t)sizeof(double);
}
int launch_gn_stats(const void* x, int64_t P, int C, int G, double* sums, double* part, int dtype, hipStream_t st) {
    const int VN = dtype == S2V_F32 ? 4 : 8;
    S2V_REQUIRE(C % VN == 0 && C <= 1024 && 256 % (C / VN) == 0 && C % G == 0, "gn_stats: unsupported channel count");
    S2V_REQUIRE(part != nullptr, "gn_stats: scratch missing");
    const unsigned grid = (unsigned)((P + GN_PIX_PER_BLOCK - 1) / GN_PIX_PER_BLOCK);
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(gn_stats_k<T>, dim3(grid), dim3(256), 0, st, (const T*)x, P, C, G, part))
    hipLaunchKernelGGL(gn_finalize_k, dim3(2 * G), dim3(256), 0, st, part, (int)grid, G, sums);
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// nearest-neighbour source frame of torch.nn.functional.interpolate under SpatialNorm3D's first-frame rule (:177-185)
__device__ __forceinline__ int zq_frame(int f, int Ff, int Fz) {
    if (Ff > 1 && (Ff & 1)) {
        if (f == 0) return 0;
        return 1 + (int)(((int64_t)(f - 1) * (Fz - 1)) / (Ff - 1));
    }
    return (int)(((int64_t)f * Fz) / Ff);
}

// out_padded[f + f_off][y+1][x+1][c] = act( GN(x)[c] * (Wy zq^ + by)[c] + (Wb zq^ + bb)[c] )
template <typename T>
__global__ __launch_bounds__(256) void snorm_apply_k(const SNormArgs a) {
    constexpr int VN = V16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // mean[G], rstd[G]
    const int C = a.C, G = a.G, cpg = C / G;
    float* mean = lds;
    float* rstd = mean + G;
    const double cnt = (double)a.F * a.H * a.W * cpg;
    for (int g = threadIdx.x; g < G; g += 256) {
        const double m = a.sums[2 * g] / cnt;
        const double var = a.sums[2 * g + 1] / cnt - m * m;
        mean[g] = (float)m;
        rstd[g] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)a.eps));
    }
    __syncthreads();
    const int vpp = C / VN;
    const int64_t P = (int64_t)a.F * a.H * a.W;
    const int64_t total = P * vpp;
    const int Hp = a.H + 2, Wp = a.W + 2;
    const T* x = (const T*)a.x;
    const T* zq = (const T*)a.zq;
    const T* gw = (const T*)a.gn_w;
    const T* gb = (const T*)a.gn_b;
    T* out = (T*)a.out;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int vi = (int)(i % vpp);
        const int64_t p = i / vpp;
        const int xx = (int)(p % a.W);
        const int yy = (int)((p / a.W) % a.H);
        const int f = (int)(p / ((int64_t)a.W * a.H));
        const int c0 = vi * VN;
        float v[VN], g_w[VN], g_b[VN];
        V16<T>::ld(x + p * C + c0, v);
        V16<T>::ld(gw + c0, g_w);
        V16<T>::ld(gb + c0, g_b);
        const int fz = zq_frame(f, a.F, a.Fz);
        const int yz = (int)(((int64_t)yy * a.hz) / a.H), xz = (int)(((int64_t)xx * a.wz) / a.W);
        // conv_y / conv_b are pointwise, so conv(nearest-upsampled zq) == nearest-upsampled conv(zq): gather the
        // latent-resolution tables built by snorm_tables_k (L2-resident, each row reused by its whole footprint)
        const int64_t pz = (((int64_t)fz * a.hz + yz) * a.wz + xz) * C + c0;
        float cy[VN], cb[VN];
        if (a.yt != nullptr) {
            V16<T>::ld((const T*)a.yt + pz, cy);
            V16<T>::ld((const T*)a.bt + pz, cb);
        }
        float o[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const int g = (c0 + e) / cpg;
            const float n = ET<T>::rnd((v[e] - mean[g]) * rstd[g] * g_w[e] + g_b[e]);
            float r = n;  // plain nn.GroupNorm (encoder resnets, yt == nullptr)
            if (a.yt != nullptr) r = ET<T>::rnd(ET<T>::rnd(n * cy[e]) + cb[e]);
            if (a.silu) r = ET<T>::rnd(silu_t<T>(r));
            o[e] = r;
        }
        V16<T>::st(out + ((((int64_t)(f + a.f_off)) * Hp + yy + 1) * Wp + xx + 1) * C + c0, o);
    }
}
// The same operator for channel counts whose 16-byte vectors per pixel divide 256 (every width of the CogVideoX VAE): a block walks
// image ROWS, so the frame / row arithmetic (and the latent row of the tables) is wave-uniform, and a thread keeps ONE channel vector
// for all its pixels: GroupNorm weight / bias, group mean / rstd are loaded once per thread instead of once per element, and no
// 64-bit division is left in the loop (snorm_apply_k spends most of its instructions on index arithmetic: 5 int64 divisions per
// vector and one int division per channel).  Same arithmetic and rounding points.
template <typename T>
__global__ __launch_bounds__(256) void snorm_apply_rows_k(const SNormArgs a) {
    constexpr int VN = V16<T>::N;
    const int C = a.C, G = a.G, cpg = C / G;
    const int vpp = C / VN;  // 256 % vpp == 0
    const int tid = threadIdx.x;
    const int vi = tid % vpp, pl = tid / vpp, ppi = 256 / vpp;  // this thread's channel vector, its pixel lane, pixels per pass
    const int c0 = vi * VN;
    const double cnt = (double)a.F * a.H * a.W * cpg;
    float g_w[VN], g_b[VN], mu[VN], rs[VN];
    V16<T>::ld((const T*)a.gn_w + c0, g_w);
    V16<T>::ld((const T*)a.gn_b + c0, g_b);
#pragma unroll
    for (int e = 0; e < VN; ++e) {
        const int g = (c0 + e) / cpg;
        const double m = a.sums[2 * g] / cnt;
        const double var = a.sums[2 * g + 1] / cnt - m * m;
        mu[e] = (float)m;
        rs[e] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)a.eps));
    }
    const int Hp = a.H + 2, Wp = a.W + 2;
    const T* __restrict__ x = (const T*)a.x;
    T* __restrict__ out = (T*)a.out;  // never aliases x or the tables: the loads of the next pixels may pass the stores
    const T* __restrict__ yt = (const T*)a.yt;
    const T* __restrict__ bt = (const T*)a.bt;
    const int rows = a.F * a.H;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const int f = r / a.H, yy = r - f * a.H;
        const int fz = zq_frame(f, a.F, a.Fz);
        const int yz = (int)(((int64_t)yy * a.hz) / a.H);
        const T* __restrict__ xr = x + (int64_t)r * a.W * C + c0;
        T* __restrict__ orow = out + ((((int64_t)(f + a.f_off)) * Hp + yy + 1) * Wp + 1) * C + c0;
        const int64_t zrow = ((int64_t)fz * a.hz + yz) * a.wz;
#pragma unroll 2
        for (int xx = pl; xx < a.W; xx += ppi) {
            float v[VN], cy[VN], cb[VN];
            V16<T>::ld(xr + (int64_t)xx * C, v);
            if (yt != nullptr) {
                const int xz = (int)(((unsigned)xx * (unsigned)a.wz) / (unsigned)a.W);
                const int64_t pz = (zrow + xz) * C + c0;
                V16<T>::ld(yt + pz, cy);
                V16<T>::ld(bt + pz, cb);
            }
            float o[VN];
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                const float n = ET<T>::rnd((v[e] - mu[e]) * rs[e] * g_w[e] + g_b[e]);
                float rr = n;
                if (yt != nullptr) rr = ET<T>::rnd(ET<T>::rnd(n * cy[e]) + cb[e]);
                if (a.silu) rr = ET<T>::rnd(silu_t<T>(rr));
                o[e] = rr;
            }
            V16<T>::st(orow + (int64_t)xx * C, o);
        }
    }
}
// latent-resolution tables: yt[p][c] = rnd(sum_j zq[p][j] wy[j][c] + by[c]), bt likewise (p over Fz*hz*wz)
template <typename T>
__global__ void snorm_tables_k(const SNormArgs a) {
    constexpr int VN = V16<T>::N;
    const int C = a.C, Cz = a.Cz, vpp = C / VN;
    const int64_t total = (int64_t)a.Fz * a.hz * a.wz * vpp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % vpp) * VN;
        const int64_t p = i / vpp;
        const T* zp = (const T*)a.zq + p * Cz;
        float cy[VN], cb[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) { cy[e] = 0.f; cb[e] = 0.f; }
        for (int j = 0; j < Cz; ++j) {
            const float z = ET<T>::ld(zp + j);
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                cy[e] = fmaf(z, a.wy[j * C + c0 + e], cy[e]);
                cb[e] = fmaf(z, a.wb[j * C + c0 + e], cb[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < VN; ++e) { cy[e] += a.by[c0 + e]; cb[e] += a.bb[c0 + e]; }
        V16<T>::st((T*)a.yt + p * C + c0, cy);
        V16<T>::st((T*)a.bt + p * C + c0, cb);
    }
}

int launch_snorm_apply(const SNormArgs& a, int dtype, hipStream_t st) {
    const int VN = dtype == S2V_F32 ? 4 : 8;
    S2V_REQUIRE(a.C % VN == 0 && a.C % a.G == 0, "snorm_apply: unsupported channel count");
    const bool plain = a.yt == nullptr;  // nn.GroupNorm (+ SiLU) without the zq modulation: the encoder's resnets / norm_out
    S2V_REQUIRE(plain || a.bt, "snorm_apply: table scratch missing");
    const size_t shmem = sizeof(float) * (size_t)2 * a.G;
    const int64_t tl = (int64_t)a.Fz * a.hz * a.wz * (a.C / VN);
    const int64_t total = (int64_t)a.F * a.H * a.W * (a.C / VN);
    const int vpp = a.C / VN;
    const bool rows_form = vpp <= 256 && 256 % vpp == 0 && (int64_t)a.W * a.wz < (1ll << 31);
    const unsigned row_grid = (unsigned)std::min<int64_t>((int64_t)a.F * a.H, 16384);
    S2V_DT_DISPATCH(dtype, {
        if (!plain) hipLaunchKernelGGL(snorm_tables_k<T>, dim3(grid_for(tl)), dim3(256), 0, st, a);
        if (rows_form) hipLaunchKernelGGL(snorm_apply_rows_k<T>, dim3(row_grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(snorm_apply_k<T>, dim3(grid_for(total, 16384)), dim3(256), shmem, st, a);
    })
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// nearest x2 in H, W (+ in time with the first-frame rule when compress_time) -> padded [Fo][2H+2][2W+2][C]
__host__ __device__ inline int upsample_out_frames(int F, int compress_time) {
    if (!compress_time || F == 1) return F;
    return (F & 1) ? 1 + 2 * (F - 1) : 2 * F;
}
template <typename T>
__global__ void upsample_k(const T* x, int F, int H, int W, int C, int compress_time, T* out) {
    constexpr int VN = V16<T>::N;
    const int Fo = upsample_out_frames(F, compress_time), Ho = 2 * H, Wo = 2 * W;
    const int vpp = C / VN;
    const int64_t total = (int64_t)Fo * Ho * Wo * vpp;
    const int Hp = Ho + 2, Wp = Wo + 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int vi = (int)(i % vpp);
        const int64_t p = i / vpp;
        const int xx = (int)(p % Wo);
        const int yy = (int)((p / Wo) % Ho);
        const int fo = (int)(p / ((int64_t)Wo * Ho));
        int fs = fo;
        if (compress_time && F > 1) fs = (F & 1) ? (fo == 0 ? 0 : 1 + (fo - 1) / 2) : fo / 2;
        float v[VN];
        V16<T>::ld(x + ((((int64_t)fs * H + yy / 2) * W) + xx / 2) * C + vi * VN, v);
        V16<T>::st(out + ((((int64_t)fo * Hp + yy + 1) * Wp) + xx + 1) * C + vi * VN, v);
    }
}
int launch_upsample(const void* x, int F, int H, int W, int C, int compress_time, void* out, int dtype, hipStream_t st) {
    const int VN = dtype == S2V_F32 ? 4 : 8;
    S2V_REQUIRE(C % VN == 0, "upsample: unsupported channel count");
    const int64_t total = (int64_t)upsample_out_frames(F, compress_time) * 4 * H * W * (C / VN);
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(upsample_k<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)x, F, H, W, C,
                       compress_time, (T*)out))
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// dense [F][H][W][Co] -> out[co][f0 + f][y][x] of a [Co][Ftot][H][W] tensor
template <typename T>
__global__ void to_ncfhw_k(const T* y, int F, int H, int W, int Co, T* out, int Ftot, int f0) {
    const int64_t total = (int64_t)F * H * W * Co;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W);
        const int yy = (int)((i / W) % H);
        const int f = (int)((i / ((int64_t)W * H)) % F);
        const int co = (int)(i / ((int64_t)W * H * F));
        out[(((int64_t)co * Ftot + f0 + f) * H + yy) * W + xx] = y[(((int64_t)f * H + yy) * W + xx) * Co + co];
    }
}
int launch_to_ncfhw(const void* y, int F, int H, int W, int Co, void* out, int Ftot, int f0, int dtype, hipStream_t st) {
    const int64_t total = (int64_t)F * H * W * Co;
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(to_ncfhw_k<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)y, F, H, W, Co,
                       (T*)out, Ftot, f0))
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// conv_out of the decoder (CogVideoXDecoder3D.forward :978-981: CogVideoXCausalConv3d(128 -> 3, k = 3)) as a DIRECT kernel (round 6).
// As an implicit GEMM it has N = 3 output columns: the 256 x 128-column tile kernel stages A twenty-seven times (once per tap) and multiplies
// 125 zero columns -- 2.3-2.6 ms per launch at 25 TFLOP/s, 3.7 % of a decode (profiles/r04_vae_conv_rates.txt).  Here a workgroup owns a
// 4-row x 32-column patch of the output for ALL frames of the batch and walks the F + 2 planes of the padded operand [2 + F][H + 2][W + 2][Cin]
// once: plane p is the dt = 0 / 1 / 2 tap of output frames p, p - 1, p - 2, so its 6 x 34-pixel halo is copied into LDS once (double-buffered:
// the next plane's global loads are in flight in registers while this one is multiplied; pixel pitch Cin * 2 + 32 bytes: conflict-free fragment
// reads, see the pitch note in the kernel) and feeds three rolling accumulators.  Every (dy, dx, 32-channel block) is one
// v_mfma_f32_16x16x32 per 16-pixel run and output frame: A = 16 consecutive pixels x 32 channels as one ds_read_b128 (lane = pixel i, channel
// octet g), shared by the three output frames; B = 32 channels x 16 output columns of which Cout are real (lanes j < Cout read their weights
// from an LDS copy, the others hold zeros); D[i][j] accumulates in fp32.  Frame p - 2 is complete after plane p: bias, round to the model
// dtype, written in the [C][Ftot][H][W] tile layout directly (the to_ncfhw pass disappears).  Per output element the sum runs over (dt, dy, dx,
// channel) in ascending order, the order of the implicit GEMM's K index.
// Lane maps of v_mfma_f32_16x16x32_{bf16,f16}: A[i][k]: i = lane & 15, k = 8 * (lane >> 4) + e; B[k][j]: j = lane & 15, same k;
// D[i][j]: j = lane & 15, i = 4 * (lane >> 4) + reg (cdna_hip_programming.md, fragment layout).
#define CO_TH 4
#define CO_TW 32
#define CO_PF 13   // 16-byte chunks of a halo plane per thread at Cin = 128: 6 * 34 * 16 / 256 = 12.75
template <typename T16, int KB /* Cin / 32 */>
__global__ __launch_bounds__(256, 1) void conv_out_direct_k(const unsigned short* __restrict__ pad, const unsigned short* __restrict__ w, int ldw,
                                                            const unsigned short* __restrict__ bias, int F, int H, int W, int Cin, int Cout,
                                                            unsigned short* __restrict__ out, int Ftot, int f0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS pitches (MI355X_MICROARCH, LDS: a ds_read_b128 is served in four groups of sixteen lanes -- {0-3, 12-15, 20-27}, ... -- over 64 banks): with
    // lane = pixel i + 16 * octet g a group mixes pixels {0-3, 12-15} of one octet with pixels {4-11} of the next; a pixel pitch of 32 bytes
    // mod 256 (8 banks) puts the first set on the multiples of 8 banks and the second (+ 16 bytes) on the odd multiples of 4: conflict-free (a pitch
    // of 16 mod 256 made lanes 11 and 12 collide in every group: 2 x).  Weight rows are Cout + 1 <= 4 distinct addresses per group: a row pitch of
    // 64 bytes mod 256 spreads them over the banks (0 mod 256 was a 4-way conflict on 60 % of the reads).
    const int pitch = Cin * 2 + 32;                               // bytes per halo pixel
    const int wpitch = 27 * Cin + 32;                             // elements per weight row in LDS
    const int halo_bytes = (CO_TH + 2) * (CO_TW + 2) * pitch;     // one plane: [CO_TH + 2][CO_TW + 2] pixels
    unsigned short* wl = (unsigned short*)(smem + 2 * halo_bytes);  // [Cout + 1][wpitch]: the real output columns, then a row of zeros
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (W + CO_TW - 1) / CO_TW;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * CO_TH, x0 = tx * CO_TW;
    const int Hp = H + 2, Wp = W + 2;
    const int K = 27 * Cin;
    for (int i = tid * 8; i < (Cout + 1) * K; i += 256 * 8) {     // K is a multiple of 8: a 16-byte chunk never straddles two rows
        const int co = i / K, k = i - co * K;
        u32x4 v = {0, 0, 0, 0};                                   // row Cout: what the lanes of the 16 - Cout padding columns, and taps of frames that do not exist, read
        if (co < Cout) v = *(const u32x4*)(w + (size_t)co * ldw + k);
        *(u32x4*)(wl + (size_t)co * wpitch + k) = v;
    }
    const int i16 = lane & 15, g = lane >> 4;
    constexpr int chunks = KB * 4;                                // 16-byte chunks per pixel (compile-time: the index arithmetic below is shifts and constant divisions)
    const int plane_chunks = (CO_TH + 2) * (CO_TW + 2) * chunks;
    u32x4 pf[CO_PF];
    auto gload = [&](int p) {                                     // plane p of the operand -> registers
        const unsigned short* plane = pad + (size_t)p * Hp * Wp * Cin;
#pragma unroll
        for (int j = 0; j < CO_PF; ++j) {
            const int i = tid + 256 * j;
            const int c = i % chunks, px = (i / chunks) % (CO_TW + 2), py = i / (chunks * (CO_TW + 2));
            const int yy = y0 + py, xx = x0 + px;                 // padded coordinates
            pf[j] = u32x4{0, 0, 0, 0};
            if (i < plane_chunks && yy < Hp && xx < Wp) pf[j] = *(const u32x4*)(plane + ((size_t)yy * Wp + xx) * Cin + c * 8);
        }
    };
    auto lstore = [&](int buf) {
        char* halo = smem + buf * halo_bytes;
#pragma unroll
        for (int j = 0; j < CO_PF; ++j) {
            const int i = tid + 256 * j;
            const int c = i % chunks, pp = i / chunks;
            if (i < plane_chunks) *(u32x4*)(halo + pp * pitch + c * 16) = pf[j];
        }
    };
    f32x4 acc[3][2];                                              // [output frame mod 3][16-pixel run]
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
        for (int r = 0; r < 2; ++r) acc[s3][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float bv = i16 < Cout ? ET<T16>::ld((const T16*)bias + i16) : 0.f;
    const int y = y0 + wave;

    // plane p (p mod 3 == R at compile time: the accumulator slots are static): dt-th tap of output frame p - dt, slot (R - dt) mod 3
    auto plane_step = [&](auto Rtag, int p) {
        constexpr int R = decltype(Rtag)::value;
        const char* halo = smem + (p & 1) * halo_bytes;
        // output frames p, p - 1, p - 2 exist? (p <= F + 1)  A tap of a frame that does not exist, like the padding columns j >= Cout, reads its B
        // fragment from the ROW OF ZEROS behind the weights: the loop body has no branch and no select (the compiler's code for a skipped MFMA
        // shuffled every accumulator through VGPRs: 1.0 ms per launch; selects in the loop: 0.63 ms)
        const bool live[3] = {p < F, p - 1 >= 0 && p - 1 < F, p - 2 >= 0};
        const unsigned short* wrow[3];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) wrow[dt] = wl + (size_t)((i16 < Cout && live[dt]) ? i16 : Cout) * wpitch + g * 8;
        // One batch = one (dy, dx) tap position: 2 * KB A fragments + 3 * KB B fragments, then 6 * KB MFMAs.  The NEXT batch's twenty ds_read_b128
        // are issued before this batch's MFMAs (two register sets, the nine batches fully unrolled): with one wave per SIMD nothing else hides the
        // LDS latency -- the straightforward loop waited for every pair of reads and ran at 31 k cycles per plane instead of 6 k
        struct Frags { u32x4 a[KB][2]; u32x4 b[KB][3]; };
        auto load_batch = [&](Frags& fr, int dy, int dx) {
            const char* arow = halo + ((wave + dy) * (CO_TW + 2) + dx + i16) * pitch + g * 16;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                fr.a[kb][0] = *(const u32x4*)(arow + kb * 64);
                fr.a[kb][1] = *(const u32x4*)(arow + 16 * pitch + kb * 64);
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) fr.b[kb][dt] = *(const u32x4*)(wrow[dt] + ((dt * 3 + dy) * 3 + dx) * Cin + kb * 32);
            }
        };
        auto mfma_batch = [&](const Frags& fr) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    f32x4* ac = acc[(R + 3 - dt) % 3];
                    if constexpr (__is_same(T16, f16_t)) {
                        ac[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fr.a[kb][0]), __builtin_bit_cast(f16x8, fr.b[kb][dt]), ac[0], 0, 0, 0);
                        ac[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fr.a[kb][1]), __builtin_bit_cast(f16x8, fr.b[kb][dt]), ac[1], 0, 0, 0);
                    } else {
                        ac[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr.a[kb][0]), __builtin_bit_cast(bf16x8, fr.b[kb][dt]), ac[0], 0, 0, 0);
                        ac[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr.a[kb][1]), __builtin_bit_cast(bf16x8, fr.b[kb][dt]), ac[1], 0, 0, 0);
                    }
                }
        };
        Frags f0_, f1_;
        load_batch(f0_, 0, 0);
#pragma unroll
        for (int bt = 0; bt < 9; ++bt) {
            Frags& cur = (bt & 1) ? f1_ : f0_;
            Frags& nxt = (bt & 1) ? f0_ : f1_;
            if (bt + 1 < 9) load_batch(nxt, (bt + 1) / 3, (bt + 1) % 3);
            __builtin_amdgcn_sched_barrier(0);   // the scheduler otherwise sinks the reads back between the MFMAs that need them (fewer live registers, every read waited for)
            mfma_batch(cur);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (p >= 2) {  // output frame p - 2 is complete: D[i][j]: this lane holds column j = i16 for pixels 4 g .. 4 g + 3 of each run
            f32x4* ac = acc[(R + 1) % 3];
            if (i16 < Cout && y < H) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int x = x0 + r * 16 + 4 * g;
                    T16* o = (T16*)out + (((size_t)i16 * Ftot + f0 + (p - 2)) * H + y) * W + x;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (x + e < W) ET<T16>::st(o + e, ac[r][e] + bv);
                }
            }
            ac[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            ac[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    using R0 = std::integral_constant<int, 0>; using R1 = std::integral_constant<int, 1>; using R2 = std::integral_constant<int, 2>;
    const int P = F + 2;
    gload(0);
    lstore(0);
    __syncthreads();                                              // plane 0 and the weights are in LDS
    for (int p = 0; p < P; ++p) {
        if (p + 1 < P) gload(p + 1);                              // in flight while plane p is multiplied
        switch (p % 3) {
            case 0: plane_step(R0{}, p); break;
            case 1: plane_step(R1{}, p); break;
            default: plane_step(R2{}, p); break;
        }
        if (p + 1 < P) lstore((p + 1) & 1);                       // the other buffer: last read as plane p - 1, before the barrier below of step p - 1
        __syncthreads();
    }
}
// conv_out as the direct kernel when it qualifies (16-bit dtype, Cin % 32 == 0, Cout <= 16, the whole LDS image fits); returns 1 when launched, 0 when
// the caller has to take the implicit-GEMM path, < 0 on error
int launch_conv_out_direct(const void* pad, const void* w, int ldw, const void* bias, int F, int H, int W, int Cin, int Cout, void* out, int Ftot,
                           int f0, int dtype, hipStream_t st) {
    if ((dtype != S2V_BF16 && dtype != S2V_F16) || Cin % 32 != 0 || Cin > 128 || Cout > 16 || Cout < 1 || ldw % 8 != 0) return 0;
    const size_t shmem = (size_t)2 * (CO_TH + 2) * (CO_TW + 2) * (Cin * 2 + 32) + (size_t)(Cout + 1) * (27 * Cin + 32) * 2;
    if (shmem > 160 * 1024) return 0;
    const int tiles = ((W + CO_TW - 1) / CO_TW) * ((H + CO_TH - 1) / CO_TH);
#define S2V_CO_LAUNCH(T, KBV)                                                                                                               \
    {                                                                                                                                        \
        S2V_TRY(ensure_lds_attr((const void*)conv_out_direct_k<T, KBV>, 160 * 1024));                                                        \
        hipLaunchKernelGGL((conv_out_direct_k<T, KBV>), dim3(tiles), dim3(256), shmem, st, (const unsigned short*)pad, (const unsigned short*)w, \
                           ldw, (const unsigned short*)bias, F, H, W, Cin, Cout, (unsigned short*)out, Ftot, f0);                             \
    }
    const int kbv = Cin / 32;
    if (dtype == S2V_F16) {
        if (kbv == 4) S2V_CO_LAUNCH(f16_t, 4) else if (kbv == 2) S2V_CO_LAUNCH(f16_t, 2) else if (kbv == 1) S2V_CO_LAUNCH(f16_t, 1) else return 0;
    } else {
        if (kbv == 4) S2V_CO_LAUNCH(bf16_t, 4) else if (kbv == 2) S2V_CO_LAUNCH(bf16_t, 2) else if (kbv == 1) S2V_CO_LAUNCH(bf16_t, 1) else return 0;
    }
#undef S2V_CO_LAUNCH
    S2V_CHECK_HIP(hipGetLastError());
    return 1;
}

// blend_v / blend_h of tiled_decode, in place on tile b ([C][F][Hb][Wb]) from its already-blended neighbour a:
//   vertical  : b[.., y, :] = a[.., Ha-E+y, :]*(1-y/E) + b[.., y, :]*(y/E),  y < E
//   horizontal: b[.., :, x] = a[.., :, Wa-E+x]*(1-x/E) + b[.., :, x]*(x/E),  x < E
// (python-float weights stay fp32 scalars in torch; each product and the sum round to the tensor dtype)
template <typename T>
__global__ void blend_k(const T* a, int Ha, int Wa, T* b, int Hb, int Wb, int CF, int E, int vertical) {
    const int rows = vertical ? E : min(Ha, Hb), cols = vertical ? min(Wa, Wb) : E;
    const int64_t total = (int64_t)CF * rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % cols);
        const int y = (int)((i / cols) % rows);
        const int cf = (int)(i / ((int64_t)cols * rows));
        const int k = vertical ? y : x;
        const float wb_ = (float)((double)k / (double)E), wa_ = (float)(1.0 - (double)k / (double)E);
        const int ya = vertical ? Ha - E + y : y, xa = vertical ? x : Wa - E + x;
        const float va = ET<T>::ld(a + ((int64_t)cf * Ha + ya) * Wa + xa);
        T* pb = b + ((int64_t)cf * Hb + y) * Wb + x;
        const float vb = ET<T>::ld(pb);
        ET<T>::st(pb, ET<T>::rnd(va * wa_) + ET<T>::rnd(vb * wb_));
    }
}
int launch_blend(const void* a, int Ha, int Wa, void* b, int Hb, int Wb, int CF, int E, int vertical, int dtype,
                 hipStream_t st) {
    const int64_t total = (int64_t)CF * (vertical ? E : (Ha < Hb ? Ha : Hb)) * (vertical ? (Wa < Wb ? Wa : Wb) : E);
    if (total <= 0) return 0;
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(blend_k<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)a, Ha, Wa, (T*)b,
                       Hb, Wb, CF, E, vertical))
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// crop-copy of a tile [CF][Ht][Wt] (first ch x cw pixels) into the frame [CF][H][W] at (y0, x0)
template <typename T>
__global__ void paste_k(const T* tile, int Ht, int Wt, int ch, int cw, T* out, int H, int W, int y0, int x0, int CF) {
    const int64_t total = (int64_t)CF * ch * cw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % cw);
        const int y = (int)((i / cw) % ch);
        const int cf = (int)(i / ((int64_t)cw * ch));
        out[((int64_t)cf * H + y0 + y) * W + x0 + x] = tile[((int64_t)cf * Ht + y) * Wt + x];
    }
}
int launch_paste(const void* tile, int Ht, int Wt, int ch, int cw, void* out, int H, int W, int y0, int x0, int CF,
                 int dtype, hipStream_t st) {
    const int64_t total = (int64_t)CF * ch * cw;
    if (total <= 0) return 0;
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(paste_k<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)tile, Ht, Wt, ch, cw,
                       (T*)out, H, W, y0, x0, CF))
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// video [C][F][H][W] (model dtype) -> [F][H][W][C] float32, clamp(x/2 + 0.5, 0, 1)
template <typename T>
__global__ void postprocess_k(const T* v, int C, int F, int H, int W, float* out) {
    const int64_t total = (int64_t)C * F * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t p = i / C;  // (f, y, x)
        const int64_t fhw = (int64_t)F * H * W;
        float x = ET<T>::ld(v + (int64_t)c * fhw + p);
        x = ET<T>::rnd(ET<T>::rnd(x / 2.0f) + 0.5f);
        out[i] = fminf(fmaxf(x, 0.f), 1.f);
    }
}
// the same, followed by export_to_video's `(frame * 255).astype(np.uint8)` (utils/export_utils.py:175): uint8 [F][H][W][C]
template <typename T>
__global__ void postprocess_u8_k(const T* v, int C, int F, int H, int W, unsigned char* out) {
    const int64_t total = (int64_t)C * F * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        const int64_t fhw = (int64_t)F * H * W;
        float x = ET<T>::ld(v + (int64_t)c * fhw + p);
        x = ET<T>::rnd(ET<T>::rnd(x / 2.0f) + 0.5f);
        x = fminf(fmaxf(x, 0.f), 1.f) * 255.0f;   // float32 product, then truncation toward zero
        out[i] = (unsigned char)(int)x;
    }
}
int launch_postprocess_u8(const void* v, int C, int F, int H, int W, unsigned char* out, int dtype, hipStream_t st) {
    const int64_t total = (int64_t)C * F * H * W;
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(postprocess_u8_k<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)v, C, F, H, W, out))
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_postprocess(const void* v, int C, int F, int H, int W, float* out, int dtype, hipStream_t st) {
    const int64_t total = (int64_t)C * F * H * W;
    S2V_DT_DISPATCH(dtype, hipLaunchKernelGGL(postprocess_k<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)v, C, F, H, W, out))
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}

// conv weight re-pack: src [cout][cin][kt][3][3] (or [cout][cin][3][3], or 1x1...) -> dst [cout][(dt,dy,dx)][cin]
template <typename TS, typename TD>
__global__ void conv_w_repack_k(const TS* src, int cout, int cin, int taps, TD* dst) {
    const int64_t total = (int64_t)cout * cin * taps;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin);
        const int tap = (int)((i / cin) % taps);
        const int co = (int)(i / ((int64_t)cin * taps));
        ET<TD>::st(dst + i, ET<TS>::ld(src + ((int64_t)co * cin + ci) * taps + tap));
    }
}
int launch_conv_w_repack(const void* src, int sdt, int cout, int cin, int taps, void* dst, int ddt, hipStream_t st) {
    const int64_t total = (int64_t)cout * cin * taps;
    dim3 g(grid_for(total));
#define RP(TS, TD) hipLaunchKernelGGL((conv_w_repack_k<TS, TD>), g, dim3(256), 0, st, (const TS*)src, cout, cin, taps, (TD*)dst)
    S2V_REQUIRE(sdt >= 0 && sdt <= 2 && ddt >= 0 && ddt <= 2, "conv_w_repack: unknown dtype");
    switch (sdt * 3 + ddt) {
        case S2V_F32 * 3 + S2V_F32: RP(float, float); break;
        case S2V_F32 * 3 + S2V_BF16: RP(float, bf16_t); break;
        case S2V_F32 * 3 + S2V_F16: RP(float, f16_t); break;
        case S2V_BF16 * 3 + S2V_F32: RP(bf16_t, float); break;
        case S2V_BF16 * 3 + S2V_BF16: RP(bf16_t, bf16_t); break;
        case S2V_BF16 * 3 + S2V_F16: RP(bf16_t, f16_t); break;
        case S2V_F16 * 3 + S2V_F32: RP(f16_t, float); break;
        case S2V_F16 * 3 + S2V_BF16: RP(f16_t, bf16_t); break;
        default: RP(f16_t, f16_t); break;
    }
#undef RP
    S2V_CHECK_HIP(hipGetLastError());
    return 0;
}
