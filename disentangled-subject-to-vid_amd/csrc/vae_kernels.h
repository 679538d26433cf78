// launchers of csrc/vae.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct SNormArgs {
    const void* x; int F, H, W, C, G;       // dense [F][H][W][C]
    const double* sums; float eps;          // from gn_stats
    const void* gn_w; const void* gn_b;     // model dtype [C]
    const float* wy; const float* by; const float* wb; const float* bb;  // fp32 [Cz][C] / [C]
    const void* zq; int Fz, hz, wz, Cz;     // dense latent frames of this frame batch [Fz][hz][wz][Cz]
    void* out; int f_off;                   // padded [f_off + F][H+2][W+2][C]
    int silu;
    void* yt; void* bt;                     // scratch [Fz*hz*wz][C] model dtype: conv_y / conv_b at latent resolution
};

int launch_latent_to_zq(const void* lat, int F, int C, int h, int w, float inv_sf, void* out, int y0, int x0, int th,
                        int tw, int dtype, hipStream_t st);
int launch_image_to_padded(const void* img, int C, int F, int H, int W, int y0, int x0, int th, int tw, void* out, int f_off,
                           int dtype, hipStream_t st);
int launch_gaussian_sample(const void* mom, const void* noise, int64_t n, void* out, int dtype, hipStream_t st);
int launch_zero_border(void* pad, int F, int H, int W, int C, int esz, hipStream_t st);  // zero ring of a padded operand [F][H+2][W+2][C]
int launch_dense_to_padded(const void* in, int F, int H, int W, int C, void* out, int f_off, int dtype, hipStream_t st);
int64_t gn_stats_scratch_bytes(int64_t P, int G);
int launch_gn_stats(const void* x, int64_t P, int C, int G, double* sums, double* part_scratch, int dtype, hipStream_t st);
int launch_snorm_apply(const SNormArgs& a, int dtype, hipStream_t st);
int launch_upsample(const void* x, int F, int H, int W, int C, int compress_time, void* out, int dtype, hipStream_t st);
int launch_to_ncfhw(const void* y, int F, int H, int W, int Co, void* out, int Ftot, int f0, int dtype, hipStream_t st);
int launch_conv_out_direct(const void* pad, const void* w, int ldw, const void* bias, int F, int H, int W, int Cin, int Cout, void* out, int Ftot,
                           int f0, int dtype, hipStream_t st);  // 1 = launched, 0 = does not qualify
int launch_blend(const void* a, int Ha, int Wa, void* b, int Hb, int Wb, int CF, int E, int vertical, int dtype,
                 hipStream_t st);
int launch_paste(const void* tile, int Ht, int Wt, int ch, int cw, void* out, int H, int W, int y0, int x0, int CF,
                 int dtype, hipStream_t st);
int launch_postprocess(const void* v, int C, int F, int H, int W, float* out, int dtype, hipStream_t st);
int launch_postprocess_u8(const void* v, int C, int F, int H, int W, unsigned char* out, int dtype, hipStream_t st);
int launch_conv_w_repack(const void* src, int sdt, int cout, int cin, int taps, void* dst, int ddt, hipStream_t st);
