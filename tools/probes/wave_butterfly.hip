#include "../../disentangled-subject-to-vid_amd/csrc/common.h"
#include <cstdio>
__global__ void k(const float* in, float* out) {
    float v = in[threadIdx.x];
    float a = wave_sum(v);
    float b = v;
    for (int o = 32; o > 0; o >>= 1) b += __shfl_xor(b, o, 64);
    float c = wave_max(v), d = v;
    for (int o = 32; o > 0; o >>= 1) d = fmaxf(d, __shfl_xor(d, o, 64));
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b; out[128 + threadIdx.x] = c; out[192 + threadIdx.x] = d;
}
int main() {
    float h[64], o[256]; unsigned s = 12345;
    for (int i = 0; i < 64; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(int)(s >> 8) / 16777216.0f * 7.3f - 3.1f; }
    float *di, *dd; hipMalloc(&di, 256); hipMalloc(&dd, 1024); hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dd); hipMemcpy(o, dd, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 64; ++i) { if (o[i] != o[64 + i]) bad++; if (o[128 + i] != o[192 + i]) bad++; }
    printf("wave_butterfly vs __shfl_xor loop: %d mismatching lanes (sum %.9g / %.9g, max %.9g)\n", bad, o[0], o[64], o[128]);
    return bad != 0;
}
