// MX e4m3 quantisation of 8 bf16 values per lane, blocks of 32 = 4 lanes: the plain form (unpack, fmax, __shfl_xor, multiply,
// v_cvt_pk_fp8_f32) against the short form (v_pk_max_u16 on the magnitudes, DPP quad exchanges, v_cvt_scalef32_pk_fp8_bf16) -- which
// step of the short form differs, if any.   Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mx_quant tools/probes/mx_quant.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
typedef short i16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__global__ void k(const u32x4* in, unsigned* out) {  // out per lane: [eb1, w0_1, w1_1, eb2, w0_2, w1_2, amb1, amb2, w0_3, w1_3]
    const int l = threadIdx.x;
    const u32x4 o = in[l];
    float x[8];
    for (int e = 0; e < 4; ++e) { x[2 * e] = __uint_as_float(o[e] << 16); x[2 * e + 1] = __uint_as_float(o[e] & 0xffff0000u); }
    float am = 0.f;
    for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(x[e]));
    am = fmaxf(am, __shfl_xor(am, 1, 64));
    am = fmaxf(am, __shfl_xor(am, 2, 64));
    unsigned eb = (__float_as_uint(am * (1.0f / 448.0f)) + 0x7fffffu) >> 23;
    eb = min(max(eb, 1u), 254u);
    const float inv = __uint_as_float((254u - eb) << 23);
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[0] * inv, x[1] * inv, w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[2] * inv, x[3] * inv, w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[4] * inv, x[5] * inv, w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[6] * inv, x[7] * inv, w1, true);
    u16x2_t pm = __builtin_bit_cast(u16x2_t, o[0] & 0x7fff7fffu);
    for (int e = 1; e < 4; ++e) pm = __builtin_elementwise_max(pm, __builtin_bit_cast(u16x2_t, o[e] & 0x7fff7fffu));
    int amb = (int)max((unsigned)pm[0], (unsigned)pm[1]) << 16;
    amb = max(amb, __builtin_amdgcn_update_dpp(0, amb, 0xB1, 0xf, 0xf, true));
    amb = max(amb, __builtin_amdgcn_update_dpp(0, amb, 0x4E, 0xf, 0xf, true));
    unsigned eb2 = (__float_as_uint(__int_as_float(amb) * (1.0f / 448.0f)) + 0x7fffffu) >> 23;
    eb2 = min(max(eb2, 1u), 254u);
    i16x2_t q0 = {0, 0}, q1 = {0, 0}, r0 = {0, 0}, r1 = {0, 0};
    const float bscale = __uint_as_float(eb << 23), binv = inv;  // both conversions with the reference eb: isolates the instruction
    // (the builtin form of these conversions was mis-compiled by this hipcc: every call read the FIRST source word)
    unsigned uq0 = 0, uq1 = 0, ur0 = 0, ur1 = 0;
    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(uq0) : "v"(o[0]), "v"(bscale));
    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(uq0) : "v"(o[1]), "v"(bscale));
    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(uq1) : "v"(o[2]), "v"(bscale));
    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(uq1) : "v"(o[3]), "v"(bscale));
    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(ur0) : "v"(o[0]), "v"(binv));
    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(ur0) : "v"(o[1]), "v"(binv));
    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(ur1) : "v"(o[2]), "v"(binv));
    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(ur1) : "v"(o[3]), "v"(binv));
    q0 = __builtin_bit_cast(i16x2_t, uq0); q1 = __builtin_bit_cast(i16x2_t, uq1); r0 = __builtin_bit_cast(i16x2_t, ur0); r1 = __builtin_bit_cast(i16x2_t, ur1);
    unsigned* p = out + l * 10;
    p[0] = eb; p[1] = w0; p[2] = w1; p[3] = eb2; p[4] = __builtin_bit_cast(unsigned, q0); p[5] = __builtin_bit_cast(unsigned, q1);
    p[6] = __float_as_uint(am); p[7] = (unsigned)amb; p[8] = __builtin_bit_cast(unsigned, r0); p[9] = __builtin_bit_cast(unsigned, r1);
}
int main() {
    std::vector<unsigned> in(64 * 4), out(64 * 10);
    srand(5);
    for (auto& v : in) {
        auto bf = [](float f) { unsigned u; std::memcpy(&u, &f, 4); return (u + 0x7fff + ((u >> 16) & 1)) >> 16; };
        const float a = (rand() % 2001 - 1000) * 0.003f, b = (rand() % 2001 - 1000) * 0.0007f;
        v = bf(a) | (bf(b) << 16);
    }
    unsigned *di, *dout;
    (void)hipMalloc(&di, in.size() * 4); (void)hipMalloc(&dout, out.size() * 4);
    (void)hipMemcpy(di, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>((const u32x4*)di, dout);
    (void)hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    int bad_am = 0, bad_eb = 0, bad_q = 0, bad_r = 0;
    for (int l = 0; l < 64; ++l) {
        const unsigned* p = &out[l * 10];
        bad_am += p[6] != p[7]; bad_eb += p[0] != p[3];
        bad_q += (p[1] != p[4]) + (p[2] != p[5]); bad_r += (p[1] != p[8]) + (p[2] != p[9]);
        if (l < 4) printf("lane %d: eb %u / %u, amax bits %08x / %08x, bytes ref %08x %08x | scalef32(scale) %08x %08x | scalef32(1/scale) %08x %08x\n", l, p[0], p[3], p[6], p[7], p[1], p[2], p[4], p[5], p[8], p[9]);
    }
    printf("mismatching lanes: amax %d, exponent %d, words via cvt_scalef32(scale = 2^e) %d of 128, via cvt_scalef32(2^-e) %d of 128\n", bad_am, bad_eb, bad_q, bad_r);
    return 0;
}
