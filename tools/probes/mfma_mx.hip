// Block scales of v_mfma_scale_f32_32x32x64_f8f6f4 (MX operands): which lane and byte of the scale VGPR scales which K elements.
// Part 1 (discovery): all scales 127 except one byte of one lane = 128; the other operand is zeroed outside one half of K.
// Part 2 (confirmation) of what part 1 showed on MI355X: with lane l holding row l & 31 and 32 bytes t = 0 .. 31 (K elements
// (l >> 5) * 32 + t in THIS probe's loading), the logical block of a byte is t >> 4 -- block b = bytes 16 b .. 16 b + 15 of both lane
// halves -- its scale is byte `opsel` of the scale VGPR of lane 32 b + row (A) / 32 b + column (B).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_mx tools/probes/mfma_mx.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned char to_e4m3(float v) { return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false) & 0xff); }
template <int OA, int OB>
__global__ void run(const float* A, const float* B, const int* SA, const int* SB, float* D) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) {
        unsigned ua = 0, ub = 0;
        for (int e = 0; e < 4; ++e) {
            const int k = (l >> 5) * 32 + w * 4 + e;
            ua |= (unsigned)to_e4m3(A[(l & 31) * 64 + k]) << (8 * e);
            ub |= (unsigned)to_e4m3(B[k * 32 + (l & 31)]) << (8 * e);
        }
        a[w] = (int)ua; b[w] = (int)ub;
    }
    f32x16 c;
    for (int e = 0; e < 16; ++e) c[e] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OA, SA[l], OB, SB[l]);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
static std::vector<float> A(32 * 64), B(64 * 32), D(1024), R(1024);
static float *dA, *dB, *dD; static int *dSA, *dSB;
template <int OA, int OB>
static void go(const std::vector<int>& SA, const std::vector<int>& SB, std::vector<float>& out) {
    (void)hipMemcpy(dSA, SA.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dSB, SB.data(), 256, hipMemcpyHostToDevice);
    run<OA, OB><<<1, 64>>>(dA, dB, dSA, dSB, dD);
    (void)hipMemcpy(out.data(), dD, 4096, hipMemcpyDeviceToHost);
}
template <int OP>
static void discover(bool opA) {
    const int unit = 127 | (127 << 8) | (127 << 16) | (127 << 24);
    for (int blk = 0; blk < 2; ++blk) {
        // keep only bytes 16 blk .. 16 blk + 15 (of both lane halves) of the OTHER operand, so only those products survive
        std::vector<float> B2 = B, A2 = A;
        if (opA) { for (int k = 0; k < 64; ++k) if (((k & 31) >> 4) != blk) for (int j = 0; j < 32; ++j) B2[k * 32 + j] = 0.f; }
        else { for (int k = 0; k < 64; ++k) if (((k & 31) >> 4) != blk) for (int i = 0; i < 32; ++i) A2[i * 64 + k] = 0.f; }
        (void)hipMemcpy(dA, A2.data(), 8192, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B2.data(), 8192, hipMemcpyHostToDevice);
        std::vector<int> U(64, unit);
        if (opA) go<OP, 0>(U, U, R); else go<0, OP>(U, U, R);
        for (int lane : {0, 1, 5, 31, 32, 33, 63})
            for (int byte = 0; byte < 4; ++byte) {
                std::vector<int> S(64, unit);
                S[lane] = (unit & ~(0xff << (8 * byte))) | (128 << (8 * byte));
                if (opA) go<OP, 0>(S, U, D); else go<0, OP>(U, S, D);
                int nrow = 0, first = -1;
                for (int x = 0; x < 32; ++x) {  // x = row of D (operand A) or column of D (operand B)
                    bool dbl = false;
                    for (int y = 0; y < 32; ++y) {
                        const float d = opA ? D[x * 32 + y] : D[y * 32 + x], r = opA ? R[x * 32 + y] : R[y * 32 + x];
                        if (r != 0.f && std::fabs(d - 2 * r) < 1e-6f * std::fabs(r) + 1e-9f) dbl = true;
                    }
                    if (dbl) { ++nrow; if (first < 0) first = x; }
                }
                if (nrow) printf("%s opsel %d: lane %2d byte %d doubles %d %s (first %d) of byte group %d\n", opA ? "A" : "B", OP, lane, byte, nrow, opA ? "row(s)" : "column(s)", first, blk);
            }
    }
}
int main() {
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) A[i * 64 + k] = (float)(((i * 7 + k * 3) % 9) - 4) * 0.5f + 0.25f;
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)(((k * 5 + j * 11) % 7) - 3) * 0.25f + 0.125f;
    (void)hipMalloc(&dA, 8192); (void)hipMalloc(&dB, 8192); (void)hipMalloc(&dD, 4096); (void)hipMalloc(&dSA, 256); (void)hipMalloc(&dSB, 256);
    discover<0>(true); discover<2>(true); discover<0>(false);
    // confirmation: every lane and byte different
    (void)hipMemcpy(dA, A.data(), 8192, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), 8192, hipMemcpyHostToDevice);
    auto sa = [](int l, int byte) { return 120 + ((l * 3 + byte * 5) % 13); };
    auto sb = [](int l, int byte) { return 122 + ((l * 5 + byte * 7) % 11); };
    std::vector<int> SA(64), SB(64);
    for (int l = 0; l < 64; ++l) {
        SA[l] = sa(l, 0) | (sa(l, 1) << 8) | (sa(l, 2) << 16) | (sa(l, 3) << 24);
        SB[l] = sb(l, 0) | (sb(l, 1) << 8) | (sb(l, 2) << 16) | (sb(l, 3) << 24);
    }
    auto check = [&](int oa, int ob) {
        double maxd = 0, maxr = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0;
            for (int k = 0; k < 64; ++k) {
                const int b = (k & 31) >> 4;  // logical block of byte t = k & 31 of lane half k >> 5
                s += (double)A[i * 64 + k] * std::ldexp(1.0, sa(b * 32 + i, oa) - 127) * (double)B[k * 32 + j] * std::ldexp(1.0, sb(b * 32 + j, ob) - 127);
            }
            maxd = std::fmax(maxd, std::fabs(D[i * 32 + j] - s)); maxr = std::fmax(maxr, std::fabs(s));
        }
        printf("opsel A %d B %d: max |D - expected| = %.3g (max |expected| %.3g)  %s\n", oa, ob, maxd, maxr, maxd <= 1e-5 * maxr ? "block / lane mapping confirmed" : "MISMATCH");
    };
    go<0, 0>(SA, SB, D); check(0, 0);
    go<1, 0>(SA, SB, D); check(1, 0);
    go<0, 2>(SA, SB, D); check(0, 2);
    go<3, 1>(SA, SB, D); check(3, 1);
    return 0;
}
