// Issue-port model of one gfx950 SIMD: what a stream that mixes v_mfma_f32_32x32x16_bf16, v_exp_f32 and plain VALU costs, against
// the stand-alone rates (MFMA 32 cycles, v_exp_f32 8, v_fma_f32 / v_mul_f32 / v_cvt_pk 2).  All operands independent.
// build: hipcc --offload-arch=gfx950 -O3 -o issue_mix issue_mix.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
#define M0 "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\t"
#define M1 "v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\t"
#define M2 "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\t"
#define M3 "v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n\t"
#define E(i) "v_exp_f32 %" #i ", %" #i "\n\t"
#define F(i) "v_fma_f32 %" #i ", %" #i ", %14, %15\n\t"
#define C(i, j) "v_cvt_pk_bf16_f32 %" #i ", %" #i ", %" #j "\n\t"
#define OPS : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(k0), "v"(k1)
#define E8 E(6) E(7) E(8) E(9) E(10) E(11) E(12) E(13)
#define F8 F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13)
#define EF8 E(6) F(7) E(8) F(9) E(10) F(11) E(12) F(13) E(7) F(6) E(9) F(8) E(11) F(10) E(13) F(12)
#define C4 C(6, 7) C(8, 9) C(10, 11) C(12, 13)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = c2[i] = c3[i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f); b[i] = (__bf16)1e-3f; }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
    float k0 = 0.999f, k1 = 0.001f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) asm volatile(EF8 EF8 EF8 EF8 OPS);                                   // 32 exp + 32 fma, 1:1
        if (MODE == 1) asm volatile(M0 E8 M1 E8 M2 E8 M3 E8 OPS);                            // 4 MFMA + 32 exp (1:8)
        if (MODE == 2) asm volatile(M0 E(6) E(7) E(8) E(9) M1 E(10) E(11) E(12) E(13) M2 E(6) E(7) E(8) E(9) M3 E(10) E(11) E(12) E(13) OPS);  // 1:4
        if (MODE == 3) asm volatile(M0 E(6) E(7) M1 E(8) E(9) M2 E(10) E(11) M3 E(12) E(13) OPS);  // 4 MFMA + 8 exp (1:2)
        // one dQ-pass 32x32 block: 6 MFMA, 16 exp, 16 mul, 8 cvt -- as the compiler emits it (phases) ...
        if (MODE == 4) asm volatile(M0 M0 E8 E8 M1 M1 F8 F8 C4 C4 M2 M2 OPS);
        // ... and with the MFMAs spread through the exp / VALU stream
        if (MODE == 5) asm volatile(M0 E(6) E(7) E(8) M0 E(9) E(10) E(11) M1 E(12) E(13) E(6) M1 E(7) E(8) E(9) M2 E(10) E(11) E(12) M2 E(13) F8 F8 C4 C4 OPS);
        // the exps interleaved 1:1 with the plain VALU work of the previous block, MFMAs spread
        if (MODE == 6) asm volatile(M0 E(6) F(7) E(8) F(9) M0 E(10) F(11) E(12) F(13) M1 E(7) F(6) E(9) F(8) M1 E(11) F(10) E(13) F(12)
                                    M2 E(6) F(7) E(8) F(9) M2 E(10) F(11) E(12) F(13) E(7) C(6, 8) E(9) C(10, 12) E(11) C(6, 8) E(13) C(10, 12) C4 OPS);
        // one forward 128-key x 32-query sub-block (32 keys): 4 MFMA, 16 exp, 8 max3 (as fma), 16 add, 8 cvt
        if (MODE == 7) asm volatile(M0 M0 F8 E8 E8 F8 F8 C4 C4 M1 M1 OPS);
        if (MODE == 8) asm volatile(M0 E(6) F(7) E(8) F(9) E(10) F(11) E(12) F(13) M0 E(7) F(6) E(9) F(8) E(11) F(10) E(13) F(12) M1 E(6) F(7) E(8) F(9) E(10) F(11) E(12) F(13)
                                    M1 E(7) F(6) E(9) F(8) E(11) F(10) E(13) F(12) C4 C4 OPS);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static float run(float* d, int blocks, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}
int main() {
    float* d;
    (void)hipMalloc(&d, 1 << 24);
    const int iters = 20000;
    struct { const char* name; double sum; } info[9] = {
        {"32 exp + 32 fma 1:1", 32 * 3.4 + 32 * 0.97}, {"4 MFMA + 32 exp (1:8)", 4 * 13.7 + 32 * 3.4}, {"4 MFMA + 16 exp (1:4)", 4 * 13.7 + 16 * 3.4},
        {"4 MFMA + 8 exp (1:2)", 4 * 13.7 + 8 * 3.4}, {"dQ block, phases", 6 * 13.7 + 16 * 3.4 + 24 * 0.97}, {"dQ block, MFMA spread in exp", 6 * 13.7 + 16 * 3.4 + 24 * 0.97},
        {"dQ block, exp:valu 1:1 + MFMA spread", 6 * 13.7 + 16 * 3.4 + 24 * 0.97 + 4 * 0.97}, {"fwd block, phases", 4 * 13.7 + 16 * 3.4 + 32 * 0.97},
        {"fwd block, exp:valu 1:1 + MFMA spread", 4 * 13.7 + 32 * 3.4 + 40 * 0.97}};
    for (int wps = 2; wps <= 4; ++wps) {
        const int blocks = 256 * wps;
        float ms[9] = {run<0>(d, blocks, iters), run<1>(d, blocks, iters), run<2>(d, blocks, iters), run<3>(d, blocks, iters), run<4>(d, blocks, iters),
                       run<5>(d, blocks, iters), run<6>(d, blocks, iters), run<7>(d, blocks, iters), run<8>(d, blocks, iters)};
        for (int m = 0; m < 9; ++m)
            printf("waves/SIMD=%d  %-40s %7.1f ns per wave-iteration per SIMD   (stand-alone sum %.1f)\n", wps, info[m].name,
                   ms[m] * 1e6 / ((double)wps * iters), info[m].sum);
    }
    return 0;
}
