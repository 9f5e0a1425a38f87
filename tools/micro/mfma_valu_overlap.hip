// Do the matrix pipe and the vector ALU of one gfx950 SIMD overlap (a) across waves whose instruction streams alternate
// coarse MFMA / VALU phases, (b) inside one wave with a fine MFMA / VALU interleave?
// Per loop iteration and wave: 8 x v_mfma_f32_32x32x16_bf16 (4 independent accumulators) and NV x (v_fma_f32 | v_exp_f32) on
// registers the MFMAs never touch.  MODE 0: MFMA only, 1: VALU only, 2: 8 MFMA then NV VALU, 3: 1 MFMA + NV/8 VALU, x8.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

#ifdef USE_AGPR
#define ACC(x) "+a"(x)
#else
#define ACC(x) "+v"(x)
#endif
#define MF(acc) "v_mfma_f32_32x32x16_bf16 %" #acc ", %4, %5, %" #acc "\n\t"
#define V8F "v_fma_f32 %6, %6, %14, %15\n\tv_fma_f32 %7, %7, %14, %15\n\tv_fma_f32 %8, %8, %14, %15\n\tv_fma_f32 %9, %9, %14, %15\n\t" \
            "v_fma_f32 %10, %10, %14, %15\n\tv_fma_f32 %11, %11, %14, %15\n\tv_fma_f32 %12, %12, %14, %15\n\tv_fma_f32 %13, %13, %14, %15\n\t"
#define V8E "v_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\t" \
            "v_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\tv_exp_f32 %12, %12\n\tv_exp_f32 %13, %13\n\t"
#define OPS : ACC(c0), ACC(c1), ACC(c2), ACC(c3) : "v"(a), "v"(b), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(k0), "v"(k1)

template <int MODE, int EXP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = c2[i] = c3[i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f); b[i] = (__bf16)1e-3f; }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
    float k0 = 0.999f, k1 = 0.001f;
    for (int it = 0; it < iters; ++it) {
        // x[] are read-write through "v" inputs on purpose: the asm is volatile and the values are only timing fodder
        if (MODE == 0) {
            asm volatile(MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) OPS);
        } else if (MODE == 1) {
            if (EXP) asm volatile(V8E V8E V8E V8E V8E V8E V8E V8E OPS);
            else asm volatile(V8F V8F V8F V8F V8F V8F V8F V8F OPS);
        } else if (MODE == 2) {
            if (EXP) asm volatile(MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) V8E V8E V8E V8E V8E V8E V8E V8E OPS);
            else asm volatile(MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) V8F V8F V8F V8F V8F V8F V8F V8F OPS);
        } else {
            if (EXP) asm volatile(MF(0) V8E MF(1) V8E MF(2) V8E MF(3) V8E MF(0) V8E MF(1) V8E MF(2) V8E MF(3) V8E OPS);
            else asm volatile(MF(0) V8F MF(1) V8F MF(2) V8F MF(3) V8F MF(0) V8F MF(1) V8F MF(2) V8F MF(3) V8F OPS);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int EXP>
static float run(float* d, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, EXP>), dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}
int main() {
    float* d;
    hipMalloc(&d, 1 << 24);
    const int iters = 20000;
    const char* names[4] = {"8 MFMA", "64 VALU", "8 MFMA ; 64 VALU", "8 x (MFMA ; 8 VALU)"};
    for (int e = 0; e < 2; ++e) {
        printf("---- VALU = %s\n", e ? "v_exp_f32" : "v_fma_f32");
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int blocks = 256 * wps;
            float ms[4];
            if (e) { ms[0] = run<0, 1>(d, blocks, iters); ms[1] = run<1, 1>(d, blocks, iters); ms[2] = run<2, 1>(d, blocks, iters); ms[3] = run<3, 1>(d, blocks, iters); }
            else   { ms[0] = run<0, 0>(d, blocks, iters); ms[1] = run<1, 0>(d, blocks, iters); ms[2] = run<2, 0>(d, blocks, iters); ms[3] = run<3, 0>(d, blocks, iters); }
            for (int m = 0; m < 4; ++m)
                printf("waves/SIMD=%d  %-22s %.3f ms -> %.1f ns per wave-iteration per SIMD\n", wps, names[m], ms[m],
                       ms[m] * 1e6 / ((double)wps * iters));
        }
    }
    return 0;
}
