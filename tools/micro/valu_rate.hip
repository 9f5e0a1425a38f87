// Issue rate of v_exp_f32 vs v_fma_f32 on gfx950: cycles per wave-instruction with 1 / 2 / 4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);
            else if (MODE == 1) a[i] = __builtin_fmaf(a[i], 0.999f, 0.001f);
            else a[i] = __builtin_amdgcn_rcpf(a[i]);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(t1 - t0) * 0.f;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}
int main() {
    float* d;
    hipMalloc(&d, 1 << 24);
    const int iters = 20000;
    const char* names[3] = {"v_exp_f32", "v_fma_f32", "v_rcp_f32"};
    for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD = blocks per CU (256 threads = 1 wave per SIMD)
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            const int blocks = 256 * wps;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
                else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
            // per SIMD: wps waves x iters x 8 instructions
            printf("%s  waves/SIMD=%d: %.3f ms, in-kernel %.0f ticks for %d instr/wave -> %.2f ns per wave-instruction per SIMD\n",
                   names[mode], wps, ms, cyc, iters * 8, ms * 1e6 / ((double)wps * iters * 8));
        }
    }
    return 0;
}
