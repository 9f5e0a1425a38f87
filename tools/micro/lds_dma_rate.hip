// Throughput of the L2 -> CU paths on gfx950: LDS-DMA (buffer_load ... lds, 16 B per lane) against plain global_load_dwordx4
// into VGPRs, source = a buffer that stays L2 / MALL resident, 1 / 2 / 4 workgroups of 256 threads per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_dma_rate lds_dma_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void* lds_void_ptr;

// MODE 0: LDS-DMA, MODE 1: global_load_dwordx4 -> VGPR (xor-reduced so the loads stay), MODE 2: global_load -> ds_write_b128
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, uint4* __restrict__ out, int iters, int src_kb) {
    __shared__ __attribute__((aligned(1024))) char smem[32768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_kb * 1024, 0x00020000);
    // every instruction of a wave moves 1 KiB (64 lanes x 16 B, contiguous); the workgroup walks the buffer
    const int span = src_kb * 1024;
    int off = ((blockIdx.x * 4 + wave) * 8192) % span;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int o = (off + j * 1024) % span;
            if (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)(smem + wave * 8192 + j * 1024), 16, lane * 16, o, 0, 0);
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(src) + o + lane * 16);
                if (MODE == 1) { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
                else *reinterpret_cast<uint4*>(smem + wave * 8192 + j * 1024 + lane * 16) = v;
            }
        }
        off = (off + 8192 * 4 * 7) % span;
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE != 1) acc = *reinterpret_cast<uint4*>(smem + tid * 16);
    out[blockIdx.x * 256 + tid] = acc;
}
int main() {
    uint4 *src, *out;
    hipMalloc(&src, 64 << 20);
    hipMemset(src, 1, 64 << 20);
    hipMalloc(&out, 1 << 24);
    const int iters = 2000;
    const char* names[3] = {"LDS-DMA 16 B/lane", "global_load_dwordx4 -> VGPR", "global_load_dwordx4 -> ds_write_b128"};
    for (int src_kb : {1024, 16384}) {
        for (int wgs = 1; wgs <= 4; wgs *= 2) {
            for (int mode = 0; mode < 3; ++mode) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                const int blocks = 256 * wgs;
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, src, out, iters, src_kb);
                    else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, src, out, iters, src_kb);
                    else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, src, out, iters, src_kb);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                const double bytes = (double)blocks * 4 * iters * 8 * 1024;
                printf("src %5d KiB  %d WG/CU  %-38s %8.1f GB/s  (%.1f B/clk/CU at 2.1 GHz)\n", src_kb, wgs, names[mode], bytes / ms / 1e6,
                       bytes / ms / 1e6 / 256 / 2.1);
            }
        }
    }
    return 0;
}
