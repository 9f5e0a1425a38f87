// fp32 atomic-add throughput on gfx950 when the target lines are only ever touched from ONE XCD (the dQ accumulation of a fused
// attention backward with the heads dealt to the XCDs): workgroup-scope (resolved in the XCD's L2) vs agent-scope atomics, against
// plain stores of the same pattern.  Each workgroup adds 16 KiB tiles (128 rows x 32 floats) into its XCD's private 0.8 MiB
// region, walking it like 49 query tiles; blockIdx % 8 = XCD.
// build: hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* buf, int iters, int region_floats, int heads_per_xcd) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int head = slot % heads_per_xcd;
    float* base = buf + ((size_t)xcd * heads_per_xcd + head) * region_floats;
    const int tid = threadIdx.x;
    const int ntile = region_floats / 4096;   // 16 KiB tiles
    int t = slot % ntile;
    for (int it = 0; it < iters; ++it) {
        float* p = base + (size_t)t * 4096;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float* q = p + j * 256 + tid;
            if (MODE == 0) __hip_atomic_fetch_add(q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 1) __hip_atomic_fetch_add(q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 2) atomicAdd(q, 1.0f);
            else *q = (float)it;
        }
        t = (t + 1) % ntile;
    }
}
int main() {
    const int heads_per_xcd = 8, region_floats = 6272 * 32;   // one head's dQ: 0.8 MB
    float* buf;
    (void)hipMalloc(&buf, (size_t)8 * heads_per_xcd * region_floats * 4);
    (void)hipMemset(buf, 0, (size_t)8 * heads_per_xcd * region_floats * 4);
    const int iters = 200;
    const char* names[4] = {"atomic add, workgroup scope", "atomic add, agent scope", "atomicAdd (default)", "plain store"};
    for (int wgs = 2; wgs <= 8; wgs *= 2) {
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            const int blocks = 256 * wgs;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, buf, iters, region_floats, heads_per_xcd);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, buf, iters, region_floats, heads_per_xcd);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, buf, iters, region_floats, heads_per_xcd);
                else hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, buf, iters, region_floats, heads_per_xcd);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double bytes = (double)blocks * iters * 16384;
            printf("%d WG/CU  %-30s %8.1f GB/s of 4-byte updates (%.2f ms for %.1f GB)\n", wgs, names[mode], bytes / ms / 1e6, ms, bytes / 1e9);
        }
    }
    // check: workgroup-scope adds from the workgroups of one XCD must all land (sum over a region == number of adds)
    (void)hipMemset(buf, 0, (size_t)8 * heads_per_xcd * region_floats * 4);
    hipLaunchKernelGGL(k<0>, dim3(256 * 4), dim3(256), 0, 0, buf, 50, region_floats, heads_per_xcd);
    (void)hipDeviceSynchronize();
    float* h = (float*)malloc((size_t)8 * heads_per_xcd * region_floats * 4);
    (void)hipMemcpy(h, buf, (size_t)8 * heads_per_xcd * region_floats * 4, hipMemcpyDeviceToHost);
    double s = 0;
    for (size_t i = 0; i < (size_t)8 * heads_per_xcd * region_floats; ++i) s += h[i];
    printf("workgroup-scope adds landed: %.0f of %.0f\n", s, (double)256 * 4 * 50 * 4096);
    return 0;
}
