// Shader-clock probe ("the SMI-free clock the attention kernels held"): ONE wave that samples the ratio of the
// shader-clock counter (s_memtime: ticks at the current SCLK, chip-wide) to the constant-rate wall counter (s_memrealtime) in
// windows of `period` wall ticks while other streams run the step beside it.  It sleeps between the two reads of a window (s_sleep),
// takes one wave slot of one CU, and leaves when (a) the caller sets *stop (stream-ordered fill on another stream), (b) the sample
// buffer is full, or (c) `max_ticks` wall ticks have passed — every path is reached without any other wave's help.
// bench.py runs it beside an UNTIMED block of steps right after the timed blocks: the clock those steps hold explains the
// box-to-box spread of the headline (17.1-17.7 ms on this pool), the timed region itself carries no instrumentation.
#include "common.h"

namespace {

__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* __restrict__ samples, int* __restrict__ count, int max_samples,
                                                         const int* __restrict__ stop, unsigned long long period, unsigned long long max_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    int n = 0;
    while (n < max_samples) {
        const unsigned long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long r1;
        do {
            __builtin_amdgcn_s_sleep(64);
            r1 = __builtin_amdgcn_s_memrealtime();
        } while (r1 - r0 < period && r1 - t_begin < max_ticks);
        const unsigned long long m1 = __builtin_amdgcn_s_memtime();
        samples[2 * n] = m1 - m0;
        samples[2 * n + 1] = r1 - r0;
        ++n;
        if (__atomic_load_n(stop, __ATOMIC_RELAXED) != 0 || r1 - t_begin >= max_ticks) break;
    }
    *count = n;
}

}  // namespace

extern "C" {

// samples: 2 * max_samples uint64 (shader ticks, wall ticks per window); count: int32 (written when the probe leaves); stop: int32 the
// caller sets non-zero to end the probe; period_us: window length; max_ms: hard time limit.  wall_khz: the wall counter's rate
// (hipDeviceAttributeWallClockRate), returned so that GHz = shader ticks / wall ticks * wall_khz / 1e6.
int svol_clock_probe(uint64_t* samples, int32_t* count, int32_t max_samples, const int32_t* stop, int64_t period_us, int64_t max_ms,
                     int32_t* wall_khz, void* stream) {
    if (!samples || !count || !stop || !wall_khz || max_samples < 1 || period_us < 1 || max_ms < 1 || max_ms > 60000) return SVOL_E_INVALID;
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
        return SVOL_E_LAUNCH;
    *wall_khz = khz;
    const unsigned long long period = (unsigned long long)period_us * (unsigned long long)khz / 1000ull;
    const unsigned long long max_ticks = (unsigned long long)max_ms * (unsigned long long)khz;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), (unsigned long long*)samples, count,
                       max_samples, stop, period, max_ticks);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

}  // extern "C"
