// Shared device/host helpers for libsvol_hip (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svol_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---- the 16-bit operand type of a translation unit ---------------------------------------------------------------------------
// The MFMA-path files (gemm_bf16 / gemm_ws_bf16 / gemm_n256_bf16 / gemm_tn_bf16 / attention_bf16 .hip) are written against
// h16_t and compiled TWICE (svol_amd/build.py): as they are — h16_t = bf16, v_mfma_*_bf16 — and with -DSVOL_H16_FP16 — h16_t =
// fp16, v_mfma_*_f16 (same rate, 11 mantissa bits, 5 exponent bits) — with their external launchers renamed *_bf16* -> *_f16*.
// One source per kernel family, two operand types (BASELINE configs[4] is stated in fp16).
#ifdef SVOL_H16_FP16
typedef f16_t h16_t;
typedef f16x8 h16x8;
typedef f16x4 h16x4;
typedef f16x2 h16x2;
#define SVOL_MFMA_16x16x32_H16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define SVOL_MFMA_32x32x16_H16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define SVOL_DS_READ_TR16_H16 svol_ds_read_tr16_f16
#define SVOL_H16_ASM "f16"
#define SVOL_H16_ONE2 0x3C003C00u      /* (1.0, 1.0) */
#define SVOL_H16_NINF_LO 0x0000FC00u   /* (-inf, 0) */
#define SVOL_H16_DTYPE SVOL_F16
#define SVOL_H16_NEG_BIG (-6.0e4f)   /* most negative row constant that is still finite as an fp16 operand */
#define SVOL_H16_PSUM_MAX 6.5e4f   /* a softmax numerator that would not fit fp16 (65504) makes the row sum exceed this */
#define svol_gemm_nt_bf16_fast svol_gemm_nt_f16_fast
#define svol_gemm_tn_bf16_fast svol_gemm_tn_f16_fast
#define svol_gemm_tn_bf16_grouped svol_gemm_tn_f16_grouped
#define svol_conv_wgrad_bf16_fast svol_conv_wgrad_f16_fast
#define svol_gemm_ws_bf16 svol_gemm_ws_f16
#define svol_gemm_n256_bf16 svol_gemm_n256_f16
#define svol_attn_fwd_bf16_launch svol_attn_fwd_f16_launch
#define svol_attn_bwd_bf16_launch svol_attn_bwd_f16_launch
#define svol_attn_ws_floats_bf16 svol_attn_ws_floats_f16
#define svol_attn_sp_image_bytes_bf16 svol_attn_sp_image_bytes_f16
#define svol_attn_sp_zero_bf16_launch svol_attn_sp_zero_f16_launch
#else
typedef bf16_t h16_t;
typedef bf16x8 h16x8;
typedef bf16x4 h16x4;
typedef bf16x2 h16x2;
#define SVOL_MFMA_16x16x32_H16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define SVOL_MFMA_32x32x16_H16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define SVOL_DS_READ_TR16_H16 __builtin_amdgcn_ds_read_tr16_b64_v4bf16
#define SVOL_H16_ASM "bf16"
#define SVOL_H16_ONE2 0x3F803F80u
#define SVOL_H16_NINF_LO 0x0000FF80u
#define SVOL_H16_DTYPE SVOL_BF16
#define SVOL_H16_NEG_BIG (-3.0e38f)
#define SVOL_H16_PSUM_MAX 1.0e30f
#endif
// (the v4f16 form of the builtin is typed on __fp16, not _Float16: go through the 16-bit integer form)
typedef __attribute__((ext_vector_type(4))) short i16x4;
__device__ __forceinline__ f16x4 svol_ds_read_tr16_f16(__attribute__((address_space(3))) f16x4* p) {
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)p));
}
// two fp32 -> one packed pair of the TU's 16-bit type in ONE instruction (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32).  hipcc finds the bf16
// form by itself from scalar casts; the fp16 form only from a vector conversion (scalar casts became 2 x v_cvt_f16_f32 + v_pack).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ h16x2 cvt_pk_h16(float a, float b) {
#ifdef SVOL_H16_FP16
    return __builtin_convertvector((f32x2){a, b}, f16x2);
#else
    return h16x2{(bf16_t)a, (bf16_t)b};
#endif
}
static inline bool svol_is16(int dtype) { return dtype == SVOL_BF16 || dtype == SVOL_F16; }

// per-type MFMA / transposed-LDS-read for kernels that are C++ templates on the element type (gemm.hip, attention.hip)
template <typename T> struct H16;
template <> struct H16<bf16_t> {
    typedef bf16x8 v8;
    typedef bf16x4 v4;
    typedef __attribute__((address_space(3))) bf16x4* lds_v4_ptr;
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ v4 read_tr(lds_v4_ptr p) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p); }
};
template <> struct H16<f16_t> {
    typedef f16x8 v8;
    typedef f16x4 v4;
    typedef __attribute__((address_space(3))) f16x4* lds_v4_ptr;
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ v4 read_tr(lds_v4_ptr p) { return svol_ds_read_tr16_f16(p); }
};

#define SVOL_CHECK_LAUNCH()                                   \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return SVOL_E_LAUNCH; \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// SVOL_DETERMINISTIC=1 (read once): every reduction that normally meets its partial sums through floating-point atomics in ARRIVAL
// order takes a form with ONE adder per output element instead — the row-loop kernels (LayerNorm / gate backward, column sums) store
// per-workgroup partial rows that are folded in index order (DetScratch / det_fold below), the split weight-gradient GEMMs do not split the
// contraction, fused bias-gradient column sums become a separate folded pass, the attention backward uses its atomic-free kernels.  A test
// mode: the step is ~2 x slower (38 against 18 ms at cfg2), and bit-identical from run to run
// (tests/test_gpu_training.py::test_training_steps_are_bit_reproducible_in_deterministic_mode).
#include <cstdlib>
static inline bool svol_deterministic() {
    static const bool det = getenv("SVOL_DETERMINISTIC") != nullptr;
    return det;
}

// Two-level reductions of the deterministic mode: instead of adding its partial vector to the sink with atomics, a workgroup (or wave)
// stores it into a row of a stream-ordered scratch; det_fold then adds the rows to the sink in index order — one adder per element,
// the same order in every run, at the parallelism of the default mode.  (hipMallocAsync / hipFreeAsync: the scratch lives between
// the two launches on the caller's stream; the mode is a debugging aid, not for graph capture.)
struct DetScratch {
    float* p = nullptr;
    hipStream_t s;
    DetScratch(size_t floats, hipStream_t s_, bool zero = false) : s(s_) {
        if (floats == 0) return;   // the default (non-deterministic) mode asks for nothing: no runtime call on the hot launches
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;   // a debugging mode, not for graph capture: refuse (p stays null -> the entry returns an error)
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
        if (hipMallocAsync(reinterpret_cast<void**>(&p), floats * sizeof(float), s) != hipSuccess) p = nullptr;
        if (p && zero && hipMemsetAsync(p, 0, floats * sizeof(float), s) != hipSuccess) { (void)hipFreeAsync(p, s); p = nullptr; }
    }
    ~DetScratch() { if (p) (void)hipFreeAsync(p, s); }
    DetScratch(const DetScratch&) = delete;
    DetScratch& operator=(const DetScratch&) = delete;
};
#ifdef __HIPCC__
namespace {
// dst[b][j] += part[b][0][j] + part[b][1][j] + ... (rows `pstride` floats apart, batches bpart / bdst floats apart)
__global__ __launch_bounds__(256) void det_fold_kernel(const float* __restrict__ part, int parts, int64_t pstride, float* __restrict__ dst,
                                                       int64_t n, int64_t bpart, int64_t bdst) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float* q = part + (int64_t)blockIdx.y * bpart + j;
    float* d = dst + (int64_t)blockIdx.y * bdst + j;
    float acc = *d;
    for (int i = 0; i < parts; ++i) acc += q[(int64_t)i * pstride];
    *d = acc;
}
inline void det_fold(const float* part, int parts, int64_t pstride, float* dst, int64_t n, hipStream_t s, int batches = 1, int64_t bpart = 0,
                     int64_t bdst = 0) {
    if (!dst || n <= 0 || parts <= 0) return;
    hipLaunchKernelGGL(det_fold_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)batches), dim3(256), 0, s, part, parts, pstride, dst, n,
                       bpart, bdst);
}
}  // namespace
#endif

// v_permlane32_swap of a value with itself: lo = the value held by lane (i & 31), hi = the value held by lane (i | 32), in
// EVERY lane i.  The second operand is laundered through an empty asm: with two identical SSA operands hipcc (ROCm 7.2) folds the
// two results of the builtin into one register (v_permlane32_swap v1, v3 ; v_add_f32 v3, v1, v1) and the partner's value is lost.
struct HalfPair { unsigned lo, hi; };
__device__ __forceinline__ HalfPair swap_halves(unsigned x) {
    unsigned y = x;
    asm volatile("" : "+v"(y));
    const auto sw = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    return HalfPair{sw[0], sw[1]};
}

// ---- scalar conversions ---------------------------------------------------
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
__device__ __forceinline__ float to_f32(f16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float x) { return (f16_t)x; }

// ---- 4-element vector access (8 B for bf16, 16 B for f32) ------------------
template <typename T> struct Vec4;
template <> struct Vec4<float> {
    f32x4 v;
    __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const f32x4*>(p); }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<f32x4*>(p) = v; }
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec4<bf16_t> {
    bf16x4 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x4*>(p); }
    __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<bf16x4*>(p) = v; }
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16_t)x; }
};

template <> struct Vec4<f16_t> {
    f16x4 v;
    __device__ __forceinline__ void load(const f16_t* p) { v = *reinterpret_cast<const f16x4*>(p); }
    __device__ __forceinline__ void store(f16_t* p) const { *reinterpret_cast<f16x4*>(p) = v; }
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (f16_t)x; }
};

// ---- wave-level reductions (64 lanes) --------------------------------------
// Wave-wide butterfly reductions without the LDS crossbar: four DPP lane permutations inside the 16-lane rows (quad xor 1,
// quad xor 2, row_half_mirror, row_mirror — after the quad steps all lanes of a quad agree, so the mirrors pair disjoint
// groups), then v_permlane16_swap and v_permlane32_swap (gfx950) across rows.  Every lane ends with the full result.
// (__shfl_xor lowers to ds_bpermute_b32: six dependent LDS round trips per reduction, which bounded the row kernels.)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ HalfPair swap_rows16(unsigned x) {  // lo: rows 0,0,2,2   hi: rows 1,1,3,3  (16-lane rows)
    unsigned y = x;
    asm volatile("" : "+v"(y));
    const auto sw = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    return HalfPair{sw[0], sw[1]};
}
__device__ __forceinline__ float wave_sum(float x) {
    x += dpp_f32<0xB1>(x);
    x += dpp_f32<0x4E>(x);
    x += dpp_f32<0x141>(x);
    x += dpp_f32<0x140>(x);
    const HalfPair a = swap_rows16(__builtin_bit_cast(unsigned, x));
    x = __builtin_bit_cast(float, a.lo) + __builtin_bit_cast(float, a.hi);
    const HalfPair b = swap_halves(__builtin_bit_cast(unsigned, x));
    return __builtin_bit_cast(float, b.lo) + __builtin_bit_cast(float, b.hi);
}
__device__ __forceinline__ float wave_max(float x) {
    x = fmaxf(x, dpp_f32<0xB1>(x));
    x = fmaxf(x, dpp_f32<0x4E>(x));
    x = fmaxf(x, dpp_f32<0x141>(x));
    x = fmaxf(x, dpp_f32<0x140>(x));
    const HalfPair a = swap_rows16(__builtin_bit_cast(unsigned, x));
    x = fmaxf(__builtin_bit_cast(float, a.lo), __builtin_bit_cast(float, a.hi));
    const HalfPair b = swap_halves(__builtin_bit_cast(unsigned, x));
    return fmaxf(__builtin_bit_cast(float, b.lo), __builtin_bit_cast(float, b.hi));
}

// ---- counter-based RNG for dropout (stateless: same mask in fwd and bwd) ----
__device__ __forceinline__ uint32_t hash_u64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return (uint32_t)x;
}
// keep-mask scale of element (row r, column k) of a [rows, row_len] tensor: 0 (dropped) or 1/(1-p).  Round 4: the mask used to be a
// 64-bit hash of the flat element index — two 64 x 64 multiplies, i.e. ~8 quarter-rate integer multiplies per ELEMENT, five times
// the rest of an attention score's work (bench.py --workload encdec --dropout 0.1: 54 ms per step against 22 without dropout).
// Now: the seed goes through the 64-bit hash once per kernel (uniform), a row contributes one multiply that is loop-invariant
// wherever a lane walks along its row (and strength-reduced where it walks along rows), the column likewise, and a PAIR of neighbouring
// columns pays the two multiplies of a 32-bit finaliser (lowbias32): 16 mask bits per element.  Keep rate 0.9000 +- 2e-4 and neighbour / lag correlations within noise on
// 8M samples; tests/gpu_checks.py::check_dropout_mask pins the function against a numpy twin.
__device__ __forceinline__ uint32_t drop_seed32(uint64_t seed) { return hash_u64(seed * 0x9E3779B97F4A7C15ULL + 0x632BE59BD9B4E019ULL); }
__device__ __forceinline__ uint32_t drop_row(uint32_t s0, uint64_t r) {
    return s0 ^ ((uint32_t)r * 0x9E3779B1u) ^ ((uint32_t)(r >> 32) * 0x7F4A7C15u);
}
// 32 mask bits of the column PAIR (2 kh, 2 kh + 1) of a row: 16 per column
__device__ __forceinline__ uint32_t drop_bits(uint32_t rowmix, uint32_t khalf) {
    uint32_t x = rowmix ^ (khalf * 0x85EBCA6Bu);
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
// dropped iff the column's 16-bit field < ceil(p * 65536): drop probability p rounded UP to a multiple of 2^-16 (0.1 -> 0.100006)
__device__ __forceinline__ uint32_t drop_thr16(float p) { return (uint32_t)ceilf(p * 65536.0f); }
__device__ __forceinline__ float drop_pick(uint32_t bits, uint32_t odd, uint32_t thr16, float inv_keep) {
    const uint32_t u = odd ? bits >> 16 : bits & 0xffffu;
    return u < thr16 ? 0.0f : inv_keep;
}
__device__ __forceinline__ float drop_scale_rk(uint32_t rowmix, uint32_t k, float p, float inv_keep) {
    return drop_pick(drop_bits(rowmix, k >> 1), k & 1u, drop_thr16(p), inv_keep);
}
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t r, uint32_t k, float p, float inv_keep) {
    return drop_scale_rk(drop_row(drop_seed32(seed), r), k, p, inv_keep);
}

// erf-GELU for the bf16 GEMM epilogues: Phi(x) from the Abramowitz-Stegun 7.1.26 rational form of erfc
// (|abs error| < 1.5e-7 on Phi, no cancellation for x < 0), 2 transcendental + ~10 plain VALU ops instead of the
// branchy library erff.  The Gaussian exp(-x^2/2) it needs is the same one gelu'(x) needs.
__device__ __forceinline__ void gelu_parts(float x, float& Phi, float& E) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    E = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);  // exp(-x^2/2)
    float q = fmaf(1.061405429f, t, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f);
    const float h = 0.5f * q * t * E;  // = 0.5 erfc(|x|/sqrt2)
    Phi = x >= 0.f ? 1.0f - h : h;
}
__device__ __forceinline__ float gelu_fast(float x) {
    float Phi, E;
    gelu_parts(x, Phi, E);
    return x * Phi;
}
__device__ __forceinline__ float dgelu_fast(float x) {
    float Phi, E;
    gelu_parts(x, Phi, E);
    return fmaf(x * 0.39894228040143267794f, E, Phi);
}

// derivative of the MLP activation for the fused backward epilogue: aux is the PRE-activation for GELU and the
// POST-activation for ReLU (relu'(pre) = [hid > 0])
__device__ __forceinline__ float dact_fast(float aux, int act) {
    return act == SVOL_ACT_RELU ? (aux > 0.f ? 1.f : 0.f) : (act == SVOL_ACT_GELU_D ? aux : dgelu_fast(aux));
}
__device__ __forceinline__ bool act_is_gelu(int act) { return act == SVOL_ACT_GELU || act == SVOL_ACT_GELU_D; }
// what svol_gemm_nt saves next to a GELU output: the pre-activation, or (SVOL_ACT_GELU_D) its derivative there
__device__ __forceinline__ float pre_save_fast(float v, int act) { return act == SVOL_ACT_GELU_D ? dgelu_fast(v) : v; }

// exact (erf) GELU and derivative
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}
