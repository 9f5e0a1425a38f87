// Convolutional backbone pieces (torchvision ResNet-18/34, backbone.py:65-89,133-152; inference / frozen):
// a convolution is  im2col -> the MFMA GEMMs of gemm*.hip  with the folded BatchNorm shift (+ identity) (+ ReLU) in the GEMM
// epilogue; activations stay NHWC ([n*h*w, c] row-major = the GEMM's own output layout) from the stem to the tokens.
//
//   svol_im2col         cols[(n,ho,wo), (ky,kx,c)] = x[n, ho*s-p+ky, wo*s-p+kx, c]  (0 outside), K padded with zeros to ldcols.
//                       Source given by element strides, so the NCHW fp32 pixel tensor of the stem and the NHWC activations use
//                       the same entry point; NHWC sources with c % 8 == 0 take a 16-byte-per-thread path.
//   svol_maxpool_nhwc   k x k / stride / pad max pooling (the stem's 3x3 s2 p1), 8 channels per thread
//   svol_avgpool_nhwc   mean over the h*w positions -> [n, c] fp32 (the sketch branch keeps torchvision's avgpool)
#include "common.h"

namespace {

template <typename TS, typename TD>
__global__ void im2col_generic_kernel(const TS* __restrict__ x, int64_t sn, int64_t sh, int64_t sw, int64_t sc, TD* __restrict__ cols,
                                      int64_t ldcols, int H, int W, int C, int kh, int kw, int stride, int pad, int Ho, int Wo,
                                      int64_t rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (row, padded column)
    if (i >= rows * ldcols) return;
    const int64_t row = i / ldcols;
    const int k = (int)(i - row * ldcols);
    float v = 0.f;
    if (k < kh * kw * C) {
        const int c = k % C, kx = (k / C) % kw, ky = k / (C * kw);
        const int wo = (int)(row % Wo), ho = (int)((row / Wo) % Ho);
        const int64_t n = row / ((int64_t)Wo * Ho);
        const int hi = ho * stride - pad + ky, wi = wo * stride - pad + kx;
        if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = to_f32(x[n * sn + hi * sh + wi * sw + c * sc]);
    }
    cols[i] = from_f32<TD>(v);
}

// The stem: few channels (C <= 4), any source strides (NCHW fp32 pixels).  A workgroup owns WT consecutive output positions of
// one output row: the kh input rows they touch are staged in LDS with reads that run along the image row (coalesced in the
// source), then the WT destination rows are written as 16-byte chunks (coalesced in the destination).
constexpr int STEM_WT = 32;
template <typename TS>
__global__ __launch_bounds__(256) void im2col_stem_kernel(const TS* __restrict__ x, int64_t sn, int64_t sh, int64_t sw, int64_t sc,
                                                           bf16_t* __restrict__ cols, int64_t ldcols, int H, int W, int C, int kh,
                                                           int kw, int stride, int pad, int Ho, int Wo) {
    extern __shared__ float tile[];  // [C][kh][span]
    const int span = (STEM_WT - 1) * stride + kw;
    const int wo0 = blockIdx.x * STEM_WT, ho = blockIdx.y;
    const int64_t n = blockIdx.z;
    const int wi0 = wo0 * stride - pad, hi0 = ho * stride - pad;
    for (int i = threadIdx.x; i < C * kh * span; i += 256) {
        const int xx = i % span, ky = (i / span) % kh, c = i / (span * kh);
        const int hi = hi0 + ky, wi = wi0 + xx;
        tile[i] = (hi >= 0 && hi < H && wi >= 0 && wi < W) ? to_f32(x[n * sn + hi * sh + wi * sw + c * sc]) : 0.f;
    }
    __syncthreads();
    const int K = kh * kw * C;
    const int chunks = (int)(ldcols / 8);
    for (int j = threadIdx.x; j < STEM_WT * chunks; j += 256) {
        const int r = j / chunks, k8 = (j - r * chunks) * 8;
        if (wo0 + r >= Wo) continue;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k8 + e;
            float v = 0.f;
            if (k < K) {
                const int c = k % C, kx = (k / C) % kw, ky = k / (C * kw);
                v = tile[(c * kh + ky) * span + r * stride + kx];
            }
            o[e] = (bf16_t)v;
        }
        *reinterpret_cast<bf16x8*>(cols + ((n * Ho + ho) * (int64_t)Wo + wo0 + r) * ldcols + k8) = o;
    }
}

// NHWC contiguous source, C % 8 == 0 (bf16) / C % 4 == 0 (fp32): one 16-byte chunk per thread, consecutive threads walk
// c, then kx, then ky of one output position (contiguous in the destination row)
template <typename T>
__global__ void im2col_nhwc_vec_kernel(const T* __restrict__ x, T* __restrict__ cols, int64_t ldcols, int H, int W, int C, int kh,
                                       int kw, int stride, int pad, int Ho, int Wo, int64_t rows) {
    constexpr int V = 16 / sizeof(T);
    const int cv = C / V;
    const int64_t per_row = (int64_t)kh * kw * cv;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * per_row) return;
    const int64_t row = i / per_row;
    const int r = (int)(i - row * per_row);
    const int c = (r % cv) * V, kx = (r / cv) % kw, ky = r / (cv * kw);
    const int wo = (int)(row % Wo), ho = (int)((row / Wo) % Ho);
    const int64_t n = row / ((int64_t)Wo * Ho);
    const int hi = ho * stride - pad + ky, wi = wo * stride - pad + kx;
    uint4 v = {0u, 0u, 0u, 0u};
    if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = *reinterpret_cast<const uint4*>(x + ((n * H + hi) * (int64_t)W + wi) * C + c);
    *reinterpret_cast<uint4*>(cols + row * ldcols + ((int64_t)ky * kw + kx) * C + c) = v;
}

template <typename T>
__global__ void zero_tail_kernel(T* __restrict__ cols, int64_t ldcols, int k0, int64_t rows) {
    const int tail = (int)(ldcols - k0);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * tail) return;
    cols[(i / tail) * ldcols + k0 + (i % tail)] = from_f32<T>(0.f);
}

template <typename T>
__global__ void maxpool_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C, int k, int stride, int pad,
                                    int Ho, int Wo, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (n, ho, wo, 8 channels)
    if (i >= total) return;
    const int cv = C / 8;
    const int c = (int)(i % cv) * 8;
    const int64_t pos = i / cv;
    const int wo = (int)(pos % Wo), ho = (int)((pos / Wo) % Ho);
    const int64_t n = pos / ((int64_t)Wo * Ho);
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int ky = 0; ky < k; ++ky) {
        const int hi = ho * stride - pad + ky;
        if (hi < 0 || hi >= H) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int wi = wo * stride - pad + kx;
            if (wi < 0 || wi >= W) continue;
            const T* src = x + ((n * H + hi) * (int64_t)W + wi) * C + c;
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], to_f32(src[e]));
        }
    }
    T* dst = y + pos * C + c;
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = from_f32<T>(m[e]);
}

template <typename T>
__global__ void avgpool_nhwc_kernel(const T* __restrict__ x, float* __restrict__ y, int HW, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;  // grid (ceil(C/256), n): coalesced over channels
    if (c >= C) return;
    const T* src = x + (int64_t)blockIdx.y * HW * C + c;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += to_f32(src[(int64_t)p * C]);
    y[(int64_t)blockIdx.y * C + c] = s / (float)HW;
}

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int svol_im2col(const void* x, int64_t sn, int64_t sh, int64_t sw, int64_t sc, int src_dtype, void* cols, int64_t ldcols,
                           int64_t N, int64_t H, int64_t W, int64_t C, int64_t kh, int64_t kw, int64_t stride, int64_t pad, int dtype,
                           void* stream) {
    if (!x || !cols || N < 0 || H <= 0 || W <= 0 || C <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return SVOL_E_INVALID;
    const int64_t Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    const int64_t K = kh * kw * C;
    if (Ho <= 0 || Wo <= 0 || ldcols < K) return SVOL_E_INVALID;
    const int64_t rows = N * Ho * Wo;
    if (rows == 0) return SVOL_OK;
    if (rows * ldcols / 256 >= (1ll << 31)) return SVOL_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool nhwc = sc == 1 && sw == C && sh == W * C && sn == H * W * C;
    const int V = dtype == SVOL_BF16 ? 8 : 4;
    if (nhwc && src_dtype == dtype && C % V == 0 && ldcols % V == 0 && aligned16(x) && aligned16(cols)) {
        const int64_t total = rows * kh * kw * (C / V);
        if (dtype == SVOL_BF16)
            hipLaunchKernelGGL(im2col_nhwc_vec_kernel<bf16_t>, dim3(nblk(total)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)cols, ldcols,
                               (int)H, (int)W, (int)C, (int)kh, (int)kw, (int)stride, (int)pad, (int)Ho, (int)Wo, rows);
        else
            hipLaunchKernelGGL(im2col_nhwc_vec_kernel<float>, dim3(nblk(total)), dim3(256), 0, s, (const float*)x, (float*)cols, ldcols,
                               (int)H, (int)W, (int)C, (int)kh, (int)kw, (int)stride, (int)pad, (int)Ho, (int)Wo, rows);
        if (ldcols > K) {
            if (dtype == SVOL_BF16)
                hipLaunchKernelGGL(zero_tail_kernel<bf16_t>, dim3(nblk(rows * (ldcols - K))), dim3(256), 0, s, (bf16_t*)cols, ldcols, (int)K, rows);
            else
                hipLaunchKernelGGL(zero_tail_kernel<float>, dim3(nblk(rows * (ldcols - K))), dim3(256), 0, s, (float*)cols, ldcols, (int)K, rows);
        }
        SVOL_CHECK_LAUNCH();
        return SVOL_OK;
    }
    if (dtype == SVOL_BF16 && C <= 4 && ldcols % 8 == 0 && aligned16(cols) && Ho <= 65535 && N <= 65535) {
        const size_t sh_bytes = (size_t)C * kh * ((STEM_WT - 1) * stride + kw) * sizeof(float);
        if (sh_bytes <= 64 * 1024) {
            const dim3 gs((unsigned)((Wo + STEM_WT - 1) / STEM_WT), (unsigned)Ho, (unsigned)N);
            if (src_dtype == SVOL_F32)
                hipLaunchKernelGGL(im2col_stem_kernel<float>, gs, dim3(256), sh_bytes, s, (const float*)x, sn, sh, sw, sc, (bf16_t*)cols,
                                   ldcols, (int)H, (int)W, (int)C, (int)kh, (int)kw, (int)stride, (int)pad, (int)Ho, (int)Wo);
            else if (src_dtype == SVOL_BF16)
                hipLaunchKernelGGL(im2col_stem_kernel<bf16_t>, gs, dim3(256), sh_bytes, s, (const bf16_t*)x, sn, sh, sw, sc, (bf16_t*)cols,
                                   ldcols, (int)H, (int)W, (int)C, (int)kh, (int)kw, (int)stride, (int)pad, (int)Ho, (int)Wo);
            else return SVOL_E_INVALID;
            SVOL_CHECK_LAUNCH();
            return SVOL_OK;
        }
    }
    const dim3 g(nblk(rows * ldcols));
#define SVOL_I2C(TS, TD)                                                                                                             \
    hipLaunchKernelGGL((im2col_generic_kernel<TS, TD>), g, dim3(256), 0, s, (const TS*)x, sn, sh, sw, sc, (TD*)cols, ldcols, (int)H, \
                       (int)W, (int)C, (int)kh, (int)kw, (int)stride, (int)pad, (int)Ho, (int)Wo, rows)
    if (src_dtype == SVOL_F32 && dtype == SVOL_BF16) SVOL_I2C(float, bf16_t);
    else if (src_dtype == SVOL_F32 && dtype == SVOL_F32) SVOL_I2C(float, float);
    else if (src_dtype == SVOL_BF16 && dtype == SVOL_BF16) SVOL_I2C(bf16_t, bf16_t);
    else return SVOL_E_INVALID;
#undef SVOL_I2C
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

extern "C" int svol_maxpool_nhwc(const void* x, void* y, int64_t N, int64_t H, int64_t W, int64_t C, int64_t k, int64_t stride,
                                 int64_t pad, int dtype, void* stream) {
    if (!x || !y || N < 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || stride <= 0 || pad < 0 || 2 * pad > k) return SVOL_E_INVALID;
    if (C % 8) return SVOL_E_UNSUPPORTED;
    const int64_t Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const int64_t total = N * Ho * Wo * (C / 8);
    if (total == 0) return SVOL_OK;
    if (total / 256 >= (1ll << 31)) return SVOL_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SVOL_BF16)
        hipLaunchKernelGGL(maxpool_nhwc_kernel<bf16_t>, dim3(nblk(total)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, (int)H, (int)W,
                           (int)C, (int)k, (int)stride, (int)pad, (int)Ho, (int)Wo, total);
    else if (dtype == SVOL_F32)
        hipLaunchKernelGGL(maxpool_nhwc_kernel<float>, dim3(nblk(total)), dim3(256), 0, s, (const float*)x, (float*)y, (int)H, (int)W,
                           (int)C, (int)k, (int)stride, (int)pad, (int)Ho, (int)Wo, total);
    else return SVOL_E_INVALID;
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

extern "C" int svol_avgpool_nhwc(const void* x, float* y, int64_t N, int64_t HW, int64_t C, int dtype, void* stream) {
    if (!x || !y || N < 0 || HW <= 0 || C <= 0) return SVOL_E_INVALID;
    if (N == 0) return SVOL_OK;
    if (N > 65535) return SVOL_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 g((unsigned)((C + 255) / 256), (unsigned)N);
    if (dtype == SVOL_BF16) hipLaunchKernelGGL(avgpool_nhwc_kernel<bf16_t>, g, dim3(256), 0, s, (const bf16_t*)x, y, (int)HW, (int)C);
    else if (dtype == SVOL_F32) hipLaunchKernelGGL(avgpool_nhwc_kernel<float>, g, dim3(256), 0, s, (const float*)x, y, (int)HW, (int)C);
    else return SVOL_E_INVALID;
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}
