// Training pieces of the convolutional backbone (VERDICT r4 "What's missing 2" / "Next round 8": the reference optimises EVERY parameter
// of build_model(args), train.py:72, and its backbone is torchvision's ResNet-34 / ResNet-18 in train mode, backbone.py:133-152 — i.e.
// BatchNorm normalises with the statistics of the batch and the convolutions have gradients).  The convolutions themselves stay the
// GEMMs of gemm*.hip / svol_conv_nhwc (forward: implicit GEMM; weight gradient: svol_gemm_tn(dz, im2col(x)); data gradient:
// svol_gemm_nt(dz, W^T) -> svol_col2im_nhwc).  Here: what stands between them, all on NHWC activations [M = n*h*w, C], 16-bit in
// memory, fp32 arithmetic, 8 channels (16 bytes) per thread:
//
//   svol_bn_colstats    per-channel sum(z - shift), sum((z - shift)^2)             (batch mean, then the centred second moment)
//   svol_bn_apply       y = act(z * scale[c] + shift[c] + residual)                (scale = gamma * rstd, shift = beta - mean * scale)
//   svol_bn_bwd_reduce  g = dy * [y > 0];  sum_g[c], sum_gx[c] = sum g * xhat      (xhat = (z - mean) * rstd)
//   svol_bn_bwd_apply   dz = gamma * rstd * (g - sum_g / M - xhat * sum_gx / M),  dres = g
//   svol_col2im_nhwc    dx[n, iy, ix, c] = sum over the (ky, kx) whose window holds (iy, ix) of dcols[(n, oy, ox), (ky, kx, c)]  (gather)
//   svol_maxpool_idx_nhwc / svol_maxpool_bwd_nhwc   nn.MaxPool2d with the window position of the FIRST maximum (PyTorch's tie rule:
//                       the scan keeps a later element only if it is strictly greater), and its gather-form backward
//
// The column reductions meet their workgroups' partials through fp32 atomics; under SVOL_DETERMINISTIC=1 through rows of a scratch
// folded in index order (common.h).
#include "common.h"

namespace {

template <typename T> struct V8;
template <> struct V8<bf16_t> { typedef bf16x8 type; };
template <> struct V8<f16_t> { typedef f16x8 type; };

template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&v)[8]) {
    const typename V8<T>::type r = *reinterpret_cast<const typename V8<T>::type*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)r[e];
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&v)[8]) {
    typename V8<T>::type r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (T)v[e];
    *reinterpret_cast<typename V8<T>::type*>(p) = r;
}

// Column reductions over [M, C]: thread -> (row lane = tid / G, channel group = tid % G), G = C / 8 groups of 8 channels, 256 / G rows
// per sweep; a workgroup owns `rows_per_wg` consecutive rows.  K partial sums per channel, folded over the row lanes in LDS, then one
// atomic per (k, channel) and workgroup — or the workgroup's row of the deterministic scratch.
template <int K>
__device__ __forceinline__ void col_reduce_finish(float (&acc)[K][8], int G, int C, float* const (&out)[K], float* det_part) {
    __shared__ float red[K][256 * 8];
    const int tid = threadIdx.x, g = tid % G, rl = tid / G, RL = 256 / G;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[k][(rl * G + g) * 8 + e] = acc[k][e];
    __syncthreads();
    for (int i = tid; i < K * C; i += 256) {
        const int k = i / C, c = i - k * C;
        float s = 0.f;
        for (int r = 0; r < RL; ++r) s += red[k][r * C + c];   // (row lane r holds channels [0, C) at r*G*8 = r*C)
        if (det_part) det_part[(int64_t)blockIdx.x * K * C + i] = s;
        else atomicAdd(out[k] + c, s);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_colstats_kernel(const T* __restrict__ z, const float* __restrict__ shift, float shift_scale, float* sum,
                                                          float* sumsq, int64_t M, int C, int64_t rows_per_wg, float* det_part) {
    const int G = C / 8, tid = threadIdx.x, g = tid % G, rl = tid / G, RL = 256 / G;
    float sh[8], acc[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sh[e] = shift ? shift[g * 8 + e] * shift_scale : 0.f; acc[0][e] = 0.f; acc[1][e] = 0.f; }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg, r1 = min(M, r0 + rows_per_wg);
    for (int64_t r = r0 + rl; r < r1; r += RL) {
        float v[8];
        load8(z + r * C + g * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[e] - sh[e]; acc[0][e] += d; acc[1][e] += d * d; }
    }
    float* const out[2] = {sum, sumsq};
    col_reduce_finish<2>(acc, G, C, out, det_part);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ z, const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const T* __restrict__ res, int relu, T* __restrict__ y, int64_t n8, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int c0 = (int)((i * 8) % C);
    float v[8], r[8];
    load8(z + i * 8, v);
    if (res) load8(res + i * 8, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float t = fmaf(v[e], scale[c0 + e], shift[c0 + e]);
        if (res) t += r[e];
        v[e] = relu ? fmaxf(t, 0.f) : t;
    }
    store8(y + i * 8, v);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ z,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd, float* sum_g,
                                                            float* sum_gx, int64_t M, int C, int64_t rows_per_wg, float* det_part) {
    const int G = C / 8, tid = threadIdx.x, g = tid % G, rl = tid / G, RL = 256 / G;
    float mu[8], rs[8], acc[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { mu[e] = mean[g * 8 + e]; rs[e] = rstd[g * 8 + e]; acc[0][e] = 0.f; acc[1][e] = 0.f; }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg, r1 = min(M, r0 + rows_per_wg);
    for (int64_t r = r0 + rl; r < r1; r += RL) {
        float d[8], zz[8], yy[8];
        load8(dy + r * C + g * 8, d);
        load8(z + r * C + g * 8, zz);
        if (y) load8(y + r * C + g * 8, yy);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gg = (!y || yy[e] > 0.f) ? d[e] : 0.f;
            acc[0][e] += gg;
            acc[1][e] += gg * ((zz[e] - mu[e]) * rs[e]);
        }
    }
    float* const out[2] = {sum_g, sum_gx};
    col_reduce_finish<2>(acc, G, C, out, det_part);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ sum_g,
                                                           const float* __restrict__ sum_gx, T* __restrict__ dz, T* __restrict__ dres,
                                                           int64_t n8, int C, float inv_m) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int c0 = (int)((i * 8) % C);
    float d[8], zz[8], yy[8], o[8];
    load8(dy + i * 8, d);
    load8(z + i * 8, zz);
    if (y) load8(y + i * 8, yy);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        const float gg = (!y || yy[e] > 0.f) ? d[e] : 0.f;
        const float xh = (zz[e] - mean[c]) * rstd[c];
        d[e] = gg;
        o[e] = gamma[c] * rstd[c] * (gg - sum_g[c] * inv_m - xh * (sum_gx[c] * inv_m));
    }
    store8(dz + i * 8, o);
    if (dres) store8(dres + i * 8, d);
}

// gather form of the transposed im2col: no atomics, every dx element has one owner
template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ dcols, int64_t ldcols, T* __restrict__ dx, int H, int W, int C,
                                                     int kh, int kw, int stride, int pad, int Ho, int Wo, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int G = C / 8;
    const int g = (int)(i % G);
    const int64_t pix = i / G;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H);
    const int64_t n = pix / ((int64_t)W * H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int ky = 0; ky < kh; ++ky) {
        const int ty = iy + pad - ky;
        if (ty < 0 || ty % stride) continue;
        const int oy = ty / stride;
        if (oy >= Ho) continue;
        for (int kx = 0; kx < kw; ++kx) {
            const int tx = ix + pad - kx;
            if (tx < 0 || tx % stride) continue;
            const int ox = tx / stride;
            if (ox >= Wo) continue;
            float v[8];
            load8(dcols + ((n * Ho + oy) * Wo + ox) * ldcols + (int64_t)(ky * kw + kx) * C + g * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
    }
    store8(dx + i * 8, acc);
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_idx_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int H, int W,
                                                          int C, int k, int stride, int pad, int Ho, int Wo, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int G = C / 8;
    const int g = (int)(i % G);
    const int64_t pix = i / G;
    const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho);
    const int64_t n = pix / ((int64_t)Wo * Ho);
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 255; }
    for (int ky = 0; ky < k; ++ky) {
        const int iy = oy * stride - pad + ky;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int ix = ox * stride - pad + kx;
            if (ix < 0 || ix >= W) continue;
            float v[8];
            load8(x + ((n * H + iy) * W + ix) * C + g * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (v[e] > best[e] || bi[e] == 255) { best[e] = v[e]; bi[e] = ky * k + kx; }   // first maximum of the scan wins (ATen max_pool2d)
        }
    }
    store8(y + i * 8, best);
    uint8_t* q = idx + i * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] = (uint8_t)bi[e];
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, T* __restrict__ dx, int H,
                                                          int W, int C, int k, int stride, int pad, int Ho, int Wo, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int G = C / 8;
    const int g = (int)(i % G);
    const int64_t pix = i / G;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H);
    const int64_t n = pix / ((int64_t)W * H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int ky = 0; ky < k; ++ky) {
        const int ty = iy + pad - ky;
        if (ty < 0 || ty % stride) continue;
        const int oy = ty / stride;
        if (oy >= Ho) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int tx = ix + pad - kx;
            if (tx < 0 || tx % stride) continue;
            const int ox = tx / stride;
            if (ox >= Wo) continue;
            const int64_t o = (((n * Ho + oy) * Wo + ox) * G + g) * 8;
            float v[8];
            load8(dy + o, v);
            const uint8_t* q = idx + o;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (q[e] == ky * k + kx) acc[e] += v[e];
        }
    }
    store8(dx + i * 8, acc);
}

// everything [C]-sized between the two column passes and the apply, in one launch: batch mean / rstd, the affine pair of svol_bn_apply,
// nn.BatchNorm2d's running-statistics update (momentum m, UNBIASED variance)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ sum, const float* __restrict__ sumsq_c,
                                                          const float* __restrict__ pivot, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* running_mean, float* running_var, float momentum,
                                                          float eps, float inv_m, float unbias, int C, float* mean, float* rstd, float* scale,
                                                          float* shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    // pivot == NULL: sum = sum z, sumsq_c = sum (z - mean)^2 (two passes).  pivot != NULL: ONE pass of sums about a per-channel pivot s
    // (a sample of the channel: |mean - s| is a few standard deviations at most, so the subtraction below loses a few bits, not the
    // variance): mean = s + sum / M, var = sumsq / M - (sum / M)^2
    const float d = sum[c] * inv_m;
    const float mu = pivot ? pivot[c] + d : d;
    const float var = pivot ? fmaxf(sumsq_c[c] * inv_m - d * d, 0.f) : sumsq_c[c] * inv_m;
    const float rs = 1.0f / sqrtf(var + eps);
    const float sc = gamma[c] * rs;
    mean[c] = mu;
    rstd[c] = rs;
    scale[c] = sc;
    shift[c] = beta[c] - mu * sc;
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mu;
    if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * var * unbias;
}

// fp32 [Cout, Cin, kh, kw] (nn.Conv2d) -> 16-bit [Cout, Kp], (ky, kx, c) order, columns K .. Kp-1 zero (the GEMMs' layout)
template <typename T>
__global__ __launch_bounds__(256) void conv_weight_pack_kernel(const float* __restrict__ w, T* __restrict__ out, int Cin, int khw, int K, int Kp,
                                                               int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t co = i / Kp;
    const int k = (int)(i - co * Kp);
    float v = 0.f;
    if (k < K) { const int c = k % Cin, t = k / Cin; v = w[(co * Cin + c) * khw + t]; }
    out[i] = (T)v;
}
// the weight gradient back: grad[Cout, Cin, kh, kw] (fp32) += dWp[Cout, Kp] (fp32, (ky, kx, c) order)
__global__ __launch_bounds__(256) void conv_weight_unpack_add_kernel(const float* __restrict__ dwp, float* __restrict__ grad, int Cin, int khw,
                                                                     int K, int Kp, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // over grad elements
    if (i >= n) return;
    const int t = (int)(i % khw);
    const int64_t r = i / khw;
    const int c = (int)(r % Cin);
    const int64_t co = r / Cin;
    grad[i] += dwp[co * Kp + (int64_t)t * Cin + c];
}
// the data gradient's weights for stride-1 convolutions: out[ci, (ky, kx, co)] = w[co, ci, kh-1-ky, kw-1-kx] (16-bit, Kp2 = pad32(kh*kw*Cout))
template <typename T>
__global__ __launch_bounds__(256) void conv_weight_flip_kernel(const float* __restrict__ w, T* __restrict__ out, int Cin, int Cout, int kh, int kw,
                                                               int Kp2, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t ci = i / Kp2;
    const int k = (int)(i - ci * Kp2);
    float v = 0.f;
    if (k < kh * kw * Cout) {
        const int co = k % Cout, t = k / Cout, ky = t / kw, kx = t % kw;
        v = w[(((int64_t)co * Cin + ci) * kh + (kh - 1 - ky)) * kw + (kw - 1 - kx)];
    }
    out[i] = (T)v;
}

bool chan_ok(int64_t C) { return C >= 8 && C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0; }
int64_t reduce_wgs(int64_t M, int64_t C, int64_t& rows_per_wg) {
    const int64_t RL = 256 / (C / 8);
    int64_t wgs = (M + RL * 8 - 1) / (RL * 8);   // >= 8 sweeps per workgroup
    if (wgs > 1024) wgs = 1024;
    if (wgs < 1) wgs = 1;
    rows_per_wg = (M + wgs - 1) / wgs;
    return (M + rows_per_wg - 1) / rows_per_wg;
}

}  // namespace

// gemm_tn_bf16.hip (compiled once per 16-bit type)
int svol_conv_wgrad_bf16_fast(const void* dz, const void* x, float* dwp, int64_t N, int64_t H, int64_t W, int64_t C, int64_t Cout, int64_t kh,
                              int64_t kw, int64_t stride, int64_t pad, int64_t Kp, hipStream_t stream);
int svol_conv_wgrad_f16_fast(const void* dz, const void* x, float* dwp, int64_t N, int64_t H, int64_t W, int64_t C, int64_t Cout, int64_t kh,
                             int64_t kw, int64_t stride, int64_t pad, int64_t Kp, hipStream_t stream);

extern "C" {

int svol_conv_wgrad_nhwc(const void* dz, const void* x, float* dwp, int64_t N, int64_t H, int64_t W, int64_t C, int64_t Cout, int64_t kh, int64_t kw,
                         int64_t stride, int64_t pad, int64_t Kp, int dtype, void* stream) {
    if (!dz || !x || !dwp || N <= 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return SVOL_E_INVALID;
    if (!svol_is16(dtype)) return SVOL_E_UNSUPPORTED;
    return (dtype == SVOL_BF16 ? svol_conv_wgrad_bf16_fast : svol_conv_wgrad_f16_fast)(dz, x, dwp, N, H, W, C, Cout, kh, kw, stride, pad, Kp,
                                                                                       reinterpret_cast<hipStream_t>(stream));
}

int svol_bn_colstats(const void* z, const float* shift, float shift_scale, float* sum, float* sumsq, int64_t M, int64_t C, int dtype,
                     void* stream) {
    if (!z || !sum || !sumsq || M <= 0) return SVOL_E_INVALID;
    if (!chan_ok(C) || !svol_is16(dtype) || M > (1ll << 40)) return SVOL_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int64_t rpw;
    const int64_t wgs = reduce_wgs(M, C, rpw);
    const bool det_mode = svol_deterministic();
    DetScratch det(det_mode ? (size_t)(wgs * 2 * C) : 0, s);
    if (det_mode && !det.p) return SVOL_E_LAUNCH;
    if (dtype == SVOL_BF16)
        hipLaunchKernelGGL(bn_colstats_kernel<bf16_t>, dim3((unsigned)wgs), dim3(256), 0, s, (const bf16_t*)z, shift, shift_scale, sum, sumsq, M, (int)C, rpw, det.p);
    else
        hipLaunchKernelGGL(bn_colstats_kernel<f16_t>, dim3((unsigned)wgs), dim3(256), 0, s, (const f16_t*)z, shift, shift_scale, sum, sumsq, M, (int)C, rpw, det.p);
    if (det_mode) {
        det_fold(det.p, (int)wgs, 2 * C, sum, C, s);
        det_fold(det.p + C, (int)wgs, 2 * C, sumsq, C, s);
    }
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_bn_finalize(const float* sum, const float* sumsq_centered, const float* pivot, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, int64_t M, int64_t C, float* mean, float* rstd,
                     float* scale, float* shift, void* stream) {
    if (!sum || !sumsq_centered || !gamma || !beta || !mean || !rstd || !scale || !shift || M <= 0 || C <= 0) return SVOL_E_INVALID;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), sum,
                       sumsq_centered, pivot, gamma, beta, running_mean, running_var, momentum, eps, 1.0f / (float)M,
                       M > 1 ? (float)M / (float)(M - 1) : 1.0f, (int)C, mean, rstd, scale, shift);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_conv_weight_pack(const float* w, void* out, int64_t Cout, int64_t Cin, int64_t kh, int64_t kw, int64_t Kp, int flip, int dtype,
                          void* stream) {
    if (!w || !out || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0) return SVOL_E_INVALID;
    if (!svol_is16(dtype)) return SVOL_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!flip) {
        const int64_t K = kh * kw * Cin;
        if (Kp < K) return SVOL_E_INVALID;
        const int64_t n = Cout * Kp;
        const unsigned g = (unsigned)((n + 255) / 256);
        if (dtype == SVOL_BF16) hipLaunchKernelGGL(conv_weight_pack_kernel<bf16_t>, dim3(g), dim3(256), 0, s, w, (bf16_t*)out, (int)Cin, (int)(kh * kw), (int)K, (int)Kp, n);
        else hipLaunchKernelGGL(conv_weight_pack_kernel<f16_t>, dim3(g), dim3(256), 0, s, w, (f16_t*)out, (int)Cin, (int)(kh * kw), (int)K, (int)Kp, n);
    } else {
        if (Kp < kh * kw * Cout) return SVOL_E_INVALID;
        const int64_t n = Cin * Kp;
        const unsigned g = (unsigned)((n + 255) / 256);
        if (dtype == SVOL_BF16) hipLaunchKernelGGL(conv_weight_flip_kernel<bf16_t>, dim3(g), dim3(256), 0, s, w, (bf16_t*)out, (int)Cin, (int)Cout, (int)kh, (int)kw, (int)Kp, n);
        else hipLaunchKernelGGL(conv_weight_flip_kernel<f16_t>, dim3(g), dim3(256), 0, s, w, (f16_t*)out, (int)Cin, (int)Cout, (int)kh, (int)kw, (int)Kp, n);
    }
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_conv_weight_unpack_add(const float* dwp, float* grad, int64_t Cout, int64_t Cin, int64_t kh, int64_t kw, int64_t Kp, void* stream) {
    if (!dwp || !grad || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0 || Kp < kh * kw * Cin) return SVOL_E_INVALID;
    const int64_t n = Cout * Cin * kh * kw;
    hipLaunchKernelGGL(conv_weight_unpack_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dwp,
                       grad, (int)Cin, (int)(kh * kw), (int)(kh * kw * Cin), (int)Kp, n);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_bn_apply(const void* z, const float* scale, const float* shift, const void* residual, int relu, void* y, int64_t M, int64_t C,
                  int dtype, void* stream) {
    if (!z || !scale || !shift || !y || M <= 0) return SVOL_E_INVALID;
    if (C < 8 || C % 8 || !svol_is16(dtype)) return SVOL_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t n8 = M * C / 8;
    const unsigned g = (unsigned)((n8 + 255) / 256);
    if (dtype == SVOL_BF16)
        hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)z, scale, shift, (const bf16_t*)residual, relu, (bf16_t*)y, n8, (int)C);
    else
        hipLaunchKernelGGL(bn_apply_kernel<f16_t>, dim3(g), dim3(256), 0, s, (const f16_t*)z, scale, shift, (const f16_t*)residual, relu, (f16_t*)y, n8, (int)C);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_bn_bwd_reduce(const void* dy, const void* y, const void* z, const float* mean, const float* rstd, float* sum_g, float* sum_gx,
                       int64_t M, int64_t C, int dtype, void* stream) {
    if (!dy || !z || !mean || !rstd || !sum_g || !sum_gx || M <= 0) return SVOL_E_INVALID;
    if (!chan_ok(C) || !svol_is16(dtype)) return SVOL_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int64_t rpw;
    const int64_t wgs = reduce_wgs(M, C, rpw);
    const bool det_mode = svol_deterministic();
    DetScratch det(det_mode ? (size_t)(wgs * 2 * C) : 0, s);
    if (det_mode && !det.p) return SVOL_E_LAUNCH;
    if (dtype == SVOL_BF16)
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3((unsigned)wgs), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)z, mean, rstd, sum_g, sum_gx, M, (int)C, rpw, det.p);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<f16_t>, dim3((unsigned)wgs), dim3(256), 0, s, (const f16_t*)dy, (const f16_t*)y, (const f16_t*)z, mean, rstd, sum_g, sum_gx, M, (int)C, rpw, det.p);
    if (det_mode) {
        det_fold(det.p, (int)wgs, 2 * C, sum_g, C, s);
        det_fold(det.p + C, (int)wgs, 2 * C, sum_gx, C, s);
    }
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_bn_bwd_apply(const void* dy, const void* y, const void* z, const float* mean, const float* rstd, const float* gamma,
                      const float* sum_g, const float* sum_gx, void* dz, void* dres, int64_t M, int64_t C, int dtype, void* stream) {
    if (!dy || !z || !mean || !rstd || !gamma || !sum_g || !sum_gx || !dz || M <= 0) return SVOL_E_INVALID;
    if (C < 8 || C % 8 || !svol_is16(dtype)) return SVOL_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t n8 = M * C / 8;
    const unsigned g = (unsigned)((n8 + 255) / 256);
    const float inv_m = 1.0f / (float)M;
    if (dtype == SVOL_BF16)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)z, mean, rstd, gamma, sum_g, sum_gx, (bf16_t*)dz, (bf16_t*)dres, n8, (int)C, inv_m);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<f16_t>, dim3(g), dim3(256), 0, s, (const f16_t*)dy, (const f16_t*)y, (const f16_t*)z, mean, rstd, gamma, sum_g, sum_gx, (f16_t*)dz, (f16_t*)dres, n8, (int)C, inv_m);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_col2im_nhwc(const void* dcols, int64_t ldcols, void* dx, int64_t N, int64_t H, int64_t W, int64_t C, int64_t kh, int64_t kw,
                     int64_t stride, int64_t pad, int dtype, void* stream) {
    if (!dcols || !dx || N <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return SVOL_E_INVALID;
    if (C < 8 || C % 8 || ldcols % 8 || ldcols < kh * kw * C || !svol_is16(dtype)) return SVOL_E_UNSUPPORTED;
    const int64_t Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return SVOL_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t n8 = N * H * W * C / 8;
    const unsigned g = (unsigned)((n8 + 255) / 256);
    if (dtype == SVOL_BF16)
        hipLaunchKernelGGL(col2im_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)dcols, ldcols, (bf16_t*)dx, (int)H, (int)W, (int)C, (int)kh, (int)kw, (int)stride, (int)pad, (int)Ho, (int)Wo, n8);
    else
        hipLaunchKernelGGL(col2im_kernel<f16_t>, dim3(g), dim3(256), 0, s, (const f16_t*)dcols, ldcols, (f16_t*)dx, (int)H, (int)W, (int)C, (int)kh, (int)kw, (int)stride, (int)pad, (int)Ho, (int)Wo, n8);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_maxpool_idx_nhwc(const void* x, void* y, uint8_t* idx, int64_t N, int64_t H, int64_t W, int64_t C, int64_t k, int64_t stride,
                          int64_t pad, int dtype, void* stream) {
    if (!x || !y || !idx || N <= 0 || H <= 0 || W <= 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0) return SVOL_E_INVALID;
    if (C < 8 || C % 8 || !svol_is16(dtype)) return SVOL_E_UNSUPPORTED;
    const int64_t Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return SVOL_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t n8 = N * Ho * Wo * C / 8;
    const unsigned g = (unsigned)((n8 + 255) / 256);
    if (dtype == SVOL_BF16)
        hipLaunchKernelGGL(maxpool_idx_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, idx, (int)H, (int)W, (int)C, (int)k, (int)stride, (int)pad, (int)Ho, (int)Wo, n8);
    else
        hipLaunchKernelGGL(maxpool_idx_kernel<f16_t>, dim3(g), dim3(256), 0, s, (const f16_t*)x, (f16_t*)y, idx, (int)H, (int)W, (int)C, (int)k, (int)stride, (int)pad, (int)Ho, (int)Wo, n8);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_maxpool_bwd_nhwc(const void* dy, const uint8_t* idx, void* dx, int64_t N, int64_t H, int64_t W, int64_t C, int64_t k,
                          int64_t stride, int64_t pad, int dtype, void* stream) {
    if (!dy || !idx || !dx || N <= 0 || H <= 0 || W <= 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0) return SVOL_E_INVALID;
    if (C < 8 || C % 8 || !svol_is16(dtype)) return SVOL_E_UNSUPPORTED;
    const int64_t Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return SVOL_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t n8 = N * H * W * C / 8;
    const unsigned g = (unsigned)((n8 + 255) / 256);
    if (dtype == SVOL_BF16)
        hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)dy, idx, (bf16_t*)dx, (int)H, (int)W, (int)C, (int)k, (int)stride, (int)pad, (int)Ho, (int)Wo, n8);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<f16_t>, dim3(g), dim3(256), 0, s, (const f16_t*)dy, idx, (f16_t*)dx, (int)H, (int)W, (int)C, (int)k, (int)stride, (int)pad, (int)Ho, (int)Wo, n8);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

}  // extern "C"
