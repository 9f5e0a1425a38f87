// LayerNorm (+dropout, +positional add) and sine positional encoding — HBM-bound row kernels.
// One wave (64 lanes) per row, 4 contiguous elements per lane per pass (8 B bf16 / 16 B f32
// coalesced loads), fp32 statistics, wave shuffle reductions (no LDS).
#include "common.h"

namespace {

constexpr int LN_MAX_PASSES = 4;  // D <= 4 * 64 * 4 = 1024

// T  = element type of the compute-dtype outputs / gradients, TX = element type of the LN input
// (float when the input is the fp32 residual stream).
template <typename T, typename TX, int NP>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y32,
                                                     T* __restrict__ y, T* __restrict__ ypos, const T* __restrict__ pos,
                                                     int64_t pos_rows, float* __restrict__ mean, float* __restrict__ rstd,
                                                     int64_t M, int D, float p, float inv_keep, uint64_t seed0,
                                                     const int64_t* __restrict__ soff) {
    const uint64_t seed = seed0 + (soff ? ((uint64_t)soff[0] << 8) : 0ull);  // device-side step counter (same stride as the host-side one): graph replays draw fresh masks
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const TX* xr = x + row * D;
    float v[NP][4];
    float s = 0.f;
    // every global load of the wave's one row goes out before the first reduction: one memory latency per wave, not two
    // (x, then pos / gamma / beta behind the statistics: 39 us per [50176, 256] launch; NP = passes of 256 columns)
    const T* pr = pos ? pos + (row % pos_rows) * D : nullptr;
    Vec4<TX> t[NP];
    Vec4<T> pv[NP];
    Vec4<float> gv[NP], bv[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < D) {
            t[j].load(xr + c);
            if (pr) pv[j].load(pr + c);
            gv[j].load(gamma + c);
            bv[j].load(beta + c);
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int c = (lane + 64 * j) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[j][e] = c < D ? t[j].get(e) : 0.f;
            s += v[j][e];
        }
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < D) {
            Vec4<T> o, op;
            Vec4<float> o32;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float r = (v[j][e] - mu) * rs * gv[j].get(e) + bv[j].get(e);
                if (p > 0.f) r *= dropout_scale(seed, (uint64_t)row, (uint32_t)(c + e), p, inv_keep);
                o.set(e, r);
                o32.set(e, r);
                if (pr) op.set(e, r + pv[j].get(e));
            }
            if (y32) o32.store(y32 + row * D + c);
            if (y) o.store(y + row * D + c);
            if (ypos) op.store(ypos + row * D + c);
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = (dy32 + dy + dypos) * gamma (dropout mask applied first)
template <typename T, typename TX, int NP>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy32, const T* __restrict__ dy,
                                                     const T* __restrict__ dy2, const TX* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float* __restrict__ dx32,
                                                     T* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ dxsum, int64_t M, int D,
                                                     float p, float inv_keep, uint64_t seed0, const int64_t* __restrict__ soff, int rows_per_wave,
                                                     float* __restrict__ det_part) {
    const uint64_t seed = seed0 + (soff ? ((uint64_t)soff[0] << 8) : 0ull);
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t r0 = wave_id * rows_per_wave;
    float dg[NP][4], db[NP][4], dxs[NP][4], gm[NP][4];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int c = (lane + 64 * j) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { dg[j][e] = 0.f; db[j][e] = 0.f; dxs[j][e] = 0.f; gm[j][e] = c < D ? gamma[c + e] : 0.f; }
    }
    // The row loop is a dependent load -> reduce -> store chain per wave: the NEXT row's operand vectors and statistics are
    // fetched before the current row is reduced (round 2: 58 -> 4x us per [50176, 256] launch)
    struct Row { Vec4<float> a32[NP]; Vec4<T> a[NP], b[NP]; Vec4<TX> xv[NP]; float mu, rs; };
    auto fetch = [&](int64_t row, Row& r) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (c < D) {
                if (dy32) r.a32[j].load(dy32 + row * D + c);
                if (dy) r.a[j].load(dy + row * D + c);
                if (dy2) r.b[j].load(dy2 + row * D + c);
                r.xv[j].load(x + row * D + c);
            }
        }
        r.mu = mean[row];
        r.rs = rstd[row];
    };
    const int64_t rend = (r0 + rows_per_wave < M) ? r0 + rows_per_wave : M;
    Row cur, nxt;
    if (r0 < rend) fetch(r0, cur);
    for (int64_t row = r0; row < rend; ++row) {
        if (row + 1 < rend) fetch(row + 1, nxt);
        const float mu = cur.mu, rs = cur.rs;
        float g[NP][4], xh[NP][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (c < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d = (dy32 ? cur.a32[j].get(e) : 0.f) + (dy ? cur.a[j].get(e) : 0.f) + (dy2 ? cur.b[j].get(e) : 0.f);
                    if (p > 0.f) d *= dropout_scale(seed, (uint64_t)row, (uint32_t)(c + e), p, inv_keep);
                    const float h = (cur.xv[j].get(e) - mu) * rs;
                    xh[j][e] = h;
                    dg[j][e] += d * h;
                    db[j][e] += d;
                    const float gg = d * gm[j][e];
                    g[j][e] = gg;
                    s1 += gg;
                    s2 += gg * h;
                }
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (c < D) {
                Vec4<T> o;
                Vec4<float> o32;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r = rs * (g[j][e] - s1 - xh[j][e] * s2);
                    dxs[j][e] += r;
                    o.set(e, r);
                    o32.set(e, r);
                }
                if (dx32) o32.store(dx32 + row * D + c);
                if (dx) o.store(dx + row * D + c);
            }
        }
        cur = nxt;
    }
    // one set of atomics per WORKGROUP: the 4 waves' partial sums are combined in LDS first (every wave
    // of every workgroup adding to the same D addresses is the contended-atomic worst case)
    __shared__ float red[3][NP * 256];
    const int wave = threadIdx.x >> 6;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int c = (lane + 64 * j) * 4;
                if (c < D) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (w == 0) { red[0][c + e] = dg[j][e]; red[1][c + e] = db[j][e]; red[2][c + e] = dxs[j][e]; }
                        else { red[0][c + e] += dg[j][e]; red[1][c + e] += db[j][e]; red[2][c + e] += dxs[j][e]; }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (det_part) {   // deterministic mode: this workgroup's row of the scratch, folded in index order by det_fold
        float* row = det_part + (int64_t)blockIdx.x * 3 * D;
        for (int c = threadIdx.x; c < D; c += 256) { row[c] = red[0][c]; row[D + c] = red[1][c]; row[2 * D + c] = red[2][c]; }
        return;
    }
    for (int c = threadIdx.x; c < D; c += 256) {
        atomicAdd(dgamma + c, red[0][c]);
        atomicAdd(dbeta + c, red[1][c]);
        if (dxsum) atomicAdd(dxsum + c, red[2][c]);
    }
}

// position_encoding.py:51-71 — one workgroup per batch row: inclusive scan of the mask over L,
// then pos[l, 2i] = sin(x / t_i), pos[l, 2i+1] = cos(x / t_i) with x = cumsum/(last+1e-6)*2pi.
// grid = (ceil(L/64), B): every block re-reduces the (tiny) mask prefix it needs, then writes 64 tokens.
template <typename T>
__global__ __launch_bounds__(256) void posenc_kernel(const float* __restrict__ mask, T* __restrict__ pos, int L, int D) {
    __shared__ float red[2][4];
    __shared__ float xe[64];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* mr = mask + (int64_t)b * L;
    const int l0 = blockIdx.x * 64;
    float before = 0.f, all = 0.f;
    for (int l = tid; l < L; l += 256) {
        const float v = (mr[l] != 0.f) ? 1.f : 0.f;
        all += v;
        if (l < l0) before += v;
    }
    before = wave_sum(before);
    all = wave_sum(all);
    if (lane == 0) { red[0][wave] = before; red[1][wave] = all; }
    __syncthreads();
    if (tid == 0) {
        float run = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const float denom = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) + 1e-6f;
        for (int i = 0; i < 64 && l0 + i < L; ++i) {
            run += (mr[l0 + i] != 0.f) ? 1.f : 0.f;
            xe[i] = run / denom * 6.283185307179586f;
        }
    }
    __syncthreads();
    const int ntok = min(64, L - l0);
    // a thread keeps its column(s): the frequency 10000^(2*floor(i/2)/D) is computed once, not per element
    for (int i = tid; i < D; i += 256) {
        const float dim_t = powf(10000.f, (float)(2 * (i / 2)) / (float)D);
        const bool odd = i & 1;
        T* out = pos + ((int64_t)b * L + l0) * D + i;
        if constexpr (sizeof(T) == 4) {   // fp32 (the parity mode): the reference's own arithmetic — a true division, libm sin / cos
            for (int t = 0; t < ntok; ++t) {
                const float a = xe[t] / dim_t;
                out[(int64_t)t * D] = from_f32<T>(odd ? cosf(a) : sinf(a));
            }
        } else {
            // 16-bit outputs (8 / 11 significant bits): the argument lies in [0, 2 pi] (x is normalised to 2 pi, dim_t >= 1), where the
            // hardware sine / cosine (v_sin_f32 / v_cos_f32 on revolutions) are good to ~1e-6 absolute — libm's range reduction and
            // polynomial made this kernel VALU-bound at 51 us for a 26 MB output (round 6: the step's first kernels are a serial chain)
            const float inv = 0.15915494309189535f / dim_t;   // revolutions per unit of x
            for (int t = 0; t < ntok; ++t) {
                const float rev = xe[t] * inv;
                out[(int64_t)t * D] = from_f32<T>(odd ? __builtin_amdgcn_cosf(rev) : __builtin_amdgcn_sinf(rev));
            }
        }
    }
}


// ---- stand-alone dropout (the enc/dec Transformer's residual / FFN dropouts, transformer.py:165-215,225-295) -------------------------
// Stateless keep mask of (row, column) of the [n / row_len, row_len] view (dropout_scale, common.h): forward and backward regenerate
// the same mask from (seed, row, column).  A thread walks along a row chunk: the row's share of the hash is computed once per chunk.
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, int64_t row_len, float p,
                                                      float inv_keep, uint64_t seed) {
    const uint32_t s0 = drop_seed32(seed);
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i < n; i += stride) {
        int64_t r = i / row_len;
        int64_t k = i - r * row_len;
        uint32_t rm = drop_row(s0, (uint64_t)r);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (i + e >= n) break;
            y[i + e] = from_f32<T>(to_f32(x[i + e]) * drop_scale_rk(rm, (uint32_t)k, p, inv_keep));
            if (++k == row_len) { k = 0; ++r; rm = drop_row(s0, (uint64_t)r); }
        }
    }
}
// (t, res and out may alias — ops.dropout_add runs in place on t: no __restrict__)
__global__ __launch_bounds__(256) void dropout_add_kernel(const float* t, const float* res, float* out, int64_t n, int64_t row_len, float p,
                                                          float inv_keep, uint64_t seed) {
    const uint32_t s0 = drop_seed32(seed);
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i < n; i += stride) {
        int64_t r = i / row_len;
        int64_t k = i - r * row_len;
        uint32_t rm = drop_row(s0, (uint64_t)r);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (i + e >= n) break;
            out[i + e] = res[i + e] + t[i + e] * drop_scale_rk(rm, (uint32_t)k, p, inv_keep);
            if (++k == row_len) { k = 0; ++r; rm = drop_row(s0, (uint64_t)r); }
        }
    }
}
}  // namespace

extern "C" {

int svol_layernorm_fwd(const void* x, int x_f32, const float* gamma, const float* beta, float* y32, void* y,
                       void* ypos, const void* pos, int64_t pos_rows, float* mean, float* rstd, int64_t M, int64_t D,
                       float dropout_p, uint64_t seed, const int64_t* seed_offset_dev, int dtype, void* stream) {
    if (!x || !gamma || !beta || (!y && !y32) || !mean || !rstd || M < 0 || D <= 0) return SVOL_E_INVALID;
    if (!aligned16(gamma) || !aligned16(beta)) return SVOL_E_INVALID;   // (16-byte vector loads of the affine parameters)
    if ((ypos != nullptr) != (pos != nullptr)) return SVOL_E_INVALID;
    if (pos && pos_rows <= 0) return SVOL_E_INVALID;
    if (D % 4 || D > LN_MAX_PASSES * 256) return SVOL_E_UNSUPPORTED;
    if (dropout_p < 0.f || dropout_p >= 1.f) return SVOL_E_INVALID;
    if (!svol_is16(dtype) && dtype != SVOL_F32) return SVOL_E_INVALID;
    if (M == 0) return SVOL_OK;
    const float inv_keep = 1.f / (1.f - dropout_p);
    const unsigned grid = (unsigned)((M + 3) / 4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define SVOL_LNF(TT, TXX, NPP)                                                                                              \
    hipLaunchKernelGGL((ln_fwd_kernel<TT, TXX, NPP>), dim3(grid), dim3(256), 0, s, (const TXX*)x, gamma, beta, y32, (TT*)y,      \
                       (TT*)ypos, (const TT*)pos, pos_rows, mean, rstd, M, (int)D, dropout_p, inv_keep, seed, seed_offset_dev)
#define SVOL_LNF_NP(TT, TXX)                     \
    do {                                         \
        if (D <= 256) SVOL_LNF(TT, TXX, 1);      \
        else if (D <= 512) SVOL_LNF(TT, TXX, 2); \
        else SVOL_LNF(TT, TXX, 4);               \
    } while (0)
    if (dtype == SVOL_BF16 && x_f32) SVOL_LNF_NP(bf16_t, float);
    else if (dtype == SVOL_BF16) SVOL_LNF_NP(bf16_t, bf16_t);
    else if (dtype == SVOL_F16 && x_f32) SVOL_LNF_NP(f16_t, float);
    else if (dtype == SVOL_F16) SVOL_LNF_NP(f16_t, f16_t);
    else SVOL_LNF_NP(float, float);
#undef SVOL_LNF_NP
#undef SVOL_LNF
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_layernorm_bwd(const float* dy32, const void* dy, const void* dy2, const void* x, int x_f32, const float* gamma,
                       const float* mean, const float* rstd, float* dx32, void* dx, float* dgamma, float* dbeta,
                       float* dx_colsum, int64_t M, int64_t D, float dropout_p, uint64_t seed, const int64_t* seed_offset_dev,
                       int dtype, void* stream) {
    if ((!dy32 && !dy && !dy2) || !x || !gamma || !mean || !rstd || (!dx && !dx32) || !dgamma || !dbeta || M < 0 || D <= 0)
        return SVOL_E_INVALID;
    if (D % 4 || D > LN_MAX_PASSES * 256) return SVOL_E_UNSUPPORTED;
    if (dropout_p < 0.f || dropout_p >= 1.f) return SVOL_E_INVALID;
    if (!svol_is16(dtype) && dtype != SVOL_F32) return SVOL_E_INVALID;
    if (M == 0) return SVOL_OK;
    const float inv_keep = 1.f / (1.f - dropout_p);
    // ~2048 waves (512 workgroups); each wave walks `rpw` consecutive rows; one set of atomics per workgroup
    int64_t rpw = (M + 2047) / 2048;
    if (rpw < 1) rpw = 1;
    const int64_t waves = (M + rpw - 1) / rpw;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool det_mode = svol_deterministic();
    DetScratch det(det_mode ? (size_t)grid * 3 * (size_t)D : 0, s);   // deterministic mode: per-workgroup rows, folded below
    if (det_mode && !det.p) return SVOL_E_LAUNCH;
    // NP = passes of 256 columns per row: the accumulator / operand arrays are sized by it (NP = 1 at d = 256: 76 VGPRs
    // instead of the 166 of the 4-pass instantiation; 46 us instead of 50 at [50176, 256] = 5.0 TB/s.  More waves or a next-row
    // prefetch do not help further: the end-of-workgroup atomics, then HBM, bound it)
#define SVOL_LNB(TT, TXX, NPP)                                                                                                      \
    hipLaunchKernelGGL((ln_bwd_kernel<TT, TXX, NPP>), dim3(grid), dim3(256), 0, s, dy32, (const TT*)dy, (const TT*)dy2, (const TXX*)x, \
                       gamma, mean, rstd, dx32, (TT*)dx, dgamma, dbeta, dx_colsum, M, (int)D, dropout_p, inv_keep, seed,              \
                       seed_offset_dev, (int)rpw, det.p)
#define SVOL_LNB_NP(TT, TXX)                  \
    do {                                      \
        if (D <= 256) SVOL_LNB(TT, TXX, 1);   \
        else if (D <= 512) SVOL_LNB(TT, TXX, 2); \
        else SVOL_LNB(TT, TXX, 4);            \
    } while (0)
    if (dtype == SVOL_BF16 && x_f32) SVOL_LNB_NP(bf16_t, float);
    else if (dtype == SVOL_BF16) SVOL_LNB_NP(bf16_t, bf16_t);
    else if (dtype == SVOL_F16 && x_f32) SVOL_LNB_NP(f16_t, float);
    else if (dtype == SVOL_F16) SVOL_LNB_NP(f16_t, f16_t);
    else SVOL_LNB_NP(float, float);
#undef SVOL_LNB_NP
#undef SVOL_LNB
    if (det_mode) {
        det_fold(det.p, (int)grid, 3 * D, dgamma, D, s);
        det_fold(det.p + D, (int)grid, 3 * D, dbeta, D, s);
        det_fold(det.p + 2 * D, (int)grid, 3 * D, dx_colsum, D, s);
    }
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_dropout(const void* x, void* y, int64_t n, int64_t row_len, float p, uint64_t seed, int dtype, void* stream) {
    if (!x || !y || n < 0 || row_len <= 0 || row_len > 0xffffffffll || !(p >= 0.f && p < 1.f)) return SVOL_E_INVALID;
    if (n == 0) return SVOL_OK;
    const float inv = 1.f / (1.f - p);
    int64_t g = (n + 1023) / 1024;
    if (g > 8192) g = 8192;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SVOL_F32) hipLaunchKernelGGL(dropout_kernel<float>, dim3((unsigned)g), dim3(256), 0, s, (const float*)x, (float*)y, n, row_len, p, inv, seed);
    else if (dtype == SVOL_BF16) hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, n, row_len, p, inv, seed);
    else if (dtype == SVOL_F16) hipLaunchKernelGGL(dropout_kernel<f16_t>, dim3((unsigned)g), dim3(256), 0, s, (const f16_t*)x, (f16_t*)y, n, row_len, p, inv, seed);
    else return SVOL_E_INVALID;
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_dropout_add(const float* t32, const float* res32, float* out32, int64_t n, int64_t row_len, float p, uint64_t seed, void* stream) {
    if (!t32 || !res32 || !out32 || n < 0 || row_len <= 0 || row_len > 0xffffffffll || !(p >= 0.f && p < 1.f)) return SVOL_E_INVALID;
    if (n == 0) return SVOL_OK;
    int64_t g = (n + 1023) / 1024;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(dropout_add_kernel, dim3((unsigned)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), t32, res32, out32, n,
                       row_len, p, 1.f / (1.f - p), seed);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_posenc_sine(const float* mask, void* pos, int64_t B, int64_t L, int64_t D, int dtype, void* stream) {
    if (!mask || !pos || B <= 0 || L <= 0 || D <= 0) return SVOL_E_INVALID;
    if (L > (1 << 24) || D > 4096) return SVOL_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (B > 65535) return SVOL_E_UNSUPPORTED;
    dim3 grid((unsigned)((L + 63) / 64), (unsigned)B);
    if (dtype == SVOL_BF16)
        hipLaunchKernelGGL(posenc_kernel<bf16_t>, grid, dim3(256), 0, s, mask, (bf16_t*)pos, (int)L, (int)D);
    else if (dtype == SVOL_F16)
        hipLaunchKernelGGL(posenc_kernel<f16_t>, grid, dim3(256), 0, s, mask, (f16_t*)pos, (int)L, (int)D);
    else if (dtype == SVOL_F32)
        hipLaunchKernelGGL(posenc_kernel<float>, grid, dim3(256), 0, s, mask, (float*)pos, (int)L, (int)D);
    else return SVOL_E_INVALID;
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

}  // extern "C"

// ---- AdamW over a flat fp32 range (torch.optim.AdamW semantics; reference train.py:98-99) ---------------------------------
// One streaming pass: p, g, m, v read once, p, m, v written once (28 bytes per parameter); 4 parameters per thread.
namespace {
// ZERO: the gradient range is zeroed behind its read (svol_adamw_flat_zero: the step boundary loses the caller's fill launches)
template <bool ZERO>
__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, int64_t n4, int64_t n, float decay, float b1, float b2,
                                                         float step_size, float inv_bc2_sqrt, float eps, float gscale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        f32x4 pp = *reinterpret_cast<f32x4*>(p + 4 * i), mm = *reinterpret_cast<f32x4*>(m + 4 * i), vv = *reinterpret_cast<f32x4*>(v + 4 * i);
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] * gscale;
            pp[e] *= decay;                                   // param.mul_(1 - lr * weight_decay)
            mm[e] = mm[e] + (gr - mm[e]) * (1.f - b1);        // exp_avg.lerp_(grad, 1 - beta1)
            vv[e] = vv[e] * b2 + (1.f - b2) * gr * gr;        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
            pp[e] -= step_size * (mm[e] / (sqrtf(vv[e]) * inv_bc2_sqrt + eps));
        }
        *reinterpret_cast<f32x4*>(p + 4 * i) = pp;
        *reinterpret_cast<f32x4*>(m + 4 * i) = mm;
        *reinterpret_cast<f32x4*>(v + 4 * i) = vv;
        if constexpr (ZERO) *reinterpret_cast<f32x4*>(g + 4 * i) = f32x4{0.f, 0.f, 0.f, 0.f};
    } else if (i == n4) {  // scalar tail (n % 4 elements)
        for (int64_t j = 4 * n4; j < n; ++j) {
            const float gr = g[j] * gscale;
            float pj = p[j] * decay;
            const float mj = m[j] + (gr - m[j]) * (1.f - b1);
            const float vj = v[j] * b2 + (1.f - b2) * gr * gr;
            pj -= step_size * (mj / (sqrtf(vj) * inv_bc2_sqrt + eps));
            p[j] = pj; m[j] = mj; v[j] = vj;
            if constexpr (ZERO) g[j] = 0.f;
        }
    }
}
}  // namespace

// ---- dynamic loss scaling (fp16 operands; the reference's fp16 mode is apex amp with dynamic scaling and overflow skip, configs.py:60-61) ----
// scaler state, four floats on the device: [0] loss scale, [1] overflow flag of the current step (0 / 1), [2] clean steps since the
// last change of the scale, [3] optimizer steps really taken (bias-correction exponent).  No host synchronisation anywhere.
namespace {
__global__ __launch_bounds__(256) void grad_finite_kernel(const float* __restrict__ g, int64_t n4, int64_t n, float* __restrict__ state) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) bad |= !(fabsf(v[e]) <= 3.0e38f);   // inf or NaN
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t j = 4 * n4; j < n; ++j) bad |= !(fabsf(g[j]) <= 3.0e38f);
    if (__any(bad) && (threadIdx.x & 63) == 0) state[1] = 1.f;   // (benign race: every writer stores the same value)
}
__global__ __launch_bounds__(256) void adamw_flat_scaled_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                                float* __restrict__ v, int64_t n4, int64_t n, float lr, float wd, float b1,
                                                                float b2, float eps, float gmul, const float* __restrict__ state) {
    if (state[1] != 0.f) return;   // an overflowed step is skipped whole: parameters and moments keep their values
    const float gscale = gmul / state[0];
    const float step = state[3] + 1.f;
    const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
    const float decay = 1.f - lr * wd, step_size = lr / bc1, inv_bc2_sqrt = 1.f / sqrtf(bc2);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        f32x4 pp = *reinterpret_cast<f32x4*>(p + 4 * i), mm = *reinterpret_cast<f32x4*>(m + 4 * i), vv = *reinterpret_cast<f32x4*>(v + 4 * i);
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] * gscale;
            pp[e] *= decay;
            mm[e] = mm[e] + (gr - mm[e]) * (1.f - b1);
            vv[e] = vv[e] * b2 + (1.f - b2) * gr * gr;
            pp[e] -= step_size * (mm[e] / (sqrtf(vv[e]) * inv_bc2_sqrt + eps));
        }
        *reinterpret_cast<f32x4*>(p + 4 * i) = pp;
        *reinterpret_cast<f32x4*>(m + 4 * i) = mm;
        *reinterpret_cast<f32x4*>(v + 4 * i) = vv;
    } else if (i == n4) {
        for (int64_t j = 4 * n4; j < n; ++j) {
            const float gr = g[j] * gscale;
            float pj = p[j] * decay;
            const float mj = m[j] + (gr - m[j]) * (1.f - b1);
            const float vj = v[j] * b2 + (1.f - b2) * gr * gr;
            pj -= step_size * (mj / (sqrtf(vj) * inv_bc2_sqrt + eps));
            p[j] = pj; m[j] = mj; v[j] = vj;
        }
    }
}
__global__ void loss_scaler_update_kernel(float* state, float growth, float backoff, float interval, float min_scale, float max_scale) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (state[1] != 0.f) {
        state[0] = fmaxf(state[0] * backoff, min_scale);
        state[2] = 0.f;
    } else {
        state[3] += 1.f;
        state[2] += 1.f;
        if (state[2] >= interval) { state[0] = fminf(state[0] * growth, max_scale); state[2] = 0.f; }
    }
    state[1] = 0.f;
}
}  // namespace

extern "C" int svol_grad_finite(const float* g, int64_t n, float* scaler_state, void* stream) {
    if (!g || !scaler_state || n < 0) return SVOL_E_INVALID;
    if (n == 0) return SVOL_OK;
    if (!aligned16(g)) return SVOL_E_UNSUPPORTED;
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(grad_finite_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, n4, n, scaler_state);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}
extern "C" int svol_adamw_flat_scaled(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                      float weight_decay, float grad_mul, const float* scaler_state, void* stream) {
    if (!p || !g || !m || !v || !scaler_state || n < 0) return SVOL_E_INVALID;
    if (n == 0) return SVOL_OK;
    if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) return SVOL_E_UNSUPPORTED;
    const int64_t n4 = n / 4;
    const int64_t blocks = (n4 + 1 + 255) / 256;
    if (blocks >= (1ll << 31)) return SVOL_E_UNSUPPORTED;
    hipLaunchKernelGGL(adamw_flat_scaled_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v, n4, n,
                       lr, weight_decay, beta1, beta2, eps, grad_mul, scaler_state);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}
extern "C" int svol_loss_scaler_update(float* scaler_state, float growth_factor, float backoff_factor, int64_t growth_interval, float min_scale,
                                       float max_scale, void* stream) {
    if (!scaler_state || growth_factor < 1.f || backoff_factor <= 0.f || backoff_factor > 1.f || growth_interval < 1) return SVOL_E_INVALID;
    hipLaunchKernelGGL(loss_scaler_update_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), scaler_state, growth_factor,
                       backoff_factor, (float)growth_interval, min_scale, max_scale);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

static int adamw_flat_launch(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int64_t step, float grad_scale, bool zero, void* stream) {
    if (!p || !g || !m || !v || n < 0 || step <= 0) return SVOL_E_INVALID;
    if (n == 0) return SVOL_OK;
    if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) return SVOL_E_UNSUPPORTED;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const int64_t n4 = n / 4;
    const int64_t blocks = (n4 + 1 + 255) / 256;
    if (blocks >= (1ll << 31)) return SVOL_E_UNSUPPORTED;
    if (zero)
        hipLaunchKernelGGL(adamw_flat_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v, n4, n,
                           1.f - lr * weight_decay, beta1, beta2, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), eps, grad_scale);
    else
        hipLaunchKernelGGL(adamw_flat_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v, n4, n,
                           1.f - lr * weight_decay, beta1, beta2, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), eps, grad_scale);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}
extern "C" int svol_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int64_t step, float grad_scale, void* stream) {
    return adamw_flat_launch(p, const_cast<float*>(g), m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, false, stream);
}
extern "C" int svol_adamw_flat_zero(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, int64_t step, float grad_scale, void* stream) {
    return adamw_flat_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, true, stream);
}

