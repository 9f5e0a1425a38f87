// Class / box heads of SVANet (svanet.py:42-44,125-127,144-156) as ONE launch per direction, and the two scalar-sized kernels of the
// criterion's autograd glue — the forward -> backward TURN of the training step.
//
// Why: between the last forward kernel of the video stream (layer 5's LN3) and its first backward kernel lies a dependency chain of
// 800-row work (DESIGN.md section 5 "In-step timeline"): the last query block, the heads, cost matrix, LSAP, losses, the heads'
// backward, that block's backward.  Round 5 measured 0.57 ms for the heads / criterion / heads-backward part of it on one clock —
// LSAP 0.18 ms, the rest ~45 launches of 3-17 us (four generic fp32 GEMMs forward; a cat, zero-pads, slices, act_bwd, three dX GEMMs,
// nine torch elementwise kernels backward), each waiting for the one before.  Here the chain is
//     heads_fwd -> match_cost -> lsap -> set_loss -> weighted_total | weighted_total' -> set_loss' -> heads_bwd
// eight launches; the two D x D weight gradients leave the chain for the weight-gradient stream (svol_gemm_tn), the 2- and 4-row ones
// are summed inside heads_bwd.
//
// Arithmetic: exact fp32 (v_mfma_f32_32x32x2_f32: fp32 products, fp32 accumulation) — the heads stay in the reference's precision in
// every compute mode.  A workgroup = 32 rows of the [R, D] query-state matrix (R = layers * B * N); the tile and
// the hidden activations live in LDS, a wave owns 32-column chunks of a layer's output and streams the weight rows from L2.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/svol_hip.h"
#include "common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int HT = 32;    // rows per workgroup
constexpr int HPAD = 4;   // floats of padding per LDS row

struct HeadsArgs {
    const float *hs, *Wc, *bc, *W0, *b0, *W1, *b1, *W2, *b2;
    float *logits, *h1, *h2, *boxes;
    int R, D;
};

struct HeadsBwdArgs {
    const float *dlogits, *dboxes;          // [R,2], [R,4]: gradients of the loss w.r.t. the heads' outputs
    const float *hs, *h1, *h2, *boxes;      // saved by the forward
    const float *Wc, *W0, *W1, *W2;
    float *dhs;                             // [R,D]
    float *gp0, *gp1;                       // [R,D]: gradients w.r.t. the pre-activations of layers 0 / 1 (operands of the D x D weight gradients)
    float *dWc, *dbc, *dW2, *db2;           // [2,D], [2], [4,D], [4]: ACCUMULATED (atomics; the caller zeroes or owns running sums)
    int R, D;
};

// Y[32 rows][n0 + 0..31] = Xs[32][K] . W[n0 + j][0..K)^T : A = the LDS tile (row m = lane % 32), B = weight row n0 + lane % 32; lane half
// h = lane / 32 walks k in [h K/2, (h+1) K/2) — the same k for both operands of every MFMA, which is all the contraction needs.
__device__ inline f32x16 tile_nt(const float* __restrict__ Xs, int ldx, const float* __restrict__ W, int ldw, int n0, int K, int r, int h) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float* pa = Xs + r * ldx + h * (K >> 1);
    const float* pb = W + (int64_t)(n0 + r) * ldw + h * (K >> 1);
    const int nq = K >> 5;   // batches of 16 k per half
    f32x4 b[4], nb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const f32x4*>(pb + 4 * c);
    for (int q = 0; q < nq; ++q) {
        if (q + 1 < nq) {
#pragma unroll
            for (int c = 0; c < 4; ++c) nb[c] = *reinterpret_cast<const f32x4*>(pb + 16 * (q + 1) + 4 * c);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(pa + 16 * q + 4 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[c][e], acc, 0, 0, 0);
        }
        if (q + 1 < nq) {
#pragma unroll
            for (int c = 0; c < 4; ++c) b[c] = nb[c];
        }
    }
    return acc;
}

// Y[32 rows][n0 + 0..31] = Gs[32][K] . W[0..K)[n0 + j] (the data gradient dX = dY W of a Linear whose weight is W [K = out, N = in]):
// B[k][n] = W[k][n0 + n], one dword per lane and MFMA, coalesced over n
__device__ inline f32x16 tile_nn(const float* __restrict__ Gs, int ldg, const float* __restrict__ W, int ldw, int n0, int K, int r, int h) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float* pa = Gs + r * ldg + h * (K >> 1);
    const float* pb = W + (int64_t)(h * (K >> 1)) * ldw + n0 + r;
    const int nq = K >> 5;
    float b[16], nb[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) b[j] = pb[(int64_t)j * ldw];
    for (int q = 0; q < nq; ++q) {
        if (q + 1 < nq) {
#pragma unroll
            for (int j = 0; j < 16; ++j) nb[j] = pb[(int64_t)(16 * (q + 1) + j) * ldw];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(pa + 16 * q + 4 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[4 * c + e], acc, 0, 0, 0);
        }
        if (q + 1 < nq) {
#pragma unroll
            for (int j = 0; j < 16; ++j) b[j] = nb[j];
        }
    }
    return acc;
}

// out[row][o] (o < NO) = Xs[row] . W[o] + b[o] for the 32 rows of the tile: 8 threads per row, partial sums met by three lane exchanges
template <int NO>
__device__ inline void small_out(const float* __restrict__ Xs, int ldx, const float* __restrict__ W, const float* __restrict__ b, int K,
                                 int tid, float (&out)[NO]) {
    const int row = tid >> 3, part = tid & 7;
#pragma unroll
    for (int o = 0; o < NO; ++o) out[o] = 0.f;
    for (int k = part * 4; k < K; k += 32) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(Xs + row * ldx + k);
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(W + (int64_t)o * K + k);
            out[o] += x[0] * w[0] + x[1] * w[1] + x[2] * w[2] + x[3] * w[3];
        }
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        out[o] += __shfl_xor(out[o], 1);
        out[o] += __shfl_xor(out[o], 2);
        out[o] += __shfl_xor(out[o], 4);
        out[o] += b ? b[o] : 0.f;
    }
}

__device__ inline void load_tile(float* __restrict__ Xs, int ldx, const float* __restrict__ X, int row0, int R, int D, int tid) {
    const int q = D >> 2;   // float4 per row
    for (int i = tid; i < HT * q; i += (int)blockDim.x) {
        const int row = i / q, c = (i - row * q) * 4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row0 + row < R) v = *reinterpret_cast<const f32x4*>(X + (int64_t)(row0 + row) * D + c);
        *reinterpret_cast<f32x4*>(Xs + row * ldx + c) = v;
    }
}

// 256 or 512 threads (round 6, second half): a 32-column chunk is 128 dependent MFMAs (~4 us) behind weight rows that arrive as 64
// scattered 16-byte pieces per load instruction; with 8 waves a wave owns ONE chunk per layer at D = 256 instead of two and a SIMD's
// two waves cover each other's load latency.  The row-per-8-threads sections run on the first 256 threads.
__global__ __launch_bounds__(512) void heads_fwd_kernel(HeadsArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = p.D, ld = D + HPAD;
    float* Xs = lds;             // hs tile, later h2
    float* Hs = lds + HT * ld;   // h1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * HT;
    const int cstride = ((int)blockDim.x >> 6) * 32;   // columns the workgroup's waves cover per round
    load_tile(Xs, ld, p.hs, row0, p.R, D, tid);
    __syncthreads();
    if (tid < 256) {   // class logits: Linear(d, 2) (svanet.py:44,125)
        float lg[2];
        small_out<2>(Xs, ld, p.Wc, p.bc, D, tid, lg);
        const int row = tid >> 3;
        if ((tid & 7) == 0 && row0 + row < p.R) { p.logits[(int64_t)(row0 + row) * 2] = lg[0]; p.logits[(int64_t)(row0 + row) * 2 + 1] = lg[1]; }
    }
    // box MLP layers 0 and 1: Linear -> ReLU (svanet.py:144-156); the activations are kept for the backward
    for (int layer = 0; layer < 2; ++layer) {
        const float* src = layer == 0 ? Xs : Hs;
        float* dst = layer == 0 ? Hs : Xs;
        const float* W = layer == 0 ? p.W0 : p.W1;
        const float* bias = layer == 0 ? p.b0 : p.b1;
        float* out = layer == 0 ? p.h1 : p.h2;
        // (layer 1 overwrites the hs tile with h2: every wave has passed the barrier behind layer 0, i.e. has read the tile for the last time)
        for (int n0 = wave * 32; n0 < D; n0 += cstride) {
            const f32x16 acc = tile_nt(src, ld, W, D, n0, D, r, h);
            const float bn = bias[n0 + r];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = 8 * g + 4 * h + e;
                    const float v = fmaxf(acc[4 * g + e] + bn, 0.f);
                    dst[m * ld + n0 + r] = v;
                    if (row0 + m < p.R) out[(int64_t)(row0 + m) * D + n0 + r] = v;
                }
        }
        if (layer == 0) __syncthreads();   // h1 complete before layer 1 reads it
    }
    __syncthreads();
    if (tid < 256) {   // layer 2: Linear(d, 4) -> sigmoid (svanet.py:126-127)
        float bx[4];
        small_out<4>(Xs, ld, p.W2, p.b2, D, tid, bx);
        const int row = tid >> 3;
        if ((tid & 7) == 0 && row0 + row < p.R) {
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = 1.f / (1.f + __expf(-bx[c]));
            *reinterpret_cast<f32x4*>(p.boxes + (int64_t)(row0 + row) * 4) = o;
        }
    }
}

__global__ __launch_bounds__(512) void heads_bwd_kernel(HeadsBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = p.D, ld = D + HPAD;
    float* As = lds;                 // g_pre1, later the hs tile (for dWc)
    float* Bs = lds + HT * ld;       // h2 tile (for dW2), later g_pre0
    float* sm = lds + 2 * HT * ld;   // [32][8]: g_pre2 (4) | dlogits (2) per row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * HT;
    if (tid < HT) {
        const int row = row0 + tid;
        float g2[4] = {0.f, 0.f, 0.f, 0.f}, gl[2] = {0.f, 0.f};
        if (row < p.R) {
            const f32x4 db = *reinterpret_cast<const f32x4*>(p.dboxes + (int64_t)row * 4);
            const f32x4 s = *reinterpret_cast<const f32x4*>(p.boxes + (int64_t)row * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) g2[c] = db[c] * s[c] * (1.f - s[c]);   // sigmoid'
            gl[0] = p.dlogits[(int64_t)row * 2];
            gl[1] = p.dlogits[(int64_t)row * 2 + 1];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) sm[tid * 8 + c] = g2[c];
        sm[tid * 8 + 4] = gl[0];
        sm[tid * 8 + 5] = gl[1];
    }
    const int nthr = (int)blockDim.x, cstride = (nthr >> 6) * 32;
    load_tile(Bs, ld, p.h2, row0, p.R, D, tid);
    __syncthreads();
    {   // dh2 = g_pre2 W2 (K = 4), g_pre1 = dh2 * [h2 > 0]  ->  As, and the 4-row weight gradient dW2 += g_pre2^T h2, db2 += colsum(g_pre2)
        const int row = (tid & 255) >> 3, part = tid & 7;
        const float g0 = sm[row * 8], g1 = sm[row * 8 + 1], g2 = sm[row * 8 + 2], g3 = sm[row * 8 + 3];
        for (int c = part * 4; c < (tid < 256 ? D : 0); c += 32) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(p.W2 + c), w1 = *reinterpret_cast<const f32x4*>(p.W2 + D + c);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(p.W2 + 2 * D + c), w3 = *reinterpret_cast<const f32x4*>(p.W2 + 3 * D + c);
            const f32x4 hv = *reinterpret_cast<const f32x4*>(Bs + row * ld + c);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = hv[e] > 0.f ? g0 * w0[e] + g1 * w1[e] + g2 * w2[e] + g3 * w3[e] : 0.f;
            *reinterpret_cast<f32x4*>(As + row * ld + c) = v;
            if (row0 + row < p.R) *reinterpret_cast<f32x4*>(p.gp1 + (int64_t)(row0 + row) * D + c) = v;
        }
        for (int c = tid; c < D; c += nthr) {   // column c of dW2's four rows: sum over the tile's rows
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int m = 0; m < HT; ++m) {
                const float x = Bs[m * ld + c];
                a0 += sm[m * 8] * x; a1 += sm[m * 8 + 1] * x; a2 += sm[m * 8 + 2] * x; a3 += sm[m * 8 + 3] * x;
            }
            unsafeAtomicAdd(p.dW2 + c, a0); unsafeAtomicAdd(p.dW2 + D + c, a1);
            unsafeAtomicAdd(p.dW2 + 2 * D + c, a2); unsafeAtomicAdd(p.dW2 + 3 * D + c, a3);
        }
        if (tid < 6) {   // db2 (4) and dbc (2): column sums of g_pre2 / dlogits
            float a = 0.f;
            for (int m = 0; m < HT; ++m) a += sm[m * 8 + tid];
            unsafeAtomicAdd(tid < 4 ? p.db2 + tid : p.dbc + (tid - 4), a);
        }
    }
    __syncthreads();
    // dh1 = g_pre1 W1, g_pre0 = dh1 * [h1 > 0]  ->  Bs
    for (int n0 = wave * 32; n0 < D; n0 += cstride) {
        const f32x16 acc = tile_nn(As, ld, p.W1, D, n0, D, r, h);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 8 * g + 4 * h + e;
                const bool ok = row0 + m < p.R;
                const float hv = ok ? p.h1[(int64_t)(row0 + m) * D + n0 + r] : 0.f;
                const float v = hv > 0.f ? acc[4 * g + e] : 0.f;
                // (Bs still holds h2 for the waves that have not finished the dW2 sums: they have — the barrier above)
                Bs[m * ld + n0 + r] = v;
                if (ok) p.gp0[(int64_t)(row0 + m) * D + n0 + r] = v;
            }
    }
    __syncthreads();
    // dhs = g_pre0 W0 + dlogits Wc; then the hs tile for the 2-row weight gradient dWc += dlogits^T hs
    for (int n0 = wave * 32; n0 < D; n0 += cstride) {
        const f32x16 acc = tile_nn(Bs, ld, p.W0, D, n0, D, r, h);
        const float wc0 = p.Wc[n0 + r], wc1 = p.Wc[D + n0 + r];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 8 * g + 4 * h + e;
                if (row0 + m < p.R) p.dhs[(int64_t)(row0 + m) * D + n0 + r] = acc[4 * g + e] + sm[m * 8 + 4] * wc0 + sm[m * 8 + 5] * wc1;
            }
    }
    load_tile(As, ld, p.hs, row0, p.R, D, tid);   // (As = g_pre1 was last read by the dh1 products, two barriers ago)
    __syncthreads();
    for (int c = tid; c < D; c += nthr) {
        float a0 = 0.f, a1 = 0.f;
        for (int m = 0; m < HT; ++m) {
            const float x = As[m * ld + c];
            a0 += sm[m * 8 + 4] * x; a1 += sm[m * 8 + 5] * x;
        }
        unsafeAtomicAdd(p.dWc + c, a0); unsafeAtomicAdd(p.dWc + D + c, a1);
    }
}

// dlogits = g_label * dl[layer][0], dboxes = g_bbox * dl[layer][1] + g_giou * dl[layer][2]   (the criterion's backward: its forward kept the
// UNIT gradients of the three losses of every layer, svol_set_loss; dl = d(whatever the caller built from the loss table) / d losses)
__global__ __launch_bounds__(256) void set_loss_bwd_kernel(const float* __restrict__ g_label, const float* __restrict__ g_bbox,
                                                           const float* __restrict__ g_giou, const float* __restrict__ dl, float* __restrict__ dlogits,
                                                           float* __restrict__ dboxes, int rows_per_layer, int64_t rows) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const int layer = (int)(row / rows_per_layer);
    const float a = dl[layer * 4], b = dl[layer * 4 + 1], c = dl[layer * 4 + 2];
    dlogits[row * 2] = g_label[row * 2] * a;
    dlogits[row * 2 + 1] = g_label[row * 2 + 1] * a;
    const f32x4 gb = *reinterpret_cast<const f32x4*>(g_bbox + row * 4), gg = *reinterpret_cast<const f32x4*>(g_giou + row * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = gb[e] * b + gg[e] * c;
    *reinterpret_cast<f32x4*>(dboxes + row * 4) = o;
}

// total = sum_i x[i] * w[i]  (train.py:227-228 on the loss table), in index order by one lane: n is 4 * layers
__global__ void weighted_total_kernel(const float* __restrict__ x, const float* __restrict__ w, int n, float* __restrict__ out) {
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int i = 0; i < n; ++i) a += x[i] * w[i];
        *out = a;
    }
}
// dx[i] = w[i] * dout
__global__ void weighted_total_bwd_kernel(const float* __restrict__ w, const float* __restrict__ dout, int n, float* __restrict__ dx) {
    const int i = threadIdx.x;
    if (i < n) dx[i] = w[i] * dout[0];
}

// 8 waves when the layer has at least 8 chunks of 32 columns (D >= 256); SVOL_HEADS_WAVES=4: round 6's first form (A/B)
inline unsigned heads_threads(int64_t D) {
    static const int w = getenv("SVOL_HEADS_WAVES") ? atoi(getenv("SVOL_HEADS_WAVES")) : 8;
    return (w >= 8 && D >= 256) ? 512u : 256u;
}

}  // namespace

extern "C" {

int svol_heads_fwd(const float* hs, const float* Wc, const float* bc, const float* W0, const float* b0, const float* W1, const float* b1,
                   const float* W2, const float* b2, float* logits, float* h1, float* h2, float* boxes, int64_t R, int64_t D, void* stream) {
    if (!hs || !Wc || !bc || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !logits || !h1 || !h2 || !boxes || R < 0) return SVOL_E_INVALID;
    if (D <= 0 || D % 32 || D > 512) return SVOL_E_UNSUPPORTED;
    for (const void* q : {(const void*)hs, (const void*)Wc, (const void*)W0, (const void*)W1, (const void*)W2, (const void*)h1, (const void*)h2,
                          (const void*)boxes})
        if (!aligned16(q)) return SVOL_E_INVALID;
    if (R == 0) return SVOL_OK;
    HeadsArgs p{hs, Wc, bc, W0, b0, W1, b1, W2, b2, logits, h1, h2, boxes, (int)R, (int)D};
    const size_t lds = (size_t)2 * HT * (D + HPAD) * sizeof(float);
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(heads_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SVOL_E_LAUNCH;
    hipLaunchKernelGGL(heads_fwd_kernel, dim3((unsigned)((R + HT - 1) / HT)), dim3(heads_threads(D)), lds, reinterpret_cast<hipStream_t>(stream), p);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_heads_bwd(const float* dlogits, const float* dboxes, const float* hs, const float* h1, const float* h2, const float* boxes,
                   const float* Wc, const float* W0, const float* W1, const float* W2, float* dhs, float* gp0, float* gp1, float* dWc,
                   float* dbc, float* dW2, float* db2, int64_t R, int64_t D, void* stream) {
    if (!dlogits || !dboxes || !hs || !h1 || !h2 || !boxes || !Wc || !W0 || !W1 || !W2 || !dhs || !gp0 || !gp1 || !dWc || !dbc || !dW2 ||
        !db2 || R < 0)
        return SVOL_E_INVALID;
    if (D <= 0 || D % 32 || D > 512) return SVOL_E_UNSUPPORTED;
    if (svol_deterministic()) return SVOL_E_UNSUPPORTED;   // the 2- / 4-row weight gradients meet through atomics: the caller takes the per-Linear path
    for (const void* q : {(const void*)dboxes, (const void*)hs, (const void*)h1, (const void*)h2, (const void*)boxes, (const void*)W2, (const void*)gp1})
        if (!aligned16(q)) return SVOL_E_INVALID;
    if (R == 0) return SVOL_OK;
    HeadsBwdArgs p{dlogits, dboxes, hs, h1, h2, boxes, Wc, W0, W1, W2, dhs, gp0, gp1, dWc, dbc, dW2, db2, (int)R, (int)D};
    const size_t lds = ((size_t)2 * HT * (D + HPAD) + HT * 8) * sizeof(float);
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(heads_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SVOL_E_LAUNCH;
    hipLaunchKernelGGL(heads_bwd_kernel, dim3((unsigned)((R + HT - 1) / HT)), dim3(heads_threads(D)), lds, reinterpret_cast<hipStream_t>(stream), p);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_set_loss_bwd(const float* g_label, const float* g_bbox, const float* g_giou, const float* dl, float* dlogits, float* dboxes,
                      int64_t n_layers, int64_t rows_per_layer, void* stream) {
    if (!g_label || !g_bbox || !g_giou || !dl || !dlogits || !dboxes || n_layers < 0 || rows_per_layer <= 0) return SVOL_E_INVALID;
    if (!aligned16(g_bbox) || !aligned16(g_giou) || !aligned16(dboxes)) return SVOL_E_INVALID;
    const int64_t rows = n_layers * rows_per_layer;
    if (rows == 0) return SVOL_OK;
    hipLaunchKernelGGL(set_loss_bwd_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g_label,
                       g_bbox, g_giou, dl, dlogits, dboxes, (int)rows_per_layer, rows);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_weighted_total(const float* x, const float* w, int64_t n, float* out, void* stream) {
    if (!x || !w || !out || n <= 0 || n > 1024) return SVOL_E_INVALID;
    hipLaunchKernelGGL(weighted_total_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), x, w, (int)n, out);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_weighted_total_bwd(const float* w, const float* dout, int64_t n, float* dx, void* stream) {
    if (!w || !dout || !dx || n <= 0 || n > 1024) return SVOL_E_INVALID;
    hipLaunchKernelGGL(weighted_total_bwd_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), w, dout, (int)n, dx);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

}  // extern "C"
