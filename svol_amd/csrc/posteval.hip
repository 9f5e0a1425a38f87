// Post-processing and evaluation on the device — the step right after the hot path in the
// reference's test.py, restated for gfx950:
//
//   svol_postprocess     test.py:133-155   foreground score = softmax(logits)[..., 0], cxcywh -> xyxy clamped to [0, 1],
//                                          chunk(num_frames), stable descending sort by score inside each chunk
//   svol_eval_max_iou    eval.py:72-90     per ground-truth box, max IoU over the top-k predictions of its frame
//                                          (including the reference's tile/repeat pair layout, utils.py:88-96)
//   svol_eval_ap         utils.py:121-201  per (video, sketch): predictions sorted by score, greedy one-to-one matching
//                                          at every IoU threshold, cumulative precision / recall, interpolated AP
//
// Everything the reference does in fp64 numpy is done in fp64 here, operation for operation (this file is compiled
// with -ffp-contract=off; numpy's pairwise summation order is reproduced), so the AP arrays and max-IoU vectors are
// bit-identical to the reference's and the formatted metrics follow.  These are tiny, latency-bound kernels (one
// wave per (video, sketch) group / one thread per box); the point is that evaluation no longer leaves the device
// for a Python loop over every prediction.
#include "common.h"

namespace {

// ---- svol_postprocess ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ logits, const float* __restrict__ boxes,
                                                          float* __restrict__ out, int N, int chunk) {
    extern __shared__ float sc[];  // [chunk] scores of this chunk
    const int b = blockIdx.y, c0 = blockIdx.x * chunk;
    const int n = min(chunk, N - c0);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float l0 = logits[((int64_t)b * N + c0 + i) * 2], l1 = logits[((int64_t)b * N + c0 + i) * 2 + 1];
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        sc[i] = e0 / (e0 + e1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float s = sc[i];
        int rank = 0;  // stable descending: earlier rows win ties (Python's sorted(..., reverse=True) keeps their order)
        for (int j = 0; j < n; ++j) rank += (sc[j] > s) || (sc[j] == s && j < i);
        const float* bx = boxes + ((int64_t)b * N + c0 + i) * 4;
        const float cx = bx[0], cy = bx[1], w = bx[2], h = bx[3];
        float* o = out + ((int64_t)b * N + c0 + rank) * 5;
        o[0] = fminf(fmaxf(cx - 0.5f * w, 0.f), 1.f);
        o[1] = fminf(fmaxf(cy - 0.5f * h, 0.f), 1.f);
        o[2] = fminf(fmaxf(cx + 0.5f * w, 0.f), 1.f);
        o[3] = fminf(fmaxf(cy + 0.5f * h, 0.f), 1.f);
        o[4] = s;
    }
}

// ---- IoU of two xyxy boxes, fp64, as compute_iou_batch_paired (utils.py:35-71) -----------------------------------
__device__ __forceinline__ double iou_xyxy(const double* a, const double* b) {
    const double xmin = fmax(a[0], b[0]), ymin = fmax(a[1], b[1]);
    const double xmax = fmin(a[2], b[2]), ymax = fmin(a[3], b[3]);
    const double inter = (xmax - xmin) * (ymax - ymin);
    const double a1 = (a[2] - a[0]) * (a[3] - a[1]);
    const double a2 = (b[2] - b[0]) * (b[3] - b[1]);
    const double uni = (a1 + a2) - inter;
    const bool valid = xmin <= xmax && ymin <= ymax;
    return valid ? inter / uni : 0.0;
}

// one thread per ground-truth box j of record r: max over i < n of iou[i][j] where the (n x m) matrix is the RESHAPE of
// the pair list laid out [gt][pred] (np.tile / np.repeat, utils.py:90-96): entry (i, j) is pair p = i*m + j =
// (pred p % n, gt p / n).  np.max propagates NaN.
__global__ void max_iou_kernel(const double* __restrict__ pred, const int32_t* __restrict__ pred_off,
                               const double* __restrict__ gt, const int32_t* __restrict__ gt_off,
                               const int32_t* __restrict__ gt_rec, double* __restrict__ out, int G, int k) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const int r = gt_rec[g];
    const int p0 = pred_off[r], n = min(k, pred_off[r + 1] - p0);
    const int g0 = gt_off[r], m = gt_off[r + 1] - g0, j = g - g0;
    double best = -INFINITY;
    bool nan = false;
    for (int i = 0; i < n; ++i) {
        const int p = i * m + j;
        const double v = iou_xyxy(pred + (int64_t)(p0 + p % n) * 4, gt + (int64_t)(g0 + p / n) * 4);
        if (v != v) nan = true;
        else if (v > best) best = v;
    }
    out[g] = nan ? NAN : best;
}

// numpy's pairwise summation of a contiguous double array (loops_utils.h.src, PW_BLOCKSIZE = 128)
__device__ double np_pairwise_sum(const double* a, int n) {
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

// one wave per (video, sketch) group; lanes 0..K-1 each own one IoU threshold after the common sort
__global__ __launch_bounds__(64) void ap_kernel(const double* __restrict__ pred_box, const double* __restrict__ pred_score,
                                                const int32_t* __restrict__ pred_frame, const int32_t* __restrict__ grp_pred_off,
                                                const double* __restrict__ gt_box, const int32_t* __restrict__ gt_frame,
                                                const int32_t* __restrict__ grp_gt_off, const double* __restrict__ thr, int K,
                                                int32_t* __restrict__ ws_order, unsigned char* __restrict__ ws_flag,
                                                unsigned char* __restrict__ ws_lock, double* __restrict__ ws_f64,
                                                double* __restrict__ ap, int P, int G, int NG) {
    const int grp = blockIdx.x, lane = threadIdx.x;
    const int p0 = grp_pred_off[grp], M = grp_pred_off[grp + 1] - p0;
    const int g0 = grp_gt_off[grp], Ng = grp_gt_off[grp + 1] - g0;
    if (M == 0) {  // "if len(prediction) == 0: return ap" (zeros)
        if (lane < K) ap[(int64_t)grp * K + lane] = 0.0;
        return;
    }
    // 1. prediction.sort(key=-score): stable ascending on -score
    int32_t* order = ws_order + p0;
    for (int i = lane; i < M; i += 64) {
        const double s = pred_score[p0 + i];
        int rank = 0;
        for (int j = 0; j < M; ++j) {
            const double sj = pred_score[p0 + j];
            rank += (sj > s) || (sj == s && j < i);
        }
        order[rank] = i;
    }
    __threadfence();
    __syncthreads();
    if (lane >= K) return;
    const double th = thr[lane];
    unsigned char* flag = ws_flag + (int64_t)lane * P + p0;  // tp flag per sorted prediction (fp = !tp)
    unsigned char* lock = ws_lock + (int64_t)lane * G + g0;
    for (int g = 0; g < Ng; ++g) lock[g] = 0;
    // 2. greedy matching in score order: the unlocked ground truth of the same frame with the highest IoU >= threshold
    for (int idx = 0; idx < M; ++idx) {
        const int p = p0 + order[idx];
        const int f = pred_frame[p];
        int best = -1;
        double bestkey = -INFINITY;
        for (int g = 0; g < Ng; ++g) {
            if (gt_frame[g0 + g] != f) continue;
            const double v = iou_xyxy(pred_box + (int64_t)p * 4, gt_box + (int64_t)(g0 + g) * 4);
            if (v < th) continue;                      // NaN is not < th: it stays a candidate, as in the reference loop
            const double key = (v != v) ? INFINITY : v;  // argsort()[::-1] visits NaN first
            if (!lock[g] && key >= bestkey) { best = g; bestkey = key; }  // ties: the later index comes first in [::-1]
        }
        flag[idx] = best >= 0;
        if (best >= 0) lock[best] = 1;
    }
    // 3. cumulative precision / recall, interpolated AP (utils.py:101-118, 191-200)
    const int64_t stride = (int64_t)P + 2 * (int64_t)NG;
    double* mprec = ws_f64 + (int64_t)lane * stride + p0 + 2 * grp;            // [M + 2]
    double* terms = ws_f64 + ((int64_t)K + lane) * stride + p0 + 2 * grp;      // [<= M + 1]
    const double npos = (double)Ng;
    double tp = 0.0, fp = 0.0;
    mprec[0] = 0.0;
    for (int i = 0; i < M; ++i) {
        if (flag[i]) tp += 1.0; else fp += 1.0;
        mprec[i + 1] = tp / (tp + fp);
    }
    mprec[M + 1] = 0.0;
    for (int i = M; i >= 0; --i) mprec[i] = fmax(mprec[i], mprec[i + 1]);
    // mrecall = [0, recall..., 1]; terms where mrecall[i] != mrecall[i-1], i = 1..M+1
    int nt = 0;
    double prev = 0.0;
    tp = 0.0;
    for (int i = 1; i <= M + 1; ++i) {
        double cur;
        if (i <= M) {
            if (flag[i - 1]) tp += 1.0;
            cur = tp / npos;
        } else cur = 1.0;
        if (cur != prev) terms[nt++] = (cur - prev) * mprec[i];
        prev = cur;
    }
    ap[(int64_t)grp * K + lane] = np_pairwise_sum(terms, nt);
}

}  // namespace

extern "C" {

int svol_postprocess(const float* logits, const float* boxes, float* out, int64_t B, int64_t N, int64_t chunk, void* stream) {
    if (!logits || !boxes || !out || B <= 0 || N <= 0 || chunk <= 0) return SVOL_E_INVALID;
    if (chunk > 12288 || B > 65535 || N > (1 << 30)) return SVOL_E_UNSUPPORTED;
    const int64_t nchunks = (N + chunk - 1) / chunk;
    hipLaunchKernelGGL(postprocess_kernel, dim3((unsigned)nchunks, (unsigned)B), dim3(chunk >= 256 ? 256 : 64), (size_t)chunk * 4,
                       reinterpret_cast<hipStream_t>(stream), logits, boxes, out, (int)N, (int)chunk);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_eval_max_iou(const double* pred_box, const int32_t* pred_off, const double* gt_box, const int32_t* gt_off,
                      const int32_t* gt_rec, double* out, int64_t n_gt, int32_t k, void* stream) {
    if (!pred_box || !pred_off || !gt_box || !gt_off || !gt_rec || !out || n_gt < 0 || k <= 0) return SVOL_E_INVALID;
    if (n_gt == 0) return SVOL_OK;
    if (n_gt > (1 << 30)) return SVOL_E_UNSUPPORTED;
    hipLaunchKernelGGL(max_iou_kernel, dim3((unsigned)((n_gt + 127) / 128)), dim3(128), 0, reinterpret_cast<hipStream_t>(stream),
                       pred_box, pred_off, gt_box, gt_off, gt_rec, out, (int)n_gt, (int)k);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_eval_ap(const double* pred_box, const double* pred_score, const int32_t* pred_frame, const int32_t* grp_pred_off,
                 const double* gt_box, const int32_t* gt_frame, const int32_t* grp_gt_off, const double* thresholds,
                 int32_t n_thresholds, int32_t* ws_order, unsigned char* ws_u8, double* ws_f64, double* ap, int64_t n_pred,
                 int64_t n_gt, int64_t n_groups, void* stream) {
    if (!pred_box || !pred_score || !pred_frame || !grp_pred_off || !grp_gt_off || !thresholds || !ws_order || !ws_u8 ||
        !ws_f64 || !ap)
        return SVOL_E_INVALID;
    if (n_groups <= 0 || n_pred < 0 || n_gt < 0 || n_thresholds <= 0) return SVOL_E_INVALID;
    if (n_gt > 0 && (!gt_box || !gt_frame)) return SVOL_E_INVALID;
    if (n_thresholds > 64 || n_pred > (1 << 28) || n_gt > (1 << 28) || n_groups > (1 << 24)) return SVOL_E_UNSUPPORTED;
    unsigned char* ws_flag = ws_u8;                                   // [K][n_pred]
    unsigned char* ws_lock = ws_u8 + (int64_t)n_thresholds * n_pred;  // [K][n_gt]
    hipLaunchKernelGGL(ap_kernel, dim3((unsigned)n_groups), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), pred_box,
                       pred_score, pred_frame, grp_pred_off, gt_box, gt_frame, grp_gt_off, thresholds, (int)n_thresholds,
                       ws_order, ws_flag, ws_lock, ws_f64, ap, (int)n_pred, (int)n_gt, (int)n_groups);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

}  // extern "C"
