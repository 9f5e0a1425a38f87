// Composite block programs of libsvol_hip (include/svol_hip.h, "composite block programs"): one C call enqueues every kernel of a
// block of a CrossModalTransformerLayer (reference cross_modal_transformer.py:105-160) on the caller's stream.
//
// Why: the per-op C-ABI was driven from Python — a ctypes call, a torch.empty and an autograd node per kernel: 618 launches and
// 19.6 ms of host issue time for a 20.8 ms step, main-queue idle 3.3 ms.  The kernels and their order are unchanged;
// what moves into C is the sequencing and the pointer arithmetic on caller-owned buffers (slot tables), ~4 us per launch instead of
// ~30.  No allocation, no synchronisation, no global state: every function is a straight line of launches.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace {

// ---- SVOL_BLOCK_TRACE=1: in-step duration of every launch group of a block program, WITHOUT a profiler attached (rocprofv3 makes
// the host the bottleneck of this step, which moves the streams against each other).  An event after every entry point a program
// calls; svol_block_trace_dump() turns consecutive pairs into per-call-site sums.  Off (the default): one predictable branch per call.
const bool kTrace = getenv("SVOL_BLOCK_TRACE") != nullptr;
struct TraceRec { const char* prog; const char* site; hipEvent_t e0, e1; };
std::mutex g_trace_mu;
std::vector<TraceRec> g_trace_log;
struct Trace {
    const char* prog;
    hipStream_t s;
    hipEvent_t last = nullptr;
    Trace* prev;
    static thread_local Trace* cur;
    Trace(const char* prog_, void* s_) : prog(prog_), s(static_cast<hipStream_t>(s_)), prev(cur) {
        cur = this;
        if (kTrace && hipEventCreate(&last) == hipSuccess) (void)hipEventRecord(last, s);
    }
    ~Trace() { cur = prev; }
    void mark(const char* site) {
        if (!kTrace || !last) return;
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        (void)hipEventRecord(e, s);
        std::lock_guard<std::mutex> lk(g_trace_mu);
        g_trace_log.push_back({prog, site, last, e});
        last = e;
    }
};
thread_local Trace* Trace::cur = nullptr;

struct Dims {
    int64_t B, L, N, D, H, F, ws_bytes;
    int dt, qdt;
    explicit Dims(const int64_t* d)
        : B(d[SVOL_DIM_B]), L(d[SVOL_DIM_L]), N(d[SVOL_DIM_N]), D(d[SVOL_DIM_D]), H(d[SVOL_DIM_H]), F(d[SVOL_DIM_F]),
          ws_bytes(d[SVOL_DIM_ATTN_WS_BYTES]), dt((int)d[SVOL_DIM_DTYPE]), qdt((int)d[SVOL_DIM_QDTYPE]) {}
    bool ok() const {
        return B > 0 && D > 0 && H > 0 && D % H == 0 && (dt == SVOL_F32 || svol_is16(dt)) && (qdt == SVOL_F32 || svol_is16(qdt));
    }
};

inline int esz(int dtype) { return svol_is16(dtype) ? 2 : 4; }
// element offset into a typed buffer
inline void* at(void* p, int64_t elems, int dtype) { return p ? static_cast<char*>(p) + elems * esz(dtype) : nullptr; }
inline const void* at(const void* p, int64_t elems, int dtype) { return p ? static_cast<const char*>(p) + elems * esz(dtype) : nullptr; }
inline float* f32(void* p) { return static_cast<float*>(p); }

// the attention kernels' scale conventions (svol_amd/ops.py AttnLNFn): bf16 q leaves its projection pre-multiplied by d_h^-1/2 log2(e)
inline float attn_scale(int64_t dh) { return (float)(1.0 / sqrt((double)dh)); }
inline float attn_premul(int64_t dh, int dtype) { return svol_is16(dtype) ? (float)(1.4426950408889634 / sqrt((double)dh)) : 0.f; }

// plain C = A B^T (+ bias) (* colscale)
inline int nt(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* colscale,
              int64_t M, int64_t N, int64_t K, int dtype, void* s) {
    return svol_gemm_nt(A, lda, nullptr, 0, W, ldw, C, ldc, static_cast<const float*>(bias), static_cast<const float*>(colscale),
                        SVOL_ACT_NONE, nullptr, 0, nullptr, 0, 0, M, N, K, dtype, s);
}
// fp32 stream: C32 = A B^T + bias + residual32
inline int nt_res(const void* A, int64_t lda, const void* W, int64_t ldw, void* C32, const void* bias, const void* res32, int64_t M,
                  int64_t N, int64_t K, int dtype, void* s) {
    return svol_gemm_nt(A, lda, nullptr, 0, W, ldw, C32, N, static_cast<const float*>(bias), nullptr, SVOL_ACT_NONE, nullptr, 0, res32, N,
                        1, M, N, K, dtype, s);
}

#define RUN(expr)                                 \
    do {                                          \
        const int rc_ = (expr);                   \
        if (rc_) return rc_;                      \
        if (kTrace && Trace::cur) Trace::cur->mark(#expr); \
    } while (0)

#define SLOT_NAME(n) #n ","
const char* const kVhNames = SVOL_VH_SLOTS(SLOT_NAME);
const char* const kQsNames = SVOL_QS_SLOTS(SLOT_NAME);
const char* const kQcNames = SVOL_QC_SLOTS(SLOT_NAME);

// the video half's MLP saves gelu' instead of the pre-activation (SVOL_VH_GELU_PRE=1: round 2's form, for A/B runs)
const int kVhGelu = getenv("SVOL_VH_GELU_PRE") ? SVOL_ACT_GELU : SVOL_ACT_GELU_D;

inline void record(void* ev, void* stream) {
    if (ev) (void)hipEventRecord(static_cast<hipEvent_t>(ev), static_cast<hipStream_t>(stream));
}

}  // namespace


extern "C" {

int svol_block_trace_dump(char* buf, int64_t cap) {
    if (!buf || cap <= 0) return SVOL_E_INVALID;
    buf[0] = 0;
    if (!kTrace) return SVOL_OK;
    if (hipDeviceSynchronize() != hipSuccess) return SVOL_E_LAUNCH;
    std::lock_guard<std::mutex> lk(g_trace_mu);
    static const bool timeline = getenv("SVOL_BLOCK_TRACE") && atoi(getenv("SVOL_BLOCK_TRACE")) == 2;
    if (timeline && !g_trace_log.empty()) {   // every record with its start / end relative to the first one (all streams, one clock)
        std::string out;
        char line[256];
        const hipEvent_t ref = g_trace_log.front().e0;
        for (const TraceRec& r : g_trace_log) {
            float t0 = 0.f, t1 = 0.f;
            if (hipEventElapsedTime(&t0, ref, r.e0) != hipSuccess || hipEventElapsedTime(&t1, ref, r.e1) != hipSuccess) continue;
            std::string site(r.site);
            const size_t c = site.find(',');
            if (c != std::string::npos) site.resize(c);
            snprintf(line, sizeof line, "%9.3f %9.3f  %-20s %s\n", t0, t1, r.prog, site.c_str());
            out += line;
        }
        std::map<hipEvent_t, int> seen_;
        for (const TraceRec& r : g_trace_log) { seen_[r.e0] = 1; seen_[r.e1] = 1; }
        for (auto& kv : seen_) (void)hipEventDestroy(kv.first);
        g_trace_log.clear();
        strncpy(buf, out.c_str(), (size_t)cap - 1);
        buf[cap - 1] = 0;
        return SVOL_OK;
    }
    std::map<std::string, std::pair<double, int>> acc;
    std::vector<std::string> order;
    std::map<hipEvent_t, int> seen;
    for (const TraceRec& r : g_trace_log) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        std::string site(r.site);
        const size_t c = site.find(',');
        if (c != std::string::npos) site.resize(c);
        const std::string key = std::string(r.prog) + " | " + site;
        if (!acc.count(key)) order.push_back(key);
        acc[key].first += ms;
        acc[key].second += 1;
    }
    for (const TraceRec& r : g_trace_log) { seen[r.e0] = 1; seen[r.e1] = 1; }
    for (auto& kv : seen) (void)hipEventDestroy(kv.first);
    g_trace_log.clear();
    std::string out;
    char line[256];
    for (const std::string& k : order) {
        snprintf(line, sizeof line, "%-110s calls %6d  avg_us %9.1f  total_ms %9.3f\n", k.c_str(), acc[k].second,
                 acc[k].first / acc[k].second * 1e3, acc[k].first);
        out += line;
    }
    strncpy(buf, out.c_str(), (size_t)cap - 1);
    buf[cap - 1] = 0;
    return SVOL_OK;
}

const char* svol_block_slot_names(int block) {
    switch (block) {
        case SVOL_BLK_VIDEO_HALF: return kVhNames;
        case SVOL_BLK_QUERY_SELF: return kQsNames;
        case SVOL_BLK_QUERY_CROSS: return kQcNames;
        default: return "";
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// video half
// ---------------------------------------------------------------------------------------------------------------------
int svol_video_half_fwd(const int64_t* dims, void* const* p, void* s) {
    if (!dims || !p) return SVOL_E_INVALID;
    Trace tr_("video_half_fwd", s);
    const Dims d(dims);
    if (!d.ok() || d.L <= 0 || d.F <= 0) return SVOL_E_INVALID;
    const int64_t M = d.B * d.L, D = d.D, F = d.F, dh = D / d.H;
    const int dt = d.dt, xf = dt != SVOL_F32;
#define P(n) p[SVOL_VH_##n]
    // gate + LN1 (:122-127)
    // (GATE_PRE: the scores are in GATE_WS already — the LN3 of the layer before wrote them, see the end of this function)
    if (P(GATE_PRE))
        RUN(svol_gate_fwd_scored(f32(P(X32)), P(POS), f32(P(U)), f32(P(G1)), f32(P(BT1)), f32(P(Y1_32)), P(Y1), P(Y1POS), f32(P(A)),
                                 f32(P(MEAN1)), f32(P(RSTD1)), f32(P(GATE_WS)), d.B, d.L, D, d.H, dt, s));
    else
        RUN(svol_gate_fwd(f32(P(X32)), P(POS), f32(P(U)), f32(P(G1)), f32(P(BT1)), f32(P(Y1_32)), P(Y1), P(Y1POS), f32(P(A)), f32(P(MEAN1)),
                          f32(P(RSTD1)), f32(P(GATE_WS)), d.B, d.L, D, d.H, dt, s));
    // q | k from y + pos, v from y (:137-139)
    RUN(nt(P(Y1POS), D, P(W_IN), D, P(QKV), 3 * D, P(B_IN), P(QSCALE), M, 2 * D, D, dt, s));
    if (P(WV_HILO))
        RUN(svol_gemm_nt_split(P(Y1), D, P(WV_HILO), 2 * D, at(P(QKV), 2 * D, dt), 3 * D, f32(P(B_IN)) + 2 * D, M, D, D, s));
    else
        RUN(nt(P(Y1), D, at(P(W_IN), 2 * D * D, dt), D, at(P(QKV), 2 * D, dt), 3 * D, f32(P(B_IN)) + 2 * D, nullptr, M, D, D, dt, s));
    record(P(EV_A0), s);
    RUN(svol_attn_fwd(P(QKV), 3 * D, at(P(QKV), D, dt), 3 * D, at(P(QKV), 2 * D, dt), 3 * D, P(O), D, f32(P(LSE)), nullptr, d.B, d.H, d.L,
                      d.L, dh, attn_scale(dh), attn_premul(dh, dt), P(ATTN_WS), d.ws_bytes, dt, s));
    record(P(EV_A1), s);
    // out-proj + residual -> LN2 (:140-141)
    RUN(nt_res(P(O), D, P(W_O), D, P(S2), P(B_O), P(Y1_32), M, D, D, dt, s));
    RUN(svol_layernorm_fwd(P(S2), xf, f32(P(G2)), f32(P(BT2)), f32(P(Y2_32)), P(Y2), nullptr, nullptr, 0, f32(P(MEAN2)), f32(P(RSTD2)), M,
                           D, 0.f, 0, nullptr, dt, s));
    // MLP1 + residual -> LN3 (+pos) (:142-143)
    // (PRE holds gelu'(pre-activation): SVOL_ACT_GELU_D — the backward's epilogue is then a multiply)
    // (the MLP as ONE launch per direction — round 4's svol_mlp_chain — measured +-0.05 ms in the step in two rounds and left the
    // library in round 5: tools/micro/mlp_chain_bf16.hip, profiles/round4_mlp_chain_lab.md)
    RUN(svol_gemm_nt(P(Y2), D, nullptr, 0, P(W_FC1), D, P(HID), F, f32(P(B_FC1)), nullptr, kVhGelu, P(PRE), F, nullptr, 0, 0, M, F, D, dt, s));
    RUN(nt_res(P(HID), F, P(W_FC2), F, P(S3), P(B_FC2), P(Y2_32), M, D, F, dt, s));
    if (P(U_NEXT) && P(GATE_WS_NEXT))   // ... and the NEXT layer's gate scores from the row just normalised (S3 is fp32 in every mode)
        RUN(svol_layernorm_gate_scores_fwd(f32(P(S3)), f32(P(G3)), f32(P(BT3)), f32(P(M32)), P(M), P(MPOS), P(POS), f32(P(MEAN3)),
                                           f32(P(RSTD3)), f32(P(U_NEXT)), f32(P(GATE_WS_NEXT)), d.B, d.L, D, d.H, dt, s));
    else
        RUN(svol_layernorm_fwd(P(S3), xf, f32(P(G3)), f32(P(BT3)), f32(P(M32)), P(M), P(MPOS), P(POS), M, f32(P(MEAN3)), f32(P(RSTD3)), M, D,
                               0.f, 0, nullptr, dt, s));
    return SVOL_OK;
}

int svol_video_half_bwd(const int64_t* dims, void* const* p, int phase, void* s) {
    if (!dims || !p || phase < 0 || phase > 2) return SVOL_E_INVALID;
    Trace tr_(phase == 2 ? "video_half_bwd.2" : "video_half_bwd.1", s);
    const Dims d(dims);
    if (!d.ok() || d.L <= 0 || d.F <= 0) return SVOL_E_INVALID;
    const int64_t M = d.B * d.L, D = d.D, F = d.F, dh = D / d.H;
    const int dt = d.dt, xf = dt != SVOL_F32;
    if (phase != 2) {
        // LN3' -> (ds32, ds), b_fc2' ; (ds W2) * gelu'(pre), b_fc1' ; dpre W1 -> dy2
        RUN(svol_layernorm_bwd(f32(P(DM32)), P(DM), P(DMPOS), P(S3), xf, f32(P(G3)), f32(P(MEAN3)), f32(P(RSTD3)), f32(P(DS32_3)), P(DS3),
                               f32(P(DG3)), f32(P(DBT3)), f32(P(DB_FC2)), M, D, 0.f, 0, nullptr, dt, s));
        RUN(svol_gemm_nt_dact(P(DS3), D, P(W_FC2_T), D, P(DPRE), F, P(PRE), F, kVhGelu, f32(P(DB_FC1)), M, F, D, dt, s));
        RUN(nt(P(DPRE), F, P(W_FC1_T), F, P(DY2), D, nullptr, nullptr, M, D, F, dt, s));
        // LN2' -> (ds32_2, g), b_o' ; do = g Wo
        RUN(svol_layernorm_bwd(f32(P(DS32_3)), P(DY2), nullptr, P(S2), xf, f32(P(G2)), f32(P(MEAN2)), f32(P(RSTD2)), f32(P(DS32_2)), P(G2D),
                               f32(P(DG2)), f32(P(DBT2)), f32(P(DB_O)), M, D, 0.f, 0, nullptr, dt, s));
        RUN(nt(P(G2D), D, P(W_O_T), D, P(DO), D, nullptr, nullptr, M, D, D, dt, s));
    }
    if (phase != 1) {
        // the fp32 dQ image of the single pass: zeroed beside the PREVIOUS layer's launch when the caller alternates two workspaces
        int aflags = 0;
        if (P(EV_CLEAN_IN)) {
            if (hipStreamWaitEvent(static_cast<hipStream_t>(s), static_cast<hipEvent_t>(P(EV_CLEAN_IN)), 0) != hipSuccess) return SVOL_E_LAUNCH;
            aflags = SVOL_ATTN_DQ_PREZEROED;
        }
        record(P(EV_A0), s);
        RUN(svol_attn_bwd_ex(P(QKV), 3 * D, at(P(QKV), D, dt), 3 * D, at(P(QKV), 2 * D, dt), 3 * D, P(O), D, P(DO), D, f32(P(LSE)),
                             f32(P(DELTA)), nullptr, P(DQKV), 3 * D, at(P(DQKV), D, dt), 3 * D, at(P(DQKV), 2 * D, dt), 3 * D, d.B, d.H, d.L,
                             d.L, dh, attn_scale(dh), attn_premul(dh, dt), P(ATTN_WS), d.ws_bytes, dt, aflags, P(EV_PREP), s));
        record(P(EV_A1), s);
        if (P(ATTN_WS_NEXT) && P(ZERO_STREAM) && P(EV_PREP) && P(EV_CLEAN_OUT)) {   // ... and the next layer's beside this one
            hipStream_t zs = static_cast<hipStream_t>(P(ZERO_STREAM));
            if (hipStreamWaitEvent(zs, static_cast<hipEvent_t>(P(EV_PREP)), 0) != hipSuccess) return SVOL_E_LAUNCH;
            const int zrc = svol_attn_bwd_zero_ws(P(ATTN_WS_NEXT), d.ws_bytes, d.B, d.H, d.L, d.L, dh, dt, zs);
            if (zrc) return zrc;
            record(P(EV_CLEAN_OUT), zs);
        }
        // d(y + pos) = [dq dk] W_qk ; dy = dv W_v
        RUN(nt(P(DQKV), 3 * D, P(W_IN_T), 3 * D, P(DXQP), D, nullptr, nullptr, M, D, 2 * D, dt, s));
        RUN(nt(at(P(DQKV), 2 * D, dt), 3 * D, at(P(W_IN_T), 2 * D, dt), 3 * D, P(DXQ), D, nullptr, nullptr, M, D, D, dt, s));
        RUN(svol_gate_bwd(f32(P(DS32_2)), P(DXQ), P(DXQP), f32(P(X32)), P(POS), f32(P(U)), f32(P(G1)), f32(P(A)), f32(P(MEAN1)),
                          f32(P(RSTD1)), f32(P(GATE_WS)), f32(P(GATE_WS2)), f32(P(DX32)), f32(P(DU)), f32(P(DG1)), f32(P(DBT1)), d.B, d.L,
                          D, d.H, dt, s));
    }
    return SVOL_OK;
}

int svol_video_half_wgrad_part(const int64_t* dims, void* const* p, int part, void* s) {
    if (!dims || !p || part < 0 || part > 2) return SVOL_E_INVALID;
    Trace tr_(part == 1 ? "video_half_wgrad.1" : part == 2 ? "video_half_wgrad.2" : "video_half_wgrad", s);
    const Dims d(dims);
    if (!d.ok()) return SVOL_E_INVALID;
    const int64_t M = d.B * d.L, D = d.D, F = d.F;
    const int dt = d.dt;
    // the block's five weight gradients in ONE launch (svol_gemm_tn_grouped) — or in two: the MLP / out-proj ones exist after
    // phase 1 of the backward (part 1: they can run beside THIS layer's attention backward), the in_proj ones after phase 2 (part 2)
    const svol_tn_problem pr[5] = {
        {P(DS3), D, P(HID), F, f32(P(DW_FC2)), F, nullptr, M, D, F},
        {P(DPRE), F, P(Y2), D, f32(P(DW_FC1)), D, nullptr, M, F, D},
        {P(G2D), D, P(O), D, f32(P(DW_O)), D, nullptr, M, D, D},
        {P(DQKV), 3 * D, P(Y1POS), D, f32(P(DW_IN)), D, f32(P(DB_IN)), M, 2 * D, D},
        {at(P(DQKV), 2 * D, dt), 3 * D, P(Y1), D, f32(P(DW_IN)) + 2 * D * D, D, f32(P(DB_IN)) + 2 * D, M, D, D}};
    if (part == 1) RUN(svol_gemm_tn_grouped(pr, 3, dt, s));
    else if (part == 2) RUN(svol_gemm_tn_grouped(pr + 3, 2, dt, s));
    else RUN(svol_gemm_tn_grouped(pr, 5, dt, s));
    return SVOL_OK;
}

int svol_video_half_wgrad(const int64_t* dims, void* const* p, void* s) { return svol_video_half_wgrad_part(dims, p, 0, s); }
#undef P

// ---------------------------------------------------------------------------------------------------------------------
// query self-attention
// ---------------------------------------------------------------------------------------------------------------------
int svol_query_self_fwd(const int64_t* dims, void* const* p, void* s) {
    if (!dims || !p) return SVOL_E_INVALID;
    Trace tr_("query_self_fwd", s);
    const Dims d(dims);
    if (!d.ok() || d.N <= 0) return SVOL_E_INVALID;
    const int64_t R = d.B * d.N, D = d.D, dh = D / d.H;
    const int qdt = d.qdt, xf = qdt != SVOL_F32;
#define P(n) p[SVOL_QS_##n]
    RUN(nt(P(OPOS), D, P(W_IN), D, P(QKV), 3 * D, P(B_IN), P(QSCALE), R, 2 * D, D, qdt, s));
    if (P(WV_HILO))
        RUN(svol_gemm_nt_split(P(O), D, P(WV_HILO), 2 * D, at(P(QKV), 2 * D, qdt), 3 * D, f32(P(B_IN)) + 2 * D, R, D, D, s));
    else
        RUN(nt(P(O), D, at(P(W_IN), 2 * D * D, qdt), D, at(P(QKV), 2 * D, qdt), 3 * D, f32(P(B_IN)) + 2 * D, nullptr, R, D, D, qdt, s));
    RUN(svol_attn_fwd(P(QKV), 3 * D, at(P(QKV), D, qdt), 3 * D, at(P(QKV), 2 * D, qdt), 3 * D, P(OA), D, f32(P(LSE)), nullptr, d.B, d.H,
                      d.N, d.N, dh, attn_scale(dh), attn_premul(dh, qdt), P(ATTN_WS), d.ws_bytes, qdt, s));
    RUN(nt_res(P(OA), D, P(W_O), D, P(S4), P(B_O), P(O32), R, D, D, qdt, s));
    RUN(svol_layernorm_fwd(P(S4), xf, f32(P(G4)), f32(P(BT4)), f32(P(Y32)), P(Y), P(YPOS), P(QPOS), d.N, f32(P(MEAN4)), f32(P(RSTD4)), R, D,
                           0.f, 0, nullptr, qdt, s));
    return SVOL_OK;
}

int svol_query_self_bwd(const int64_t* dims, void* const* p, void* s) {
    if (!dims || !p) return SVOL_E_INVALID;
    Trace tr_("query_self_bwd", s);
    const Dims d(dims);
    if (!d.ok() || d.N <= 0) return SVOL_E_INVALID;
    const int64_t R = d.B * d.N, D = d.D, dh = D / d.H;
    const int qdt = d.qdt, xf = qdt != SVOL_F32;
    RUN(svol_layernorm_bwd(f32(P(DY32)), P(DY), P(DYPOS), P(S4), xf, f32(P(G4)), f32(P(MEAN4)), f32(P(RSTD4)), f32(P(DO32)), P(G),
                           f32(P(DG4)), f32(P(DBT4)), f32(P(DB_O)), R, D, 0.f, 0, nullptr, qdt, s));
    if (P(DQPOS) && P(DYPOS))   // gradient of the broadcast query_pos operand: sum over the batch
        RUN(svol_colsum(P(DYPOS), d.N * D, f32(P(DQPOS)), d.B, d.N * D, qdt, s));
    RUN(nt(P(G), D, P(W_O_T), D, P(DOA), D, nullptr, nullptr, R, D, D, qdt, s));
    RUN(svol_attn_bwd(P(QKV), 3 * D, at(P(QKV), D, qdt), 3 * D, at(P(QKV), 2 * D, qdt), 3 * D, P(OA), D, P(DOA), D, f32(P(LSE)),
                      f32(P(DELTA)), nullptr, P(DQKV), 3 * D, at(P(DQKV), D, qdt), 3 * D, at(P(DQKV), 2 * D, qdt), 3 * D, d.B, d.H, d.N,
                      d.N, dh, attn_scale(dh), attn_premul(dh, qdt), P(ATTN_WS), d.ws_bytes, qdt, s));
    RUN(nt(P(DQKV), 3 * D, P(W_IN_T), 3 * D, P(DXQP), D, nullptr, nullptr, R, D, 2 * D, qdt, s));
    RUN(nt(at(P(DQKV), 2 * D, qdt), 3 * D, at(P(W_IN_T), 2 * D, qdt), 3 * D, P(DXQ), D, nullptr, nullptr, R, D, D, qdt, s));
    return SVOL_OK;
}

int svol_query_self_wgrad(const int64_t* dims, void* const* p, void* s) {
    if (!dims || !p) return SVOL_E_INVALID;
    Trace tr_("query_self_wgrad", s);
    const Dims d(dims);
    if (!d.ok()) return SVOL_E_INVALID;
    const int64_t R = d.B * d.N, D = d.D;
    const int qdt = d.qdt;
    const svol_tn_problem pr[3] = {
        {P(G), D, P(OA), D, f32(P(DW_O)), D, nullptr, R, D, D},
        {P(DQKV), 3 * D, P(OPOS), D, f32(P(DW_IN)), D, f32(P(DB_IN)), R, 2 * D, D},
        {at(P(DQKV), 2 * D, qdt), 3 * D, P(O), D, f32(P(DW_IN)) + 2 * D * D, D, f32(P(DB_IN)) + 2 * D, R, D, D}};
    RUN(svol_gemm_tn_grouped(pr, 3, qdt, s));
    return SVOL_OK;
#undef P
}

// ---------------------------------------------------------------------------------------------------------------------
// query -> video cross-attention + MLP2
// ---------------------------------------------------------------------------------------------------------------------
int svol_query_cross_fwd(const int64_t* dims, void* const* p, void* s) {
    if (!dims || !p) return SVOL_E_INVALID;
    Trace tr_("query_cross_fwd", s);
    const Dims d(dims);
    if (!d.ok() || d.N <= 0 || d.L <= 0 || d.F <= 0) return SVOL_E_INVALID;
    const int64_t R = d.B * d.N, M = d.B * d.L, D = d.D, F = d.F, dh = D / d.H;
    const int dt = d.dt, qdt = d.qdt, xf = qdt != SVOL_F32;
    const bool mixed = qdt != dt;
#define P(n) p[SVOL_QC_##n]
    if (mixed) {   // q in the query stream's precision, then ONE rounding into the attention core's
        RUN(nt(P(OPOS), D, P(W_INQ), D, P(Q), D, P(B_IN), P(QSCALE), R, D, D, qdt, s));
        RUN(svol_cast(P(Q), qdt, P(QC), dt, R * D, s));
    } else {
        RUN(nt(P(OPOS), D, P(W_KV), D, P(QC), D, P(B_IN), P(QSCALE), R, D, D, dt, s));
    }
    RUN(nt(P(MPOS), D, at(P(W_KV), D * D, dt), D, P(KV), 2 * D, f32(P(B_IN)) + D, nullptr, M, D, D, dt, s));
    if (P(WV_HILO))
        RUN(svol_gemm_nt_split(P(MV), D, P(WV_HILO), 2 * D, at(P(KV), D, dt), 2 * D, f32(P(B_IN)) + 2 * D, M, D, D, s));
    else
        RUN(nt(P(MV), D, at(P(W_KV), 2 * D * D, dt), D, at(P(KV), D, dt), 2 * D, f32(P(B_IN)) + 2 * D, nullptr, M, D, D, dt, s));
    RUN(svol_attn_fwd(P(QC), D, P(KV), 2 * D, at(P(KV), D, dt), 2 * D, P(OA), D, f32(P(LSE)), f32(P(KBIAS)), d.B, d.H, d.N, d.L, dh,
                      attn_scale(dh), attn_premul(dh, dt), P(ATTN_WS), d.ws_bytes, dt, s));
    const void* oq = P(OA);
    if (mixed) {
        RUN(svol_cast(P(OA), dt, P(OAQ), qdt, R * D, s));
        oq = P(OAQ);
    }
    RUN(nt_res(oq, D, P(W_O), D, P(S5), P(B_O), P(O32), R, D, D, qdt, s));
    RUN(svol_layernorm_fwd(P(S5), xf, f32(P(G5)), f32(P(BT5)), f32(P(Y5_32)), P(Y5), nullptr, nullptr, 0, f32(P(MEAN5)), f32(P(RSTD5)), R,
                           D, 0.f, 0, nullptr, qdt, s));
    RUN(svol_gemm_nt(P(Y5), D, nullptr, 0, P(W_FC1), D, P(HID), F, f32(P(B_FC1)), nullptr, SVOL_ACT_GELU, P(PRE), F, nullptr, 0, 0, R, F, D,
                     qdt, s));
    RUN(nt_res(P(HID), F, P(W_FC2), F, P(S6), P(B_FC2), P(Y5_32), R, D, F, qdt, s));
    RUN(svol_layernorm_fwd(P(S6), xf, f32(P(G6)), f32(P(BT6)), f32(P(Y32)), P(Y), P(YPOS), P(QPOS), d.N, f32(P(MEAN6)), f32(P(RSTD6)), R, D,
                           0.f, 0, nullptr, qdt, s));
    return SVOL_OK;
}

int svol_query_cross_bwd(const int64_t* dims, void* const* p, void* s) {
    if (!dims || !p) return SVOL_E_INVALID;
    Trace tr_("query_cross_bwd", s);
    const Dims d(dims);
    if (!d.ok() || d.N <= 0 || d.L <= 0 || d.F <= 0) return SVOL_E_INVALID;
    const int64_t R = d.B * d.N, M = d.B * d.L, D = d.D, F = d.F, dh = D / d.H;
    const int dt = d.dt, qdt = d.qdt, xf = qdt != SVOL_F32;
    const bool mixed = qdt != dt;
    // MLP2
    RUN(svol_layernorm_bwd(f32(P(DY32)), P(DY), P(DYPOS), P(S6), xf, f32(P(G6)), f32(P(MEAN6)), f32(P(RSTD6)), f32(P(DS32_6)), P(DS6),
                           f32(P(DG6)), f32(P(DBT6)), f32(P(DB_FC2)), R, D, 0.f, 0, nullptr, qdt, s));
    if (P(DQPOS) && P(DYPOS)) RUN(svol_colsum(P(DYPOS), d.N * D, f32(P(DQPOS)), d.B, d.N * D, qdt, s));
    RUN(svol_gemm_nt_dact(P(DS6), D, P(W_FC2_T), D, P(DPRE), F, P(PRE), F, SVOL_ACT_GELU, f32(P(DB_FC1)), R, F, D, qdt, s));
    RUN(nt(P(DPRE), F, P(W_FC1_T), F, P(DY5), D, nullptr, nullptr, R, D, F, qdt, s));
    // LN5' ; do = g Wo
    RUN(svol_layernorm_bwd(f32(P(DS32_6)), P(DY5), nullptr, P(S5), xf, f32(P(G5)), f32(P(MEAN5)), f32(P(RSTD5)), f32(P(DO32)), P(G5D),
                           f32(P(DG5)), f32(P(DBT5)), f32(P(DB_O)), R, D, 0.f, 0, nullptr, qdt, s));
    RUN(nt(P(G5D), D, P(W_O_T), D, P(DOAQ), D, nullptr, nullptr, R, D, D, qdt, s));
    const void* d_o = P(DOAQ);
    if (mixed) {
        RUN(svol_cast(P(DOAQ), qdt, P(DOA), dt, R * D, s));
        d_o = P(DOA);
    }
    RUN(svol_attn_bwd(P(QC), D, P(KV), 2 * D, at(P(KV), D, dt), 2 * D, P(OA), D, d_o, D, f32(P(LSE)), f32(P(DELTA)), f32(P(KBIAS)), P(DQC), D,
                      P(DKV), 2 * D, at(P(DKV), D, dt), 2 * D, d.B, d.H, d.N, d.L, dh, attn_scale(dh), attn_premul(dh, dt), P(ATTN_WS),
                      d.ws_bytes, dt, s));
    // the two video-sized products first: the video half's backward waits for them
    RUN(nt(P(DKV), 2 * D, at(P(W_KV_T), D, dt), 3 * D, P(DMPOS), D, nullptr, nullptr, M, D, D, dt, s));
    RUN(nt(at(P(DKV), D, dt), 2 * D, at(P(W_KV_T), 2 * D, dt), 3 * D, P(DMV), D, nullptr, nullptr, M, D, D, dt, s));
    const void* dq = P(DQC);
    if (mixed) {
        RUN(svol_cast(P(DQC), dt, P(DQ), qdt, R * D, s));
        dq = P(DQ);
    }
    RUN(nt(dq, D, mixed ? P(W_INQ_T) : P(W_KV_T), 3 * D, P(DXQP), D, nullptr, nullptr, R, D, D, qdt, s));
    return SVOL_OK;
}

int svol_query_cross_wgrad(const int64_t* dims, void* const* p, void* s) {
    if (!dims || !p) return SVOL_E_INVALID;
    Trace tr_("query_cross_wgrad", s);
    const Dims d(dims);
    if (!d.ok()) return SVOL_E_INVALID;
    const int64_t R = d.B * d.N, M = d.B * d.L, D = d.D, F = d.F;
    const int dt = d.dt, qdt = d.qdt;
    const bool mixed = qdt != dt;
    // the video-sized ones first (they are what the stream they share with the video half's weight gradients is sized for)
    const svol_tn_problem pv[2] = {
        {P(DKV), 2 * D, P(MPOS), D, f32(P(DW_IN)) + D * D, D, f32(P(DB_IN)) + D, M, D, D},
        {at(P(DKV), D, dt), 2 * D, P(MV), D, f32(P(DW_IN)) + 2 * D * D, D, f32(P(DB_IN)) + 2 * D, M, D, D}};
    RUN(svol_gemm_tn_grouped(pv, 2, dt, s));
    const svol_tn_problem pq[4] = {
        {P(DS6), D, P(HID), F, f32(P(DW_FC2)), F, nullptr, R, D, F},
        {P(DPRE), F, P(Y5), D, f32(P(DW_FC1)), D, nullptr, R, F, D},
        {P(G5D), D, mixed ? P(OAQ) : P(OA), D, f32(P(DW_O)), D, nullptr, R, D, D},
        {mixed ? P(DQ) : P(DQC), D, P(OPOS), D, f32(P(DW_IN)), D, f32(P(DB_IN)), R, D, D}};
    RUN(svol_gemm_tn_grouped(pq, 4, qdt, s));
    return SVOL_OK;
#undef P
}

}  // extern "C"
