// Set matching + criterion on device (reference lib/modeling/matcher.py, lib/modeling/loss.py,
// lib/utils/box_utils.py and scipy.optimize.linear_sum_assignment).
//
// The reference builds the FULL [B*N, sumM] cost matrix, copies it to the host (.cpu() sync,
// matcher.py:86/156), and calls scipy once per block in a Python loop, once per decoder layer.
// Here only the diagonal blocks the reference actually uses are computed, and every block of every
// layer is solved in one launch (one wavefront per block) with no host round trip.
//
// Compiled with -ffp-contract=off: the cost arithmetic mirrors torch's separate fp32 ops (no FMA
// fusion) and the solver mirrors scipy's fp64 arithmetic, so assignments are bit-exact.
#include "common.h"

namespace {

struct Box { float x0, y0, x1, y1; };
__device__ __forceinline__ Box to_xyxy(const float* b) {  // box_utils.py:9-13
    return Box{b[0] - 0.5f * b[2], b[1] - 0.5f * b[3], b[0] + 0.5f * b[2], b[1] + 0.5f * b[3]};
}
__device__ __forceinline__ float giou_pair(const Box& a, const Box& b) {  // box_utils.py:24-61
    const float area1 = (a.x1 - a.x0) * (a.y1 - a.y0);
    const float area2 = (b.x1 - b.x0) * (b.y1 - b.y0);
    const float iw = fmaxf(fminf(a.x1, b.x1) - fmaxf(a.x0, b.x0), 0.f);
    const float ih = fmaxf(fminf(a.y1, b.y1) - fmaxf(a.y0, b.y0), 0.f);
    const float inter = iw * ih;
    const float uni = area1 + area2 - inter;
    const float iou = inter / uni;
    const float ew = fmaxf(fmaxf(a.x1, b.x1) - fminf(a.x0, b.x0), 0.f);
    const float eh = fmaxf(fmaxf(a.y1, b.y1) - fminf(a.y0, b.y0), 0.f);
    const float area = ew * eh;
    return iou - (area - uni) / area;
}

__global__ __launch_bounds__(256) void match_cost_kernel(const float* __restrict__ logits, const float* __restrict__ boxes,
                                                         const float* __restrict__ tgt, const int32_t* __restrict__ pred_off,
                                                         const int32_t* __restrict__ pred_cnt,
                                                         const int32_t* __restrict__ tgt_off,
                                                         const int32_t* __restrict__ tgt_cnt,
                                                         const int64_t* __restrict__ cost_off, float* __restrict__ cost,
                                                         float w_bbox, float w_giou, float w_class,
                                                         int32_t* __restrict__ box_status) {
    const int p = blockIdx.x;
    const int np = pred_cnt[p], nt = tgt_cnt[p];
    const int po = pred_off[p], to = tgt_off[p];
    float* C = cost + cost_off[p];
    if (box_status) {
        // generalized_box_iou's early check (box_utils.py:51-52): every box must satisfy x1 >= x0 and y1 >= y0 (a NaN
        // coordinate fails it too).  The reference asserts on the host; here the problem is flagged and the host raises
        // the AssertionError lazily (PackedTargets.check_status) — the criterion itself never synchronises.
        int bad = 0;
        for (int i = threadIdx.x; i < np; i += blockDim.x) {
            const Box b = to_xyxy(boxes + (int64_t)(po + i) * 4);
            bad |= !(b.x1 >= b.x0) || !(b.y1 >= b.y0);
        }
        for (int j = threadIdx.x; j < nt; j += blockDim.x) {
            const Box b = to_xyxy(tgt + (int64_t)(to + j) * 4);
            bad |= !(b.x1 >= b.x0) || !(b.y1 >= b.y0);
        }
        bad = __syncthreads_or(bad);
        if (threadIdx.x == 0) box_status[p] = bad ? 1 : 0;
    }
    for (int e = threadIdx.x; e < np * nt; e += blockDim.x) {
        const int i = e / nt, j = e - i * nt;
        const float* lg = logits + (int64_t)(po + i) * 2;
        const float* pb = boxes + (int64_t)(po + i) * 4;
        const float* tb = tgt + (int64_t)(to + j) * 4;
        // softmax(-1)[0]  (matcher.py:59)
        const float mx = fmaxf(lg[0], lg[1]);
        const float e0 = expf(lg[0] - mx), e1 = expf(lg[1] - mx);
        const float p0 = e0 / (e0 + e1);
        // torch.cdist(p=1) (matcher.py:79)
        const float l1 = ((fabsf(pb[0] - tb[0]) + fabsf(pb[1] - tb[1])) + fabsf(pb[2] - tb[2])) + fabsf(pb[3] - tb[3]);
        const float g = giou_pair(to_xyxy(pb), to_xyxy(tb));
        C[e] = (w_bbox * l1 + w_giou * (-g)) + w_class * (-p0);  // matcher.py:85
    }
}

// ---------------------------------------------------------------------------
// Batched LSAP: one wave64 per problem.  Restates scipy's rectangular_lsap (Crouse 2016).
__device__ __forceinline__ double wave_min_d(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmin(x, __shfl_xor(x, o, 64));
    return x;
}
__device__ __forceinline__ int wave_min_i(int x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = min(x, __shfl_xor(x, o, 64));
    return x;
}
__device__ __forceinline__ int wave_max_i(int x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = max(x, __shfl_xor(x, o, 64));
    return x;
}

// (value, preference key) arg-min over the wave in ONE butterfly: smaller value wins, ties go to the LARGER key.  The four
// intra-row steps are DPP lane permutations (quad xor 1 / xor 2, row_half_mirror, row_mirror: no LDS traffic, a few cycles
// each), the row-pair step is v_permlane16_swap and the last step v_permlane32_swap (gfx950); every lane
// ends up with the wave's winner.  (Three __shfl_xor reductions — a double min, an int max, an int min = 24 dependent
// ds_bpermutes — were 0.3 ms of this kernel's 0.55 ms critical path at the benchmark size.)
struct MinKey { double s; int key; };
__device__ __forceinline__ MinKey better(const MinKey a, const MinKey b) { return (b.s < a.s || (b.s == a.s && b.key > a.key)) ? b : a; }
template <int CTRL>
__device__ __forceinline__ MinKey dpp_step(const MinKey a) {
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, a.s);
    const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)bits, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(bits >> 32), CTRL, 0xf, 0xf, false);
    MinKey o;
    o.s = __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    o.key = __builtin_amdgcn_mov_dpp(a.key, CTRL, 0xf, 0xf, false);
    return better(a, o);
}
__device__ __forceinline__ MinKey wave_argmin(MinKey a) {
    a = dpp_step<0xB1>(a);   // quad_perm [1,0,3,2]
    a = dpp_step<0x4E>(a);   // quad_perm [2,3,0,1]
    a = dpp_step<0x141>(a);  // row_half_mirror: quads 0<->1, 2<->3 (every lane of a quad already holds the quad's winner)
    a = dpp_step<0x140>(a);  // row_mirror: halves of the 16-lane row
    {   // rows 0 <-> 1 and 2 <-> 3 of the wave: v_permlane16_swap (no LDS round trip)
        const unsigned long long ab = __builtin_bit_cast(unsigned long long, a.s);
        const HalfPair l16 = swap_rows16((unsigned)ab), h16 = swap_rows16((unsigned)(ab >> 32)), k16 = swap_rows16((unsigned)a.key);
        MinKey e, o;
        e.s = __builtin_bit_cast(double, ((unsigned long long)h16.lo << 32) | l16.lo);
        e.key = (int)k16.lo;
        o.s = __builtin_bit_cast(double, ((unsigned long long)h16.hi << 32) | l16.hi);
        o.key = (int)k16.hi;
        a = better(e, o);
    }
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, a.s);
    const HalfPair l = swap_halves((unsigned)bits), hw = swap_halves((unsigned)(bits >> 32)), k = swap_halves((unsigned)a.key);
    MinKey p0, p1;  // the winners of lanes 0-31 and of lanes 32-63, both visible in every lane
    p0.s = __builtin_bit_cast(double, ((unsigned long long)hw.lo << 32) | l.lo);
    p0.key = (int)k.lo;
    p1.s = __builtin_bit_cast(double, ((unsigned long long)hw.hi << 32) | l.hi);
    p1.key = (int)k.hi;
    return better(p0, p1);
}

// Register-resident solver for the common sizes (after scipy's transpose: nr <= 64 rows, nc <= 128 columns — the 100 object
// queries against up to 64 boxes of a video).  The LDS version below walks four LDS arrays per column per step, every access a
// dependent ~100-cycle round trip in a single wave; here lane l owns columns l and l + 64 (v, shortest-path cost, path, row4col,
// scanned flag, position in scipy's `remaining` list) and row l (u, col4row, scanned flag) in registers, the list's swap-remove
// is a position update in the two lanes involved, lookups by index are v_readlane, and only the cost row comes from LDS.
// Arithmetic, visiting order and the tie rule are those of the LDS version (and of scipy): results are bit-identical.
__device__ __forceinline__ double readlane_f64(double x, int l) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, l), hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// wave-wide min / max of a 32-bit key, result uniform (SGPR): DPP quad / row permutations, then row_bcast15 / row_bcast31
// gather the four 16-lane rows into lane 63.  With the folded DPP operand each stage is one v_min_u32 / v_max_u32.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned x) {
    if constexpr (ROW_MASK == 0xf) return (unsigned)__builtin_amdgcn_mov_dpp((int)x, CTRL, 0xf, 0xf, true);  // folds into the consumer
    else return (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, ROW_MASK, 0xf, false);         // masked rows keep x
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned x) {
    x = min(x, dpp_u32<0xB1, 0xf>(x));
    x = min(x, dpp_u32<0x4E, 0xf>(x));
    x = min(x, dpp_u32<0x141, 0xf>(x));
    x = min(x, dpp_u32<0x140, 0xf>(x));
    x = min(x, dpp_u32<0x142, 0xa>(x));   // row_bcast15 into rows 1 and 3
    x = min(x, dpp_u32<0x143, 0xc>(x));   // row_bcast31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned x) {
    x = max(x, dpp_u32<0xB1, 0xf>(x));
    x = max(x, dpp_u32<0x4E, 0xf>(x));
    x = max(x, dpp_u32<0x141, 0xf>(x));
    x = max(x, dpp_u32<0x140, 0xf>(x));
    x = max(x, dpp_u32<0x142, 0xa>(x));
    x = max(x, dpp_u32<0x143, 0xc>(x));
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}
// order-preserving map double -> uint64 (x < y  <=>  mono(x) < mono(y); -0.0 is folded onto +0.0 first so that equal doubles
// map to equal keys): the arg-min below compares integers, 32 bits at a time
__device__ __forceinline__ unsigned long long mono_f64(double x) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x + 0.0);
    const unsigned long long m = (unsigned long long)((long long)b >> 63);
    return b ^ (m | 0x8000000000000000ull);
}
__device__ __forceinline__ double unmono_f64(unsigned long long k) {
    const unsigned long long m = (k >> 63) ? 0x8000000000000000ull : ~0ull;
    return __builtin_bit_cast(double, k ^ m);
}
__device__ __forceinline__ bool lsap_reg(const float* __restrict__ Cs, int nr, int nc, int lane, int (&col4row_out)) {
    const int j0 = lane, j1 = lane + 64;
    const bool ok0 = j0 < nc, ok1 = j1 < nc;
    const int j0c = ok0 ? j0 : 0, j1c = ok1 ? j1 : 0;  // clamped cost-row offsets (the value is discarded for a missing column)
    double v0 = 0.0, v1 = 0.0, spc0 = INFINITY, spc1 = INFINITY, u = 0.0;
    int path0 = -1, path1 = -1, r4c0 = -1, r4c1 = -1, pos0 = -1, pos1 = -1, c4r = -1;
    bool sc0 = false, sc1 = false, sr = false;
    const unsigned long long K_INF = 0xFFF0000000000000ull;  // mono_f64(+inf)
    for (int cur = 0; cur < nr; ++cur) {
        pos0 = ok0 ? nc - 1 - j0 : -1;  // remaining[it] = nc - it - 1
        pos1 = ok1 ? nc - 1 - j1 : -1;
        sc0 = sc1 = sr = false;
        spc0 = spc1 = INFINITY;
        int num_remaining = nc, i = cur, sink = -1;
        double min_val = 0.0;
        while (sink == -1) {
            sr = sr || lane == i;
            const double ui = readlane_f64(u, i);
            const float* crow = Cs + i * nc;
            const double r0 = min_val + (double)crow[j0c] - ui - v0;
            const double r1 = min_val + (double)crow[j1c] - ui - v1;
            const bool a0 = pos0 >= 0, a1 = pos1 >= 0;
            const bool up0 = a0 && r0 < spc0, up1 = a1 && r1 < spc1;
            spc0 = up0 ? r0 : spc0;
            path0 = up0 ? i : path0;
            spc1 = up1 ? r1 : spc1;
            path1 = up1 ? i : path1;
            // (cost, preference) of the lane's two columns.  scipy's rule over the remaining columns: lowest cost; among equal costs the
            // LAST unassigned column visited, else the FIRST column visited -> preference = [unassigned][pos | 127 - pos], larger wins;
            // slot and lane ride in the low bits (positions are unique, they never decide)
            const unsigned long long k0 = a0 ? mono_f64(spc0) : ~0ull, k1 = a1 ? mono_f64(spc1) : ~0ull;
            const unsigned t0 = a0 ? (0x100000u | ((r4c0 == -1 ? 0x80u | (unsigned)pos0 : 127u - (unsigned)pos0) << 8) | (unsigned)lane) : 0u;
            const unsigned t1 = a1 ? (0x100000u | ((r4c1 == -1 ? 0x80u | (unsigned)pos1 : 127u - (unsigned)pos1) << 8) | 0x40u | (unsigned)lane) : 0u;
            const bool second = k1 < k0 || (k1 == k0 && t1 > t0);
            const unsigned long long k = second ? k1 : k0;
            const unsigned t = second ? t1 : t0;
            const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
            const unsigned mh = wave_min_u32(hi);
            // usually ONE lane holds the smallest upper word (sign, exponent, 20 mantissa bits): it is the arg-min, whatever the rest says
            const unsigned long long cand = __ballot(hi == mh);
            unsigned ml, mt;
            if ((cand & (cand - 1)) == 0) {
                const int w = __builtin_ctzll(cand);
                ml = (unsigned)__builtin_amdgcn_readlane((int)lo, w);
                mt = (unsigned)__builtin_amdgcn_readlane((int)t, w);
            } else {
                ml = wave_min_u32(hi == mh ? lo : 0xffffffffu);
                mt = wave_max_u32((hi == mh && lo == ml) ? t : 0u);
            }
            const unsigned long long kmin = ((unsigned long long)mh << 32) | ml;
            if (kmin >= K_INF) return false;  // infeasible: the cheapest remaining column costs +inf
            min_val = unmono_f64(kmin);
            const unsigned pf = (mt >> 8) & 0xffu;
            const int index = (pf & 0x80u) ? (int)(pf & 0x7fu) : 127 - (int)pf;
            const int lj = (int)(mt & 0x3fu), slot = (int)((mt >> 6) & 1u);
            const int j = lj + 64 * slot;
            const int r4 = __builtin_amdgcn_readlane(slot ? r4c1 : r4c0, lj);
            if (r4 == -1) sink = j; else i = r4;
            const int last = num_remaining - 1;
            // SC[j] = true; remaining[index] = remaining[last]; --num_remaining
            const bool me0 = lane == lj && slot == 0, me1 = lane == lj && slot == 1;
            sc0 = sc0 || me0;
            sc1 = sc1 || me1;
            pos0 = me0 ? -1 : (pos0 == last ? index : pos0);
            pos1 = me1 ? -1 : (pos1 == last ? index : pos1);
            num_remaining = last;
        }
        // dual update
        if (lane == cur) u += min_val;
        {
            const int c = c4r < 0 ? 0 : c4r;
            const double s0 = __shfl(spc0, c & 63, 64), s1 = __shfl(spc1, c & 63, 64);
            if (sr && lane != cur && lane < nr) u += min_val - ((c >> 6) ? s1 : s0);
        }
        if (sc0) v0 -= min_val - spc0;
        if (sc1) v1 -= min_val - spc1;
        // augment along the path
        int j = sink;
        for (;;) {
            const int lj = j & 63, slot = j >> 6;
            const int ii = slot ? __builtin_amdgcn_readlane(path1, lj) : __builtin_amdgcn_readlane(path0, lj);
            if (lane == lj) { if (slot) r4c1 = ii; else r4c0 = ii; }
            const int t = __builtin_amdgcn_readlane(c4r, ii);
            if (lane == ii) c4r = j;
            j = t;
            if (ii == cur) break;
        }
    }
    col4row_out = c4r;
    return true;
}

// Round 6: the same solver with a shorter step.  One search step of lsap_reg is ~150 instructions of ONE wave (~900 cycles: the 48
// problems of a training step run one wave each on 48 CUs, and the launch — 0.2 ms — sits on the forward -> backward dependency
// chain with the rest of the chip idle).  What the step needs per column pair is the smallest UPPER word of the two keys; the lower
// word, the tie preference and the per-lane pre-selection only matter when two candidates share an upper word (sign, exponent, 20
// mantissa bits) — then the full procedure of lsap_reg runs (`slow`), bit for bit.  Otherwise the one candidate IS the arg-min:
// its value, list position and row are four v_readlane.  Scanned-column flags are the removed positions (pos < 0), scanned rows a
// scalar bit mask, the row index a scalar (v_readfirstlane: the cost row's address is SALU), the second column's cost an immediate
// offset from the first (the staged block is padded: the value of a column >= nc is discarded).  Same arithmetic in the same
// order, same visiting order, same tie rule: assignments bit-identical to lsap_reg's (tests: scipy blocks incl. ties, goldens).
__device__ __forceinline__ unsigned mono_hi32(unsigned hi) { return hi ^ ((unsigned)((int)hi >> 31) | 0x80000000u); }
__device__ __forceinline__ bool lsap_reg2(const float* __restrict__ Cs, int nr, int nc, int lane, int (&col4row_out)) {
    const int j0 = lane, j1 = lane + 64;
    const bool ok0 = j0 < nc, ok1 = j1 < nc;
    double v0 = 0.0, v1 = 0.0, u = 0.0;
    int path0 = -1, path1 = -1, r4c0 = -1, r4c1 = -1, c4r = -1;
    const unsigned long long K_INF = 0xFFF0000000000000ull;  // mono_f64(+inf)
    for (int cur = 0; cur < nr; ++cur) {
        int pos0 = ok0 ? nc - 1 - j0 : -1;  // remaining[it] = nc - it - 1; < 0: not in the list (scanned, or no such column)
        int pos1 = ok1 ? nc - 1 - j1 : -1;
        double spc0 = INFINITY, spc1 = INFINITY;
        unsigned long long sr = 0;           // scanned rows
        int num_remaining = nc, i = cur, sink = -1;
        double min_val = 0.0;
        while (sink == -1) {
            i = __builtin_amdgcn_readfirstlane(i);
            sr |= 1ull << i;
            const double ui = readlane_f64(u, i);
            const float* crow = Cs + i * nc + lane;
            const double r0 = min_val + (double)crow[0] - ui - v0;
            const double r1 = min_val + (double)crow[64] - ui - v1;
            const bool up0 = pos0 >= 0 && r0 < spc0, up1 = pos1 >= 0 && r1 < spc1;
            spc0 = up0 ? r0 : spc0;
            path0 = up0 ? i : path0;
            spc1 = up1 ? r1 : spc1;
            path1 = up1 ? i : path1;
            const double f0 = spc0 + 0.0, f1 = spc1 + 0.0;   // -0.0 folded onto +0.0 (equal doubles, equal keys)
            const unsigned h0 = mono_hi32((unsigned)(__builtin_bit_cast(unsigned long long, f0) >> 32)) | (unsigned)(pos0 >> 31);
            const unsigned h1 = mono_hi32((unsigned)(__builtin_bit_cast(unsigned long long, f1) >> 32)) | (unsigned)(pos1 >> 31);
            const unsigned mh = wave_min_u32(min(h0, h1));
            if (mh >= (unsigned)(K_INF >> 32)) return false;  // infeasible: the cheapest remaining column costs +inf (or none remains)
            const unsigned long long c0 = __ballot(h0 == mh), c1 = __ballot(h1 == mh);
            int lj, slot, index, r4;
            if (__builtin_popcountll(c0) + __builtin_popcountll(c1) == 1) {
                if (c0) {
                    lj = __builtin_ctzll(c0); slot = 0;
                    min_val = readlane_f64(f0, lj);
                    index = __builtin_amdgcn_readlane(pos0, lj);
                    r4 = __builtin_amdgcn_readlane(r4c0, lj);
                } else {
                    lj = __builtin_ctzll(c1); slot = 1;
                    min_val = readlane_f64(f1, lj);
                    index = __builtin_amdgcn_readlane(pos1, lj);
                    r4 = __builtin_amdgcn_readlane(r4c1, lj);
                }
            } else {   // `slow`: several candidates share the upper word — lsap_reg's full (cost, preference) arg-min
                const bool a0 = pos0 >= 0, a1 = pos1 >= 0;
                const unsigned long long k0 = a0 ? mono_f64(spc0) : ~0ull, k1 = a1 ? mono_f64(spc1) : ~0ull;
                const unsigned t0 = a0 ? (0x100000u | ((r4c0 == -1 ? 0x80u | (unsigned)pos0 : 127u - (unsigned)pos0) << 8) | (unsigned)lane) : 0u;
                const unsigned t1 = a1 ? (0x100000u | ((r4c1 == -1 ? 0x80u | (unsigned)pos1 : 127u - (unsigned)pos1) << 8) | 0x40u | (unsigned)lane) : 0u;
                const bool second = k1 < k0 || (k1 == k0 && t1 > t0);
                const unsigned long long k = second ? k1 : k0;
                const unsigned t = second ? t1 : t0;
                const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
                const unsigned ml = wave_min_u32(hi == mh ? lo : 0xffffffffu);
                const unsigned mt = wave_max_u32((hi == mh && lo == ml) ? t : 0u);
                min_val = unmono_f64(((unsigned long long)mh << 32) | ml);
                const unsigned pf = (mt >> 8) & 0xffu;
                index = (pf & 0x80u) ? (int)(pf & 0x7fu) : 127 - (int)pf;
                lj = (int)(mt & 0x3fu);
                slot = (int)((mt >> 6) & 1u);
                r4 = __builtin_amdgcn_readlane(slot ? r4c1 : r4c0, lj);
            }
            if (r4 == -1) sink = lj + 64 * slot; else i = r4;
            const int last = num_remaining - 1;
            // SC[j] = true; remaining[index] = remaining[last]; --num_remaining
            const bool me = lane == lj;
            pos0 = (me && slot == 0) ? -1 : (pos0 == last ? index : pos0);
            pos1 = (me && slot == 1) ? -1 : (pos1 == last ? index : pos1);
            num_remaining = last;
        }
        // dual update
        if (lane == cur) u += min_val;
        {
            const int c = c4r < 0 ? 0 : c4r;
            const double s0 = __shfl(spc0, c & 63, 64), s1 = __shfl(spc1, c & 63, 64);
            if (((sr >> lane) & 1ull) && lane != cur && lane < nr) u += min_val - ((c >> 6) ? s1 : s0);
        }
        if (ok0 && pos0 < 0) v0 -= min_val - spc0;
        if (ok1 && pos1 < 0) v1 -= min_val - spc1;
        // augment along the path
        int j = sink;
        for (;;) {
            const int lj = j & 63, slot = j >> 6;
            const int ii = slot ? __builtin_amdgcn_readlane(path1, lj) : __builtin_amdgcn_readlane(path0, lj);
            if (lane == lj) { if (slot) r4c1 = ii; else r4c0 = ii; }
            const int t = __builtin_amdgcn_readlane(c4r, ii);
            if (lane == ii) c4r = j;
            j = t;
            if (ii == cur) break;
        }
    }
    col4row_out = c4r;
    return true;
}

__global__ __launch_bounds__(64) void lsap_kernel(const float* __restrict__ cost, const int64_t* __restrict__ cost_off,
                                                  const int32_t* __restrict__ pred_off, const int32_t* __restrict__ pred_cnt,
                                                  const int32_t* __restrict__ tgt_off, const int32_t* __restrict__ tgt_cnt,
                                                  int32_t* __restrict__ match, int32_t* __restrict__ status, int max_dim,
                                                  int stage_floats, int v1) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    double* u = reinterpret_cast<double*>(lds);
    double* v = u + max_dim;
    double* spc = v + max_dim;
    int* path = reinterpret_cast<int*>(spc + max_dim);
    int* col4row = path + max_dim;
    int* row4col = col4row + max_dim;
    int* remaining = row4col + max_dim;
    unsigned char* SR = reinterpret_cast<unsigned char*>(remaining + max_dim);
    unsigned char* SC = SR + max_dim;
    // the cost block, staged in LDS in the orientation the solver walks it (row i contiguous): every step of the
    // augmenting-path search is a dependent read of one cost row, ~1-2 us from global memory vs ~100 cycles from LDS
    float* Cs = reinterpret_cast<float*>(lds + (size_t)max_dim * (3 * 8 + 4 * 4 + 2));

    const int p = blockIdx.x, lane = threadIdx.x;
    const int np = pred_cnt[p], nt = tgt_cnt[p];
    const int po = pred_off[p], to = tgt_off[p];
    const float* C = cost + cost_off[p];
    for (int i = lane; i < np; i += 64) match[po + i] = -1;
    if (lane == 0) status[p] = 0;
    if (np == 0 || nt == 0) return;
    const bool tr = nt < np;  // tall matrix -> solve the transpose (scipy)
    const int nr = tr ? nt : np, nc = tr ? np : nt;
    const bool staged = np * nt <= stage_floats;
    auto cst = [&](int i, int j) -> double {
        if (staged) return (double)Cs[i * nc + j];
        return (double)(tr ? C[(int64_t)j * nt + i] : C[(int64_t)i * nt + j]);
    };

    // NaN / -inf -> "matrix contains invalid numeric entries"
    int bad = 0;
    for (int e = lane; e < np * nt; e += 64) {
        const float c = C[e];
        if (c != c || c == -INFINITY) bad = 1;
        if (staged) {
            const int pi = e / nt, tj = e - pi * nt;  // C is [np][nt]
            Cs[tr ? tj * nc + pi : pi * nc + tj] = c;
        }
    }
    if (__any(bad)) {
        if (lane == 0) status[p] = 1;
        return;
    }
    if (staged && nr <= 64 && nc <= 128) {  // register-resident path (see lsap_reg)
        __syncthreads();
        int c4 = -1;
        if (!(v1 ? lsap_reg(Cs, nr, nc, lane, c4) : lsap_reg2(Cs, nr, nc, lane, c4))) {
            if (lane == 0) status[p] = 2;
            return;
        }
        if (lane < nr) {
            if (tr) match[po + c4] = to + lane;
            else match[po + lane] = to + c4;
        }
        return;
    }
    for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = lane; j < nc; j += 64) { v[j] = 0.0; path[j] = -1; row4col[j] = -1; }
    __syncthreads();

    for (int cur = 0; cur < nr; ++cur) {
        for (int it = lane; it < nc; it += 64) { remaining[it] = nc - it - 1; SC[it] = 0; spc[it] = INFINITY; }
        for (int i = lane; i < nr; i += 64) SR[i] = 0;
        __syncthreads();
        int num_remaining = nc;
        double min_val = 0.0;
        int i = cur, sink = -1;
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            const double ui = u[i];
            double lmin = INFINITY;
            int first_it = 0x7fffffff, last_una = -1;
            for (int it = lane; it < num_remaining; it += 64) {
                const int j = remaining[it];
                const double r = min_val + cst(i, j) - ui - v[j];
                double s = spc[j];
                if (r < s) { path[j] = i; spc[j] = r; s = r; }
                const bool una = row4col[j] == -1;
                if (s < lmin) { lmin = s; first_it = it; last_una = una ? it : -1; }
                else if (s == lmin) {
                    if (first_it == 0x7fffffff) first_it = it;  // s == lmin == +inf on the first visit
                    if (una) last_una = it;
                }
            }
            // scipy's rule over the remaining columns: lowest cost; among equal costs the LAST unassigned column visited,
            // else the FIRST column visited.  Per lane that is (lmin, last_una >= 0 ? last_una : first_it); the key makes
            // any unassigned candidate beat any assigned one, larger `it` better among the former, smaller among the latter.
            MinKey mk;
            mk.s = lmin;
            mk.key = last_una >= 0 ? 0x40000000 + last_una : 0x3fffffff - min(first_it, 0x3fffffff);
            mk = wave_argmin(mk);
            const double gmin = mk.s;
            const int index = mk.key >= 0x40000000 ? mk.key - 0x40000000 : 0x3fffffff - mk.key;
            min_val = gmin;
            if (min_val == INFINITY) {  // infeasible
                if (lane == 0) status[p] = 2;
                return;
            }
            __syncthreads();
            const int j = remaining[index];
            const int r4c = row4col[j];
            if (r4c == -1) sink = j; else i = r4c;
            __syncthreads();
            if (lane == 0) { SC[j] = 1; remaining[index] = remaining[num_remaining - 1]; }
            --num_remaining;
            __syncthreads();
        }
        // dual update
        if (lane == 0) u[cur] += min_val;
        for (int ii = lane; ii < nr; ii += 64)
            if (SR[ii] && ii != cur) u[ii] += min_val - spc[col4row[ii]];
        for (int j = lane; j < nc; j += 64)
            if (SC[j]) v[j] -= min_val - spc[j];
        __syncthreads();
        // augment
        if (lane == 0) {
            int j = sink;
            for (;;) {
                const int ii = path[j];
                row4col[j] = ii;
                const int t = col4row[ii];
                col4row[ii] = j;
                j = t;
                if (ii == cur) break;
            }
        }
        __syncthreads();
    }
    if (tr) {
        for (int ii = lane; ii < nr; ii += 64) match[po + col4row[ii]] = to + ii;
    } else {
        for (int ii = lane; ii < nr; ii += 64) match[po + ii] = to + col4row[ii];
    }
}

// ---------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_d(double x, double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    __syncthreads();
    if (lane == 0) red[wave] = x;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// d(giou)/d(pred cxcywh), pred = b[0..3], target t[0..3]
__device__ void giou_grad(const float* b, const float* t, float& giou, float (&gb)[4]) {
    const float x0 = b[0] - 0.5f * b[2], y0 = b[1] - 0.5f * b[3], x1 = b[0] + 0.5f * b[2], y1 = b[1] + 0.5f * b[3];
    const float tx0 = t[0] - 0.5f * t[2], ty0 = t[1] - 0.5f * t[3], tx1 = t[0] + 0.5f * t[2], ty1 = t[1] + 0.5f * t[3];
    const float area1 = (x1 - x0) * (y1 - y0), area2 = (tx1 - tx0) * (ty1 - ty0);
    const float dix = fminf(x1, tx1) - fmaxf(x0, tx0), diy = fminf(y1, ty1) - fmaxf(y0, ty0);
    const float iw = fmaxf(dix, 0.f), ih = fmaxf(diy, 0.f);
    const float inter = iw * ih;
    const float uni = area1 + area2 - inter;
    const float iou = inter / uni;
    const float dex = fmaxf(x1, tx1) - fminf(x0, tx0), dey = fmaxf(y1, ty1) - fminf(y0, ty0);
    const float ew = fmaxf(dex, 0.f), eh = fmaxf(dey, 0.f);
    const float area = ew * eh;
    giou = iou - (area - uni) / area;
    // reverse mode, upstream d/dgiou = 1
    const float G_union = -inter / (uni * uni) + 1.f / area;
    const float G_inter = 1.f / uni - G_union;
    const float G_area = -uni / (area * area);
    const float G_area1 = G_union;
    auto gsel = [](float a, float c, bool want_greater) -> float {  // share of the gradient going to `a`
        if (a == c) return 0.5f;
        return ((a > c) == want_greater) ? 1.f : 0.f;
    };
    const float G_iw = dix >= 0.f ? G_inter * ih : 0.f, G_ih = diy >= 0.f ? G_inter * iw : 0.f;
    const float G_ew = dex >= 0.f ? G_area * eh : 0.f, G_eh = dey >= 0.f ? G_area * ew : 0.f;
    float gx0 = -G_iw * gsel(x0, tx0, true) - G_ew * gsel(x0, tx0, false) - G_area1 * (y1 - y0);
    float gx1 = G_iw * gsel(x1, tx1, false) + G_ew * gsel(x1, tx1, true) + G_area1 * (y1 - y0);
    float gy0 = -G_ih * gsel(y0, ty0, true) - G_eh * gsel(y0, ty0, false) - G_area1 * (x1 - x0);
    float gy1 = G_ih * gsel(y1, ty1, false) + G_eh * gsel(y1, ty1, true) + G_area1 * (x1 - x0);
    gb[0] = gx0 + gx1;
    gb[1] = gy0 + gy1;
    gb[2] = 0.5f * (gx1 - gx0);
    gb[3] = 0.5f * (gy1 - gy0);
}

__global__ __launch_bounds__(256) void set_loss_kernel(const float* __restrict__ logits, const float* __restrict__ boxes,
                                                       const float* __restrict__ tgt, const int32_t* __restrict__ match,
                                                       float* __restrict__ losses, float* __restrict__ g_label,
                                                       float* __restrict__ g_bbox, float* __restrict__ g_giou, int rows,
                                                       float eos, const int32_t* __restrict__ vid_off, int rows_per_video,
                                                       const int32_t* __restrict__ status,
                                                       const int32_t* __restrict__ box_status, int problems_per_layer) {
    __shared__ double red[4];
    extern __shared__ int vmin[];  // per-video minimum matched target id (PerFrameMatcher re-basing quirk)
    const int layer = blockIdx.x;
    const int64_t base = (int64_t)layer * rows;
    const int nvid = vid_off ? rows / rows_per_video : 0;
    for (int b = threadIdx.x; b < nvid; b += 256) vmin[b] = 0x7fffffff;
    __syncthreads();
    double cnt = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) {
        const int m = match[base + r];
        cnt += m >= 0 ? 1.0 : 0.0;
        if (vid_off && m >= 0) atomicMin(&vmin[r / rows_per_video], m);
    }
    const double K = block_sum_d(cnt, red);
    const float invK = K > 0 ? (float)(1.0 / K) : 0.f;
    const float invR = 1.f / (float)rows;
    double s_nll = 0.0, s_l1 = 0.0, s_g = 0.0, s_ok = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) {
        const int64_t row = base + r;
        int m = match[row];
        // matcher.py:114-115 hands the criterion target ids re-based by the smallest MATCHED id of the
        // video; loss.py:87 then indexes the video's own box list with them.  Identical to the true
        // target unless the video's first box is unmatched (a frame with more boxes than queries).
        if (vid_off && m >= 0) { const int b = r / rows_per_video; m = vid_off[b] + (m - vmin[b]); }
        const float l0 = logits[row * 2], l1 = logits[row * 2 + 1];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        const float lse = mx + logf(e0 + e1);
        const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
        const int cls = m >= 0 ? 0 : 1;  // foreground = 0, background = 1 (loss.py:32-33)
        const float w = cls == 0 ? 1.f : eos;
        s_nll += (double)(-w * ((cls == 0 ? l0 : l1) - lse));
        g_label[row * 2] = w * (p0 - (cls == 0 ? 1.f : 0.f)) * invR;
        g_label[row * 2 + 1] = w * (p1 - (cls == 1 ? 1.f : 0.f)) * invR;
        float gb[4] = {0.f, 0.f, 0.f, 0.f}, gg[4] = {0.f, 0.f, 0.f, 0.f};
        if (m >= 0) {
            const float* b = boxes + row * 4;
            const float* t = tgt + (int64_t)m * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float d = b[c] - t[c];
                s_l1 += (double)fabsf(d);
                gb[c] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * invK * 0.25f;
            }
            float gi, dgi[4];
            giou_grad(b, t, gi, dgi);
            s_g += (double)(1.f - gi);
#pragma unroll
            for (int c = 0; c < 4; ++c) gg[c] = -dgi[c] * invK;
            s_ok += (l0 >= l1) ? 1.0 : 0.0;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { g_bbox[row * 4 + c] = gb[c]; g_giou[row * 4 + c] = gg[c]; }
    }
    const double t_nll = block_sum_d(s_nll, red);
    const double t_l1 = block_sum_d(s_l1, red);
    const double t_g = block_sum_d(s_g, red);
    const double t_ok = block_sum_d(s_ok, red);
    // a layer whose matching is not scipy's (NaN / -inf costs: scipy raises ValueError; infeasible) or whose boxes fail
    // generalized_box_iou's check (the reference raises AssertionError) must not look like a healthy step: its losses
    // become NaN, which every training loop's finite-loss check sees without a host sync here
    int flagged = 0;
    for (int q = threadIdx.x; q < problems_per_layer; q += 256) {
        const int p = layer * problems_per_layer + q;
        flagged |= (status && status[p] != 0) || (box_status && box_status[p] != 0);
    }
    flagged = __syncthreads_or(flagged);
    if (flagged) {   // ... and its unit gradients too, so that backward() / optimizer.step() cannot proceed on a non-scipy matching
        const float qn = __builtin_nanf("");
        for (int r = threadIdx.x; r < rows; r += 256) {
            const int64_t row = base + r;
            g_label[row * 2] = g_label[row * 2 + 1] = qn;
#pragma unroll
            for (int c = 0; c < 4; ++c) { g_bbox[row * 4 + c] = qn; g_giou[row * 4 + c] = qn; }
        }
    }
    if (threadIdx.x == 0 && flagged) {
        const float qnan = __builtin_nanf("");
        losses[layer * 4 + 0] = losses[layer * 4 + 1] = losses[layer * 4 + 2] = losses[layer * 4 + 3] = qnan;
    } else if (threadIdx.x == 0) {
        losses[layer * 4 + 0] = (float)(t_nll / (double)rows);
        losses[layer * 4 + 1] = K > 0 ? (float)(t_l1 / (4.0 * K)) : 0.f;
        losses[layer * 4 + 2] = K > 0 ? (float)(t_g / K) : 0.f;
        losses[layer * 4 + 3] = K > 0 ? (float)(100.0 - t_ok * (100.0 / K)) : 0.f;
    }
}

}  // namespace

extern "C" {

int svol_match_cost(const float* logits, const float* boxes, const float* tgt_boxes, const int32_t* pred_off,
                    const int32_t* pred_cnt, const int32_t* tgt_off, const int32_t* tgt_cnt, const int64_t* cost_off,
                    float* cost, int32_t n_problems, float w_bbox, float w_giou, float w_class, int32_t* box_status,
                    void* stream) {
    if (!logits || !boxes || !tgt_boxes || !pred_off || !pred_cnt || !tgt_off || !tgt_cnt || !cost_off || !cost)
        return SVOL_E_INVALID;
    if (n_problems < 0) return SVOL_E_INVALID;
    if (n_problems == 0) return SVOL_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(match_cost_kernel, dim3((unsigned)n_problems), dim3(256), 0, s, logits, boxes, tgt_boxes, pred_off,
                       pred_cnt, tgt_off, tgt_cnt, cost_off, cost, w_bbox, w_giou, w_class, box_status);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_lsap_batched(const float* cost, const int64_t* cost_off, const int32_t* pred_off, const int32_t* pred_cnt,
                      const int32_t* tgt_off, const int32_t* tgt_cnt, int32_t* match, int32_t* status, int32_t n_problems,
                      int32_t max_dim, void* stream) {
    if (!cost || !cost_off || !pred_off || !pred_cnt || !tgt_off || !tgt_cnt || !match || !status) return SVOL_E_INVALID;
    if (n_problems < 0 || max_dim < 0) return SVOL_E_INVALID;
    if (n_problems == 0) return SVOL_OK;
    if (max_dim > 2048) return SVOL_E_UNSUPPORTED;
    int md = ((max_dim + 15) / 16) * 16;
    if (md < 16) md = 16;
    const size_t base = (size_t)md * (3 * 8 + 4 * 4 + 2);  // multiple of 16 bytes (md % 16 == 0)
    size_t stage = (size_t)max_dim * max_dim * 4;             // room for a max_dim x max_dim block, capped by 64 KiB of LDS
    const size_t pad = 512;   // lsap_reg2 reads column lane + 64 of a cost row unconditionally (discarded when >= nc)
    if (base + stage + pad > 65536) stage = base + pad < 65536 ? (65536 - base - pad) / 16 * 16 : 0;
    const size_t lds = base + stage + pad;
    static const int v1 = getenv("SVOL_LSAP_V1") != nullptr;   // round 2's search step (A/B)
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(lsap_kernel, dim3((unsigned)n_problems), dim3(64), lds, s, cost, cost_off, pred_off, pred_cnt, tgt_off,
                       tgt_cnt, match, status, md, (int)(stage / 4), v1);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_set_loss(const float* logits, const float* boxes, const float* tgt_boxes, const int32_t* match, float* losses,
                  float* g_label, float* g_bbox, float* g_giou, int32_t n_layers, int32_t rows_per_layer, float eos_coef,
                  const int32_t* rebase_vid_off, int32_t rows_per_video, const int32_t* status, const int32_t* box_status,
                  int32_t problems_per_layer, void* stream) {
    if (!logits || !boxes || !tgt_boxes || !match || !losses || !g_label || !g_bbox || !g_giou) return SVOL_E_INVALID;
    if (n_layers <= 0 || rows_per_layer <= 0) return SVOL_E_INVALID;
    if ((status || box_status) && problems_per_layer <= 0) return SVOL_E_INVALID;
    size_t lds = 0;
    if (rebase_vid_off) {
        if (rows_per_video <= 0 || rows_per_layer % rows_per_video) return SVOL_E_INVALID;
        lds = sizeof(int) * (size_t)(rows_per_layer / rows_per_video);
        if (lds > 32768) return SVOL_E_UNSUPPORTED;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(set_loss_kernel, dim3((unsigned)n_layers), dim3(256), lds, s, logits, boxes, tgt_boxes, match, losses,
                       g_label, g_bbox, g_giou, (int)rows_per_layer, eos_coef, rebase_vid_off, (int)rows_per_video, status,
                       box_status, (int)problems_per_layer);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

}  // extern "C"
