// Sketch -> video attention gate (reference cross_modal_transformer.py:122-127), HBM-bound row kernels.
//
// The reference runs a full nn.MultiheadAttention with ONE query (the sketch token) over the L video
// tokens and keeps only the head-averaged attention weights att1[b,l]; mem = LN1(x + att1 * x).
// With a single query the key projection collapses: score[b,h,l] = (x+pos)[b,l,:] . u[b,h,:] with
// u[b,h,:] = d_h^-1/2 * W_k,h^T q[b,h,:] (the k-bias adds a per-(b,h) constant that cancels in the
// softmax), so the [L,d]x[d,d] K projection, the V projection, P.V and out_proj of the reference are
// never computed.  Forward = 3 passes over x (scores, per-(b,h) max/sum, apply+LN1); backward = 3.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int GP = 4;  // max passes: D <= 1024
constexpr int GH = 8;  // max heads

// 8 head sums over 64 lanes as ONE reduce-scatter butterfly: the xor-1 partners split the heads (each keeps four and adds the
// partner's four), the xor-2 partners split again (two each), then two values ride the remaining four steps — 14 exchanges instead
// of 8 x 6.  Lane q of quad 0 ends with heads {4*(q&1) + 2*(q>>1), +1} and stores them: scores[b, hh, l].
__device__ __forceinline__ void store_head_sums(const float (&part)[GH], float* __restrict__ scores, int b, int l, int L, int H, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    float h4[4], h2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float send = b0 ? part[i] : part[4 + i], keep = b0 ? part[4 + i] : part[i];
        h4[i] = keep + dpp_f32<0xB1>(send);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float send = b1 ? h4[i] : h4[2 + i], keep = b1 ? h4[2 + i] : h4[i];
        h2[i] = keep + dpp_f32<0x4E>(send);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float v = h2[i];
        // the lanes of a quad hold DIFFERENT heads now: only lanes with the same position in their quad may be added — rotations of
        // the 16-lane row by 4 and by 8 lanes.  (Until round 6 these two steps were wave_sum's row_half_mirror / row_mirror, which
        // pair lane q with lane 3 - q of the neighbouring quad: every score was a mixture of heads.  LN1(x (1 + a)) is invariant to
        // the per-token scale up to its epsilon, so no output, loss or parameter gradient moved beyond 1e-6 and nothing noticed;
        // tests/gpu_checks.py::check_gate now compares the scores and the gate weights themselves.)
        v += dpp_f32<0x124>(v);   // row_ror:4
        v += dpp_f32<0x128>(v);   // row_ror:8
        const HalfPair r16 = swap_rows16(__builtin_bit_cast(unsigned, v));
        v = __builtin_bit_cast(float, r16.lo) + __builtin_bit_cast(float, r16.hi);
        const HalfPair r32 = swap_halves(__builtin_bit_cast(unsigned, v));
        h2[i] = __builtin_bit_cast(float, r32.lo) + __builtin_bit_cast(float, r32.hi);
    }
    if (lane < 4) {
        const int hh0 = 4 * (lane & 1) + 2 * (lane >> 1);
        if (hh0 < H) scores[((int64_t)b * H + hh0) * L + l] = h2[0];
        if (hh0 + 1 < H) scores[((int64_t)b * H + hh0 + 1) * L + l] = h2[1];
    }
}

// scores[b,hh,l] = (x+pos)[b,l,:] . u[b,hh,:]      grid = (ceil(L / (4*rpw)), B), one wave per row
template <typename T, int NP>
__global__ __launch_bounds__(256) void gate_scores_kernel(const float* __restrict__ x, const T* __restrict__ pos,
                                                          const float* __restrict__ u, float* __restrict__ scores, int L,
                                                          int D, int H, int rpw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int l0 = (blockIdx.x * 4 + wave) * rpw;
    float ur[GH][NP][4];
#pragma unroll
    for (int hh = 0; hh < GH; ++hh)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = (lane + 64 * j) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) ur[hh][j][e] = (hh < H && c < D) ? u[((int64_t)b * H + hh) * D + c + e] : 0.f;
        }
    const int lend = min(l0 + rpw, L);
    Vec4<float> xv[NP], nxv[NP];
    Vec4<T> pv[NP], npv[NP];
    auto fetch = [&](int l, Vec4<float> (&xr)[NP], Vec4<T> (&pr)[NP]) {
        const int64_t row = (int64_t)b * L + l;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (c < D) {
                xr[j].load(x + row * D + c);
                pr[j].load(pos + row * D + c);
            }
        }
    };
    if (l0 < lend) fetch(l0, xv, pv);
    for (int l = l0; l < lend; ++l) {
        if (l + 1 < lend) fetch(l + 1, nxv, npv);  // next row in flight while this one is reduced
        float part[GH];
#pragma unroll
        for (int hh = 0; hh < GH; ++hh) part[hh] = 0.f;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (c < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = xv[j].get(e) + pv[j].get(e);
#pragma unroll
                    for (int hh = 0; hh < GH; ++hh) part[hh] += t * ur[hh][j][e];
                }
            }
        }
        store_head_sums(part, scores, b, l, L, H, lane);
#pragma unroll
        for (int j = 0; j < NP; ++j) { xv[j] = nxv[j]; pv[j] = npv[j]; }
    }
}

// per (b,hh): mx = max_l score, sm = sum_l exp(score - mx);  optionally also
// c = sum_l p[l] * da[b,l] / H (backward).   grid = B*H blocks of 256
__global__ __launch_bounds__(256) void gate_stats_kernel(const float* __restrict__ scores, float* __restrict__ mx_out,
                                                         float* __restrict__ sm_out, int L) {
    __shared__ float red[4];
    const int bh = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* s = scores + (int64_t)bh * L;
    float m = -INFINITY;
    for (int l = tid; l < L; l += 256) m = fmaxf(m, s[l]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float acc = 0.f;
    for (int l = tid; l < L; l += 256) acc += __expf(s[l] - m);
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        mx_out[bh] = m;
        sm_out[bh] = red[0] + red[1] + red[2] + red[3];
    }
}

// a[b,l] = mean_h softmax ; s1 = x*(1+a) ; y = LN(s1) ; ypos = y + pos        one wave per row
template <typename T, int NP>
__global__ __launch_bounds__(256) void gate_apply_kernel(const float* __restrict__ x, const T* __restrict__ pos,
                                                         const float* __restrict__ scores, const float* __restrict__ mx,
                                                         const float* __restrict__ sm, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y32,
                                                         T* __restrict__ y,
                                                         T* __restrict__ ypos, float* __restrict__ a_out,
                                                         float* __restrict__ mean, float* __restrict__ rstd, int L, int D,
                                                         int H, int64_t M) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int b = (int)(row / L), l = (int)(row % L);
    // every global load of the wave's one row goes out before anything is reduced (scores, x, pos, gamma, beta): one memory
    // latency per wave instead of three (41 -> 3x us per [50176, 256] launch)
    Vec4<float> t[NP], gv[NP], bv[NP];
    Vec4<T> pv[NP];
    float sc[GH], mxv[GH], smv[GH];
#pragma unroll
    for (int hh = 0; hh < GH; ++hh) {
        const int bh = b * H + (hh < H ? hh : 0);
        sc[hh] = scores[(int64_t)bh * L + l];
        mxv[hh] = mx[bh];
        smv[hh] = sm[bh];
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < D) {
            t[j].load(x + row * D + c);
            pv[j].load(pos + row * D + c);
            gv[j].load(gamma + c);
            bv[j].load(beta + c);
        }
    }
    float a = 0.f;
#pragma unroll
    for (int hh = 0; hh < GH; ++hh)
        if (hh < H) a += __expf(sc[hh] - mxv[hh]) / smv[hh];
    a /= (float)H;
    float v[NP][4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int c = (lane + 64 * j) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[j][e] = c < D ? t[j].get(e) * (1.f + a) : 0.f;
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
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; a_out[row] = a; }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < D) {
            Vec4<T> o, op;
            Vec4<float> o32;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float r = (v[j][e] - mu) * rs * gv[j].get(e) + bv[j].get(e);
                o.set(e, r);
                o32.set(e, r);
                op.set(e, r + pv[j].get(e));
            }
            if (y32) o32.store(y32 + row * D + c);
            if (y) o.store(y + row * D + c);
            if (ypos) op.store(ypos + row * D + c);
        }
    }
}

// LN3 of layer i with the gate scores of layer i + 1 as its epilogue: the row this wave has just normalised IS
// the next layer's x, pos is already in registers for the + pos output, and every layer's gate vectors exist before the layer loop
// (cross_modal_transformer.py: all_gate_vectors) — so the next layer's score pass (one more read of x32 + pos: 77 MB, 33 us at cfg2)
// is 32 FMAs per lane here.  One row per wave as in ln_fwd_kernel (norm.hip: same arithmetic, same order -> the same bits), the four
// rows of a workgroup belong to one batch element (launcher: L % 4 == 0) whose u[b] is staged in LDS once per workgroup; the
// products and the head butterfly are gate_scores_kernel's, so the scores carry the same bits as the stand-alone pass too.
template <typename T, int NP>
__global__ __launch_bounds__(256) void ln_scores_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y32, T* __restrict__ y,
                                                        T* __restrict__ ypos, const T* __restrict__ pos, float* __restrict__ mean,
                                                        float* __restrict__ rstd, const float* __restrict__ u,
                                                        float* __restrict__ scores, int L, int D, int H) {
    extern __shared__ __attribute__((aligned(16))) float su[];   // u[b]: [GH][D], rows >= H zero (no per-head branch below)
    __builtin_assume(D % 4 == 0);   // (launcher) -> the rows of su are 16-byte aligned: ds_read_b128
    const int tid = threadIdx.x, lane = tid & 63;
    const int b = blockIdx.y, l = blockIdx.x * 4 + (tid >> 6);   // grid = (L / 4, B)
    const int64_t row = (int64_t)b * L + l;
    {
        const int nu = H * D;
        const float* ub = u + (int64_t)b * nu;
        for (int i = tid * 4; i < GH * D; i += 1024) {
            f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < nu) w = *reinterpret_cast<const f32x4*>(ub + i);
            *reinterpret_cast<f32x4*>(su + i) = w;
        }
    }
    const float* xr = x + row * D;
    const T* pr = pos + row * D;
    Vec4<float> t[NP], gv[NP], bv[NP];
    Vec4<T> pv[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < D) {
            t[j].load(xr + c);
            pv[j].load(pr + c);
            gv[j].load(gamma + c);
            bv[j].load(beta + c);
        }
    }
    float v[NP][4];
    float s = 0.f;
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
    float tp[NP][4];   // (y + pos) in fp32: what gate_scores_kernel forms from the stored fp32 row and pos
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int c = (lane + 64 * j) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) tp[j][e] = 0.f;
        if (c < D) {
            Vec4<T> o, op;
            Vec4<float> o32;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float r = (v[j][e] - mu) * rs * gv[j].get(e) + bv[j].get(e);
                o.set(e, r);
                o32.set(e, r);
                op.set(e, r + pv[j].get(e));
                tp[j][e] = r + pv[j].get(e);
            }
            if (y32) o32.store(y32 + row * D + c);
            if (y) o.store(y + row * D + c);
            if (ypos) op.store(ypos + row * D + c);
        }
    }
    __syncthreads();
    float part[GH];
#pragma unroll
    for (int hh = 0; hh < GH; ++hh) part[hh] = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int c = (lane + 64 * j) * 4;
        if (c < D) {
            f32x4 ur[GH];
#pragma unroll
            for (int hh = 0; hh < GH; ++hh) ur[hh] = *reinterpret_cast<const f32x4*>(su + hh * D + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int hh = 0; hh < GH; ++hh) part[hh] += tp[j][e] * ur[hh][e];
            }
        }
    }
    store_head_sums(part, scores, b, l, L, H, lane);
}

// backward pass 1: LN1 backward, dx_part = ds1*(1+a), da[row] = sum_d ds1*x ; dgamma/dbeta atomics.
// NP = 16-byte column groups per lane (D <= 256 * NP): the row loop is latency-bound (one dependent load -> reduce -> store chain
// per wave), so the NEXT row's four operand vectors are fetched before the current row is reduced, and the register arrays are sized
// for the real width (the untemplated version carried four groups whatever D was: twice the registers, half the waves per SIMD).
template <typename T, int NP>
__global__ __launch_bounds__(256) void gate_bwd_ln_kernel(const float* __restrict__ dy32, const T* __restrict__ dy,
                                                          const T* __restrict__ dy2, const float* __restrict__ x,
                                                          const float* __restrict__ a_in,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, float* __restrict__ dx,
                                                          float* __restrict__ da, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, const float* __restrict__ scores,
                                                          const float* __restrict__ mx, const float* __restrict__ sm,
                                                          float* __restrict__ cc, int L, int H, int64_t M, int D, int rpw,
                                                          float* __restrict__ det_cc, float* __restrict__ det_gb) {
    // det_cc / det_gb: deterministic mode (common.h): this WAVE's row of [B*H] partial c sums (zero-filled scratch) and this
    // WORKGROUP's row of [2*D] dgamma | dbeta partials, folded in index order by the launcher instead of the atomics
    const int lane = threadIdx.x & 63;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw;
    const int64_t rend = (r0 + rpw < M) ? r0 + rpw : M;
    float dg[NP][4], db[NP][4], gm[NP][4];
    // c[b,h] = sum_l p[b,h,l] * da[b,l] / H (what the softmax backward of the next pass needs) is folded in here: lane h < H keeps
    // head h's partial sum over this wave's rows and adds it to cc[b,h] (zeroed by the launcher) when the batch element changes and at
    // the end — the separate B*H-block statistics kernel between the two row passes cost 24 us per layer inside the step
    int bcur = r0 < rend ? (int)(r0 / L) : 0, lcur = r0 < rend ? (int)(r0 % L) : 0;
    float cacc = 0.f, mxl = 0.f, isl = 0.f;
    if (lane < H && r0 < rend) { mxl = mx[bcur * H + lane]; isl = 1.f / sm[bcur * H + lane]; }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int c = (lane + 64 * j) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { dg[j][e] = 0.f; db[j][e] = 0.f; gm[j][e] = c < D ? gamma[c + e] : 0.f; }
    }
    struct Row { Vec4<float> p32[NP], xv[NP]; Vec4<T> p[NP], p2[NP]; float mu, rs, a, sc; };
    auto fetch = [&](int64_t row, int b, int l, Row& r) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (c < D) {
                if (dy32) r.p32[j].load(dy32 + row * D + c);
                if (dy) r.p[j].load(dy + row * D + c);
                if (dy2) r.p2[j].load(dy2 + row * D + c);
                r.xv[j].load(x + row * D + c);
            }
        }
        r.mu = mean[row]; r.rs = rstd[row]; r.a = a_in[row];
        r.sc = lane < H ? scores[((int64_t)b * H + lane) * L + l] : 0.f;
    };
    Row cur, nxt;
    if (r0 < rend) fetch(r0, bcur, lcur, cur);
    for (int64_t row = r0; row < rend; ++row) {
        int bn = bcur, ln = lcur + 1;
        if (ln == L) { ln = 0; ++bn; }
        if (row + 1 < rend) fetch(row + 1, bn, ln, nxt);
        const float mu = cur.mu, rs = cur.rs, a = cur.a;
        float g[NP][4], xh[NP][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (c < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = (dy32 ? cur.p32[j].get(e) : 0.f) + (dy ? cur.p[j].get(e) : 0.f) + (dy2 ? cur.p2[j].get(e) : 0.f);
                    const float hv = (cur.xv[j].get(e) * (1.f + a) - mu) * rs;
                    xh[j][e] = hv;
                    dg[j][e] += d * hv;
                    db[j][e] += d;
                    const float gg = d * gm[j][e];
                    g[j][e] = gg;
                    s1 += gg;
                    s2 += gg * hv;
                }
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
        float dacc = 0.f;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (c < D) {
                Vec4<float> o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ds = rs * (g[j][e] - s1 - xh[j][e] * s2);
                    dacc += ds * cur.xv[j].get(e);
                    o.set(e, ds * (1.f + a));
                }
                o.store(dx + row * D + c);
            }
        }
        dacc = wave_sum(dacc);
        if (lane == 0) da[row] = dacc;
        cacc += __expf(cur.sc - mxl) * isl * dacc;
        if (bn != bcur || row + 1 == rend) {   // wave-uniform: leaving this batch element (or done)
            if (lane < H) {
                if (det_cc) det_cc[(((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (M / L) + bcur) * H + lane] = cacc / (float)H;
                else atomicAdd(cc + bcur * H + lane, cacc / (float)H);
                cacc = 0.f;
                if (row + 1 < rend) { mxl = mx[bn * H + lane]; isl = 1.f / sm[bn * H + lane]; }
            }
        }
        bcur = bn;
        lcur = ln;
        cur = nxt;
    }
    __shared__ float red[2][NP * 256];
    const int wave = threadIdx.x >> 6;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int c = (lane + 64 * j) * 4;
                if (c < D) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (w == 0) { red[0][c + e] = dg[j][e]; red[1][c + e] = db[j][e]; }
                        else { red[0][c + e] += dg[j][e]; red[1][c + e] += db[j][e]; }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (det_gb) {
        float* row = det_gb + (int64_t)blockIdx.x * 2 * D;
        for (int c = threadIdx.x; c < D; c += 256) { row[c] = red[0][c]; row[D + c] = red[1][c]; }
        return;
    }
    for (int c = threadIdx.x; c < D; c += 256) {
        atomicAdd(dgamma + c, red[0][c]);
        atomicAdd(dbeta + c, red[1][c]);
    }
}

// backward pass 3: dscore[hh] = p_hh[l] * (da[l]/H - c[b,hh]) ; dx += sum_hh dscore*u ; du += dscore*(x+pos)
template <typename T, int NP>
__global__ __launch_bounds__(256) void gate_bwd_apply_kernel(const float* __restrict__ x, const T* __restrict__ pos,
                                                             const float* __restrict__ u, const float* __restrict__ scores,
                                                             const float* __restrict__ mx, const float* __restrict__ sm,
                                                             const float* __restrict__ da, const float* __restrict__ cc,
                                                             float* __restrict__ dx, float* __restrict__ du, int L, int D,
                                                             int H, int rpw, float* __restrict__ det_du) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int l0 = (blockIdx.x * 4 + wave) * rpw;
    float ur[GH][NP][4], dur[GH][NP][4];
#pragma unroll
    for (int hh = 0; hh < GH; ++hh)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = (lane + 64 * j) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ur[hh][j][e] = (hh < H && c < D) ? u[((int64_t)b * H + hh) * D + c + e] : 0.f;
                dur[hh][j][e] = 0.f;
            }
        }
    // The row loop is latency-bound (one dependent load -> compute -> store chain per wave, ~1.5 KB in flight): the NEXT row's
    // operands are fetched before the current row is processed (rows are distinct, so hoisting the dx load over the dx store
    // is safe — the compiler cannot know that)
    const int lend = min(l0 + rpw, L);
    Vec4<float> xv[NP], dv[NP], nxv[NP], ndv[NP];
    Vec4<T> pv[NP], npv[NP];
    // per-head constants of this batch element, and the per-row scalars (8 head scores + da): the row scalars are nine more
    // dependent loads per row — fetched one row ahead with the vectors, or every row pays their full latency
    float mxr[GH], isr[GH], ccr[GH], sc[GH], nsc[GH], dar = 0.f, ndar = 0.f;
#pragma unroll
    for (int hh = 0; hh < GH; ++hh) {
        mxr[hh] = hh < H ? mx[b * H + hh] : 0.f;
        isr[hh] = hh < H ? 1.f / sm[b * H + hh] : 0.f;
        ccr[hh] = hh < H ? cc[b * H + hh] : 0.f;
        sc[hh] = nsc[hh] = 0.f;
    }
    auto fetch = [&](int l, Vec4<float> (&xr)[NP], Vec4<T> (&pr)[NP], Vec4<float> (&dr)[NP], float (&sr)[GH], float& dr1) {
        const int64_t row = (int64_t)b * L + l;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (c < D) {
                xr[j].load(x + row * D + c);
                pr[j].load(pos + row * D + c);
                dr[j].load(dx + row * D + c);
            }
        }
#pragma unroll
        for (int hh = 0; hh < GH; ++hh)
            if (hh < H) sr[hh] = scores[((int64_t)b * H + hh) * L + l];
        dr1 = da[row];
    };
    if (l0 < lend) fetch(l0, xv, pv, dv, sc, dar);
    for (int l = l0; l < lend; ++l) {
        const int64_t row = (int64_t)b * L + l;
        if (l + 1 < lend) fetch(l + 1, nxv, npv, ndv, nsc, ndar);
        float ds[GH];
        const float dal = dar / (float)H;
#pragma unroll
        for (int hh = 0; hh < GH; ++hh) {
            if (hh < H) {
                const float pp = __expf(sc[hh] - mxr[hh]) * isr[hh];
                ds[hh] = pp * (dal - ccr[hh]);
            } else ds[hh] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (c < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = xv[j].get(e) + pv[j].get(e);
                    float add = 0.f;
#pragma unroll
                    for (int hh = 0; hh < GH; ++hh) {
                        add += ds[hh] * ur[hh][j][e];
                        dur[hh][j][e] += ds[hh] * t;
                    }
                    dv[j].set(e, dv[j].get(e) + add);
                }
                dv[j].store(dx + row * D + c);
            }
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) { xv[j] = nxv[j]; pv[j] = npv[j]; dv[j] = ndv[j]; }
#pragma unroll
        for (int hh = 0; hh < GH; ++hh) sc[hh] = nsc[hh];
        dar = ndar;
    }
    // du: fold the workgroup's four waves in LDS first (every wave of a batch element hits the same H*D addresses)
    __shared__ float red[GH * NP * 256];
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int hh = 0; hh < GH; ++hh)
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const int c = (lane + 64 * j) * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float* r = red + hh * NP * 256 + c + e;
                        if (w == 0) *r = dur[hh][j][e];
                        else *r += dur[hh][j][e];
                    }
                }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < GH * NP * 256; i += 256) {
        const int hh = i / (NP * 256), c = i % (NP * 256);
        if (hh < H && c < D) {
            // deterministic mode: the workgroup's row [H*D] of batch element b's block of the scratch (folded by the launcher)
            if (det_du) det_du[(((int64_t)b * gridDim.x + blockIdx.x) * H + hh) * D + c] = red[i];
            else atomicAdd(du + ((int64_t)b * H + hh) * D + c, red[i]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Gate VECTORS: u[b,h,:] = d_h^-1/2 * W_k,h^T (W_q,h s_b + b_q,h) — the one-query attention's key projection folded into one
// d-vector per (batch, head) (B*d-sized algebra on the packed in_proj parameters; cross_modal_transformer.py:122-123 through
// nn.MultiheadAttention's in_proj).  One launch forward, two backward, instead of a dozen BLAS / elementwise launches.
// forward: block = one (head, batch element), 256 threads.  Phase 1: the head's d_h rows of W_q, one wave per row (lanes split the
// d columns, 16-byte loads, wave reduction) -> q[b, h*dh .. ]; phase 2: thread = column c, u[b,h,c] = sc * sum_e q[e] W_k[h*dh+e][c]
// (row reads coalesced over the threads).
__device__ __forceinline__ void gate_vec_fwd_body(const float* __restrict__ skch, const float* __restrict__ W,
                                                  const float* __restrict__ bias, float* __restrict__ q_out,
                                                  float* __restrict__ u_out, int D, int H) {
    __shared__ float s_q[GP * 256];
    const int hh = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, dh = D / H;
    float sv[GP][4];
#pragma unroll
    for (int j = 0; j < GP; ++j) {
        const int c = (lane + 64 * j) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) sv[j][e] = c < D ? skch[(int64_t)b * D + c + e] : 0.f;
    }
    for (int e = wave; e < dh; e += 4) {
        const int row = hh * dh + e;
        const float* w = W + (int64_t)row * D;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < GP; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (c < D) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c);
                acc += wv[0] * sv[j][0] + wv[1] * sv[j][1] + wv[2] * sv[j][2] + wv[3] * sv[j][3];
            }
        }
        acc = wave_sum(acc) + bias[row];
        if (lane == 0) { s_q[e] = acc; q_out[(int64_t)b * D + row] = acc; }
    }
    __syncthreads();
    const float sc = rsqrtf((float)dh);
    const float* Wk = W + (int64_t)D * D + (int64_t)hh * dh * D;
    for (int c = tid; c < D; c += 256) {
        float acc = 0.f;
        for (int e = 0; e < dh; ++e) acc += s_q[e] * Wk[(int64_t)e * D + c];
        u_out[((int64_t)b * H + hh) * D + c] = acc * sc;
    }
}
__global__ __launch_bounds__(256) void gate_vec_fwd_kernel(const float* skch, const float* W, const float* bias, float* q_out,
                                                           float* u_out, int D, int H) {
    gate_vec_fwd_body(skch, W, bias, q_out, u_out, D, H);
}
// backward A: block = one (head, batch element): dq[b, h*dh+e] = sc * du[b,h,:] . W_k[h*dh+e,:], one wave per row
__device__ __forceinline__ void gate_vec_bwd_a_body(const float* __restrict__ du, const float* __restrict__ W,
                                                    float* __restrict__ dq_out, int D, int H) {
    const int hh = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, dh = D / H;
    const float sc = rsqrtf((float)dh);
    float gv[GP][4];
#pragma unroll
    for (int j = 0; j < GP; ++j) {
        const int c = (lane + 64 * j) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[j][e] = c < D ? du[((int64_t)b * H + hh) * D + c + e] : 0.f;
    }
    const float* Wk = W + (int64_t)D * D;
    for (int e = wave; e < dh; e += 4) {
        const int row = hh * dh + e;
        const float* w = Wk + (int64_t)row * D;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < GP; ++j) {
            const int c = (lane + 64 * j) * 4;
            if (c < D) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c);
                acc += wv[0] * gv[j][0] + wv[1] * gv[j][1] + wv[2] * gv[j][2] + wv[3] * gv[j][3];
            }
        }
        acc = wave_sum(acc) * sc;
        if (lane == 0) dq_out[(int64_t)b * D + row] = acc;
    }
}
// backward B: blocks 0 .. D-1 = one row e of W_q and of W_k: dW_q[e][c] += sum_b dq[b][e] s[b][c] ; dW_k[e][c] += sc * sum_b q[b][e]
// du[b][h(e)][c] ; db_q[e] += sum_b dq[b][e] (every output element has one owner: plain read-modify-write into the gradient, sink
// or zeroed buffer); blocks D .. D+B-1 = one batch element: dskch[b][c] = sum_e dq[b][e] W_q[e][c] (row reads coalesced over c).
__global__ __launch_bounds__(256) void gate_vec_bwd_a_kernel(const float* du, const float* W, float* dq_out, int D, int H) {
    gate_vec_bwd_a_body(du, W, dq_out, D, H);
}
__device__ __forceinline__ void gate_vec_bwd_b_body(const float* __restrict__ du, const float* __restrict__ skch,
                                                    const float* __restrict__ q, const float* __restrict__ dq,
                                                    const float* __restrict__ W, float* __restrict__ dW,
                                                    float* __restrict__ db, float* __restrict__ dskch, int B, int D, int H) {
    const int tid = threadIdx.x, dh = D / H;
    if ((int)blockIdx.x >= D) {
        const int b = blockIdx.x - D;
        if (!dskch) return;
        // 4 waves x 64 lanes: a lane owns 4 consecutive columns (one 16-byte load per row of W), a wave every 4th row; the
        // four partial sums meet in LDS (256 dependent row reads per thread in the one-column-per-thread form: 40 us)
        __shared__ float s_dq[GP * 256];
        __shared__ float4 s_red[4][GP * 64];
        for (int e = tid; e < D; e += 256) s_dq[e] = dq[(int64_t)b * D + e];
        __syncthreads();
        const int w = tid >> 6, l = tid & 63;
        for (int c = 4 * l, k = l; c < D; c += 256, k += 64) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
            for (int e = w; e < D; e += 4) {
                const float4 wv = *reinterpret_cast<const float4*>(W + (int64_t)e * D + c);
                const float g = s_dq[e];
                acc.x += g * wv.x; acc.y += g * wv.y; acc.z += g * wv.z; acc.w += g * wv.w;
            }
            s_red[w][k] = acc;
        }
        __syncthreads();
        for (int c = tid; c < D; c += 256) {
            const float* r = reinterpret_cast<const float*>(&s_red[0][0]);
            const int stride = GP * 64 * 4;
            dskch[(int64_t)b * D + c] = (r[c] + r[stride + c]) + (r[2 * stride + c] + r[3 * stride + c]);
        }
        return;
    }
    const int e = blockIdx.x, hh = e / dh;
    const float sc = rsqrtf((float)dh);
    for (int c = tid; c < D; c += 256) {
        float aq = 0.f, ak = 0.f;
        for (int b = 0; b < B; ++b) {
            aq += dq[(int64_t)b * D + e] * skch[(int64_t)b * D + c];
            ak += q[(int64_t)b * D + e] * du[((int64_t)b * H + hh) * D + c];
        }
        dW[(int64_t)e * D + c] += aq;
        dW[(int64_t)(D + e) * D + c] += ak * sc;
    }
    if (tid == 0) {
        float bsum = 0.f;
        for (int b = 0; b < B; ++b) bsum += dq[(int64_t)b * D + e];
        db[e] += bsum;
    }
}
__global__ __launch_bounds__(256) void gate_vec_bwd_b_kernel(const float* du, const float* skch, const float* q, const float* dq,
                                                             const float* W, float* dW, float* db, float* dskch, int B, int D, int H) {
    gate_vec_bwd_b_body(du, skch, q, dq, W, dW, db, dskch, B, D, H);
}

// All layers' gate vectors in one launch each way (blockIdx.z = layer).  Round 4: the six per-layer backward pairs (13 + 50 us of
// latency-bound B*d-sized algebra each) ran one after the other at the very END of backward — their autograd nodes are the oldest —
// with nothing else left to overlap: 0.38 ms of the step.  The layers are independent; side by side they take one pair's time.
struct GateVecMulti {
    const float* du[SVOL_GATE_VEC_MAX_LAYERS];
    const float* W[SVOL_GATE_VEC_MAX_LAYERS];
    const float* bias[SVOL_GATE_VEC_MAX_LAYERS];
    float* q[SVOL_GATE_VEC_MAX_LAYERS];
    float* u[SVOL_GATE_VEC_MAX_LAYERS];
    float* dq[SVOL_GATE_VEC_MAX_LAYERS];
    float* dW[SVOL_GATE_VEC_MAX_LAYERS];
    float* db[SVOL_GATE_VEC_MAX_LAYERS];
    float* dskch[SVOL_GATE_VEC_MAX_LAYERS];
};
__global__ __launch_bounds__(256) void gate_vec_fwd_multi_kernel(const float* skch, GateVecMulti g, int D, int H) {
    const int l = blockIdx.z;
    gate_vec_fwd_body(skch, g.W[l], g.bias[l], g.q[l], g.u[l], D, H);
}
__global__ __launch_bounds__(256) void gate_vec_bwd_a_multi_kernel(GateVecMulti g, int D, int H) {
    const int l = blockIdx.z;
    gate_vec_bwd_a_body(g.du[l], g.W[l], g.dq[l], D, H);
}
__global__ __launch_bounds__(256) void gate_vec_bwd_b_multi_kernel(const float* skch, GateVecMulti g, int B, int D, int H) {
    const int l = blockIdx.z;
    gate_vec_bwd_b_body(g.du[l], skch, g.q[l], g.dq[l], g.W[l], g.dW[l], g.db[l], g.dskch[l], B, D, H);
}

int rows_per_wave(int64_t rows, int64_t target_waves) {
    int64_t r = (rows + target_waves - 1) / target_waves;
    return (int)(r < 1 ? 1 : r);
}
// total waves of the row-loop kernels: enough resident waves per SIMD to cover HBM latency (each wave walks its rows
// one after the other), few enough that the per-wave u registers / LDS reductions stay amortised
int64_t gate_waves(int which) {
    static const char* names[3] = {"SVOL_GATE_WAVES_SC", "SVOL_GATE_WAVES_LN", "SVOL_GATE_WAVES_AP"};
    static int64_t w[3] = {0, 0, 0};
    if (!w[which]) w[which] = getenv(names[which]) ? atoll(getenv(names[which])) : 4096;
    return w[which] < 64 ? 64 : w[which];
}

}  // namespace

extern "C" {

static int gate_fwd_impl(const float* x32, const void* pos, const float* u, const float* gamma, const float* beta, float* y32,
                         void* y, void* ypos, float* a, float* mean, float* rstd, float* ws, int64_t B, int64_t L, int64_t D, int64_t H,
                         int dtype, void* stream, bool scored) {
    if (!x32 || !pos || !u || !gamma || !beta || (!y && !y32) || !a || !mean || !rstd || !ws) return SVOL_E_INVALID;
    if (!aligned16(gamma) || !aligned16(beta)) return SVOL_E_INVALID;   // (16-byte vector loads of the affine parameters)
    if (B <= 0 || L <= 0 || D <= 0 || H <= 0) return SVOL_E_INVALID;
    if (D % 4 || D > GP * 256 || H > GH || B > 65535 || L > (1 << 24)) return SVOL_E_UNSUPPORTED;
    if (!svol_is16(dtype) && dtype != SVOL_F32) return SVOL_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int np_ = D <= 256 ? 1 : (D <= 512 ? 2 : 4);
    float* scores = ws;
    float* mx = ws + B * H * L;
    float* sm = mx + B * H;
    const int rpw = rows_per_wave(L, gate_waves(0) / B + 1);
    dim3 g1((unsigned)((L + 4 * rpw - 1) / (4 * rpw)), (unsigned)B);
    const int64_t M = B * L;
    const unsigned g3 = (unsigned)((M + 3) / 4);
#define SVOL_GATE_FWD(TT)                                                                                                   \
    do {                                                                                                                    \
        if (scored) { }                                                                                                      \
        else if (np_ == 1) hipLaunchKernelGGL((gate_scores_kernel<TT, 1>), g1, dim3(256), 0, s, x32, (const TT*)pos, u, scores, (int)L, (int)D, (int)H, rpw); \
        else if (np_ == 2) hipLaunchKernelGGL((gate_scores_kernel<TT, 2>), g1, dim3(256), 0, s, x32, (const TT*)pos, u, scores, (int)L, (int)D, (int)H, rpw); \
        else hipLaunchKernelGGL((gate_scores_kernel<TT, 4>), g1, dim3(256), 0, s, x32, (const TT*)pos, u, scores, (int)L, (int)D, (int)H, rpw); \
        hipLaunchKernelGGL(gate_stats_kernel, dim3((unsigned)(B * H)), dim3(256), 0, s, scores, mx, sm, (int)L);            \
        if (np_ == 1) hipLaunchKernelGGL((gate_apply_kernel<TT, 1>), dim3(g3), dim3(256), 0, s, x32, (const TT*)pos, scores, mx, sm, gamma, beta, y32, (TT*)y, (TT*)ypos, a, mean, rstd, (int)L, (int)D, (int)H, M); \
        else if (np_ == 2) hipLaunchKernelGGL((gate_apply_kernel<TT, 2>), dim3(g3), dim3(256), 0, s, x32, (const TT*)pos, scores, mx, sm, gamma, beta, y32, (TT*)y, (TT*)ypos, a, mean, rstd, (int)L, (int)D, (int)H, M); \
        else hipLaunchKernelGGL((gate_apply_kernel<TT, 4>), dim3(g3), dim3(256), 0, s, x32, (const TT*)pos, scores, mx, sm, gamma, beta, y32, (TT*)y, (TT*)ypos, a, mean, rstd, (int)L, (int)D, (int)H, M); \
    } while (0)
    if (dtype == SVOL_BF16) SVOL_GATE_FWD(bf16_t);
    else if (dtype == SVOL_F16) SVOL_GATE_FWD(f16_t);
    else SVOL_GATE_FWD(float);
#undef SVOL_GATE_FWD
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_gate_fwd(const float* x32, const void* pos, const float* u, const float* gamma, const float* beta, float* y32,
                  void* y, void* ypos, float* a, float* mean, float* rstd, float* ws, int64_t B, int64_t L, int64_t D, int64_t H,
                  int dtype, void* stream) {
    return gate_fwd_impl(x32, pos, u, gamma, beta, y32, y, ypos, a, mean, rstd, ws, B, L, D, H, dtype, stream, false);
}

// ws already holds scores[b,h,l] (svol_layernorm_gate_scores_fwd of the layer before): statistics + apply only
int svol_gate_fwd_scored(const float* x32, const void* pos, const float* u, const float* gamma, const float* beta, float* y32,
                         void* y, void* ypos, float* a, float* mean, float* rstd, float* ws, int64_t B, int64_t L, int64_t D,
                         int64_t H, int dtype, void* stream) {
    return gate_fwd_impl(x32, pos, u, gamma, beta, y32, y, ypos, a, mean, rstd, ws, B, L, D, H, dtype, stream, true);
}

int svol_layernorm_gate_scores_fwd(const float* x32, const float* gamma, const float* beta, float* y32, void* y, void* ypos,
                                   const void* pos, float* mean, float* rstd, const float* u_next, float* scores_next, int64_t B,
                                   int64_t L, int64_t D, int64_t H, int dtype, void* stream) {
    if (!x32 || !gamma || !beta || (!y && !y32 && !ypos) || !pos || !mean || !rstd || !u_next || !scores_next) return SVOL_E_INVALID;
    if (!aligned16(gamma) || !aligned16(beta) || !aligned16(u_next)) return SVOL_E_INVALID;
    if (B <= 0 || L <= 0 || D <= 0 || H <= 0) return SVOL_E_INVALID;
    // (L % 4: the four rows of a workgroup share u[b]; D <= 256: one 16-byte column group per lane — the width at which this kernel's
    // LayerNorm and svol_layernorm_fwd's compile to the same arithmetic, checked bit for bit by tests/gpu_checks.py)
    if (D % 4 || D > 256 || H > GH || L % 4 || L > (1 << 24) || B > 65535) return SVOL_E_UNSUPPORTED;
    if (!svol_is16(dtype) && dtype != SVOL_F32) return SVOL_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(L / 4), (unsigned)B);
    const size_t lds = (size_t)GH * D * sizeof(float);
#define SVOL_LNS(TT)                                                                                                          \
    hipLaunchKernelGGL((ln_scores_kernel<TT, 1>), grid, dim3(256), lds, s, x32, gamma, beta, y32, (TT*)y, (TT*)ypos,               \
                       (const TT*)pos, mean, rstd, u_next, scores_next, (int)L, (int)D, (int)H)
    if (dtype == SVOL_BF16) SVOL_LNS(bf16_t);
    else if (dtype == SVOL_F16) SVOL_LNS(f16_t);
    else SVOL_LNS(float);
#undef SVOL_LNS
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_gate_bwd(const float* dy32, const void* dy, const void* dy2, const float* x32, const void* pos, const float* u,
                  const float* gamma, const float* a, const float* mean, const float* rstd, const float* ws, float* ws2,
                  float* dx32, float* du, float* dgamma, float* dbeta, int64_t B, int64_t L, int64_t D, int64_t H, int dtype,
                  void* stream) {
    if ((!dy32 && !dy && !dy2) || !x32 || !pos || !u || !gamma || !a || !mean || !rstd || !ws || !ws2 || !dx32 || !du ||
        !dgamma || !dbeta)
        return SVOL_E_INVALID;
    if (B <= 0 || L <= 0 || D <= 0 || H <= 0) return SVOL_E_INVALID;
    if (D % 4 || D > GP * 256 || H > GH || B > 65535 || L > (1 << 24)) return SVOL_E_UNSUPPORTED;
    if (!svol_is16(dtype) && dtype != SVOL_F32) return SVOL_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int np_ = D <= 256 ? 1 : (D <= 512 ? 2 : 4);
    const float* scores = ws;
    const float* mx = ws + B * H * L;
    const float* sm = mx + B * H;
    float* da = ws2;
    float* cc = ws2 + B * L;
    const int64_t M = B * L;
    const int rpw1 = rows_per_wave(M, gate_waves(1));
    const unsigned g1 = (unsigned)(((M + rpw1 - 1) / rpw1 + 3) / 4);
    const int rpw3 = rows_per_wave(L, gate_waves(2) / B + 1);
    dim3 g3((unsigned)((L + 4 * rpw3 - 1) / (4 * rpw3)), (unsigned)B);
    if (hipMemsetAsync(cc, 0, sizeof(float) * (size_t)(B * H), s) != hipSuccess) return SVOL_E_LAUNCH;
    // deterministic mode (common.h): the three atomic reductions of the two kernels go through per-wave / per-workgroup rows of a
    // stream-ordered scratch and are folded in index order: [waves][B*H] (zero-filled: a wave visits few batch elements) |
    // [workgroups][2*D] | [B][workgroups per batch element][H*D]
    const bool det_mode = svol_deterministic();
    const size_t n_cc = (size_t)g1 * 4 * (size_t)(B * H), n_gb = (size_t)g1 * 2 * (size_t)D, n_du = (size_t)B * g3.x * (size_t)(H * D);
    DetScratch det(det_mode ? n_cc + n_gb + n_du : 0, s, true);
    if (det_mode && !det.p) return SVOL_E_LAUNCH;
    float* det_cc = det_mode ? det.p : nullptr;
    float* det_gb = det_mode ? det.p + n_cc : nullptr;
    float* det_du = det_mode ? det.p + n_cc + n_gb : nullptr;
#define SVOL_GATE_BWD(TT)                                                                                                   \
    do {                                                                                                                    \
        if (np_ == 1) hipLaunchKernelGGL((gate_bwd_ln_kernel<TT, 1>), dim3(g1), dim3(256), 0, s, dy32, (const TT*)dy, (const TT*)dy2, x32, a, gamma, mean, rstd, dx32, da, dgamma, dbeta, scores, mx, sm, cc, (int)L, (int)H, M, (int)D, rpw1, det_cc, det_gb); \
        else if (np_ == 2) hipLaunchKernelGGL((gate_bwd_ln_kernel<TT, 2>), dim3(g1), dim3(256), 0, s, dy32, (const TT*)dy, (const TT*)dy2, x32, a, gamma, mean, rstd, dx32, da, dgamma, dbeta, scores, mx, sm, cc, (int)L, (int)H, M, (int)D, rpw1, det_cc, det_gb); \
        else hipLaunchKernelGGL((gate_bwd_ln_kernel<TT, 4>), dim3(g1), dim3(256), 0, s, dy32, (const TT*)dy, (const TT*)dy2, x32, a, gamma, mean, rstd, dx32, da, dgamma, dbeta, scores, mx, sm, cc, (int)L, (int)H, M, (int)D, rpw1, det_cc, det_gb); \
        if (det_mode) {                                                                                                     \
            det_fold(det_cc, (int)(g1 * 4), B * H, cc, B * H, s);                                                           \
            det_fold(det_gb, (int)g1, 2 * D, dgamma, D, s);                                                                 \
            det_fold(det_gb + D, (int)g1, 2 * D, dbeta, D, s);                                                              \
        }                                                                                                                   \
        if (np_ == 1) hipLaunchKernelGGL((gate_bwd_apply_kernel<TT, 1>), g3, dim3(256), 0, s, x32, (const TT*)pos, u, scores, mx, sm, da, cc, dx32, du, (int)L, (int)D, (int)H, rpw3, det_du); \
        else if (np_ == 2) hipLaunchKernelGGL((gate_bwd_apply_kernel<TT, 2>), g3, dim3(256), 0, s, x32, (const TT*)pos, u, scores, mx, sm, da, cc, dx32, du, (int)L, (int)D, (int)H, rpw3, det_du); \
        else hipLaunchKernelGGL((gate_bwd_apply_kernel<TT, 4>), g3, dim3(256), 0, s, x32, (const TT*)pos, u, scores, mx, sm, da, cc, dx32, du, (int)L, (int)D, (int)H, rpw3, det_du); \
    } while (0)
    if (dtype == SVOL_BF16) SVOL_GATE_BWD(bf16_t);
    else if (dtype == SVOL_F16) SVOL_GATE_BWD(f16_t);
    else SVOL_GATE_BWD(float);
#undef SVOL_GATE_BWD
    if (det_mode) det_fold(det_du, (int)g3.x, H * D, du, H * D, s, (int)B, (int64_t)g3.x * H * D, H * D);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_gate_vectors_fwd(const float* skch, const float* W_in, const float* b_in, float* q_out, float* u_out, int64_t B, int64_t D,
                          int64_t H, void* stream) {
    if (!skch || !W_in || !b_in || !q_out || !u_out || B <= 0 || D <= 0 || H <= 0) return SVOL_E_INVALID;
    if (D % 4 || D > GP * 256 || H > GH || D % H || B > 65535) return SVOL_E_UNSUPPORTED;
    if (!aligned16(W_in)) return SVOL_E_INVALID;
    hipLaunchKernelGGL(gate_vec_fwd_kernel, dim3((unsigned)H, (unsigned)B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), skch,
                       W_in, b_in, q_out, u_out, (int)D, (int)H);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_gate_vectors_bwd(const float* du, const float* skch, const float* W_in, const float* q, float* dq_ws, float* dskch,
                          float* dW_in, float* db_in, int64_t B, int64_t D, int64_t H, void* stream) {
    if (!du || !skch || !W_in || !q || !dq_ws || !dW_in || !db_in || B <= 0 || D <= 0 || H <= 0) return SVOL_E_INVALID;
    if (D % 4 || D > GP * 256 || H > GH || D % H || B > 65535) return SVOL_E_UNSUPPORTED;
    if (!aligned16(W_in)) return SVOL_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gate_vec_bwd_a_kernel, dim3((unsigned)H, (unsigned)B), dim3(256), 0, s, du, W_in, dq_ws, (int)D, (int)H);
    hipLaunchKernelGGL(gate_vec_bwd_b_kernel, dim3((unsigned)(D + B)), dim3(256), 0, s, du, skch, q, dq_ws, W_in, dW_in, db_in, dskch,
                       (int)B, (int)D, (int)H);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

// n_layers problems that share skch: per-layer pointer arrays (host arrays of n_layers device pointers).  dskch: per-layer [B, D]
// outputs (the caller sums them: every layer's sketch gradient has its own owner, no atomics), or NULL.
int svol_gate_vectors_fwd_multi(const float* skch, const float* const* W_in, const float* const* b_in, float* const* q_out,
                                float* const* u_out, int64_t n_layers, int64_t B, int64_t D, int64_t H, void* stream) {
    if (!skch || !W_in || !b_in || !q_out || !u_out || B <= 0 || D <= 0 || H <= 0 || n_layers <= 0) return SVOL_E_INVALID;
    if (D % 4 || D > GP * 256 || H > GH || D % H || B > 65535 || n_layers > SVOL_GATE_VEC_MAX_LAYERS) return SVOL_E_UNSUPPORTED;
    GateVecMulti g{};
    for (int l = 0; l < n_layers; ++l) {
        if (!W_in[l] || !b_in[l] || !q_out[l] || !u_out[l] || !aligned16(W_in[l])) return SVOL_E_INVALID;
        g.W[l] = W_in[l]; g.bias[l] = b_in[l]; g.q[l] = q_out[l]; g.u[l] = u_out[l];
    }
    hipLaunchKernelGGL(gate_vec_fwd_multi_kernel, dim3((unsigned)H, (unsigned)B, (unsigned)n_layers), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), skch, g, (int)D, (int)H);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_gate_vectors_bwd_multi(const float* const* du, const float* skch, const float* const* W_in, const float* const* q,
                                float* const* dq_ws, float* const* dskch, float* const* dW_in, float* const* db_in, int64_t n_layers,
                                int64_t B, int64_t D, int64_t H, void* stream) {
    if (!du || !skch || !W_in || !q || !dq_ws || !dW_in || !db_in || B <= 0 || D <= 0 || H <= 0 || n_layers <= 0) return SVOL_E_INVALID;
    if (D % 4 || D > GP * 256 || H > GH || D % H || B > 65535 || n_layers > SVOL_GATE_VEC_MAX_LAYERS) return SVOL_E_UNSUPPORTED;
    GateVecMulti g{};
    for (int l = 0; l < n_layers; ++l) {
        if (!du[l] || !W_in[l] || !q[l] || !dq_ws[l] || !dW_in[l] || !db_in[l] || !aligned16(W_in[l])) return SVOL_E_INVALID;
        g.du[l] = du[l]; g.W[l] = W_in[l]; g.q[l] = const_cast<float*>(q[l]); g.dq[l] = dq_ws[l]; g.dW[l] = dW_in[l]; g.db[l] = db_in[l];
        g.dskch[l] = dskch ? dskch[l] : nullptr;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gate_vec_bwd_a_multi_kernel, dim3((unsigned)H, (unsigned)B, (unsigned)n_layers), dim3(256), 0, s, g, (int)D, (int)H);
    hipLaunchKernelGGL(gate_vec_bwd_b_multi_kernel, dim3((unsigned)(D + B), 1, (unsigned)n_layers), dim3(256), 0, s, skch, g, (int)B,
                       (int)D, (int)H);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

}  // extern "C"
