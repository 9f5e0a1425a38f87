// Multi-head attention core for the SVOL head (gfx950 / CDNA4): flash-style forward and backward that
// never materialise the Lq x Lk score matrix (the reference's nn.MultiheadAttention writes
// [B*h, L, L] = 9.4 GiB per layer at the benchmark size; cross_modal_transformer.py:139).
//
// Head dim is small (d_h = 32 at the benchmark size, 8 at the CPU config), so ONE 32x32 MFMA tile
// spans the whole head dim.  All three kernels share one scheme built on the "swapped" product so
// that softmax statistics are lane-local (MI355X guide: swapped QK^T, accumulator-as-next-operand):
//
//   "lane side"  : a block of 32 rows held in registers as the MFMA B operand (one row per lane&31)
//   "reg side"   : 64-row tiles streamed through LDS, used as the MFMA A operand
//   first product : X[reg row][lane row] = A_tile[reg row][:] . B_block[lane row][:]   (contract d)
//   second product: Y[d][lane row]      += A_tileT[d][reg row] * f(X)[reg row][lane row]
//                   f(X) is fed straight from the accumulator registers (no LDS round trip).
//
//   forward : lane = queries, reg = keys : X = K Q^T -> P ; O^T += V^T P
//   dq      : lane = queries, reg = keys : P, dP = V dO^T -> dS ; dQ^T += K^T dS
//   dk/dv   : lane = keys, reg = queries : P, dV^T += dO^T P ; dP = dO V^T -> dS ; dK^T += Q^T dS
//
// dQ is produced by its own pass instead of fp32 atomics: with d_h = 32 there are only 64 FLOP per
// atomic byte and the chip-wide atomic rate (~1.3 TB/s) would bound the kernel (guide, Guideline 12).
//
// Element types: this file's generic template serves f32 (v_mfma_f32_32x32x2_f32, exact fp32 — the parity
// mode); bf16 (v_mfma_f32_32x32x16_bf16) is dispatched to the tuned kernels of attention_bf16.hip.
// Softmax runs in the log2 domain with fp32 statistics; exp via v_exp_f32.
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

template <typename T> struct AT;
template <> struct AT<float> {
    static constexpr int EPC = 4, NA = 4, CPR = 8, NAT_ROW = 128, TR_STRIDE = 64 * 4 + 16;
};

// ---- staging ---------------------------------------------------------------
// natural image: [64 rows][32 d] (zero padded), 16-byte chunks XOR-swizzled against bank conflicts
template <typename T>
__device__ __forceinline__ int nat_off(int row, int ch) {
    if constexpr (sizeof(T) == 2) return row * 64 + ((ch ^ ((row >> 2) & 3)) << 4);
    else return row * 128 + ((ch ^ (row & 7)) << 4);
}

// Stage 64 rows starting at row0 of a [rows, ld] matrix (head column offset already applied) into the
// natural image `nat` and/or the transposed image `tr` ([32 d][64 rows], padded row stride).
template <typename T>
__device__ __forceinline__ void stage_tile(char* nat, char* tr, const T* g, int64_t ld, int row0, int limit, int dh,
                                           int tid) {
    constexpr int EPC = AT<T>::EPC, CPR = AT<T>::CPR, ST = AT<T>::TR_STRIDE;
    for (int c = tid; c < 64 * CPR; c += 256) {
        const int row = c / CPR, ch = c % CPR;
        const int d0 = ch * EPC;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row0 + row < limit && d0 < dh) v = *reinterpret_cast<const uint4*>(g + (int64_t)(row0 + row) * ld + d0);
        if (nat) *reinterpret_cast<uint4*>(nat + nat_off<T>(row, ch)) = v;
        if (tr) {
            if constexpr (sizeof(T) == 2) {
                const bf16x8 e = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) *reinterpret_cast<bf16_t*>(tr + (d0 + j) * ST + row * 2) = e[j];
            } else {
                const f32x4 e = __builtin_bit_cast(f32x4, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<float*>(tr + (d0 + j) * ST + row * 4) = e[j];
            }
        }
    }
}

// lane-side block: this lane's row (r = lane & 31), the d-chunks its half (h = lane >> 5) feeds to the MFMA
template <typename T>
__device__ __forceinline__ void load_lane_block(uint4 (&out)[AT<T>::NA], const T* g, int64_t ld, int row, bool valid,
                                                int dh, int h) {
    constexpr int EPC = AT<T>::EPC, NA = AT<T>::NA;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int ch = (sizeof(T) == 2) ? (2 * i + h) : (4 * h + i);
        const int d0 = ch * EPC;
        out[i] = (valid && d0 < dh) ? *reinterpret_cast<const uint4*>(g + (int64_t)row * ld + d0) : make_uint4(0, 0, 0, 0);
    }
}

// A operand of the first product from the natural image: tile row `row`, contraction over d
template <typename T>
__device__ __forceinline__ void read_nat(uint4 (&a)[AT<T>::NA], const char* nat, int row, int h) {
#pragma unroll
    for (int i = 0; i < AT<T>::NA; ++i) {
        const int ch = (sizeof(T) == 2) ? (2 * i + h) : (4 * h + i);
        a[i] = *reinterpret_cast<const uint4*>(nat + nat_off<T>(row, ch));
    }
}

// A operand of the second product from the transposed image: row r = d, contraction over the 32 reg-side
// rows of sub-tile t, in the order the accumulator registers of the first product present them
template <typename T>
__device__ __forceinline__ void read_tr(uint4 (&a)[AT<T>::NA], const char* tr, int r, int h, int t) {
    constexpr int ST = AT<T>::TR_STRIDE;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int col = 32 * t + 16 * s + 4 * h;
            const uint2 p0 = *reinterpret_cast<const uint2*>(tr + r * ST + col * 2);
            const uint2 p1 = *reinterpret_cast<const uint2*>(tr + r * ST + (col + 8) * 2);
            a[s] = make_uint4(p0.x, p0.y, p1.x, p1.y);
        }
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = 32 * t + 8 * g + 4 * h;
            a[g] = *reinterpret_cast<const uint4*>(tr + r * ST + col * 4);
        }
    }
}

template <typename T>
__device__ __forceinline__ void mma_first(f32x16& acc, const uint4 (&a)[AT<T>::NA], const uint4 (&b)[AT<T>::NA]) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[s]), __builtin_bit_cast(bf16x8, b[s]),
                                                          acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 af = __builtin_bit_cast(f32x4, a[j]), bf = __builtin_bit_cast(f32x4, b[j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
        }
    }
}

// acc += A_tr * X, X = accumulator tile of the first product used directly as B operand
template <typename T>
__device__ __forceinline__ void mma_second(f32x16& acc, const uint4 (&a)[AT<T>::NA], const f32x16& x) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 b;
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = (bf16_t)x[8 * s + j];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[s]), b, acc, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 af = __builtin_bit_cast(f32x4, a[g]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], x[4 * g + e], acc, 0, 0, 0);
        }
    }
}

// accumulator [d][lane row] -> out[row][d] (T), optionally scaled
template <typename T>
__device__ __forceinline__ void store_acc_T(const f32x16& acc, T* out, int64_t ld, int row, bool valid, int dh, int h,
                                            float mul) {
    if (!valid) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int d0 = 8 * g + 4 * h;
        if (d0 < dh) {
            Vec4<T> v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v.set(e, acc[4 * g + e] * mul);
            v.store(out + (int64_t)row * ld + d0);
        }
    }
}

struct AttnArgs {
    const void *q, *k, *v, *o, *d_o;
    void *out_o, *dq, *dk, *dv;
    const float* kbias;
    float *lse2, *delta;
    int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int B, H, Lq, Lk, dh;
    float scale;
    float premul;
    // attention-probability dropout (nn.MultiheadAttention(dropout=p) in training mode, transformer.py:158-160,216-222): the
    // softmax numerators that feed P V are multiplied by a stateless keep mask (0 or 1/(1-p)) of the element index
    // ((b*H + h)*Lq + q)*Lk + key — the row sums / lse stay those of the undropped softmax; backward regenerates the mask
    float drop_p, drop_inv;
    uint64_t drop_seed;
};

// keep-mask scale of score element (row, key); row = (b*H + h)*Lq + q
__device__ __forceinline__ float attn_drop(uint64_t seed, uint64_t row, int key, float p, float inv) {
    return dropout_scale(seed, row, (uint32_t)key, p, inv);
}
// the same for keys key0 (EVEN) and key0 + 1 of one row: one generator call for the pair
__device__ __forceinline__ void attn_drop2(uint32_t rowmix, int key0, uint32_t thr16, float inv, float& s0, float& s1) {
    const uint32_t bits = drop_bits(rowmix, (uint32_t)key0 >> 1);
    s0 = drop_pick(bits, 0u, thr16, inv);
    s1 = drop_pick(bits, 1u, thr16, inv);
}

template <typename T> struct Smem {
    static constexpr int NAT = 64 * AT<T>::NAT_ROW;
    static constexpr int TR = 32 * AT<T>::TR_STRIDE;
};

// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[Smem<T>::NAT + Smem<T>::TR + 64 * 4];
    char* sK = smem;
    char* sVt = smem + Smem<T>::NAT;
    float* sbias = reinterpret_cast<float*>(smem + Smem<T>::NAT + Smem<T>::TR);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, hh = blockIdx.y;
    const int qrow = blockIdx.x * 128 + wave * 32 + r;
    const bool qvalid = qrow < p.Lq;
    const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * p.dh;
    const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * p.dh;
    const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * p.dh;
    const float* kb = p.kbias ? p.kbias + (int64_t)b * p.Lk : nullptr;
    const float sc = p.premul != 0.f ? 1.f : p.scale * LOG2E;

    uint4 qb[AT<T>::NA];
    load_lane_block<T>(qb, Q, p.ldq, qrow, qvalid, p.dh, h);

    float m = -INFINITY, l = 0.f;
    f32x16 O;
#pragma unroll
    for (int i = 0; i < 16; ++i) O[i] = 0.f;

    for (int kt = 0; kt < p.Lk; kt += 64) {
        __syncthreads();
        stage_tile<T>(sK, nullptr, K, p.ldk, kt, p.Lk, p.dh, tid);
        stage_tile<T>(nullptr, sVt, V, p.ldv, kt, p.Lk, p.dh, tid);
        if (tid < 64) {
            const int key = kt + tid;
            sbias[tid] = key < p.Lk ? (kb ? kb[key] * LOG2E : 0.f) : -INFINITY;
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            uint4 ka[AT<T>::NA];
            read_nat<T>(ka, sK, sub * 32 + r, h);
            f32x16 S;
#pragma unroll
            for (int i = 0; i < 16; ++i) S[i] = 0.f;
            mma_first<T>(S, ka, qb);
            float mloc = -INFINITY;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(sbias + sub * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    S[4 * g + e] = S[4 * g + e] * sc + bb[e];
                    mloc = fmaxf(mloc, S[4 * g + e]);
                }
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float m_new = fmaxf(m, mloc);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m - m_use);
            float ls = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                S[i] = __builtin_amdgcn_exp2f(S[i] - m_use);
                ls += S[i];
            }
            l = l * alpha + ls;
            m = m_new;
#pragma unroll
            for (int i = 0; i < 16; ++i) O[i] *= alpha;
            if (p.drop_p > 0.f) {
                const uint32_t rm = drop_row(drop_seed32(p.drop_seed), (((uint64_t)b * p.H + hh) * p.Lq + (qvalid ? qrow : 0)));
                const uint32_t thr = drop_thr16(p.drop_p);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {   // (the lane's keys 8g + 4h + e: pairs share one generator call)
                        float s0, s1;
                        attn_drop2(rm, kt + sub * 32 + 8 * g + 4 * h + e, thr, p.drop_inv, s0, s1);
                        S[4 * g + e] *= s0;
                        S[4 * g + e + 1] *= s1;
                    }
            }
            uint4 va[AT<T>::NA];
            read_tr<T>(va, sVt, r, h, sub);
            mma_second<T>(O, va, S);
        }
    }
    const float lt = l + __shfl_xor(l, 32, 64);
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    T* Oo = reinterpret_cast<T*>(p.out_o) + (int64_t)b * p.Lq * p.ldo + hh * p.dh;
    store_acc_T<T>(O, Oo, p.ldo, qrow, qvalid, p.dh, h, inv);
    if (qvalid && h == 0) p.lse2[((int64_t)b * p.H + hh) * p.Lq + qrow] = m + __builtin_amdgcn_logf(lt);
}

// delta[b,h,q] = sum_d dO[q, h*dh + d] * O[q, h*dh + d]
template <typename T>
__global__ void attn_delta_kernel(AttnArgs p) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)p.B * p.Lq * p.H;
    if (idx >= total) return;
    const int hh = (int)(idx % p.H);
    const int64_t row = idx / p.H;  // b*Lq + q
    const int b = (int)(row / p.Lq), q = (int)(row % p.Lq);
    const T* o = reinterpret_cast<const T*>(p.o) + row * p.ldo + hh * p.dh;
    const T* d = reinterpret_cast<const T*>(p.d_o) + row * p.lddo + hh * p.dh;
    float s = 0.f;
    for (int i = 0; i < p.dh; i += 4) {
        Vec4<T> a, c;
        a.load(o + i);
        c.load(d + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) s += a.get(e) * c.get(e);
    }
    p.delta[((int64_t)b * p.H + hh) * p.Lq + q] = s;
}

// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * Smem<T>::NAT + Smem<T>::TR + 64 * 4];
    char* sK = smem;
    char* sV = smem + Smem<T>::NAT;
    char* sKt = smem + 2 * Smem<T>::NAT;
    float* sbias = reinterpret_cast<float*>(smem + 2 * Smem<T>::NAT + Smem<T>::TR);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, hh = blockIdx.y;
    const int qrow = blockIdx.x * 128 + wave * 32 + r;
    const bool qvalid = qrow < p.Lq;
    const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * p.dh;
    const T* dO = reinterpret_cast<const T*>(p.d_o) + (int64_t)b * p.Lq * p.lddo + hh * p.dh;
    const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * p.dh;
    const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * p.dh;
    const float* kb = p.kbias ? p.kbias + (int64_t)b * p.Lk : nullptr;
    const float sc = p.premul != 0.f ? 1.f : p.scale * LOG2E;

    uint4 qb[AT<T>::NA], dob[AT<T>::NA];
    load_lane_block<T>(qb, Q, p.ldq, qrow, qvalid, p.dh, h);
    load_lane_block<T>(dob, dO, p.lddo, qrow, qvalid, p.dh, h);
    const int64_t sidx = ((int64_t)b * p.H + hh) * p.Lq + qrow;
    const float lse = qvalid ? p.lse2[sidx] : INFINITY;
    const float dl = qvalid ? p.delta[sidx] : 0.f;

    f32x16 dQ;
#pragma unroll
    for (int i = 0; i < 16; ++i) dQ[i] = 0.f;

    for (int kt = 0; kt < p.Lk; kt += 64) {
        __syncthreads();
        stage_tile<T>(sK, sKt, K, p.ldk, kt, p.Lk, p.dh, tid);
        stage_tile<T>(sV, nullptr, V, p.ldv, kt, p.Lk, p.dh, tid);
        if (tid < 64) {
            const int key = kt + tid;
            sbias[tid] = key < p.Lk ? (kb ? kb[key] * LOG2E : 0.f) : -INFINITY;
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            uint4 a[AT<T>::NA];
            read_nat<T>(a, sK, sub * 32 + r, h);
            f32x16 S, dP;
#pragma unroll
            for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
            mma_first<T>(S, a, qb);
            read_nat<T>(a, sV, sub * 32 + r, h);
            mma_first<T>(dP, a, dob);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(sbias + sub * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pe = __builtin_amdgcn_exp2f(S[4 * g + e] * sc + bb[e] - lse);
                    float dpe = dP[4 * g + e];
                    if (p.drop_p > 0.f)
                        dpe *= attn_drop(p.drop_seed, (((uint64_t)b * p.H + hh) * p.Lq + (qvalid ? qrow : 0)),
                                         kt + sub * 32 + 8 * g + 4 * h + e, p.drop_p, p.drop_inv);
                    S[4 * g + e] = pe * (dpe - dl) * p.scale;
                }
            }
            read_tr<T>(a, sKt, r, h, sub);
            mma_second<T>(dQ, a, S);
        }
    }
    T* dQo = reinterpret_cast<T*>(p.dq) + (int64_t)b * p.Lq * p.lddq + hh * p.dh;
    store_acc_T<T>(dQ, dQo, p.lddq, qrow, qvalid, p.dh, h, 1.f);
}

// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * Smem<T>::NAT + 2 * Smem<T>::TR + 128 * 4];
    char* sQ = smem;
    char* sdO = smem + Smem<T>::NAT;
    char* sQt = smem + 2 * Smem<T>::NAT;
    char* sdOt = smem + 2 * Smem<T>::NAT + Smem<T>::TR;
    float* slse = reinterpret_cast<float*>(smem + 2 * Smem<T>::NAT + 2 * Smem<T>::TR);
    float* sdl = slse + 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, hh = blockIdx.y;
    const int krow = blockIdx.x * 128 + wave * 32 + r;
    const bool kvalid = krow < p.Lk;
    const T* Q = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * p.dh;
    const T* dO = reinterpret_cast<const T*>(p.d_o) + (int64_t)b * p.Lq * p.lddo + hh * p.dh;
    const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * p.dh;
    const T* V = reinterpret_cast<const T*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * p.dh;
    const float sc = p.premul != 0.f ? 1.f : p.scale * LOG2E;
    const float kbl = kvalid ? (p.kbias ? p.kbias[(int64_t)b * p.Lk + krow] * LOG2E : 0.f) : -INFINITY;
    const float* lse_g = p.lse2 + ((int64_t)b * p.H + hh) * p.Lq;
    const float* dl_g = p.delta + ((int64_t)b * p.H + hh) * p.Lq;

    uint4 kbk[AT<T>::NA], vbk[AT<T>::NA];
    load_lane_block<T>(kbk, K, p.ldk, krow, kvalid, p.dh, h);
    load_lane_block<T>(vbk, V, p.ldv, krow, kvalid, p.dh, h);

    f32x16 dK, dV;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dK[i] = 0.f; dV[i] = 0.f; }

    for (int qt = 0; qt < p.Lq; qt += 64) {
        __syncthreads();
        stage_tile<T>(sQ, sQt, Q, p.ldq, qt, p.Lq, p.dh, tid);
        stage_tile<T>(sdO, sdOt, dO, p.lddo, qt, p.Lq, p.dh, tid);
        if (tid < 64) {
            const int qi = qt + tid;
            slse[tid] = qi < p.Lq ? lse_g[qi] : INFINITY;
            sdl[tid] = qi < p.Lq ? dl_g[qi] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            uint4 a[AT<T>::NA];
            read_nat<T>(a, sQ, sub * 32 + r, h);
            f32x16 S, dP;
#pragma unroll
            for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
            mma_first<T>(S, a, kbk);  // S[q][key]
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ls = *reinterpret_cast<const f32x4*>(slse + sub * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) S[4 * g + e] = __builtin_amdgcn_exp2f(S[4 * g + e] * sc + kbl - ls[e]);
            }
            f32x16 MS;   // keep-mask scales of this block (all ones without dropout)
#pragma unroll
            for (int i = 0; i < 16; ++i) MS[i] = 1.f;
            if (p.drop_p > 0.f) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int qi = min(qt + sub * 32 + 8 * g + 4 * h + e, p.Lq - 1);
                        MS[4 * g + e] = attn_drop(p.drop_seed, (((uint64_t)b * p.H + hh) * p.Lq + qi), kvalid ? krow : 0,
                                                  p.drop_p, p.drop_inv);
                    }
            }
            read_tr<T>(a, sdOt, r, h, sub);
            {
                f32x16 Pd = S;
#pragma unroll
                for (int i = 0; i < 16; ++i) Pd[i] *= MS[i];
                mma_second<T>(dV, a, Pd);  // dV^T += dO^T (P . mask)
            }
            read_nat<T>(a, sdO, sub * 32 + r, h);
            mma_first<T>(dP, a, vbk);  // dP[q][key] = dO V^T
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 dd = *reinterpret_cast<const f32x4*>(sdl + sub * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) S[4 * g + e] = S[4 * g + e] * (dP[4 * g + e] * MS[4 * g + e] - dd[e]) * p.scale;
            }
            read_tr<T>(a, sQt, r, h, sub);
            mma_second<T>(dK, a, S);  // dK^T += Q^T dS
        }
    }
    T* dKo = reinterpret_cast<T*>(p.dk) + (int64_t)b * p.Lk * p.lddk + hh * p.dh;
    T* dVo = reinterpret_cast<T*>(p.dv) + (int64_t)b * p.Lk * p.lddv + hh * p.dh;
    store_acc_T<T>(dK, dKo, p.lddk, krow, kvalid, p.dh, h, p.premul != 0.f ? 1.f / p.premul : 1.f);
    store_acc_T<T>(dV, dVo, p.lddv, krow, kvalid, p.dh, h, 1.f);
}

int check_common(int64_t B, int64_t H, int64_t Lq, int64_t Lk, int64_t dh, int dtype) {
    if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || dh <= 0) return SVOL_E_INVALID;
    if (!svol_is16(dtype) && dtype != SVOL_F32) return SVOL_E_INVALID;
    if (dh > 32 || dh % 8) return SVOL_E_UNSUPPORTED;
    if (B > 65535 || H > 65535 || Lq > (1 << 24) || Lk > (1 << 24)) return SVOL_E_UNSUPPORTED;
    return SVOL_OK;
}
bool ld_ok(int64_t ld, int dtype) { return ld % (svol_is16(dtype) ? 8 : 4) == 0; }

}  // namespace

// bf16 fast path (attention_bf16.hip)
int svol_attn_fwd_bf16_launch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                              int64_t ldo, float* lse2, const float* kbias, int B, int H, int Lq, int Lk, int dh, float scale,
                              float premul, float* ws, int64_t ws_bytes, float drop_p, uint64_t drop_seed, hipStream_t s);
int svol_attn_bwd_bf16_launch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                              const void* o, int64_t ldo, const void* d_o, int64_t lddo, const float* lse2, float* delta,
                              const float* kbias, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, int B,
                              int H, int Lq, int Lk, int dh, float scale, float premul, float* ws, int64_t ws_bytes,
                              float drop_p, uint64_t drop_seed, int flags, void* ev_prep, hipStream_t s);

int64_t svol_attn_ws_floats_bf16(int B, int H, int Lq, int Lk, int dh);
int64_t svol_attn_sp_image_bytes_bf16(int B, int H, int Lq, int Lk, int dh, int64_t ws_bytes);
int svol_attn_sp_zero_bf16_launch(float* ws, int64_t ws_bytes, int B, int H, int Lq, int Lk, int dh, hipStream_t s);
int64_t svol_attn_sp_image_bytes_f16(int B, int H, int Lq, int Lk, int dh, int64_t ws_bytes);
int svol_attn_sp_zero_f16_launch(float* ws, int64_t ws_bytes, int B, int H, int Lq, int Lk, int dh, hipStream_t s);
// the same kernels compiled with fp16 operands (attention_bf16.hip with -DSVOL_H16_FP16)
int svol_attn_fwd_f16_launch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                             int64_t ldo, float* lse2, const float* kbias, int B, int H, int Lq, int Lk, int dh, float scale,
                             float premul, float* ws, int64_t ws_bytes, float drop_p, uint64_t drop_seed, hipStream_t s);
int svol_attn_bwd_f16_launch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                             const void* o, int64_t ldo, const void* d_o, int64_t lddo, const float* lse2, float* delta,
                             const float* kbias, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, int B,
                             int H, int Lq, int Lk, int dh, float scale, float premul, float* ws, int64_t ws_bytes,
                             float drop_p, uint64_t drop_seed, int flags, void* ev_prep, hipStream_t s);

extern "C" {

int64_t svol_attn_ws_bytes(int64_t B, int64_t H, int64_t Lq, int64_t Lk, int64_t dh) {
    if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || dh <= 0 || B > (1 << 24) || Lq > (1 << 24) || Lk > (1 << 24)) return 0;
    return 4 * svol_attn_ws_floats_bf16((int)B, (int)H, (int)Lq, (int)Lk, (int)dh);
}

static int attn_fwd_impl(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                         int64_t ldo, float* lse2, const float* kbias, int64_t B, int64_t H, int64_t Lq, int64_t Lk, int64_t dh,
                         float scale, float q_premul, void* ws, int64_t ws_bytes, float drop_p, uint64_t drop_seed, int dtype,
                         void* stream) {
    if (!q || !k || !v || !o || !lse2) return SVOL_E_INVALID;
    if (!(drop_p >= 0.f && drop_p < 1.f)) return SVOL_E_INVALID;
    int rc = check_common(B, H, Lq, Lk, dh, dtype);
    if (rc) return rc;
    if (!ld_ok(ldq, dtype) || !ld_ok(ldk, dtype) || !ld_ok(ldv, dtype) || !ld_ok(ldo, dtype)) return SVOL_E_UNSUPPORTED;
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(o)) return SVOL_E_INVALID;
    AttnArgs p{};
    p.q = q; p.k = k; p.v = v; p.out_o = o; p.lse2 = lse2; p.kbias = kbias;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.B = (int)B; p.H = (int)H; p.Lq = (int)Lq; p.Lk = (int)Lk; p.dh = (int)dh; p.scale = scale; p.premul = q_premul;
    p.drop_p = drop_p; p.drop_inv = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; p.drop_seed = drop_seed;
    dim3 grid((unsigned)((Lq + 127) / 128), (unsigned)H, (unsigned)B);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (svol_is16(dtype))
        return (dtype == SVOL_BF16 ? svol_attn_fwd_bf16_launch : svol_attn_fwd_f16_launch)(
            q, ldq, k, ldk, v, ldv, o, ldo, lse2, kbias, (int)B, (int)H, (int)Lq, (int)Lk, (int)dh, scale, q_premul,
            aligned16(ws) ? (float*)ws : nullptr, ws_bytes, drop_p, drop_seed, s);
    hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, dim3(256), 0, s, p);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_attn_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                  int64_t ldo, float* lse2, const float* kbias, int64_t B, int64_t H, int64_t Lq, int64_t Lk, int64_t dh,
                  float scale, float q_premul, void* ws, int64_t ws_bytes, int dtype, void* stream) {
    return attn_fwd_impl(q, ldq, k, ldk, v, ldv, o, ldo, lse2, kbias, B, H, Lq, Lk, dh, scale, q_premul, ws, ws_bytes, 0.f, 0, dtype,
                         stream);
}
int svol_attn_fwd_dropout(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                          int64_t ldo, float* lse2, const float* kbias, int64_t B, int64_t H, int64_t Lq, int64_t Lk, int64_t dh,
                          float scale, float q_premul, void* ws, int64_t ws_bytes, float dropout_p, uint64_t seed, int dtype,
                          void* stream) {
    return attn_fwd_impl(q, ldq, k, ldk, v, ldv, o, ldo, lse2, kbias, B, H, Lq, Lk, dh, scale, q_premul, ws, ws_bytes, dropout_p, seed,
                         dtype, stream);
}

static int attn_bwd_impl(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                         int64_t ldo, const void* d_o, int64_t lddo, const float* lse2, float* delta, const float* kbias, void* dq,
                         int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, int64_t B, int64_t H, int64_t Lq, int64_t Lk,
                         int64_t dh, float scale, float q_premul, void* ws, int64_t ws_bytes, float drop_p, uint64_t drop_seed,
                         int dtype, void* stream, int flags = 0, void* ev_prep = nullptr) {
    if (!q || !k || !v || !o || !d_o || !lse2 || !delta || !dq || !dk || !dv) return SVOL_E_INVALID;
    if (!(drop_p >= 0.f && drop_p < 1.f)) return SVOL_E_INVALID;
    int rc = check_common(B, H, Lq, Lk, dh, dtype);
    if (rc) return rc;
    if (!ld_ok(ldq, dtype) || !ld_ok(ldk, dtype) || !ld_ok(ldv, dtype) || !ld_ok(ldo, dtype) || !ld_ok(lddo, dtype) ||
        !ld_ok(lddq, dtype) || !ld_ok(lddk, dtype) || !ld_ok(lddv, dtype))
        return SVOL_E_UNSUPPORTED;
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(o) || !aligned16(d_o) || !aligned16(dq) ||
        !aligned16(dk) || !aligned16(dv))
        return SVOL_E_INVALID;
    AttnArgs p{};
    p.q = q; p.k = k; p.v = v; p.o = o; p.d_o = d_o; p.lse2 = const_cast<float*>(lse2); p.delta = delta; p.kbias = kbias;
    p.dq = dq; p.dk = dk; p.dv = dv;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    p.B = (int)B; p.H = (int)H; p.Lq = (int)Lq; p.Lk = (int)Lk; p.dh = (int)dh; p.scale = scale; p.premul = q_premul;
    p.drop_p = drop_p; p.drop_inv = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; p.drop_seed = drop_seed;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = B * Lq * H;
    dim3 gd((unsigned)((total + 255) / 256));
    dim3 gq((unsigned)((Lq + 127) / 128), (unsigned)H, (unsigned)B);
    dim3 gk((unsigned)((Lk + 127) / 128), (unsigned)H, (unsigned)B);
    if (svol_is16(dtype))
        return (dtype == SVOL_BF16 ? svol_attn_bwd_bf16_launch : svol_attn_bwd_f16_launch)(
            q, ldq, k, ldk, v, ldv, o, ldo, d_o, lddo, lse2, delta, kbias, dq, lddq, dk, lddk, dv, lddv, (int)B, (int)H, (int)Lq,
            (int)Lk, (int)dh, scale, q_premul, aligned16(ws) ? (float*)ws : nullptr, ws_bytes, drop_p, drop_seed, flags, ev_prep, s);
    hipLaunchKernelGGL(attn_delta_kernel<float>, gd, dim3(256), 0, s, p);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<float>, gq, dim3(256), 0, s, p);
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel<float>, gk, dim3(256), 0, s, p);
    if (ev_prep) (void)hipEventRecord(static_cast<hipEvent_t>(ev_prep), s);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_attn_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                  int64_t ldo, const void* d_o, int64_t lddo, const float* lse2, float* delta, const float* kbias, void* dq,
                  int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, int64_t B, int64_t H, int64_t Lq, int64_t Lk,
                  int64_t dh, float scale, float q_premul, void* ws, int64_t ws_bytes, int dtype, void* stream) {
    return attn_bwd_impl(q, ldq, k, ldk, v, ldv, o, ldo, d_o, lddo, lse2, delta, kbias, dq, lddq, dk, lddk, dv, lddv, B, H, Lq, Lk, dh,
                         scale, q_premul, ws, ws_bytes, 0.f, 0, dtype, stream);
}
int svol_attn_bwd_ex(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                     int64_t ldo, const void* d_o, int64_t lddo, const float* lse2, float* delta, const float* kbias, void* dq,
                     int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, int64_t B, int64_t H, int64_t Lq, int64_t Lk,
                     int64_t dh, float scale, float q_premul, void* ws, int64_t ws_bytes, int dtype, int flags, void* ev_prep,
                     void* stream) {
    if (flags & ~SVOL_ATTN_DQ_PREZEROED) return SVOL_E_INVALID;
    return attn_bwd_impl(q, ldq, k, ldk, v, ldv, o, ldo, d_o, lddo, lse2, delta, kbias, dq, lddq, dk, lddk, dv, lddv, B, H, Lq, Lk, dh,
                         scale, q_premul, ws, ws_bytes, 0.f, 0, dtype, stream, flags, ev_prep);
}
int64_t svol_attn_bwd_sp_image_bytes(int64_t B, int64_t H, int64_t Lq, int64_t Lk, int64_t dh, int64_t ws_bytes, int dtype) {
    if (!svol_is16(dtype) || check_common(B, H, Lq, Lk, dh, dtype)) return 0;
    return (dtype == SVOL_BF16 ? svol_attn_sp_image_bytes_bf16 : svol_attn_sp_image_bytes_f16)((int)B, (int)H, (int)Lq, (int)Lk, (int)dh,
                                                                                              ws_bytes);
}
int svol_attn_bwd_zero_ws(void* ws, int64_t ws_bytes, int64_t B, int64_t H, int64_t Lq, int64_t Lk, int64_t dh, int dtype, void* stream) {
    if (!ws || !aligned16(ws)) return SVOL_E_INVALID;
    if (!svol_is16(dtype)) return SVOL_E_UNSUPPORTED;
    const int rc = check_common(B, H, Lq, Lk, dh, dtype);
    if (rc) return rc;
    return (dtype == SVOL_BF16 ? svol_attn_sp_zero_bf16_launch : svol_attn_sp_zero_f16_launch)(
        static_cast<float*>(ws), ws_bytes, (int)B, (int)H, (int)Lq, (int)Lk, (int)dh, reinterpret_cast<hipStream_t>(stream));
}
int svol_attn_bwd_dropout(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                          int64_t ldo, const void* d_o, int64_t lddo, const float* lse2, float* delta, const float* kbias, void* dq,
                          int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, int64_t B, int64_t H, int64_t Lq, int64_t Lk,
                          int64_t dh, float scale, float q_premul, void* ws, int64_t ws_bytes, float dropout_p, uint64_t seed,
                          int dtype, void* stream) {
    return attn_bwd_impl(q, ldq, k, ldk, v, ldv, o, ldo, d_o, lddo, lse2, delta, kbias, dq, lddq, dk, lddk, dv, lddv, B, H, Lq, Lk, dh,
                         scale, q_premul, ws, ws_bytes, dropout_p, seed, dtype, stream);
}

}  // extern "C"
