// bf16 fast path of the attention core (gfx950): same "swapped product / accumulator-as-operand"
// scheme as attention.hip, restructured after profiling the first version (rocprofv3 PMC: VALU issue,
// not MFMA, bounds these kernels at d_h = 32 — 349 VALU instructions vs 16 MFMAs per 64-key tile):
//
//  * MFMA results stay in VGPRs (__launch_bounds__(256, 2) -> the compiler uses the VGPR form of
//    v_mfma_f32_32x32x16_bf16; the first version spent 40 % of its VALU slots on v_accvgpr_read/write);
//  * 128-row tiles, two LDS buffers, ONE barrier per tile; the next tile's global loads are issued
//    before the MFMA/softmax work of the current tile and written to the other buffer afterwards;
//  * a single row-major LDS image per operand serves both MFMA operand shapes: row reads
//    (ds_read_b128, XOR-swizzled 16-byte chunks) for the product that contracts over d, and hardware
//    transposed reads (ds_read_b64_tr_b16) for the product that contracts over the tile rows — no
//    transposing LDS stores, no second copy;
//  * softmax statistics in the raw-score domain with a LAZY rescale: O and l are rescaled only when
//    some query of the wave raises its running maximum by more than 2^4 (exactness is unaffected: P is
//    formed against the stale maximum and bf16's relative precision is scale free), exp2 arguments come
//    from one FMA (score * scale*log2e - m), no separate scale or subtract pass;
//  * backward: the 1/sqrt(d_h) factor of dS is applied once to the dQ / dK accumulators at the end.
#include <cstdint>
#include <cstdlib>

#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int KT = 128;            // rows per staged tile
constexpr int IMG = KT * 64;       // bytes per image (32 bf16 per row)
constexpr float LAZY_THR = 4.0f;   // log2 units

struct Args {
    const void *q, *k, *v, *o, *d_o;
    void *out_o, *dq, *dk, *dv;
    const float* kbias;
    float *lse2, *delta;
    int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int B, H, Lq, Lk, dh;
    float scale;
    float premul;  // != 0: q was pre-multiplied by premul = scale*log2(e) (fused into the projection GEMM epilogue)
    // key-split of the few-query (cross attention) launches: ksplit workgroups share one query tile, each takes
    // tiles_per_split key tiles; partial results meet in the fp32 workspace
    int ksplit, tiles_per_split;
    float *ws_o, *ws_ml, *ws_dq;  // [ksplit][B*Lq][H*dh] unnormalised O | [ksplit][B][H][Lq][2] (m2, l) | [B*Lq][H*dh] dQ
    const int* tile_flags;        // [B][ceil(Lk/KT)] key-tile classes of the masked fast kernels (attn_tile_flags_bf16)
    int head_xcd, nxt;            // != 0: 1-D grid of B*H*nxt workgroups with the heads dealt to the 8 XCDs (block_coords)
    int tail_last;                // != 0 (with head_xcd): every head's LAST x tile is dispatched after all the others
    int* redo;                    // [grid] written by attn_fwd_bf16_fast (1 = a row sum overflowed), read by the safe kernel behind it
    unsigned *nl2, *nd2;          // [B,H,Lq] -lse2 / -delta as (hi, lo) bf16 pairs: written by the fast dQ kernel, DMA'd by the dK/dV kernel
    // attention-probability dropout (general kernels only; see attention.hip AttnArgs): keep mask of ((b*H + h)*Lq + q)*Lk + key
    float drop_p, drop_inv;
    uint64_t drop_seed;
    int dq_rot;                   // != 0: the unmasked dQ kernel runs its rotated schedule (dq_phase; SVOL_ATTN_NO_DQ_ROT=1 clears it)
};
__device__ __forceinline__ float attn_drop(uint64_t seed, uint64_t row, int key, float p, float inv) {
    return dropout_scale(seed, row, (uint32_t)key, p, inv);
}
// the same for keys key0 (EVEN) and key0 + 1 of one row: one generator call for the pair
__device__ __forceinline__ void attn_drop2(uint32_t rowmix, int key0, uint32_t thr16, float inv, float& s0, float& s1) {
    const uint32_t bits = drop_bits(rowmix, (uint32_t)key0 >> 1);
    s0 = drop_pick(bits, 0u, thr16, inv);
    s1 = drop_pick(bits, 1u, thr16, inv);
}

typedef __attribute__((address_space(3))) h16x4* lds_bf16x4_ptr;

__device__ __forceinline__ int img_off(int row, int ch) { return row * 64 + ((ch ^ ((row >> 2) & 3)) << 4); }

struct Stage { uint4 v[2]; };

// 128 rows x 4 chunks = 512 chunks, 2 per thread
__device__ __forceinline__ void load_regs(Stage& s, const h16_t* g, int64_t ld, int row0, int limit, int dh, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i, row = c >> 2, ch = c & 3;
        s.v[i] = (row0 + row < limit && ch * 8 < dh) ? *reinterpret_cast<const uint4*>(g + (int64_t)(row0 + row) * ld + ch * 8)
                                                      : make_uint4(0, 0, 0, 0);
    }
}
__device__ __forceinline__ void store_lds(char* img, const Stage& s, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i, row = c >> 2, ch = c & 3;
        *reinterpret_cast<uint4*>(img + img_off(row, ch)) = s.v[i];
    }
}

// LDS-DMA staging (global_load_lds_dwordx4: no VGPR round trip, no ds_write): one 1-KiB piece = 16 rows x 64 B of a
// [128][32] bf16 tile straight into the XOR-swizzled image.  The LDS side of the instruction is linear (wave base + lane * 16),
// so the swizzle sits on the per-lane SOURCE address: lane i lands in (row 16*piece + i/4, slot i%4), which must hold chunk
// slot ^ ((row >> 2) & 3) = (i & 3) ^ ((i >> 4) & 3) of that row (img_off).  Rows must exist (callers: full tiles only).
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gbl_vptr;
__device__ __forceinline__ void dma_piece(char* img, const h16_t* g, int64_t ld, int row0, int piece, int lane) {
    const int row = 16 * piece + (lane >> 2);
    const int ch = (lane & 3) ^ ((lane >> 4) & 3);
    const h16_t* src = g + (int64_t)(row0 + row) * ld + ch * 8;
    __builtin_amdgcn_global_load_lds((gbl_vptr)src, (lds_vptr)(img + piece * 1024), 16, 0, 0);
}
// a whole 128-row tile by the four waves of a workgroup (two pieces each); `wave` must be wave-uniform (readfirstlane)
__device__ __forceinline__ void dma_tile(char* img, const h16_t* g, int64_t ld, int row0, int wave, int lane) {
    dma_piece(img, g, ld, row0, 2 * wave, lane);
    dma_piece(img, g, ld, row0, 2 * wave + 1, lane);
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// The same transfer issued from inline asm: hipcc models the builtin as a store to LDS and drains vmcnt in front of the NEXT LDS
// read it can see — i.e. right behind the issue, which serialises the whole load latency into every tile.  From asm the compiler
// does not know LDS is written; the caller owns the ordering (dma_wait_all + barrier before anyone reads the image).
__device__ __forceinline__ void dma_piece_async(char* img, const h16_t* g, int64_t ld, int row0, int piece, int lane) {
    const int row = 16 * piece + (lane >> 2);
    const int ch = (lane & 3) ^ ((lane >> 4) & 3);
    const h16_t* src = g + (int64_t)(row0 + row) * ld + ch * 8;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_vptr)(img + piece * 1024));
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "m0");
}
__device__ __forceinline__ void dma_tile_async(char* img, const h16_t* g, int64_t ld, int row0, int wave, int lane) {
    dma_piece_async(img, g, ld, row0, 2 * wave, lane);
    dma_piece_async(img, g, ld, row0, 2 * wave + 1, lane);
}

__device__ __forceinline__ void load_lane_block(uint4 (&out)[2], const h16_t* g, int64_t ld, int row, bool valid, int dh,
                                                int h) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int d0 = (2 * s + h) * 8;
        out[s] = (valid && d0 < dh) ? *reinterpret_cast<const uint4*>(g + (int64_t)row * ld + d0) : make_uint4(0, 0, 0, 0);
    }
}

// A operand, contraction over d: tile row `row`, k-step s covers d = 16s + 8h .. +7
__device__ __forceinline__ void read_rows(uint4 (&a)[2], const char* img, int row, int h) {
#pragma unroll
    for (int s = 0; s < 2; ++s) a[s] = *reinterpret_cast<const uint4*>(img + img_off(row, 2 * s + h));
}

// A operand, contraction over the 32 rows of sub-tile `sub` (rows presented in the order of the first
// product's accumulator registers): lane (r = lane&31 -> column d = r).  ds_read_b64_tr_b16 per 16-lane group:
// lane 4q+p supplies the address of block row q, columns 4p..4p+3; lane i receives column i of the 4 rows.
__device__ __forceinline__ void read_tr(uint4 (&a)[2], const char* img, int sub, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, hh = g >> 1;
    const int ch = 2 * (g & 1) + (p >> 1), inner = 8 * (p & 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int r1 = 32 * sub + 16 * s + 4 * hh + q;
        const h16x4 lo = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(img + img_off(r1, ch) + inner));
        const h16x4 hi = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(img + img_off(r1 + 8, ch) + inner));
        a[s] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
}

// v_max3_f32 without the canonicalising v_max_f32 x,x,x hipcc puts in front of fmaxf() on MFMA results
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// The score accumulators are first read by inline-asm v_max3_f32 (max3 above).  hipcc pads an MFMA write -> VALU read
// of the same registers with the required wait states only for instructions it can see; inline asm is opaque to
// its hazard recogniser, so without this fence the maxima were taken from accumulators still in flight whenever
// the wave was not held up by its partner: numerically harmless (the maximum is only the softmax reference point)
// but the outputs changed by an ulp from run to run.  19 wait states cover a 16-pass MFMA; the "+v" ties place the
// fence after every score MFMA and before every use.
__device__ __forceinline__ void mfma_results_ready(f32x16& a, f32x16& b, f32x16& c, f32x16& d) {
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

__device__ __forceinline__ void mfma_results_ready2(f32x16& a, f32x16& b) {
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b));
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

__device__ __forceinline__ f32x16 mma_first(const uint4 (&a)[2], const uint4 (&b)[2]) {
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < 2; ++s)
        acc = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, a[s]), __builtin_bit_cast(h16x8, b[s]), acc, 0, 0,
                                                      0);
    return acc;
}

__device__ __forceinline__ void mma_second(f32x16& acc, const uint4 (&a)[2], const f32x16& x) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        h16x8 b;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {   // packed conversion (v_cvt_pk_*): the loop is issue-bound, one instruction per pair
            const h16x2 pr = cvt_pk_h16(x[8 * s + j], x[8 * s + j + 1]);
            b[j] = pr[0];
            b[j + 1] = pr[1];
        }
        acc = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, a[s]), b, acc, 0, 0, 0);
    }
}

__device__ __forceinline__ void store_acc(const f32x16& acc, h16_t* out, int64_t ld, int row, bool valid, int dh, int h,
                                          float mul) {
    if (!valid) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int d0 = 8 * g + 4 * h;
        if (d0 < dh) {
            h16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (h16_t)(acc[4 * g + e] * mul);
            *reinterpret_cast<h16x4*>(out + (int64_t)row * ld + d0) = v;
        }
    }
}

__device__ __forceinline__ void stage_bias(float* sb, const float* kb, int row0, int limit, int tid) {
    if (tid < KT) {
        const int key = row0 + tid;
        sb[tid] = key < limit ? (kb ? kb[key] * LOG2E : 0.f) : -INFINITY;
    }
}

// ---------------------------------------------------------------------------
// forward: lane = queries (32 per wave, 128 per workgroup), tiles over keys
template <bool MASKED>
__global__ __launch_bounds__(256, 2) void attn_fwd_bf16(Args p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * IMG + 2 * KT * 4];
    char* sK = smem;                 // [2][IMG]
    char* sV = smem + 2 * IMG;       // [2][IMG]
    float* sB = reinterpret_cast<float*>(smem + 4 * IMG);  // [2][KT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, hh = blockIdx.y;
    const int qt = blockIdx.x / p.ksplit, sp = blockIdx.x % p.ksplit;
    const int qrow = qt * 128 + wave * 32 + r;
    const bool qvalid = qrow < p.Lq;
    const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * p.dh;
    const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * p.dh;
    const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * p.dh;
    const float* kb = p.kbias ? p.kbias + (int64_t)b * p.Lk : nullptr;
    const float c = p.premul != 0.f ? 1.f : p.scale * LOG2E;

    uint4 qb[2];
    load_lane_block(qb, Q, p.ldq, qrow, qvalid, p.dh, h);

    // running max m (raw-score domain when !MASKED, scaled log2 domain when MASKED), partial row sum l
    float m = -INFINITY, l = 0.f;
    f32x16 O = zero16();

    const int t0 = sp * p.tiles_per_split;
    const int nt = min((p.Lk + KT - 1) / KT, t0 + p.tiles_per_split);  // this workgroup's key tiles: [t0, nt)
    Stage sk, sv;
    load_regs(sk, K, p.ldk, t0 * KT, p.Lk, p.dh, tid);
    load_regs(sv, V, p.ldv, t0 * KT, p.Lk, p.dh, tid);
    store_lds(sK, sk, tid);
    store_lds(sV, sv, tid);
    if (MASKED) stage_bias(sB, kb, t0 * KT, p.Lk, tid);
    __syncthreads();

    for (int t = t0; t < nt; ++t) {
        const int cur = (t - t0) & 1;
        if (t + 1 < nt) {
            load_regs(sk, K, p.ldk, (t + 1) * KT, p.Lk, p.dh, tid);
            load_regs(sv, V, p.ldv, (t + 1) * KT, p.Lk, p.dh, tid);
        }
        const char* kimg = sK + cur * IMG;
        const char* vimg = sV + cur * IMG;
        const float* bias = sB + cur * KT;
        // ---- scores of the whole 128-key tile first (8 MFMAs back to back), softmax once per tile:
        // four independent 32-key chains give the VALU block ILP and amortise the max exchange / rescale
        f32x16 S[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            uint4 ka[2];
            read_rows(ka, kimg, sub * 32 + r, h);
            S[sub] = mma_first(ka, qb);
        }
        mfma_results_ready(S[0], S[1], S[2], S[3]);
        if (MASKED) {
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + sub * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) S[sub][4 * g + e] = S[sub][4 * g + e] * c + bb[e];
                }
        }
        float ml[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            float mm = max3(S[sub][0], S[sub][1], S[sub][2]);
#pragma unroll
            for (int i = 3; i < 15; i += 2) mm = max3(mm, S[sub][i], S[sub][i + 1]);
            ml[sub] = max3(mm, S[sub][15], mm);
        }
        float mloc = max3(ml[0], ml[1], fmaxf(ml[2], ml[3]));
        {   // other half-wave's maximum for the same query: v_permlane32_swap (VALU, no LDS round trip)
            const HalfPair sw = swap_halves(__builtin_bit_cast(unsigned, mloc));
            mloc = fmaxf(__builtin_bit_cast(float, sw.lo), __builtin_bit_cast(float, sw.hi));
        }
        const float thr = MASKED ? LAZY_THR : LAZY_THR / c;
        if (__any(mloc > m + thr)) {  // wave-uniform: rescale only when some query's maximum really moved
            const float m_new = fmaxf(m, mloc);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(MASKED ? (m - m_use) : (m - m_use) * c);
            l *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) O[i] *= alpha;
            m = m_new;
        }
        const float m_use = (m == -INFINITY) ? 0.f : m;
        const float mc = MASKED ? -m_use : -m_use * c;
        float ls[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                S[sub][i] = __builtin_amdgcn_exp2f(MASKED ? S[sub][i] + mc : __builtin_fmaf(S[sub][i], c, mc));
                ls[sub] += S[sub][i];
            }
        l += (ls[0] + ls[1]) + (ls[2] + ls[3]);
        if (p.drop_p > 0.f) {   // row sums stay those of the undropped softmax; only what feeds P V is masked
            const uint32_t rm = drop_row(drop_seed32(p.drop_seed), (((uint64_t)b * p.H + hh) * p.Lq + (qvalid ? qrow : 0)));
            const uint32_t thr = drop_thr16(p.drop_p);
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {   // (the lane's keys 8g + 4h + e: pairs share one generator call)
                        float s0, s1;
                        attn_drop2(rm, t * KT + sub * 32 + 8 * g + 4 * h + e, thr, p.drop_inv, s0, s1);
                        S[sub][4 * g + e] *= s0;
                        S[sub][4 * g + e + 1] *= s1;
                    }
        }
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            uint4 va[2];
            read_tr(va, vimg, sub, lane);
            mma_second(O, va, S[sub]);
        }
        if (t + 1 < nt) {
            store_lds(sK + (cur ^ 1) * IMG, sk, tid);
            store_lds(sV + (cur ^ 1) * IMG, sv, tid);
            if (MASKED) stage_bias(sB + (cur ^ 1) * KT, kb, (t + 1) * KT, p.Lk, tid);
        }
        __syncthreads();
    }
    const float lt = l + __shfl_xor(l, 32, 64);
    const float m2 = MASKED ? m : m * c;  // scaled log2 domain
    if (p.ksplit > 1) {  // partial: unnormalised O and (m2, l) to the workspace, attn_combine_bf16 finishes
        if (qvalid) {
            float* wo = p.ws_o + (((int64_t)sp * p.B + b) * p.Lq + qrow) * (p.H * p.dh) + hh * p.dh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 8 * g + 4 * h;
                if (d0 < p.dh) *reinterpret_cast<f32x4*>(wo + d0) = f32x4{O[4 * g], O[4 * g + 1], O[4 * g + 2], O[4 * g + 3]};
            }
            if (h == 0) {
                float* wm = p.ws_ml + ((((int64_t)sp * p.B + b) * p.H + hh) * p.Lq + qrow) * 2;
                wm[0] = m2;
                wm[1] = lt;
            }
        }
        return;
    }
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    h16_t* Oo = reinterpret_cast<h16_t*>(p.out_o) + (int64_t)b * p.Lq * p.ldo + hh * p.dh;
    store_acc(O, Oo, p.ldo, qrow, qvalid, p.dh, h, inv);
    if (qvalid && h == 0) p.lse2[((int64_t)b * p.H + hh) * p.Lq + qrow] = m2 + __builtin_amdgcn_logf(lt);
}

// merge of the key-split partials: O = sum_i 2^(m_i - M) O_i / sum_i 2^(m_i - M) l_i.  One thread per (row, head).
__global__ void attn_combine_bf16(Args p) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)p.B * p.Lq * p.H;
    if (idx >= total) return;
    const int hh = (int)(idx % p.H);
    const int64_t row = idx / p.H;
    const int b = (int)(row / p.Lq), q = (int)(row % p.Lq);
    float M = -INFINITY;
    for (int i = 0; i < p.ksplit; ++i) M = fmaxf(M, p.ws_ml[((((int64_t)i * p.B + b) * p.H + hh) * p.Lq + q) * 2]);
    float L = 0.f, acc[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) acc[d] = 0.f;
    for (int i = 0; i < p.ksplit; ++i) {
        const float* ml = p.ws_ml + ((((int64_t)i * p.B + b) * p.H + hh) * p.Lq + q) * 2;
        if (ml[0] == -INFINITY) continue;  // a split that saw only masked keys (or none)
        const float w = __builtin_amdgcn_exp2f(ml[0] - M);
        L += w * ml[1];
        const float* po = p.ws_o + (((int64_t)i * p.B + b) * p.Lq + q) * (p.H * p.dh) + hh * p.dh;
#pragma unroll
        for (int d = 0; d < 32; d += 4) {
            if (d < p.dh) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(po + d);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[d + e] += w * t[e];
            }
        }
    }
    const float inv = L > 0.f ? 1.f / L : 0.f;
    h16_t* o = reinterpret_cast<h16_t*>(p.out_o) + row * p.ldo + hh * p.dh;
#pragma unroll
    for (int d = 0; d < 32; d += 4) {
        if (d < p.dh) {
            h16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (h16_t)(acc[d + e] * inv);
            *reinterpret_cast<h16x4*>(o + d) = v;
        }
    }
    p.lse2[((int64_t)b * p.H + hh) * p.Lq + q] = M + __builtin_amdgcn_logf(L);
}

__global__ void attn_delta_bf16(Args p) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)p.B * p.Lq * p.H;
    if (idx >= total) return;
    const int hh = (int)(idx % p.H);
    const int64_t row = idx / p.H;
    const int b = (int)(row / p.Lq), q = (int)(row % p.Lq);
    const h16_t* o = reinterpret_cast<const h16_t*>(p.o) + row * p.ldo + hh * p.dh;
    const h16_t* d = reinterpret_cast<const h16_t*>(p.d_o) + row * p.lddo + hh * p.dh;
    float s = 0.f;
    for (int i = 0; i < p.dh; i += 8) {
        const h16x8 a = *reinterpret_cast<const h16x8*>(o + i), c = *reinterpret_cast<const h16x8*>(d + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)c[e];
    }
    p.delta[((int64_t)b * p.H + hh) * p.Lq + q] = s;
    if (p.ksplit > 1) {  // the key-split dQ pass accumulates with atomics: zero its fp32 target here
        float* z = p.ws_dq + row * (p.H * p.dh) + hh * p.dh;
        for (int i = 0; i < p.dh; i += 4) *reinterpret_cast<f32x4*>(z + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// dq (bf16) = scale * ws_dq after the key-split dQ pass.  One thread per (row, head).
__global__ void attn_dq_finish_bf16(Args p) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)p.B * p.Lq * p.H;
    if (idx >= total) return;
    const int hh = (int)(idx % p.H);
    const int64_t row = idx / p.H;
    const float* z = p.ws_dq + row * (p.H * p.dh) + hh * p.dh;
    h16_t* o = reinterpret_cast<h16_t*>(p.dq) + row * p.lddq + hh * p.dh;
    for (int i = 0; i < p.dh; i += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(z + i);
        h16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (h16_t)(t[e] * p.scale);
        *reinterpret_cast<h16x4*>(o + i) = v;
    }
}

// ---------------------------------------------------------------------------
// dQ: lane = queries, tiles over keys.  dQ^T += K^T (P * (dP - delta)); scaled by 1/sqrt(dh) at the end.
template <bool MASKED>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_bf16(Args p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * IMG + 2 * KT * 4];
    char* sK = smem;
    char* sV = smem + 2 * IMG;
    float* sB = reinterpret_cast<float*>(smem + 4 * IMG);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, hh = blockIdx.y;
    const int qt = blockIdx.x / p.ksplit, sp = blockIdx.x % p.ksplit;
    const int qrow = qt * 128 + wave * 32 + r;
    const bool qvalid = qrow < p.Lq;
    const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * p.dh;
    const h16_t* dO = reinterpret_cast<const h16_t*>(p.d_o) + (int64_t)b * p.Lq * p.lddo + hh * p.dh;
    const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * p.dh;
    const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * p.dh;
    const float* kb = p.kbias ? p.kbias + (int64_t)b * p.Lk : nullptr;
    const float c = p.premul != 0.f ? 1.f : p.scale * LOG2E;

    uint4 qb[2], dob[2];
    load_lane_block(qb, Q, p.ldq, qrow, qvalid, p.dh, h);
    load_lane_block(dob, dO, p.lddo, qrow, qvalid, p.dh, h);
    const int64_t sidx = ((int64_t)b * p.H + hh) * p.Lq + qrow;
    const float nlse = qvalid ? -p.lse2[sidx] : -INFINITY;  // exp2(x - inf) = 0 for rows past Lq
    const float dl = qvalid ? p.delta[sidx] : 0.f;

    f32x16 dQ = zero16();
    const int t0 = sp * p.tiles_per_split;
    const int nt = min((p.Lk + KT - 1) / KT, t0 + p.tiles_per_split);  // this workgroup's key tiles: [t0, nt)
    Stage sk, sv;
    load_regs(sk, K, p.ldk, t0 * KT, p.Lk, p.dh, tid);
    load_regs(sv, V, p.ldv, t0 * KT, p.Lk, p.dh, tid);
    store_lds(sK, sk, tid);
    store_lds(sV, sv, tid);
    if (MASKED) stage_bias(sB, kb, t0 * KT, p.Lk, tid);
    __syncthreads();

    for (int t = t0; t < nt; ++t) {
        const int cur = (t - t0) & 1;
        if (t + 1 < nt) {
            load_regs(sk, K, p.ldk, (t + 1) * KT, p.Lk, p.dh, tid);
            load_regs(sv, V, p.ldv, (t + 1) * KT, p.Lk, p.dh, tid);
        }
        const char* kimg = sK + cur * IMG;
        const char* vimg = sV + cur * IMG;
        const float* bias = sB + cur * KT;
#pragma unroll 2
        for (int sub = 0; sub < 4; ++sub) {
            uint4 a[2];
            read_rows(a, kimg, sub * 32 + r, h);
            f32x16 S = mma_first(a, qb);
            read_rows(a, vimg, sub * 32 + r, h);
            f32x16 dP = mma_first(a, dob);
            if (p.drop_p > 0.f) {
                const uint32_t rm = drop_row(drop_seed32(p.drop_seed), (((uint64_t)b * p.H + hh) * p.Lq + (qvalid ? qrow : 0)));
                const uint32_t thr = drop_thr16(p.drop_p);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        float s0, s1;
                        attn_drop2(rm, t * KT + sub * 32 + 8 * g + 4 * h + e, thr, p.drop_inv, s0, s1);
                        dP[4 * g + e] *= s0;
                        dP[4 * g + e + 1] *= s1;
                    }
            }
            if (MASKED) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + sub * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(S[4 * g + e], c, bb[e] + nlse));
                        S[4 * g + e] = pe * (dP[4 * g + e] - dl);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(S[i], c, nlse));
                    S[i] = pe * (dP[i] - dl);
                }
            }
            read_tr(a, kimg, sub, lane);
            mma_second(dQ, a, S);
        }
        if (t + 1 < nt) {
            store_lds(sK + (cur ^ 1) * IMG, sk, tid);
            store_lds(sV + (cur ^ 1) * IMG, sv, tid);
            if (MASKED) stage_bias(sB + (cur ^ 1) * KT, kb, (t + 1) * KT, p.Lk, tid);
        }
        __syncthreads();
    }
    if (p.ksplit > 1) {  // partial over this key range: fp32 atomics, attn_dq_finish_bf16 scales and rounds
        if (qvalid) {
            float* z = p.ws_dq + ((int64_t)b * p.Lq + qrow) * (p.H * p.dh) + hh * p.dh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 8 * g + 4 * h;
                if (d0 < p.dh) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(z + d0 + e, dQ[4 * g + e]);
                }
            }
        }
        return;
    }
    h16_t* dQo = reinterpret_cast<h16_t*>(p.dq) + (int64_t)b * p.Lq * p.lddq + hh * p.dh;
    store_acc(dQ, dQo, p.lddq, qrow, qvalid, p.dh, h, p.scale);
}

// ---------------------------------------------------------------------------
// dK, dV: lane = keys, tiles over queries.
template <bool KBIAS>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_bf16(Args p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * IMG + 4 * KT * 4];
    char* sQ = smem;
    char* sdO = smem + 2 * IMG;
    float* sL = reinterpret_cast<float*>(smem + 4 * IMG);  // [2][KT] -lse2
    float* sD = sL + 2 * KT;                               // [2][KT] delta

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, hh = blockIdx.y;
    const int krow = blockIdx.x * 128 + wave * 32 + r;
    const bool kvalid = krow < p.Lk;
    const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * p.dh;
    const h16_t* dO = reinterpret_cast<const h16_t*>(p.d_o) + (int64_t)b * p.Lq * p.lddo + hh * p.dh;
    const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * p.dh;
    const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * p.dh;
    const float c = p.premul != 0.f ? 1.f : p.scale * LOG2E;
    const float kbl = kvalid ? (p.kbias ? p.kbias[(int64_t)b * p.Lk + krow] * LOG2E : 0.f) : -INFINITY;
    const float* lse_g = p.lse2 + ((int64_t)b * p.H + hh) * p.Lq;
    const float* dl_g = p.delta + ((int64_t)b * p.H + hh) * p.Lq;

    uint4 kbk[2], vbk[2];
    load_lane_block(kbk, K, p.ldk, krow, kvalid, p.dh, h);
    load_lane_block(vbk, V, p.ldv, krow, kvalid, p.dh, h);

    f32x16 dK = zero16(), dV = zero16();
    const int nt = (p.Lq + KT - 1) / KT;
    Stage sq, sdo;
    float rl = 0.f, rd = 0.f;
    auto load_stats = [&](int row0) {
        if (tid < KT) {
            const int qi = row0 + tid;
            rl = qi < p.Lq ? -lse_g[qi] : -INFINITY;  // exp2(x - inf) = 0 for rows past Lq
            rd = qi < p.Lq ? dl_g[qi] : 0.f;
        }
    };
    auto store_stats = [&](int buf) {
        if (tid < KT) { sL[buf * KT + tid] = rl; sD[buf * KT + tid] = rd; }
    };
    load_regs(sq, Q, p.ldq, 0, p.Lq, p.dh, tid);
    load_regs(sdo, dO, p.lddo, 0, p.Lq, p.dh, tid);
    load_stats(0);
    store_lds(sQ, sq, tid);
    store_lds(sdO, sdo, tid);
    store_stats(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            load_regs(sq, Q, p.ldq, (t + 1) * KT, p.Lq, p.dh, tid);
            load_regs(sdo, dO, p.lddo, (t + 1) * KT, p.Lq, p.dh, tid);
            load_stats((t + 1) * KT);
        }
        const char* qimg = sQ + cur * IMG;
        const char* doimg = sdO + cur * IMG;
        const float* nl = sL + cur * KT;
        const float* dd = sD + cur * KT;
#pragma unroll 1
        for (int sub = 0; sub < 4; ++sub) {
            uint4 a[2];
            read_rows(a, qimg, sub * 32 + r, h);
            f32x16 S = mma_first(a, kbk);  // S[q][key]
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ls = *reinterpret_cast<const f32x4*>(nl + sub * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    S[4 * g + e] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[4 * g + e], c, KBIAS ? kbl + ls[e] : ls[e]));
            }
            read_tr(a, doimg, sub, lane);
            f32x16 MS;   // keep-mask scales of this block
#pragma unroll
            for (int i = 0; i < 16; ++i) MS[i] = 1.f;
            if (p.drop_p > 0.f) {
                // A lane holds ONE key and 16 query rows here; the generator yields the bits of a key PAIR of one row, and the pair's
                // other key sits in the neighbouring lane (krow = ... + lane % 32): of two consecutive rows the even lane draws the
                // first, the odd lane the second, one DPP swap hands each the other's word (a quarter of the integer multiplies of
                // the per-element form: 20 -> ~17 ms of mask generation per enc/dec training step).
                const uint32_t s0 = drop_seed32(p.drop_seed), thr = drop_thr16(p.drop_p), odd = (uint32_t)lane & 1u;
                const uint64_t rbase = ((uint64_t)b * p.H + hh) * p.Lq;
                const bool rows32 = rbase + (uint64_t)p.Lq <= 0xffffffffull;   // (uniform) the row index fits 32 bits: one multiply per row
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const int q0 = t * KT + sub * 32 + 8 * g + 4 * h + e;
                        const int qm = min(q0 + (int)odd, p.Lq - 1);
                        const uint32_t rm = rows32 ? (s0 ^ (((uint32_t)rbase + (uint32_t)qm) * 0x9E3779B1u)) : drop_row(s0, rbase + (uint64_t)qm);
                        const uint32_t mine = drop_bits(rm, (uint32_t)krow >> 1);
                        const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xf, 0xf, true);   // lane ^ 1
                        MS[4 * g + e] = drop_pick(odd ? other : mine, odd, thr, p.drop_inv);
                        MS[4 * g + e + 1] = drop_pick(odd ? mine : other, odd, thr, p.drop_inv);
                    }
                f32x16 Pd = S;
#pragma unroll
                for (int i = 0; i < 16; ++i) Pd[i] *= MS[i];
                mma_second(dV, a, Pd);  // dV^T += dO^T (P . mask)
            } else {
                mma_second(dV, a, S);  // dV^T += dO^T P
            }
            read_rows(a, doimg, sub * 32 + r, h);
            f32x16 dP = mma_first(a, vbk);  // dP[q][key] = dO V^T
            if (p.drop_p > 0.f) {
#pragma unroll
                for (int i = 0; i < 16; ++i) dP[i] *= MS[i];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 dl = *reinterpret_cast<const f32x4*>(dd + sub * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) S[4 * g + e] = S[4 * g + e] * (dP[4 * g + e] - dl[e]);
            }
            read_tr(a, qimg, sub, lane);
            mma_second(dK, a, S);  // dK^T += Q^T dS
        }
        if (t + 1 < nt) {
            store_lds(sQ + (cur ^ 1) * IMG, sq, tid);
            store_lds(sdO + (cur ^ 1) * IMG, sdo, tid);
            store_stats(cur ^ 1);
        }
        __syncthreads();
    }
    h16_t* dKo = reinterpret_cast<h16_t*>(p.dk) + (int64_t)b * p.Lk * p.lddk + hh * p.dh;
    h16_t* dVo = reinterpret_cast<h16_t*>(p.dv) + (int64_t)b * p.Lk * p.lddv + hh * p.dh;
    store_acc(dK, dKo, p.lddk, krow, kvalid, p.dh, h, p.premul != 0.f ? p.scale / p.premul : p.scale);
    store_acc(dV, dVo, p.lddv, krow, kvalid, p.dh, h, 1.f);
}

// ===========================================================================================
// "PRE" kernels: the big unmasked self-attention with q pre-multiplied by scale*log2(e) (fused into
// the projection GEMM epilogue, so it costs no extra rounding).  Every per-score VALU operation that
// is not an exp, a product or a conversion is moved into the MFMA: the row constants (-running max,
// -lse, -delta) are the INITIAL ACCUMULATORS of the score / dP products (guide: "row constants as the
// initial accumulator"), so   p = exp2(acc)   and   dS = p * acc2   come straight out of the matrix pipe.
// Per 32x32 block: forward 8 max3 + 16 exp + 16 add + 8 cvt (was + 16 fma); dQ 16 exp + 16 mul + 8 cvt
// (was + 16 fma + 16 sub); dK/dV 16 exp + 16 mul + 16 cvt (was + 16 fma + 16 sub).
__device__ __forceinline__ f32x16 splat16(float x) {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = x;
    return z;
}
// D = A*B + C with D and C in DIFFERENT registers, so a loop-invariant C (the row constants) is never copied:
// through the builtin hipcc picks the tied ("mac") form and re-materialises C with 16 v_mov per product.
// s_nop 1: VALU-written operand -> MFMA read wait states (nothing inside an asm statement is padded for us);
// the result feeds only the next MFMA's C (accumulate chain, no wait states needed).
__device__ __forceinline__ f32x16 mma_first_c(const uint4 (&a)[2], const uint4 (&b)[2], const f32x16& c0) {
    f32x16 acc;
    asm("s_nop 1\n\tv_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, %3"
        : "=&v"(acc)
        : "v"(__builtin_bit_cast(h16x8, a[0])), "v"(__builtin_bit_cast(h16x8, b[0])), "v"(c0));
    return SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, a[1]), __builtin_bit_cast(h16x8, b[1]), acc, 0, 0,
                                                   0);
}
// accumulator initialised from a per-ROW vector in LDS (row constants of the reg-side tile)
__device__ __forceinline__ f32x16 rows16(const float* v, int sub, int h) {
    f32x16 z;

#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(v + sub * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) z[4 * g + e] = t[e];
    }
    return z;
}

// Workgroup -> (x tile, head, batch).  Every workgroup of one (batch, head) streams that head's whole K / V (or Q / dO): 0.8 MB
// at L = 6272.  Hardware deals consecutive workgroup ids round-robin over the 8 XCDs, so with the plain 3-D grid the tiles of a
// head are spread over all eight L2s and every L2 sees every head in flight (measured: 0.56 GB memory-side reads per launch for
// 0.05 GB of K / V).  With B*H a multiple of 8 the launchers use a 1-D grid instead: id % 8 = XCD, and each XCD walks its own
// heads tile by tile, so a head's K / V is fetched into ONE L2.
__device__ __forceinline__ void block_coords(const Args& p, int& xt, int& hh, int& b) {
    if (p.head_xcd) {
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        int hl;
        const int two = p.head_xcd == 2 ? 2 : 1;   // 2: the heads of a pair alternate tile by tile (see below)
        if (p.tail_last) {  // the short last tiles (dQ pass: half a workgroup of rows) fill the launch's last, partial round
            const int nfull = p.nxt - 1, hpx = (p.B * p.H) >> 3;
            if (slot < hpx * nfull) {
                const int g = slot / (two * nfull), r = slot - g * two * nfull;
                hl = g * two + (two == 2 ? (r & 1) : 0);
                xt = two == 2 ? (r >> 1) : r;
            } else { hl = slot - hpx * nfull; xt = nfull; }
        } else {
            const int g = slot / (two * p.nxt), r = slot - g * two * p.nxt;
            hl = g * two + (two == 2 ? (r & 1) : 0);
            xt = two == 2 ? (r >> 1) : r;
        }
        // head_xcd == 2: the two heads that share every 128-byte line of the [.., H, 32]-interleaved operands (64 bytes each) are
        // walked by the SAME XCD, the same x tile of both in consecutive workgroups: a line is fetched once, into one L2
        const int head = p.head_xcd == 2 ? (((hl >> 1) * 8 + xcd) * 2 + (hl & 1)) : hl * 8 + xcd;
        hh = head % p.H;
        b = head / p.H;
    } else {
        xt = blockIdx.x;
        hh = blockIdx.y;
        b = blockIdx.z;
    }
}

// x as two bf16 (hi = rn(x), lo = rn(x - hi)) packed in one dword: |x - hi - lo| <= 2^-17 |x|
__device__ __forceinline__ unsigned split_bf16x2(float x) {
    const h16_t hi = (h16_t)x;
    const float rem = x - (float)hi;
    const h16_t lo = (h16_t)rem;
    return (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
}

// additive key bias in the log2 domain; keys past Lk are masked
__device__ __forceinline__ float key_bias_log2(const float* kb, int key, int Lk) {
    return key < Lk ? (kb ? kb[key] * LOG2E : 0.f) : -INFINITY;
}
// S (keys x queries, swapped product) += bias[key]: one more 16-deep MFMA contracts [bias_hi, bias_lo, 0...] (this lane's key) with
// [1, 1, 0...] — the mixed tiles of the masked kernels pay one matrix instruction per 32x32 block and no vector work
__device__ __forceinline__ f32x16 add_key_bias(const f32x16& S, const float* kb, int key, int Lk, int h) {
    const float bv = key_bias_log2(kb, key, Lk);
    const unsigned pair = bv == -INFINITY ? SVOL_H16_NINF_LO : split_bf16x2(bv);  // (-inf, 0): x - hi would be NaN
    const uint4 a = h == 0 ? make_uint4(pair, 0u, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
    const uint4 o = h == 0 ? make_uint4(SVOL_H16_ONE2, 0u, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
    return SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, o), S, 0, 0, 0);
}
// flags[b][t] = 0 (the 128 keys of tile t all have zero bias), 2 (all at -inf or past Lk), 1 (anything else)
__global__ __launch_bounds__(64) void attn_tile_flags_bf16(const float* __restrict__ kbias, int Lk, int nt, int* __restrict__ flags) {
    const int t = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const float* kb = kbias ? kbias + (int64_t)b * Lk : nullptr;
    const float b0 = key_bias_log2(kb, t * KT + lane, Lk), b1 = key_bias_log2(kb, t * KT + 64 + lane, Lk);
    int flag = 0;
    if (__any(b0 != 0.f || b1 != 0.f)) flag = __all(b0 == -INFINITY && b1 == -INFINITY) ? 2 : 1;
    if (lane == 0) flags[(int64_t)b * nt + t] = flag;
}

// MASKED: additive key bias and / or a key count that is not a multiple of the tile.  Every 128-key tile is classified once per
// workgroup pass (two bias loads per lane, issued one tile ahead, and two ballots): all-zero bias -> the unmasked code path;
// every key at -inf (padding, or past Lk) -> the tile is skipped outright; anything else -> the bias is staged in LDS and
// added to the scores (16 adds per 32x32 block, only in the tiles that straddle a mask boundary).
template <bool MASKED>
__device__ __forceinline__ void attn_fwd_pre_body(const Args p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * IMG];
    char* sK = smem;
    char* sV = smem + 2 * IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int xt, hh, b;
    block_coords(p, xt, hh, b);
    const int qrow = xt * 128 + wave * 32 + r;
    const bool qvalid = qrow < p.Lq;
    const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * p.dh;
    const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * p.dh;
    const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * p.dh;
    const float* kb = (MASKED && p.kbias) ? p.kbias + (int64_t)b * p.Lk : nullptr;

    uint4 qb[2];
    load_lane_block(qb, Q, p.ldq, qrow, qvalid, p.dh, h);
    float m = 0.f, l = 0.f;       // m: running reference (log2 domain), scores enter the softmax as s - m
    f32x16 O = zero16();
    f32x16 Cm = zero16();         // -m in every accumulator register of this lane's query
    bool started = false;         // the first tile that is not skipped anchors the reference

    const int nt = MASKED ? (p.Lk + KT - 1) / KT : p.Lk / KT;  // unmasked: the launcher guarantees Lk % KT == 0
    Stage sk, sv;
    load_regs(sk, K, p.ldk, 0, p.Lk, p.dh, tid);
    load_regs(sv, V, p.ldv, 0, p.Lk, p.dh, tid);
    const int* fl = MASKED ? p.tile_flags + (int64_t)b * nt : nullptr;
    unsigned long long mixed = 0, dead = 0;
    store_lds(sK, sk, tid);
    store_lds(sV, sv, tid);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            load_regs(sk, K, p.ldk, (t + 1) * KT, p.Lk, p.dh, tid);
            load_regs(sv, V, p.ldv, (t + 1) * KT, p.Lk, p.dh, tid);
        }
        // tile classes as two 64-bit wave-uniform masks (SGPR pairs), refreshed every 64 tiles with one load per lane: a scalar
        // load per tile would put an s_waitcnt lgkmcnt(0) — which also drains the LDS reads in flight — at the top of every tile
        if (MASKED && (t & 63) == 0) {
            const int f = (t + lane < nt) ? fl[t + lane] : 2;
            mixed = __ballot(f == 1);
            dead = __ballot(f == 2);
        }
        const int flag = MASKED ? (int)((mixed >> (t & 63)) & 1) + 2 * (int)((dead >> (t & 63)) & 1) : 0;  // 0 plain, 1 mixed, 2 skip
        if (!MASKED || flag != 2) {
            const char* kimg = sK + cur * IMG;
            const char* vimg = sV + cur * IMG;
            f32x16 S[4];
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                uint4 ka[2];
                read_rows(ka, kimg, sub * 32 + r, h);
                S[sub] = mma_first_c(ka, qb, Cm);  // = score - m
            }
            if (MASKED && flag == 1) {
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) S[sub] = add_key_bias(S[sub], kb, t * KT + sub * 32 + r, p.Lk, h);
            }
            mfma_results_ready(S[0], S[1], S[2], S[3]);  // (after the bias products: max3 below is inline asm)
            float ml[4];
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                float mm = max3(S[sub][0], S[sub][1], S[sub][2]);
#pragma unroll
                for (int i = 3; i < 15; i += 2) mm = max3(mm, S[sub][i], S[sub][i + 1]);
                ml[sub] = max3(mm, S[sub][15], mm);
            }
            float mloc = max3(ml[0], ml[1], fmaxf(ml[2], ml[3]));
            {
                const HalfPair sw = swap_halves(__builtin_bit_cast(unsigned, mloc));
                mloc = fmaxf(__builtin_bit_cast(float, sw.lo), __builtin_bit_cast(float, sw.hi));
            }
            const bool first = MASKED ? !started : (t == 0);
            if (first || __any(mloc > LAZY_THR)) {  // wave-uniform
                // first tile: anchor the reference at this tile's maximum (may move down); later: only upward moves
                const float dm = first ? mloc : fmaxf(mloc, 0.f);
                if (!first) {
                    const float alpha = __builtin_amdgcn_exp2f(-dm);
                    l *= alpha;
#pragma unroll
                    for (int i = 0; i < 16; ++i) O[i] *= alpha;
                }
                m += dm;
                Cm = splat16(-m);
#pragma unroll
                for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                    for (int i = 0; i < 16; ++i) S[sub][i] -= dm;
            }
            started = true;
            float ls[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    S[sub][i] = __builtin_amdgcn_exp2f(S[sub][i]);
                    ls[sub] += S[sub][i];
                }
            l += (ls[0] + ls[1]) + (ls[2] + ls[3]);
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                uint4 va[2];
                read_tr(va, vimg, sub, lane);
                mma_second(O, va, S[sub]);
            }
        }
        if (t + 1 < nt) {
            store_lds(sK + (cur ^ 1) * IMG, sk, tid);
            store_lds(sV + (cur ^ 1) * IMG, sv, tid);
        }
        __syncthreads();
    }
    const float lt = l + __shfl_xor(l, 32, 64);
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    h16_t* Oo = reinterpret_cast<h16_t*>(p.out_o) + (int64_t)b * p.Lq * p.ldo + hh * p.dh;
    store_acc(O, Oo, p.ldo, qrow, qvalid, p.dh, h, inv);
    if (qvalid && h == 0) p.lse2[((int64_t)b * p.H + hh) * p.Lq + qrow] = m + __builtin_amdgcn_logf(lt);
}
// The unmasked kernel is kept as its own function: instantiating the template above with MASKED = false compiles to 174
// VGPRs (2 waves per SIMD) instead of this body's 163 (3 waves), 4 % slower.
__global__ __launch_bounds__(256, 2) void attn_fwd_bf16_pre(Args p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * IMG];
    // behind attn_fwd_bf16_fast: only the workgroups whose rows overflowed there run (normally none)
    if (p.redo && !p.redo[blockIdx.x]) return;
    char* sK = smem;
    char* sV = smem + 2 * IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int xt, hh, b;
    block_coords(p, xt, hh, b);
    const int qrow = xt * 128 + wave * 32 + r;
    const bool qvalid = qrow < p.Lq;
    const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * p.dh;
    const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * p.dh;
    const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * p.dh;

    uint4 qb[2];
    load_lane_block(qb, Q, p.ldq, qrow, qvalid, p.dh, h);
    float m = 0.f, l = 0.f;       // m: running reference (log2 domain), scores enter the softmax as s - m
    f32x16 O = zero16();
    f32x16 Cm = zero16();         // -m in every accumulator register of this lane's query

    const int nt = p.Lk / KT;     // launcher guarantees Lk % KT == 0
    Stage sk, sv;
    load_regs(sk, K, p.ldk, 0, p.Lk, p.dh, tid);
    load_regs(sv, V, p.ldv, 0, p.Lk, p.dh, tid);
    store_lds(sK, sk, tid);
    store_lds(sV, sv, tid);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            load_regs(sk, K, p.ldk, (t + 1) * KT, p.Lk, p.dh, tid);
            load_regs(sv, V, p.ldv, (t + 1) * KT, p.Lk, p.dh, tid);
        }
        const char* kimg = sK + cur * IMG;
        const char* vimg = sV + cur * IMG;
        f32x16 S[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            uint4 ka[2];
            read_rows(ka, kimg, sub * 32 + r, h);
            S[sub] = mma_first_c(ka, qb, Cm);  // = score - m
        }
        mfma_results_ready(S[0], S[1], S[2], S[3]);
        float ml[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            float mm = max3(S[sub][0], S[sub][1], S[sub][2]);
#pragma unroll
            for (int i = 3; i < 15; i += 2) mm = max3(mm, S[sub][i], S[sub][i + 1]);
            ml[sub] = max3(mm, S[sub][15], mm);
        }
        float mloc = max3(ml[0], ml[1], fmaxf(ml[2], ml[3]));
        {
            const HalfPair sw = swap_halves(__builtin_bit_cast(unsigned, mloc));
            mloc = fmaxf(__builtin_bit_cast(float, sw.lo), __builtin_bit_cast(float, sw.hi));
        }
        if (t == 0 || __any(mloc > LAZY_THR)) {  // wave-uniform
            // first tile: anchor the reference at this tile's maximum (may move down); later: only upward moves
            const float dm = (t == 0) ? mloc : fmaxf(mloc, 0.f);
            if (t != 0) {
                const float alpha = __builtin_amdgcn_exp2f(-dm);
                l *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) O[i] *= alpha;
            }
            m += dm;
            Cm = splat16(-m);
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int i = 0; i < 16; ++i) S[sub][i] -= dm;
        }
        float ls[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                S[sub][i] = __builtin_amdgcn_exp2f(S[sub][i]);
                ls[sub] += S[sub][i];
            }
        l += (ls[0] + ls[1]) + (ls[2] + ls[3]);
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            uint4 va[2];
            read_tr(va, vimg, sub, lane);
            mma_second(O, va, S[sub]);
        }
        if (t + 1 < nt) {
            store_lds(sK + (cur ^ 1) * IMG, sk, tid);
            store_lds(sV + (cur ^ 1) * IMG, sv, tid);
        }
        __syncthreads();
    }
    const float lt = l + __shfl_xor(l, 32, 64);
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    h16_t* Oo = reinterpret_cast<h16_t*>(p.out_o) + (int64_t)b * p.Lq * p.ldo + hh * p.dh;
    store_acc(O, Oo, p.ldo, qrow, qvalid, p.dh, h, inv);
    if (qvalid && h == 0) p.lse2[((int64_t)b * p.H + hh) * p.Lq + qrow] = m + __builtin_amdgcn_logf(lt);
}
__global__ __launch_bounds__(256, 2) void attn_fwd_bf16_pre_masked(Args p) { attn_fwd_pre_body<true>(p); }

// The fast forward (round 2).  In-kernel ablations of attn_fwd_bf16_pre at the benchmark shape (tools/micro/attn_lab.hip,
// profiles/round2_attention_lab.md) showed a kernel bound by instruction ISSUE, every class of instruction costing its own
// time (v_exp ~6, any other VALU ~3, an MFMA ~17, a global load ~75 cycles of SIMD time): the per-tile running maximum (33
// v_max3 + the exchange + a 20-cycle hazard fence per 128-key tile) and the register staging (4 ds_write_b128 + 16 VGPRs)
// are pure overhead.  Here the softmax reference is anchored ONCE, at the row maximum of key tile 0, and every tile is
// exponentiated against it: P = 2^(s - m0) is exact in floating point whatever m0 is (bf16 P keeps its relative precision
// at any magnitude, l and O accumulate in fp32), as long as nothing overflows — a later score more than ~2^100 above the
// anchor.  That cannot be ruled out, so it is DETECTED: an inf / NaN row sum flags the workgroup in `redo`, and
// attn_fwd_bf16_pre, launched right behind this kernel, recomputes exactly the flagged workgroups with per-tile maxima
// (it exits at once everywhere else).  K / V tiles are staged by LDS-DMA.  0.507 -> 0.447 ms at B 8, H 8, L 6272.
__global__ __launch_bounds__(256, 2) void attn_fwd_bf16_fast(Args p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * IMG];
    char* sK = smem;
    char* sV = smem + 2 * IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int xt, hh, b;
    block_coords(p, xt, hh, b);
    const int qrow = xt * 128 + wave * 32 + r;
    const bool qvalid = qrow < p.Lq;
    const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * p.dh;
    const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * p.dh;
    const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * p.dh;
    uint4 qb[2];
    load_lane_block(qb, Q, p.ldq, qrow, qvalid, p.dh, h);
    const int nt = p.Lk / KT;     // launcher guarantees Lk % KT == 0 and dh == 32
    dma_tile(sK, K, p.ldk, 0, wave, lane);
    dma_tile(sV, V, p.ldv, 0, wave, lane);
    dma_wait_all();
    __syncthreads();
    float m;                      // the anchor: this query's largest score in key tile 0 (log2 domain)
    {
        float mm = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            uint4 ka[2];
            read_rows(ka, sK, sub * 32 + r, h);
            const f32x16 S = mma_first(ka, qb);
#pragma unroll
            for (int i = 0; i < 16; ++i) mm = fmaxf(mm, S[i]);
        }
        const HalfPair sw = swap_halves(__builtin_bit_cast(unsigned, mm));
        m = fmaxf(__builtin_bit_cast(float, sw.lo), __builtin_bit_cast(float, sw.hi));
    }
    const f32x16 Cm = splat16(-m);
    f32x16 O = zero16();
    float l = 0.f;
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {         // the other buffer was last read in tile t-1, behind that tile's barrier
            dma_tile_async(sK + (cur ^ 1) * IMG, K, p.ldk, (t + 1) * KT, wave, lane);   // (waited for at the tile boundary)
            dma_tile_async(sV + (cur ^ 1) * IMG, V, p.ldv, (t + 1) * KT, wave, lane);
        }
        const char* kimg = sK + cur * IMG;
        const char* vimg = sV + cur * IMG;
        f32x16 S[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            uint4 ka[2];
            read_rows(ka, kimg, sub * 32 + r, h);
            S[sub] = mma_first_c(ka, qb, Cm);  // = score - anchor
        }
        f32x2 ls[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};   // row sums as v_pk_add_f32 over register pairs (32 instead of 64 adds)
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                S[sub][i] = __builtin_amdgcn_exp2f(S[sub][i]);
                S[sub][i + 1] = __builtin_amdgcn_exp2f(S[sub][i + 1]);
                ls[sub] += f32x2{S[sub][i], S[sub][i + 1]};
            }
        {
            const f32x2 t2 = (ls[0] + ls[1]) + (ls[2] + ls[3]);
            l += t2[0] + t2[1];
        }
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            uint4 va[2];
            read_tr(va, vimg, sub, lane);
            mma_second(O, va, S[sub]);
        }
        dma_wait_all();           // this wave's pieces of tile t+1 have landed; the barrier publishes everyone's
        __syncthreads();
    }
    const float lt = l + __shfl_xor(l, 32, 64);
    const int bad = __syncthreads_or(qvalid && !(lt < SVOL_H16_PSUM_MAX));   // inf / NaN (fp16: any P near 65504): a score left the anchor's range
    if (tid == 0) p.redo[blockIdx.x] = bad ? 1 : 0;
    if (bad) return;              // attn_fwd_bf16_pre (next launch on the stream) recomputes this workgroup
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    h16_t* Oo = reinterpret_cast<h16_t*>(p.out_o) + (int64_t)b * p.Lq * p.ldo + hh * p.dh;
    store_acc(O, Oo, p.ldo, qrow, qvalid, p.dh, h, inv);
    if (qvalid && h == 0) p.lse2[((int64_t)b * p.H + hh) * p.Lq + qrow] = m + __builtin_amdgcn_logf(lt);
}

// ---- pieces of the ROTATED dQ loop (round 3) ----------------------------------------------------------------------------------
// The round-2 loop issued the eight score / dP products of a 32-key step back to back (256 cycles in which the wave issues nothing
// else) and then ran each block's exp / multiply / convert chain, which waits on exactly those results: MFMA and VALU of ONE wave
// never overlapped, and two co-resident waves met in the same phase as often as not (co-execution 17 %, 466 cycles per block against
// an issue floor of ~270).  Here the two query blocks of a wave run half a step apart: while block c's softmax chain issues, the
// products of the OTHER block (for this key step or the next) are in the matrix pipe — the VALU instructions only touch results
// that finished a phase ago.  The interleave is pinned by hand: one MFMA, then ~48 issue cycles of exp / multiply, fenced with
// sched_barrier(0) so that hipcc's scheduler keeps the order (it would re-cluster the MFMAs).
__device__ __forceinline__ f32x16 mma_c_first(const uint4& a0, const uint4& b0, const f32x16& c0) {
    f32x16 acc;   // D != C: the loop-invariant row constants are never copied (see mma_first_c)
    asm("s_nop 1\n\tv_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, %3"
        : "=&v"(acc)
        : "v"(__builtin_bit_cast(h16x8, a0)), "v"(__builtin_bit_cast(h16x8, b0)), "v"(c0));
    return acc;
}
__device__ __forceinline__ f32x16 mma_acc(const uint4& a1, const uint4& b1, const f32x16& acc) {
    return SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, a1), __builtin_bit_cast(h16x8, b1), acc, 0, 0, 0);
}
template <int LO>
__device__ __forceinline__ void ds_chunk(f32x16& S, const f32x16& dP) {   // dS = exp2(score - lse) * (dP - delta), 4 elements
    // the products as two v_pk_mul_f32 on adjacent accumulator registers (plain VALU does not co-issue with the matrix pipe — only
    // v_exp does, profiles/round2_pmc_attention.md — so every vector instruction saved is 4 issue cycles per block)
#pragma unroll
    for (int i = LO; i < LO + 4; i += 2) {
        const f32x2 e = {__builtin_amdgcn_exp2f(S[i]), __builtin_amdgcn_exp2f(S[i + 1])};
        const f32x2 d = {dP[i], dP[i + 1]};
        const f32x2 m = e * d;
        S[i] = m[0];
        S[i + 1] = m[1];
    }
}
#define SVOL_FENCE() __builtin_amdgcn_sched_barrier(0)
// start the two products of block n (A fragments ka / va of a key step) while block c finishes: chain, conversion, dQ_c += K^T dS_c
__device__ __forceinline__ void dq_phase(f32x16& Sn, f32x16& dPn, const uint4 (&ka)[2], const uint4 (&va)[2], const uint4 (&qbn)[2],
                                         const uint4 (&dobn)[2], const f32x16& Cln, const f32x16& Cdn, f32x16& Sc, const f32x16& dPc,
                                         f32x16& dQc, const uint4 (&kt)[2]) {
    Sn = mma_c_first(ka[0], qbn[0], Cln);
    SVOL_FENCE();
    ds_chunk<0>(Sc, dPc);
    SVOL_FENCE();
    Sn = mma_acc(ka[1], qbn[1], Sn);
    SVOL_FENCE();
    ds_chunk<4>(Sc, dPc);
    SVOL_FENCE();
    dPn = mma_c_first(va[0], dobn[0], Cdn);
    SVOL_FENCE();
    ds_chunk<8>(Sc, dPc);
    SVOL_FENCE();
    dPn = mma_acc(va[1], dobn[1], dPn);
    SVOL_FENCE();
    ds_chunk<12>(Sc, dPc);
    SVOL_FENCE();
    mma_second(dQc, kt, Sc);
}
__device__ __forceinline__ void dq_finish(f32x16& Sc, const f32x16& dPc, f32x16& dQc, const uint4 (&kt)[2]) {
    ds_chunk<0>(Sc, dPc);
    ds_chunk<4>(Sc, dPc);
    ds_chunk<8>(Sc, dPc);
    ds_chunk<12>(Sc, dPc);
    mma_second(dQc, kt, Sc);
}

// Two 32-query blocks per wave (256 queries per workgroup): every K / V fragment read from LDS feeds two MFMAs, and
// a wave always has a second, independent MFMA -> exp -> MFMA chain to issue from while the first one waits.
template <bool MASKED>
__device__ __forceinline__ void attn_bwd_dq_pre_body(const Args& p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * IMG];
    char* sK = smem;
    char* sV = smem + 2 * IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int xt, hh, b;
    block_coords(p, xt, hh, b);
    xt = __builtin_amdgcn_readfirstlane(xt);   // (the divisions of block_coords leave uniform values in VECTOR registers: back to scalars)
    hh = __builtin_amdgcn_readfirstlane(hh);
    b = __builtin_amdgcn_readfirstlane(b);
    const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * p.dh;
    const h16_t* dO = reinterpret_cast<const h16_t*>(p.d_o) + (int64_t)b * p.Lq * p.lddo + hh * p.dh;
    const h16_t* O = reinterpret_cast<const h16_t*>(p.o) + (int64_t)b * p.Lq * p.ldo + hh * p.dh;
    const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * p.dh;
    const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * p.dh;

    int qrow[2];
    bool qvalid[2];
    uint4 qb[2][2], dob[2][2];
    f32x16 Cl[2], Cd[2], dQ[2];
    // a workgroup with at most 128 rows left (the last tile of a head when Lq % 256 is in 1..128) gives every wave ONE
    // 32-query block instead of two half-empty waves with two: it finishes in about half the time, and the launcher
    // dispatches these tiles last, where they shorten the partial last round of workgroups
    const bool single = p.Lq - xt * 256 <= 128;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        qrow[u] = single ? (u == 0 ? xt * 256 + wave * 32 + r : p.Lq) : xt * 256 + wave * 64 + u * 32 + r;
        qvalid[u] = qrow[u] < p.Lq;
        load_lane_block(qb[u], Q, p.ldq, qrow[u], qvalid[u], p.dh, h);
        load_lane_block(dob[u], dO, p.lddo, qrow[u], qvalid[u], p.dh, h);
        const int64_t sidx = ((int64_t)b * p.H + hh) * p.Lq + qrow[u];
        Cl[u] = splat16(qvalid[u] ? -p.lse2[sidx] : -INFINITY);  // score - lse  (rows past Lq -> p = 0)
        // delta = rowsum(dO . O), computed here instead of in a kernel of its own (this lane and lane ^ 32 hold the two halves
        // of the row) and published for the dK/dV pass, which runs after this kernel on the same stream
        float dl = 0.f;
        {
            uint4 ob[2];
            load_lane_block(ob, O, p.ldo, qrow[u], qvalid[u], p.dh, h);
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const h16x8 a = __builtin_bit_cast(h16x8, ob[s_]), c = __builtin_bit_cast(h16x8, dob[u][s_]);
#pragma unroll
                for (int e = 0; e < 8; ++e) dl += (float)a[e] * (float)c[e];
            }
            const HalfPair sw = swap_halves(__builtin_bit_cast(unsigned, dl));
            dl = __builtin_bit_cast(float, sw.lo) + __builtin_bit_cast(float, sw.hi);
        }
        if (qvalid[u] && h == 0) {
            p.delta[sidx] = dl;
            if (p.nl2) {  // the dK/dV pass adds the row constants through the matrix pipe as (hi, lo) bf16 pairs: split them once, here
                p.nl2[sidx] = split_bf16x2(-p.lse2[sidx]);
                p.nd2[sidx] = split_bf16x2(-dl);
            }
        }
        Cd[u] = splat16(qvalid[u] ? -dl : 0.f);                  // dP - delta
        dQ[u] = zero16();
    }
    const float* kb = (MASKED && p.kbias) ? p.kbias + (int64_t)b * p.Lk : nullptr;
    const int nt = MASKED ? (p.Lk + KT - 1) / KT : p.Lk / KT;
    Stage sk, sv;
    load_regs(sk, K, p.ldk, 0, p.Lk, p.dh, tid);
    load_regs(sv, V, p.ldv, 0, p.Lk, p.dh, tid);
    const int* fl = MASKED ? p.tile_flags + (int64_t)b * nt : nullptr;
    unsigned long long mixed = 0, dead = 0;
    store_lds(sK, sk, tid);
    store_lds(sV, sv, tid);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            load_regs(sk, K, p.ldk, (t + 1) * KT, p.Lk, p.dh, tid);
            load_regs(sv, V, p.ldv, (t + 1) * KT, p.Lk, p.dh, tid);
        }
        if (MASKED && (t & 63) == 0) {  // tile classes as in the forward kernel
            const int f = (t + lane < nt) ? fl[t + lane] : 2;
            mixed = __ballot(f == 1);
            dead = __ballot(f == 2);
        }
        const int flag = MASKED ? (int)((mixed >> (t & 63)) & 1) + 2 * (int)((dead >> (t & 63)) & 1) : 0;
        const char* kimg = sK + cur * IMG;
        const char* vimg = sV + cur * IMG;
#pragma unroll 2
        for (int sub = 0; sub < ((MASKED && flag == 2) ? 0 : 4); ++sub) {
            uint4 ka[2], va[2], kt[2];
            read_rows(ka, kimg, sub * 32 + r, h);
            read_rows(va, vimg, sub * 32 + r, h);
            f32x16 S0 = mma_first_c(ka, qb[0], Cl[0]);
            f32x16 S1, dP1;
            if (!single) S1 = mma_first_c(ka, qb[1], Cl[1]);
            f32x16 dP0 = mma_first_c(va, dob[0], Cd[0]);
            if (!single) dP1 = mma_first_c(va, dob[1], Cd[1]);
            read_tr(kt, kimg, sub, lane);
            if (MASKED && flag == 1) {
                S0 = add_key_bias(S0, kb, t * KT + sub * 32 + r, p.Lk, h);
                if (!single) S1 = add_key_bias(S1, kb, t * KT + sub * 32 + r, p.Lk, h);
            }
            // hipcc (ROCm 7.2) pads an MFMA write -> VALU read only inside a basic block: with `single` the branch around the second
            // block's products lands straight on the first v_exp (fp16 build: S0[3] was read the instruction after the bias MFMA
            // issued — the masked key's bias was silently missing for accumulator row 3).  Rare paths only (tail tile / mixed tile):
            if (single || (MASKED && flag == 1)) mfma_results_ready2(S0, dP0);
#pragma unroll
            for (int i = 0; i < 16; ++i) S0[i] = __builtin_amdgcn_exp2f(S0[i]) * dP0[i];
            mma_second(dQ[0], kt, S0);
            if (!single) {
#pragma unroll
                for (int i = 0; i < 16; ++i) S1[i] = __builtin_amdgcn_exp2f(S1[i]) * dP1[i];
                mma_second(dQ[1], kt, S1);
            }
        }
        if (t + 1 < nt) {
            store_lds(sK + (cur ^ 1) * IMG, sk, tid);
            store_lds(sV + (cur ^ 1) * IMG, sv, tid);
        }
        __syncthreads();
    }
    h16_t* dQo = reinterpret_cast<h16_t*>(p.dq) + (int64_t)b * p.Lq * p.lddq + hh * p.dh;
    // the row indices are RECOMPUTED here (from a thread id hipcc cannot trace back) instead of being carried across the key loop: at
    // 256 registers their 64-bit forms were the 2 - 3 VGPRs these kernels spilled to scratch (VERDICT r4 "What's weak 8")
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int r_e = tid_e & 31, h_e = (tid_e >> 5) & 1, wave_e = tid_e >> 6;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int qrow_e = single ? (u == 0 ? xt * 256 + wave_e * 32 + r_e : p.Lq) : xt * 256 + wave_e * 64 + u * 32 + r_e;
        store_acc(dQ[u], dQo, p.lddq, qrow_e, qrow_e < p.Lq, p.dh, h_e, p.scale);
    }
}
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_bf16_pre(Args p) { attn_bwd_dq_pre_body<false>(p); }
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_bf16_pre_masked(Args p) { attn_bwd_dq_pre_body<true>(p); }

// The unmasked dQ pass on the ROTATED schedule (dq_phase above).  Same tile geometry, row constants and delta prologue as
// attn_bwd_dq_pre_body<false>; K / V tiles come by LDS-DMA (no staging registers: the rotation keeps four accumulator blocks, the
// row constants and both blocks' operands live — 256 VGPRs is the budget at two waves per SIMD).  A workgroup with at most 128 rows
// left runs the same loop with its second block empty (zero operands, -inf row constant: every term is an exact zero).
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_bf16_rot(Args p) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * IMG];
    char* sK = smem;
    char* sV = smem + 2 * IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int xt, hh, b;
    block_coords(p, xt, hh, b);
    xt = __builtin_amdgcn_readfirstlane(xt);   // (the divisions of block_coords leave uniform values in VECTOR registers: back to scalars)
    hh = __builtin_amdgcn_readfirstlane(hh);
    b = __builtin_amdgcn_readfirstlane(b);
    const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * p.dh;
    const h16_t* dO = reinterpret_cast<const h16_t*>(p.d_o) + (int64_t)b * p.Lq * p.lddo + hh * p.dh;
    const h16_t* O = reinterpret_cast<const h16_t*>(p.o) + (int64_t)b * p.Lq * p.ldo + hh * p.dh;
    const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * p.dh;
    const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * p.dh;
    dma_tile(sK, K, p.ldk, 0, wave, lane);
    dma_tile(sV, V, p.ldv, 0, wave, lane);

    int qrow[2];
    bool qvalid[2];
    uint4 qb[2][2], dob[2][2];
    f32x16 Cl[2], Cd[2], dQ[2];
    const bool single = p.Lq - xt * 256 <= 128;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        qrow[u] = single ? (u == 0 ? xt * 256 + wave * 32 + r : p.Lq) : xt * 256 + wave * 64 + u * 32 + r;
        qvalid[u] = qrow[u] < p.Lq;
        load_lane_block(qb[u], Q, p.ldq, qrow[u], qvalid[u], p.dh, h);
        load_lane_block(dob[u], dO, p.lddo, qrow[u], qvalid[u], p.dh, h);
        const int64_t sidx = ((int64_t)b * p.H + hh) * p.Lq + qrow[u];
        Cl[u] = splat16(qvalid[u] ? -p.lse2[sidx] : -INFINITY);
        float dl = 0.f;
        {
            uint4 ob[2];
            load_lane_block(ob, O, p.ldo, qrow[u], qvalid[u], p.dh, h);
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const h16x8 a = __builtin_bit_cast(h16x8, ob[s_]), c = __builtin_bit_cast(h16x8, dob[u][s_]);
#pragma unroll
                for (int e = 0; e < 8; ++e) dl += (float)a[e] * (float)c[e];
            }
            const HalfPair sw = swap_halves(__builtin_bit_cast(unsigned, dl));
            dl = __builtin_bit_cast(float, sw.lo) + __builtin_bit_cast(float, sw.hi);
        }
        if (qvalid[u] && h == 0) {
            p.delta[sidx] = dl;
            if (p.nl2) {
                p.nl2[sidx] = split_bf16x2(-p.lse2[sidx]);
                p.nd2[sidx] = split_bf16x2(-dl);
            }
        }
        Cd[u] = splat16(qvalid[u] ? -dl : 0.f);
        dQ[u] = zero16();
    }
    const int nt = p.Lk / KT;
    dma_wait_all();
    __syncthreads();
    f32x16 S0, S1, dP0, dP1;
    uint4 ka[2], va[2];
    read_rows(ka, sK, r, h);
    read_rows(va, sV, r, h);
    S0 = mma_first_c(ka, qb[0], Cl[0]);
    dP0 = mma_first_c(va, dob[0], Cd[0]);
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < nt;
        if (more) {   // the other image is free: its last reads came before the previous boundary's barrier
            dma_tile_async(sK + (cur ^ 1) * IMG, K, p.ldk, (t + 1) * KT, wave, lane);
            dma_tile_async(sV + (cur ^ 1) * IMG, V, p.ldv, (t + 1) * KT, wave, lane);
        }
        const char* kimg = sK + cur * IMG;
        const char* vimg = sV + cur * IMG;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            uint4 kt[2];
            read_tr(kt, kimg, sub, lane);
            // block 1's products of this key step start; block 0 (products issued a phase ago) finishes
            dq_phase(S1, dP1, ka, va, qb[1], dob[1], Cl[1], Cd[1], S0, dP0, dQ[0], kt);
            if (sub < 3) {
                read_rows(ka, kimg, (sub + 1) * 32 + r, h);
                read_rows(va, vimg, (sub + 1) * 32 + r, h);
                dq_phase(S0, dP0, ka, va, qb[0], dob[0], Cl[0], Cd[0], S1, dP1, dQ[1], kt);
            } else {
                dma_wait_all();      // this wave's pieces of tile t + 1 have landed; the barrier publishes everyone's
                __syncthreads();     // (kt of this step is already in registers)
                if (more) {
                    read_rows(ka, sK + (cur ^ 1) * IMG, r, h);
                    read_rows(va, sV + (cur ^ 1) * IMG, r, h);
                    dq_phase(S0, dP0, ka, va, qb[0], dob[0], Cl[0], Cd[0], S1, dP1, dQ[1], kt);
                } else {
                    dq_finish(S1, dP1, dQ[1], kt);
                }
            }
        }
    }
    h16_t* dQo = reinterpret_cast<h16_t*>(p.dq) + (int64_t)b * p.Lq * p.lddq + hh * p.dh;
    // the row indices are RECOMPUTED here (from a thread id hipcc cannot trace back) instead of being carried across the key loop: at
    // 256 registers their 64-bit forms were the 2 - 3 VGPRs these kernels spilled to scratch (VERDICT r4 "What's weak 8")
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int r_e = tid_e & 31, h_e = (tid_e >> 5) & 1, wave_e = tid_e >> 6;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int qrow_e = single ? (u == 0 ? xt * 256 + wave_e * 32 + r_e : p.Lq) : xt * 256 + wave_e * 64 + u * 32 + r_e;
        store_acc(dQ[u], dQo, p.lddq, qrow_e, qrow_e < p.Lq, p.dh, h_e, p.scale);
    }
}

// In this kernel the per-QUERY constants (-lse, -delta) run along the 16 accumulator registers of a lane (rows of
// the score tile are queries), so as initial accumulators they cost four ds_read_b128 per product — half of the
// loop's LDS traffic, and LDS was the busiest unit (rocprofv3 PMC: ~70 % of its bandwidth).  They ride the matrix
// pipe instead: one more 16-deep MFMA per product contracts [hi, lo, 0...] (per query, one dword from LDS)
// with [1, 1, 0...] (constant), i.e. adds -lse / -delta to every score of that query in fp32.
// DMA (unmasked, Lq % 128 == 0, pre-split statistics from the dQ kernel): Q / dO tiles and the two statistics vectors go
// HBM -> LDS by LDS-DMA (no VGPR round trip, no ds_write, no split arithmetic here): 0.785 -> 0.740 ms at the benchmark shape.
template <bool MASKED, bool DMA = false>
__device__ __forceinline__ void attn_bwd_dkdv_pre_body(const Args& p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * IMG + 4 * KT * 4];
    char* sQ = smem;
    char* sdO = smem + 2 * IMG;
    unsigned* sL = reinterpret_cast<unsigned*>(smem + 4 * IMG);  // [2][KT] -lse2 as (hi, lo) bf16
    unsigned* sD = sL + 2 * KT;                                  // [2][KT] -delta as (hi, lo) bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = DMA ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int xt, hh, b;
    block_coords(p, xt, hh, b);
    const int krow = xt * 128 + wave * 32 + r;
    const bool kvalid = krow < p.Lk;
    const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * p.dh;
    const h16_t* dO = reinterpret_cast<const h16_t*>(p.d_o) + (int64_t)b * p.Lq * p.lddo + hh * p.dh;
    const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * p.dh;
    const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * p.dh;
    const float* lse_g = p.lse2 + ((int64_t)b * p.H + hh) * p.Lq;
    const float* dl_g = p.delta + ((int64_t)b * p.H + hh) * p.Lq;

    uint4 kbk[2], vbk[2];
    load_lane_block(kbk, K, p.ldk, krow, kvalid, p.dh, h);
    load_lane_block(vbk, V, p.ldv, krow, kvalid, p.dh, h);
    f32x16 dK = zero16(), dV = zero16();
    const int nt = (p.Lq + KT - 1) / KT;
    Stage sq, sdo;
    float rl = 0.f, rd = 0.f;
    auto load_stats = [&](int row0) {
        if (tid < KT) {
            const int qi = row0 + tid;
            rl = qi < p.Lq ? -lse_g[qi] : SVOL_H16_NEG_BIG;  // finite IN THE OPERAND TYPE (it meets a 0 in the MFMA), still exp2 -> 0
            rd = qi < p.Lq ? -dl_g[qi] : 0.f;
        }
    };
    auto store_stats = [&](int buf) {
        if (tid < KT) { sL[buf * KT + tid] = split_bf16x2(rl); sD[buf * KT + tid] = split_bf16x2(rd); }
    };
    const uint4 ones = h == 0 ? make_uint4(SVOL_H16_ONE2, 0u, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);  // [1, 1, 0...]
    // MASKED: the key bias is a per-LANE constant here (keys are the columns of the score tile).  It rides the same extra
    // MFMA as -lse: the query side gets [hi, lo, 1, 1, 0...] and this key's side [1, 1, bias_hi, bias_lo, 0...] — no
    // per-score instruction at all.  Keys past Lk get -inf (p = 0).  A wave whose 32 keys are all masked only helps staging.
    float kbl = 0.f;
    if (MASKED) kbl = kvalid ? (p.kbias ? p.kbias[(int64_t)b * p.Lk + krow] * LOG2E : 0.f) : -INFINITY;
    const bool wave_dead = MASKED && __all(kbl == -INFINITY);
    const unsigned kb_pair = kbl == -INFINITY ? SVOL_H16_NINF_LO : split_bf16x2(kbl);  // (-inf, 0): x - hi would be NaN
    const uint4 ones_s = (MASKED && h == 0) ? make_uint4(SVOL_H16_ONE2, kb_pair, 0u, 0u) : ones;
    const unsigned q_one = MASKED ? SVOL_H16_ONE2 : 0u;
    const unsigned* nl_h = DMA ? p.nl2 + ((int64_t)b * p.H + hh) * p.Lq : nullptr;
    const unsigned* nd_h = DMA ? p.nd2 + ((int64_t)b * p.H + hh) * p.Lq : nullptr;
    auto dma_stats = [&](int buf, int row0) {   // wave 0: -lse pairs, wave 1: -delta pairs; 32 lanes x 16 bytes = 128 queries
        if (wave < 2 && lane < 32) {
            const unsigned* src = (wave == 0 ? nl_h : nd_h) + row0 + lane * 4;
            unsigned* dst = (wave == 0 ? sL : sD) + buf * KT;
            __builtin_amdgcn_global_load_lds((gbl_vptr)src, (lds_vptr)dst, 16, 0, 0);
        }
    };
    auto dma_stats_async = [&](int buf, int row0) {   // the same from inline asm (see dma_piece_async): no compiler-inserted drain
        if (wave < 2) {
            const unsigned* src = (wave == 0 ? nl_h : nd_h) + row0 + (lane & 31) * 4;
            const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_vptr)((wave == 0 ? sL : sD) + buf * KT));
            if (lane < 32) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "m0");
        }
    };
    if (DMA) {
        dma_tile(sQ, Q, p.ldq, 0, wave, lane);
        dma_tile(sdO, dO, p.lddo, 0, wave, lane);
        dma_stats(0, 0);
        dma_wait_all();
    } else {
        load_regs(sq, Q, p.ldq, 0, p.Lq, p.dh, tid);
        load_regs(sdo, dO, p.lddo, 0, p.Lq, p.dh, tid);
        load_stats(0);
        store_lds(sQ, sq, tid);
        store_lds(sdO, sdo, tid);
        store_stats(0);
    }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            if (DMA) {
                dma_tile_async(sQ + (cur ^ 1) * IMG, Q, p.ldq, (t + 1) * KT, wave, lane);   // (waited for at the tile boundary)
                dma_tile_async(sdO + (cur ^ 1) * IMG, dO, p.lddo, (t + 1) * KT, wave, lane);
                dma_stats_async(cur ^ 1, (t + 1) * KT);
            } else {
                load_regs(sq, Q, p.ldq, (t + 1) * KT, p.Lq, p.dh, tid);
                load_regs(sdo, dO, p.lddo, (t + 1) * KT, p.Lq, p.dh, tid);
                load_stats((t + 1) * KT);
            }
        }
        const char* qimg = sQ + cur * IMG;
        const char* doimg = sdO + cur * IMG;
        const unsigned* nl = sL + cur * KT;
        const unsigned* nd = sD + cur * KT;
        // software pipeline over the four 32-query sub-tiles: the score / dP products of sub-tile s+1 are issued
        // BEFORE the exp / multiply / convert work of sub-tile s, so the matrix pipe runs under the VALU phase
        // (a wave's MFMA -> VALU -> MFMA chain is strictly dependent otherwise, and the profile showed time = sum)
        auto first_products = [&](int sub, f32x16& S, f32x16& dP) {
            uint4 a[2];
            // every lane loads its query's pair; for the h = 1 lanes (k = 8..15) `ones` is zero and the pair is finite
            const uint4 el = make_uint4(nl[sub * 32 + r], q_one, 0u, 0u);
            const uint4 ed = make_uint4(nd[sub * 32 + r], 0u, 0u, 0u);
            read_rows(a, qimg, sub * 32 + r, h);
            S = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, el), __builtin_bit_cast(h16x8, ones_s),
                                                        zero16(), 0, 0, 0);
            S = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, a[0]), __builtin_bit_cast(h16x8, kbk[0]), S, 0, 0, 0);
            S = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, a[1]), __builtin_bit_cast(h16x8, kbk[1]), S, 0, 0, 0);  // score - lse[q]
            read_rows(a, doimg, sub * 32 + r, h);
            dP = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, ed), __builtin_bit_cast(h16x8, ones),
                                                         zero16(), 0, 0, 0);
            dP = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, a[0]), __builtin_bit_cast(h16x8, vbk[0]), dP, 0, 0, 0);
            dP = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, a[1]), __builtin_bit_cast(h16x8, vbk[1]), dP, 0, 0, 0);  // dO V^T - delta[q]
        };
        f32x16 S, dP, Sn, dPn;
        if (!wave_dead) first_products(0, S, dP);
#pragma unroll
        for (int sub = 0; sub < (wave_dead ? 0 : 4); ++sub) {
            if (sub + 1 < 4) first_products(sub + 1, Sn, dPn);
            uint4 a[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) S[i] = __builtin_amdgcn_exp2f(S[i]);
            read_tr(a, doimg, sub, lane);
            mma_second(dV, a, S);
#pragma unroll
            for (int i = 0; i < 16; i += 2) {   // v_pk_mul_f32 on adjacent accumulator registers (8 instead of 16 vector instructions)
                const f32x2 m = f32x2{S[i], S[i + 1]} * f32x2{dP[i], dP[i + 1]};
                S[i] = m[0];
                S[i + 1] = m[1];
            }
            read_tr(a, qimg, sub, lane);
            mma_second(dK, a, S);
            if (sub + 1 < 4) { S = Sn; dP = dPn; }
        }
        if (t + 1 < nt && !DMA) {
            store_lds(sQ + (cur ^ 1) * IMG, sq, tid);
            store_lds(sdO + (cur ^ 1) * IMG, sdo, tid);
            store_stats(cur ^ 1);
        }
        if (DMA) dma_wait_all();
        __syncthreads();
    }
    h16_t* dKo = reinterpret_cast<h16_t*>(p.dk) + (int64_t)b * p.Lk * p.lddk + hh * p.dh;
    h16_t* dVo = reinterpret_cast<h16_t*>(p.dv) + (int64_t)b * p.Lk * p.lddv + hh * p.dh;
    store_acc(dK, dKo, p.lddk, krow, kvalid, p.dh, h, p.scale / p.premul);
    store_acc(dV, dVo, p.lddv, krow, kvalid, p.dh, h, 1.f);
}
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_bf16_pre(Args p) { attn_bwd_dkdv_pre_body<false>(p); }
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_bf16_pre_dma(Args p) { attn_bwd_dkdv_pre_body<false, true>(p); }
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_bf16_pre_masked(Args p) { attn_bwd_dkdv_pre_body<true>(p); }

// ===========================================================================================
// SINGLE-PASS backward (round 4): the two-pass kernels above recompute S = Q K^T and dP = dO V^T in both passes and exponentiate
// twice — 16 MFMAs and two exp passes per 32 x 32 block where one pass needs 10 and one — and both sit within 15-25 % of their own
// instruction-issue floor.  Here a workgroup is KEY-stationary: 4 waves x 128 keys = 512 keys of one (batch, head); a wave keeps
// dK^T / dV^T of its four 32-key blocks in 128 accumulator registers and K / V as MFMA B operands in registers, and the workgroup
// streams the head's queries in 32-row steps (Q / dO tiles + the per-query constants -lse, -delta by LDS-DMA, 128 rows at a time):
//   S = Q K^T - lse, dP = dO V^T - delta (row constants = initial accumulators, read once per step for all four key blocks),
//   P = exp2(S), dV^T += dO^T P, dS = P * dP, dK^T += Q^T dS                (key on the lane: P / dS feed the next MFMA as they stand)
//   dQ[q][d] += dS[q][key] K[key][d]: the one product that contracts over the LANE index of dS — the wave writes its packed dS block
//   to its own LDS image [key][q] and reads it back transposed (ds_read_b64_tr_b16); K as that product's B operand is loop invariant.
// dQ is summed over the workgroup's 512 keys on chip (each wave's 32 x 32 fp32 partial through LDS, one row group per wave) and leaves
// as fp32 atomics with d on the lane — two 128-byte row segments per wave instruction, the full-rate shape — into a zeroed fp32 image
// that attn_dq_round_bf16 rounds afterwards: L / 512 adds per element (0.63 GB per launch at B 8, H 8, L 6272 against the 2.5 GB of
// 128-key ownership that ruled a single pass out in round 3).  The atomics of step i are issued in step i + 1, behind the barrier that
// publishes the partials, and are never waited for inside the loop (counted vmcnt).
constexpr int SP_KB = 4;                    // 32-key blocks per wave
constexpr int SP_WKEYS = 32 * SP_KB;        // keys per wave
constexpr int SP_KEYS = 4 * SP_WKEYS;       // keys per workgroup
constexpr int SP_PART = 4 * 4 * 64 * 16;    // one partial buffer: [wave][row group][lane] x 16 bytes
constexpr int SP_LDS = 4 * IMG + 4 * IMG + 2 * SP_PART + 4 * KT * 4 + 512;

// byte offset of the 8-byte unit (chunk g = q / 8, half = (q / 4) & 1) of row `key` in a [32 keys][32 q] 16-bit image with 64-byte
// rows: 16-byte chunks and 8-byte halves XOR-ed with key bits so that the 16 lanes of a ds_write_b64 group (16 keys, one unit each)
// land on 32 different banks; the transposed read takes four whole rows per 32-lane half and is conflict free either way
__device__ __forceinline__ int ds_off(int key, int g, int half) {
    return key * 64 + ((g ^ ((key >> 2) & 3)) << 4) + ((half ^ ((key >> 1) & 1)) << 3);
}
// operand whose contraction index runs over the rows of 32-row sub-tile `sub` of a [128][32] image in NATURAL order (k-step s, lane
// half h, element j <-> row 16 s + 8 h + j), lane & 31 = column: the loop-invariant B operand K[key][d] of the dQ product
__device__ __forceinline__ void read_tr_nat(uint4 (&a)[2], const char* img, int sub, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, hh = g >> 1;
    const int ch = 2 * (g & 1) + (p >> 1), inner = 8 * (p & 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int r1 = 32 * sub + 16 * s + 8 * hh + q;
        const h16x4 lo = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(img + img_off(r1, ch) + inner));
        const h16x4 hi = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(img + img_off(r1 + 4, ch) + inner));
        a[s] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
}
// A operand dS[q][key] of the dQ product from the wave's [32 keys][32 q] image (ds_off), same natural key order
__device__ __forceinline__ void read_ds_tr(uint4 (&a)[2], const char* img, int lane) {
    const int gg = lane >> 4, i = lane & 15, qq = i >> 2, p = i & 3, hh = gg >> 1;
    const int g = 2 * (gg & 1) + (p >> 1), half = p & 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int key = 16 * s + 8 * hh + qq;
        const h16x4 lo = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(img + ds_off(key, g, half)));
        const h16x4 hi = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(img + ds_off(key + 4, g, half)));
        a[s] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
}
// the 16 accumulator registers of a lane as the two B-operand fragments of the next product (v_cvt_pk per register pair)
__device__ __forceinline__ void pack16(uint4 (&o)[2], const f32x16& x) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        h16x8 b;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const h16x2 pr = cvt_pk_h16(x[8 * s + j], x[8 * s + j + 1]);
            b[j] = pr[0];
            b[j + 1] = pr[1];
        }
        o[s] = __builtin_bit_cast(uint4, b);
    }
}
__device__ __forceinline__ void mma_packed(f32x16& acc, const uint4 (&a)[2], const uint4 (&b)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
        acc = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, a[s]), __builtin_bit_cast(h16x8, b[s]), acc, 0, 0, 0);
}
__device__ __forceinline__ void sp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ===========================================================================================
// FEW-QUERY backward (round 6): the query -> video cross-attention (cross_modal_transformer.py:151-156) has N = 100 queries against
// L = 6272 keys.  It used to take the general two-pass kernels (a key-split dQ pass + a dK/dV pass, 256 + 3136 workgroups, 16 MFMAs and
// two exp passes per 32 x 32 block) — on the query stream, i.e. BESIDE the video stream's single-pass attention backward, whose
// one-wave-per-SIMD workgroups cannot share a CU with anything: a timing-only ablation (zero fills in place of these launches) moved the
// whole step from 17.20 to 16.29 ms (profiles/round6_summary.md).  Here the launch is ONE key-stationary pass, the single-pass
// algorithm in its plain, compiler-scheduled form: a workgroup owns `tiles_per_split` consecutive 128-key tiles of one (batch, head)
// (a wave: 32 keys of each), the <= 128 queries stay in LDS for its whole life,
//   S = Q K^T, P = exp2(S - lse + key bias), dV^T += dO^T P, dP = dO V^T, dS = P (dP - delta), dK^T += Q^T dS  (key on the lane), and
//   dQ[q][d] += dS[q][key] K[key][d]: the packed dS block crosses the wave's own LDS image once and comes back transposed; its B
//   operand is the K tile read in natural row order;
// dK / dV of a tile are final when its 32-query blocks are done; dQ stays in 64 accumulator registers per wave over all the
// workgroup's tiles, the four waves' partials meet in LDS and leave as fp32 atomics into the key-split path's zeroed fp32 image
// (attn_delta_bf16 zeroes it, attn_dq_finish_bf16 scales and rounds it): 10 MFMAs and one exp pass per block, ~450 short workgroups.
template <bool KBIAS>
__global__ __launch_bounds__(256, 2) void attn_bwd_fq_bf16(Args p) {
    __shared__ __attribute__((aligned(16))) char smem[3 * IMG + 2 * KT * 4 + 4 * 2048];
    char* sQ = smem;                 // [128 q][32 d]
    char* sdO = smem + IMG;
    char* sK = smem + 2 * IMG;       // the current key tile (row reads for the score product, natural-order transposed reads for dQ)
    float* sL = reinterpret_cast<float*>(smem + 3 * IMG);   // [KT] -lse2
    float* sD = sL + KT;                                    // [KT] delta
    char* sT = smem + 3 * IMG + 2 * KT * 4;                 // [wave][32 keys][32 q] dS images
    float* sRed = reinterpret_cast<float*>(smem);           // epilogue: [wave][32 q][32 d] fp32 partials (over sQ / sdO: 16 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, hh = blockIdx.y;
    const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * 32;
    const h16_t* dO = reinterpret_cast<const h16_t*>(p.d_o) + (int64_t)b * p.Lq * p.lddo + hh * 32;
    const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * 32;
    const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * 32;
    h16_t* dKo = reinterpret_cast<h16_t*>(p.dk) + (int64_t)b * p.Lk * p.lddk + hh * 32;
    h16_t* dVo = reinterpret_cast<h16_t*>(p.dv) + (int64_t)b * p.Lk * p.lddv + hh * 32;
    const float c = p.premul != 0.f ? 1.f : p.scale * LOG2E;
    const float kmul = p.premul != 0.f ? p.scale / p.premul : p.scale;
    const int nsub = (p.Lq + 31) >> 5;                      // 32-query blocks (launcher: Lq <= 128)
    const int ntk = (p.Lk + KT - 1) / KT;
    const int t0 = blockIdx.x * p.tiles_per_split, t1 = min(ntk, t0 + p.tiles_per_split);
    Stage st;
    {
        Stage sq, sdo;
        load_regs(sq, Q, p.ldq, 0, p.Lq, 32, tid);
        load_regs(sdo, dO, p.lddo, 0, p.Lq, 32, tid);
        load_regs(st, K, p.ldk, t0 * KT, p.Lk, 32, tid);
        if (tid < KT) {
            const int64_t si = ((int64_t)b * p.H + hh) * p.Lq + tid;
            sL[tid] = tid < p.Lq ? -p.lse2[si] : -INFINITY;   // exp2(x - inf) = 0 for rows past Lq
            sD[tid] = tid < p.Lq ? p.delta[si] : 0.f;
        }
        store_lds(sQ, sq, tid);
        store_lds(sdO, sdo, tid);
        store_lds(sK, st, tid);
    }
    __syncthreads();
    f32x16 dQ[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dQ[i] = zero16();
    char* img = sT + wave * 2048;
    for (int t = t0; t < t1; ++t) {
        const int krow = t * KT + wave * 32 + r;
        const bool kvalid = krow < p.Lk;
        uint4 kbk[2], vbk[2], kd[2];
        read_rows(kbk, sK, wave * 32 + r, h);
        read_tr_nat(kd, sK, wave, lane);
        load_lane_block(vbk, V, p.ldv, krow, kvalid, 32, h);
        if (t + 1 < t1) load_regs(st, K, p.ldk, (t + 1) * KT, p.Lk, 32, tid);   // next key tile in flight under this tile's products
        float kbl = kvalid ? 0.f : -INFINITY;
        if (KBIAS && kvalid) kbl = p.kbias[(int64_t)b * p.Lk + krow] * LOG2E;
        f32x16 dKa = zero16(), dVa = zero16();
        if (!__all(kbl == -INFINITY)) {   // (a wave whose 32 keys are all masked has P = 0: dK = dV = 0, nothing for dQ)
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {   // (unrolled: the dQ accumulators keep compile-time register names)
                if (sub >= nsub) break;
                uint4 a[2];
                read_rows(a, sQ, sub * 32 + r, h);
                f32x16 S = mma_first(a, kbk);   // S[q][key]
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 ls = *reinterpret_cast<const f32x4*>(sL + sub * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) S[4 * g + e] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[4 * g + e], c, kbl + ls[e]));
                }
                read_tr(a, sdO, sub, lane);
                mma_second(dVa, a, S);          // dV^T += dO^T P
                read_rows(a, sdO, sub * 32 + r, h);
                const f32x16 dP = mma_first(a, vbk);   // dP[q][key] = dO V^T
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 dl = *reinterpret_cast<const f32x4*>(sD + sub * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) S[4 * g + e] = S[4 * g + e] * (dP[4 * g + e] - dl[e]);
                }
                uint4 ds[2];
                pack16(ds, S);
                read_tr(a, sQ, sub, lane);
                mma_packed(dKa, a, ds);         // dK^T += Q^T dS
                // dS [q][key] -> this wave's [key][q] image -> back transposed as the A operand of dQ += dS K
                *reinterpret_cast<uint2*>(img + ds_off(r, 0, h)) = make_uint2(ds[0].x, ds[0].y);
                *reinterpret_cast<uint2*>(img + ds_off(r, 1, h)) = make_uint2(ds[0].z, ds[0].w);
                *reinterpret_cast<uint2*>(img + ds_off(r, 2, h)) = make_uint2(ds[1].x, ds[1].y);
                *reinterpret_cast<uint2*>(img + ds_off(r, 3, h)) = make_uint2(ds[1].z, ds[1].w);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                read_ds_tr(a, img, lane);
                mma_packed(dQ[sub], a, kd);
                __builtin_amdgcn_wave_barrier();   // (the image is rewritten by the next block only after this read)
            }
        }
        store_acc(dKa, dKo, p.lddk, krow, kvalid, 32, h, kmul);
        store_acc(dVa, dVo, p.lddv, krow, kvalid, 32, h, 1.f);
        __syncthreads();                 // every wave is done with this key tile's image
        if (t + 1 < t1) {
            store_lds(sK, st, tid);
            __syncthreads();
        }
    }
    // dQ: the four waves' partials per 32-query block meet in LDS (over the Q / dO images: every wave passed the loop's last barrier), one
    // row group per thread quad, and leave as fp32 atomics into the zeroed [B, Lq, H, 32] image
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
        if (sub >= nsub) break;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) sRed[(wave * 32 + 8 * g + 4 * h + e) * 32 + r] = dQ[sub][4 * g + e];
        __syncthreads();
        {
            const int row = tid >> 3, d0 = (tid & 7) * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(sRed + row * 32 + d0);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(sRed + (w * 32 + row) * 32 + d0);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += u[e];
            }
            const int q = sub * 32 + row;
            if (q < p.Lq) {
                float* z = p.ws_dq + ((int64_t)b * p.Lq + q) * (p.H * 32) + hh * 32 + d0;
#pragma unroll
                for (int e = 0; e < 4; ++e) unsafeAtomicAdd(z + e, v[e]);
            }
        }
        __syncthreads();
    }
}

// before the single pass: delta = rowsum(dO * O); the row constants as plain fp32 vectors nl = -lse2, nd = -delta ([B,H,Lq], the
// loop DMAs 64 of them per wave instruction); the fp32 dQ image zeroed.  One thread per (row, head), 32 rows x H heads per block;
// the [B,H,Lq] side is read / written through LDS so that both sides of the transposition are coalesced.
// ZERO = false: the caller has zeroed the image already (SVOL_ATTN_DQ_PREZEROED: on another stream, under the previous launch's
// issue-bound single pass, where the 51 MB of stores cost nothing — here they are a third of this kernel's bytes on the critical path).
template <bool ZERO>
__global__ __launch_bounds__(256) void attn_bwd_sp_prep_bf16(Args p) {
    __shared__ float s_lse[8][33], s_dl[8][33];
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 32;          // global row (b * Lq + q); Lq % 32 == 0
    const int b = (int)(row0 / p.Lq), q0 = (int)(row0 % p.Lq);
    {
        const int hh = tid >> 5, rl = tid & 31;
        if (hh < p.H) s_lse[hh][rl] = p.lse2[((int64_t)b * p.H + hh) * p.Lq + q0 + rl];
    }
    const int rl = tid >> 3, hh = tid & 7;
    float dl = 0.f;
    if (hh < p.H) {
        const int64_t row = row0 + rl;
        const h16_t* o = reinterpret_cast<const h16_t*>(p.o) + row * p.ldo + hh * 32;
        const h16_t* d = reinterpret_cast<const h16_t*>(p.d_o) + row * p.lddo + hh * 32;
#pragma unroll
        for (int i = 0; i < 32; i += 8) {
            const h16x8 a = *reinterpret_cast<const h16x8*>(o + i), c = *reinterpret_cast<const h16x8*>(d + i);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += (float)a[e] * (float)c[e];
        }
        if constexpr (ZERO) {
            float* z = p.ws_dq + row * (p.H * 32) + hh * 32;
#pragma unroll
            for (int i = 0; i < 32; i += 4) *reinterpret_cast<f32x4*>(z + i) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        s_dl[hh][rl] = dl;
    }
    __syncthreads();
    {
        const int h2 = tid >> 5, r2 = tid & 31;
        if (h2 < p.H) {
            const int64_t si = ((int64_t)b * p.H + h2) * p.Lq + q0 + r2;
            reinterpret_cast<float*>(p.nl2)[si] = -s_lse[h2][r2];
            reinterpret_cast<float*>(p.nd2)[si] = -s_dl[h2][r2];
            p.delta[si] = s_dl[h2][r2];
        }
    }
}
// after it: dq (16-bit) = scale * fp32 image; dk / dv of a head's tail keys (Lk % 512 of them) = the sum of the four query-quarter
// partials the tail workgroups left behind the image (blocks past the image's)
__global__ __launch_bounds__(256) void attn_dq_round_bf16(Args p) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;   // 8 consecutive columns of one row
    const int64_t W = (int64_t)p.H * 32, total = (int64_t)p.B * p.Lq * W;
    if (i >= total) {
        const int tk = p.Lk % (4 * 128);
        const int64_t j = i - ((total + 2047) / 2048) * 2048;               // 8 consecutive d of one (head, dk | dv, key)
        if (tk == 0 || j < 0 || j >= (int64_t)p.B * p.H * 2 * tk * 32) return;
        const int64_t per = (int64_t)2 * tk * 32;
        const int bh = (int)(j / per), rem = (int)(j % per), kv = rem / (tk * 32), key = (rem / 32) % tk, d0 = rem % 32;
        const float* src = p.ws_dq + total + (int64_t)bh * 4 * per + rem;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = a;
#pragma unroll
        for (int part = 0; part < 4; ++part) { a += *reinterpret_cast<const f32x4*>(src + part * per); c += *reinterpret_cast<const f32x4*>(src + part * per + 4); }
        const float mul = kv == 0 ? p.scale / p.premul : 1.f;
        h16x8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = (h16_t)(a[e] * mul); v[4 + e] = (h16_t)(c[e] * mul); }
        const int b_ = bh / p.H, hh_ = bh % p.H;
        const int64_t row = (int64_t)b_ * p.Lk + (p.Lk - tk) + key;
        h16_t* o = kv == 0 ? reinterpret_cast<h16_t*>(p.dk) + row * p.lddk : reinterpret_cast<h16_t*>(p.dv) + row * p.lddv;
        *reinterpret_cast<h16x8*>(o + hh_ * 32 + d0) = v;
        return;
    }
    const int64_t row = i / W, c = i % W;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p.ws_dq + i), b = *reinterpret_cast<const f32x4*>(p.ws_dq + i + 4);
    h16x8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (h16_t)(a[e] * p.scale); v[4 + e] = (h16_t)(b[e] * p.scale); }
    *reinterpret_cast<h16x8*>(reinterpret_cast<h16_t*>(p.dq) + row * p.lddq + c) = v;
}

// the fp32 dQ image zeroed by itself (svol_attn_bwd_zero_ws): few registers, no LDS — its waves fit beside a single-pass workgroup's
__global__ __launch_bounds__(256) void attn_sp_zero_image(float* __restrict__ z, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<f32x4*>(z)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// Every MFMA of the single-pass kernel is inline asm: at one wave per SIMD (512 registers) hipcc selects the AGPR-destination form
// for the builtin, so score / dP tiles that the VALU exponentiates would pay a v_accvgpr_read per element.  Register classes by
// constraint: results the VALU touches "v", the dK / dV accumulators and the loop-invariant K / V fragments "a".
// One MFMA per statement, and the statements are placed BY HAND between chunks of vector work (sched_barrier(0) on both sides): a
// single wave issues in order, so a cluster of MFMAs stalls on the matrix pipe (32 cycles each) while the vector work behind it
// waits, and a cluster of vector work leaves the pipe idle — the first version of this kernel, scheduled by hipcc, ran the two as a
// plain sum (840 cycles per 32 x 32 block).  Per block: 10 MFMAs (80 issue cycles) + ~296 cycles of exp / multiply / convert / LDS
// issue, i.e. one MFMA every ~37 cycles with ~30 cycles of fillers behind it.
// HAZARDS are ours inside and behind an asm statement (hipcc pads nothing): a VALU-written operand needs 2 wait states before the
// MFMA that reads it (every such MFMA below sits >= 2 instructions behind the conversion that feeds it); an MFMA result must not be
// read by a non-MFMA instruction for passes + 4 wait states — every such consumer is at least four MFMAs behind its producer.
// tests/test_isa_hazards.py checks both distances in the emitted code.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ u32x4 to_acc(u32x4 r) {
    asm volatile("; operand -> AGPR" : "+a"(r));
    return r;
}
__device__ __forceinline__ u32x4 as_u32x4(const uint4& v) { return u32x4{v.x, v.y, v.z, v.w}; }
#define SP_FENCE() __builtin_amdgcn_sched_barrier(0)
// D (VGPRs) = A B + C, B in AGPRs, D and C different ranges
__device__ __forceinline__ void sp_mfma_c(f32x16& d, const uint4& a, const u32x4& b, const f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, %3" : "=&v"(d) : "v"(as_u32x4(a)), "a"(b), "v"(c));
}
// D (VGPRs) = A B (C = the inline constant 0), B in AGPRs
__device__ __forceinline__ void sp_mfma_z(f32x16& d, const uint4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, 0" : "=&v"(d) : "v"(as_u32x4(a)), "a"(b));
}
// D (VGPRs) += A B, B in AGPRs
__device__ __forceinline__ void sp_mfma_v(f32x16& d, const uint4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, %0" : "+v"(d) : "v"(as_u32x4(a)), "a"(b));
}
// acc (AGPRs) += A B, both operands in VGPRs
__device__ __forceinline__ void sp_mfma_a(f32x16& acc, const uint4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, %0" : "+a"(acc) : "v"(as_u32x4(a)), "v"(b));
}
__device__ __forceinline__ unsigned cvt_pk_u32(float a, float b) { return __builtin_bit_cast(unsigned, cvt_pk_h16(a, b)); }

// A operands (rows of Q and dO) and row constants (-lse, -delta as initial accumulators) of one 32-query step
struct SpStep {
    uint4 qa[2], doa[2];
    f32x16 Cl, Cd;
};
constexpr int SP_OFF_Q = 4 * KT * 4 + 512;        // LDS map: [2][KT] -lse2 | [2][KT] -delta | [KT] -inf | Q tiles | dO tiles | dS | partials
constexpr int SP_OFF_T = SP_OFF_Q + 4 * IMG;
constexpr int SP_OFF_P = SP_OFF_Q + 8 * IMG;

// NKB = 32-key blocks per wave: 4 in every full workgroup (512 keys); the last workgroup of a head whose key count is not a multiple
// of 512 spreads its 128 / 256 / 384 keys over all four waves (NKB = 1 / 2 / 3) instead of leaving waves without work.
//
// The kernel is ONE uniform stream of blocks j = step * NKB + kb (32 queries x 32 keys each).  Block j
//   * issues the score / dP products of block j + 1 (slots 1-4; across a step boundary they use the next step's operands, which were
//     fetched from LDS as soon as the last products of the current step had been issued),
//   * exponentiates, multiplies and converts its own tiles between the MFMAs and feeds dV / dK (slots 5, 7, 9, 10), writes its packed
//     dS to the wave's LDS image and reads it back transposed,
//   * runs the dQ product of block j - 2 (slots 6, 8): a transposed read is consumed a whole block after it was issued.
// The dQ partial of a step is therefore complete in block 1 of the NEXT step (block 0 two steps later when NKB = 1): it is written to LDS
// there, published by that step's barrier and added to the fp32 image (atomics) one step later.  DW + 1 extra DRAIN steps flush the
// pipeline: their row constant is -inf, so P = dS = 0 and the accumulators do not move — no epilogue code, no special cases.
// ABL: timing-only ablations for tools/micro/attn_lab_sp (results invalid): 1 no vector fillers, 2 no MFMAs, 4 no step barrier,
// 8 no dS round trip through LDS, 16 no tile DMA / waits.  The product instantiates ABL = 0 only.
//
// hh, b: head, batch.  kbase: first key of the workgroup.  [t_begin, t_end): the 128-query tiles it sweeps — all of them for a full
// workgroup; the tail workgroup of a head (fewer than 512 keys, NKB < 4) exists FOUR times, each sweeping a quarter of the tiles and
// leaving its dK / dV as an fp32 partial in `tail_out` (attn_dq_round_bf16 adds the four): a tail of 128 keys then costs a sixth of a
// full workgroup's time instead of 0.6 of it (measured), which is what the partial fourth round of workgroups cost the launch.
template <int NKB, int ABL = 0>
__device__ __forceinline__ void attn_bwd_sp_body(const Args& p, char* smem, int hh, int b, int kbase, int t_begin, int t_end, float* tail_out) {
    // every LDS-DMA destination (row constants, Q / dO tiles, the prologue's K / V staging) sits in the first 64 KiB: the existing
    // kernels never put an M0 base above 0xFFFF, and nothing here depends on how many bits of M0 the transfer honours
    float* sL = reinterpret_cast<float*>(smem);   // [2][KT] -lse2
    float* sD = sL + 2 * KT;                      // [2][KT] -delta
    float* sNeg = sL + 4 * KT;                    // [KT]    -inf (drain steps)
    char* sQ = smem + SP_OFF_Q;            // [2][IMG]   Q tiles (128 queries); dO tiles 2 * IMG behind
    char* sPart = smem + SP_OFF_P;         // [2][SP_PART] dQ partials
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    constexpr int dqw = 8 * 32;            // launcher: H == 8 (row pitch of the fp32 dQ image: immediates instead of address arithmetic)
    const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * 32;
    const h16_t* dO = reinterpret_cast<const h16_t*>(p.d_o) + (int64_t)b * p.Lq * p.lddo + hh * 32;
    const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * 32;
    const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * 32;
    const float* nl_g = reinterpret_cast<const float*>(p.nl2) + ((int64_t)b * p.H + hh) * p.Lq;
    const float* nd_g = reinterpret_cast<const float*>(p.nd2) + ((int64_t)b * p.H + hh) * p.Lq;
    const int key0 = kbase + wave * 32 * NKB;
    char* sT = smem + SP_OFF_T + wave * IMG;   // this wave's dS images: [NKB][32 keys][32 q]
    const int ntl = t_end - t_begin;       // tiles of this workgroup (>= 1)

    u32x4 kbk[NKB][2], vbk[NKB][2], kd[NKB][2];
    f32x16 dK[NKB], dV[NKB];
    {   // this wave's K rows, then its V rows, through ITS quarter of the (still unused) Q / dO buffers
        char* sS = sQ + wave * IMG;
#pragma unroll
        for (int pc = 0; pc < 2 * NKB; ++pc) dma_piece(sS, K, p.ldk, key0, pc, lane);
        dma_wait_all();
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            uint4 t0[2], t1[2];
            read_rows(t0, sS, kb * 32 + r, h);
            read_tr_nat(t1, sS, kb, lane);
#pragma unroll
            for (int s = 0; s < 2; ++s) { kbk[kb][s] = to_acc(as_u32x4(t0[s])); kd[kb][s] = to_acc(as_u32x4(t1[s])); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int pc = 0; pc < 2 * NKB; ++pc) dma_piece(sS, V, p.ldv, key0, pc, lane);
        dma_wait_all();
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            uint4 t0[2];
            read_rows(t0, sS, kb * 32 + r, h);
#pragma unroll
            for (int s = 0; s < 2; ++s) vbk[kb][s] = to_acc(as_u32x4(t0[s]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            dK[kb] = zero16();
            dV[kb] = zero16();
            asm volatile("; accumulators -> AGPR" : "+a"(dK[kb]), "+a"(dV[kb]));
        }
    }
    // both partial slots start as zeros (the first steps "reduce" empty buffers: every step issues the same four atomics, which
    // keeps the counted waits uniform); the -inf row constants of the drain steps
#pragma unroll
    for (int buf = 0; buf < 2; ++buf)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(sPart + buf * SP_PART + ((wave * 4 + g) * 64 + lane) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < KT) sNeg[tid] = -INFINITY;

    // ---- addresses: everything lane dependent is computed ONCE; a step adds scalars ----
    // LDS-DMA sources of this wave's two 16-row pieces of a tile (dma_piece's address arithmetic), tile t_begin
    const int prow = t_begin * KT + 32 * wave + (lane >> 2), pch = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    const h16_t* gq = Q + (int64_t)prow * p.ldq + pch;
    const h16_t* gdo = dO + (int64_t)prow * p.lddo + pch;
    const float* gst = (wave < 2 ? nl_g : nd_g) + t_begin * KT + (wave & 1) * 64 + (lane & 15) * 4;
    const unsigned lds_q = (unsigned)(size_t)(lds_vptr)sQ + wave * 2048, lds_st = (unsigned)(size_t)(lds_vptr)((wave < 2 ? sL : sD) + (wave & 1) * 64);
    auto dma_tile_at = [&](int tl, int buf) {   // local tile tl -> buffer buf: 5 vector-memory instructions per wave
        const h16_t* a = gq + (int64_t)tl * KT * p.ldq;
        const h16_t* c = gdo + (int64_t)tl * KT * p.lddo;
        const unsigned dq_ = __builtin_amdgcn_readfirstlane(lds_q + buf * IMG), dd_ = dq_ + 2 * IMG;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dq_), "v"(a) : "m0");
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dq_ + 1024), "v"(a + 16 * p.ldq) : "m0");
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dd_), "v"(c) : "m0");
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dd_ + 1024), "v"(c + 16 * p.lddo) : "m0");
        const unsigned ds_ = __builtin_amdgcn_readfirstlane(lds_st + buf * KT * 4);
        const float* e = gst + tl * KT;
        if (lane < 16) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(ds_), "v"(e) : "m0");
    };
    // byte offsets inside a tile image, without the 32-row sub-tile (+ sub * 2048): img_off(32 sub + x, ch) = 2048 sub + img_off(x, ch)
    // for every x < 32 — the swizzle only looks at bits 2-3 of the row
    const unsigned o_rows0 = img_off(r, h), o_rows1 = img_off(r, 2 + h);                     // read_rows, k-step 0 / 1
    unsigned o_trl, o_trh;                                                                     // read_tr: lo / hi row groups (k-step 1: + 1024)
    {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h2 = g >> 1;
        const int ch = 2 * (g & 1) + (pp >> 1), inner = 8 * (pp & 1);
        o_trl = img_off(4 * h2 + q, ch) + inner;
        o_trh = img_off(4 * h2 + q + 8, ch) + inner;
    }
    const unsigned o_st = 16 * h;                                                              // rows16: + 128 sub + 32 g
    auto tr_pair = [&](uint4 (&a)[2], const char* img) {   // = read_tr(a, tile image, sub, lane) with img = tile image + 2048 sub
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const h16x4 lo = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(img + o_trl + 1024 * s));
            const h16x4 hi = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(img + o_trh + 1024 * s));
            a[s] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    // operands of step (local tile tl, sub-tile sub); tiles past the end are DRAIN tiles: any resident rows, -inf as the score constant
    auto load_step = [&](SpStep& f, int tl, int sub, bool drain) {
        const int cur = (drain ? ntl - 1 : tl) & 1;
        const char* qi = sQ + cur * IMG + sub * 2048;
        f.qa[0] = *reinterpret_cast<const uint4*>(qi + o_rows0);
        f.qa[1] = *reinterpret_cast<const uint4*>(qi + o_rows1);
        f.doa[0] = *reinterpret_cast<const uint4*>(qi + 2 * IMG + o_rows0);
        f.doa[1] = *reinterpret_cast<const uint4*>(qi + 2 * IMG + o_rows1);
        const char* cl = drain ? reinterpret_cast<const char*>(sNeg) : reinterpret_cast<const char*>(sL) + cur * KT * 4 + sub * 128;
        const char* cd = reinterpret_cast<const char*>(sD) + cur * KT * 4 + sub * 128;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(cl + o_st + 32 * g), c = *reinterpret_cast<const f32x4*>(cd + o_st + 32 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) { f.Cl[4 * g + e] = a[e]; f.Cd[4 * g + e] = c[e]; }
        }
    };
    // the wave's dS image: where this lane writes its four 8-byte units, and where it reads them back transposed
    unsigned o_w[4], o_rl, o_rh;
#pragma unroll
    for (int g = 0; g < 4; ++g) o_w[g] = ds_off(r, g, h);
    {
        const int gg = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3, h2 = gg >> 1;
        o_rl = ds_off(8 * h2 + qq, 2 * (gg & 1) + (pp >> 1), pp & 1);          // k-step 1: + 1024 (keys + 16: same swizzle bits)
        o_rh = ds_off(8 * h2 + qq + 4, 2 * (gg & 1) + (pp >> 1), pp & 1);
    }
    // the sum over the four waves of one row group (rows 8 wave + 4 h + e of a step's 32 queries) of the partials in `buf`, added
    // to the fp32 image: lanes 0..31 / 32..63 = the 128 contiguous bytes of two rows
    float* dq_base = p.ws_dq + ((int64_t)b * p.Lq + (int64_t)t_begin * KT) * dqw + hh * 32;   // (uniform)
    const int dq_lane = (8 * wave + 4 * h) * dqw + r;
    const char* pr_base = sPart + (wave * 64 + lane) * 16;
    auto reduce_load = [&](f32x4 (&v)[4], int buf) {
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) v[w2] = *reinterpret_cast<const f32x4*>(pr_base + buf * SP_PART + w2 * 4 * 64 * 16);
    };
    // qstep: local step whose partial this is.  < 0: an empty buffer (the first steps); >= 4 ntl: a DRAIN step's partial — exact zeros
    // (P = exp2(-inf)).  Both are added to a row of this workgroup's own range: the drain steps' natural target lies past it — for the
    // last batch element past the END of the image (cfg5, B = 1: the image ends the allocation; a write fault on a read-only page).
    const int last_step = 4 * ntl - 1;
    auto reduce_add = [&](const f32x4 (&v)[4], int qstep) {
        const f32x4 acc = (v[0] + v[1]) + (v[2] + v[3]);
        float* dst = dq_base + (int64_t)min(max(qstep, 0), last_step) * 32 * dqw + dq_lane;
#ifdef SP_ABLATE   // lab only (tools/micro/attn_lab_sp -DSP_ABLATE=1): plain stores in place of the atomics — same instruction and vmcnt counts
#pragma unroll
        for (int e = 0; e < 4; ++e) __builtin_nontemporal_store(acc[e], dst + e * dqw);
#else
#pragma unroll
        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dst + e * dqw, acc[e]);
#endif
    };

    sp_barrier();                          // every wave is done with its staging quarter; zeros / -inf are in place
    dma_tile_at(0, 0);
    if (ntl > 1) dma_tile_at(1, 1);        // (with one wave per SIMD nothing hides a late tile: two tiles ahead from the start)
    dma_wait_all();
    sp_barrier();

    constexpr int DW = NKB == 1 ? 2 : 1;   // a step's dQ partial is written DW steps later
    constexpr int KBW = NKB == 1 ? 0 : 1;  //   ... at the end of this block
    constexpr int AHEAD = NKB == 1 ? 2 : 1;   // how many steps ahead the operand fetch runs
    SpStep f;                  // operands of the step whose products are issued next
    f32x16 S, dP;              // score - lse / dP - delta of the block that is processed next
    load_step(f, 0, 0, false);
    sp_mfma_c(S, f.qa[0], kbk[0][0], f.Cl);
    sp_mfma_v(S, f.qa[1], kbk[0][1]);
    sp_mfma_c(dP, f.doa[0], vbk[0][0], f.Cd);
    sp_mfma_v(dP, f.doa[1], vbk[0][1]);
    SP_FENCE();
    if (NKB == 1) load_step(f, 0, 1, false);   // (NKB == 1: a block's "next block" is always the next step's)
    uint4 dsa0[2], dsa1[2];                // dS^T fragments of the two previous blocks (older first)
#pragma unroll
    for (int s = 0; s < 2; ++s) dsa0[s] = dsa1[s] = make_uint4(0, 0, 0, 0);
    f32x16 dQp = zero16();

    // one iteration = one 128-query tile = four steps with a compile-time sub-tile; iteration ntl is the DRAIN tile (4 >= DW + 1 steps)
    for (int tl = 0; tl <= ntl; ++tl) {
        const bool drain = tl == ntl;
        const char* qtile = sQ + ((drain ? ntl - 1 : tl) & 1) * IMG;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
        const int st = 4 * tl + sub;
        if (!(ABL & 16) && sub == 0 && tl >= 1 && tl + 1 < ntl) {   // the other tile buffers were last read in the step before, behind its barrier
            dma_tile_at(tl + 1, (tl + 1) & 1);
            asm volatile("" ::: "memory");   // the atomics below stay BEHIND these five transfers (the counted vmcnt below relies on it)
        }
        const int subn = (sub + AHEAD) & 3, tn = tl + (sub + AHEAD >= 4 ? 1 : 0);   // the step whose operands this step fetches
        uint4 qt[2], dot[2];               // transposed A operands of this step's dV / dK products
        tr_pair(qt, qtile + sub * 2048);
        tr_pair(dot, qtile + 2 * IMG + sub * 2048);
        f32x4 red[4];
        reduce_load(red, (sub - DW - 1) & 1);   // the partial published by the last barrier
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int kn = kb + 1 < NKB ? kb + 1 : 0;    // key block of the NEXT block in the stream
            const int ks = (kb + 2 * NKB - 2) % NKB;     // key block whose dQ product runs here (the block two back in the stream)
            f32x16 Sn, dPn;
            u32x4 pw0, pw1, dw0, dw1;      // P / dS of this block, packed: the B operands of the dV / dK products
            char* img = sT + kb * 2048;
            // Pure vector instructions carry no ordering against an asm statement or a sched_barrier in instruction selection — hipcc
            // moved conversions right in front of the MFMA that reads them (VALU write -> MFMA read hazard: NaNs in dK / dV) and
            // emptied the filler slots; pinning them with empty asm statements costs an s_nop behind every statement.  So the fillers
            // are asm volatile too: volatile statements keep their order, the instruction stream of a block IS the source below.
            // (Hazards then ours: v_exp result -> VALU read 1 wait state, VALU write -> MFMA operand 2: see the slot comments.)
#define SP_SLOT(stmt) do { SP_FENCE(); if (!(ABL & 2)) { stmt; } SP_FENCE(); } while (0)
#define SP_E(i) do { if (!(ABL & 1)) asm volatile("v_exp_f32 %0, %0" : "+v"(S[i])); } while (0)
#define SP_CP(j) if (!(ABL & 1)) asm volatile("v_cvt_pk_" SVOL_H16_ASM "_f32 %0, %1, %2" : "=v"(((j) < 4 ? pw0 : pw1)[(j) & 3]) : "v"(S[2 * (j)]), "v"(S[2 * (j) + 1]))
#define SP_M2(i) do { if (ABL & 1) break; f32x2 d_ = {dP[i], dP[(i) + 1]}; const f32x2 s_ = {S[i], S[(i) + 1]}; asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d_) : "v"(s_)); dP[i] = d_[0]; dP[(i) + 1] = d_[1]; } while (0)
#define SP_CD(j) if (!(ABL & 1)) asm volatile("v_cvt_pk_" SVOL_H16_ASM "_f32 %0, %1, %2" : "=v"(((j) < 4 ? dw0 : dw1)[(j) & 3]) : "v"(dP[2 * (j)]), "v"(dP[2 * (j) + 1]))
#define SP_W(g) if (!(ABL & 8)) *reinterpret_cast<uint2*>(img + o_w[g]) = (g) < 2 ? make_uint2(dw0[2 * ((g) & 1)], dw0[2 * ((g) & 1) + 1]) \
                                                                  : make_uint2(dw1[2 * ((g) & 1)], dw1[2 * ((g) & 1) + 1])
            // slot 1
            SP_SLOT(sp_mfma_c(Sn, f.qa[0], kbk[kn][0], f.Cl));
            SP_E(0); SP_E(1); SP_E(2); SP_E(3);
            // slot 2
            SP_SLOT(sp_mfma_v(Sn, f.qa[1], kbk[kn][1]));
            SP_CP(0); SP_CP(1); SP_E(4); SP_E(5); SP_M2(0);
            // slot 3
            SP_SLOT(sp_mfma_c(dPn, f.doa[0], vbk[kn][0], f.Cd));
            SP_E(6); SP_E(7); SP_CP(2); SP_CP(3);
            // slot 4
            SP_SLOT(sp_mfma_v(dPn, f.doa[1], vbk[kn][1]));
            SP_M2(2); SP_E(8); SP_E(9); SP_CD(0); SP_CD(1); SP_M2(4);
            // the products that read this step's operands are all issued: fetch the operands of the step after it
            if (NKB == 1 || kb == NKB - 2) load_step(f, tn, subn, tn >= ntl);
            if (kb == NKB - 1) reduce_add(red, st - DW - 1);   // (atomics of the partial published by the last barrier)
            // slot 5: dV^T += dO^T P, first k-step (P registers 0..7)
            SP_SLOT(sp_mfma_a(dV[kb], dot[0], pw0));
            SP_E(10); SP_E(11); SP_CP(4); SP_M2(6);
            // slot 6: dQ += dS K of the block two back, first k-step
            SP_SLOT(if (ks == 0) sp_mfma_z(dQp, dsa0[0], kd[0][0]); else sp_mfma_v(dQp, dsa0[0], kd[ks][0]));
            SP_CD(2); SP_CD(3); SP_E(12); SP_CP(5); SP_M2(8);
            // slot 7: dK^T += Q^T dS, first k-step (dS registers 0..7; the last conversion is two instructions back)
            SP_SLOT(sp_mfma_a(dK[kb], qt[0], dw0));
            SP_E(13); SP_E(14); SP_E(15); SP_CP(6);
            // slot 8
            SP_SLOT(sp_mfma_v(dQp, dsa0[1], kd[ks][1]));
            SP_CP(7); SP_M2(10); SP_M2(12); SP_M2(14); SP_CD(4); SP_CD(5);
            // slot 9: dV, second k-step
            SP_SLOT(sp_mfma_a(dV[kb], dot[1], pw1));
            SP_CD(6); SP_CD(7);
            // (the partial stays LIVE up to here on every path: where the write below is skipped — hipcc peels the first steps — a dead
            // MFMA result would have its registers reused while the MFMA is still in flight: a write-after-write hazard nobody pads)
            if (kb == KBW) asm volatile("" ::"v"(dQp));
            if (kb == KBW && st >= DW) {   // the dQ partial of step st - DW is complete (slot 8 was its last product, 7+ instructions back)
                char* pwr = sPart + ((sub - DW) & 1) * SP_PART + (wave * 4 * 64 + lane) * 16;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(pwr + g * 64 * 16) = f32x4{dQp[4 * g], dQp[4 * g + 1], dQp[4 * g + 2], dQp[4 * g + 3]};
            }
            SP_W(0); SP_W(1);
            // slot 10: dK, second k-step
            SP_SLOT(sp_mfma_a(dK[kb], qt[1], dw1));
            SP_W(2); SP_W(3);
#pragma unroll
            for (int s = 0; s < 2; ++s) { dsa0[s] = dsa1[s]; }
#pragma unroll
            for (int s = 0; s < ((ABL & 8) ? 0 : 2); ++s) {  // this block's dS, transposed: consumed in slots 6 / 8 of the block after next
                const h16x4 lo = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(img + o_rl + 1024 * s));
                const h16x4 hi = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(img + o_rh + 1024 * s));
                dsa1[s] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#undef SP_SLOT
#undef SP_E
#undef SP_CP
#undef SP_M2
#undef SP_CD
#undef SP_W
            S = Sn;
            dP = dPn;
        }
        // tile tl + 1 (5 LDS-DMA instructions, issued at the top of this tile) is first read when the operands of its step 0 are
        // fetched: two steps ahead with one block per wave, in the last step of this tile otherwise; 4 atomics per step behind it
        if (ABL & 16) { }
        else if (sub == (NKB == 1 ? 1 : 2) && tl >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NKB == 1 ? 8 : 12) : "memory");
        // the partial written in block KBW must be in LDS before the barrier.  LDS operations of a wave complete in order and the
        // step's last eight are W(0..3) and the four transposed reads of its last block: at most those may still be in flight
        if (ABL & 4) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(8)\n\ts_barrier" ::: "memory");
        }
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // the last asm MFMAs' results are read by compiler-generated code below
    if (tail_out) {   // fp32 partial [2][keys of the workgroup][32]: raw sums, attn_dq_round_bf16 adds the four parts and scales
        const int nk = 4 * 32 * NKB;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            float* ok = tail_out + (int64_t)(wave * 32 * NKB + kb * 32 + r) * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<f32x4*>(ok + 8 * g + 4 * h) = f32x4{dK[kb][4 * g], dK[kb][4 * g + 1], dK[kb][4 * g + 2], dK[kb][4 * g + 3]};
                *reinterpret_cast<f32x4*>(ok + (int64_t)nk * 32 + 8 * g + 4 * h) = f32x4{dV[kb][4 * g], dV[kb][4 * g + 1], dV[kb][4 * g + 2], dV[kb][4 * g + 3]};
            }
        }
        return;
    }
    h16_t* dKo = reinterpret_cast<h16_t*>(p.dk) + (int64_t)b * p.Lk * p.lddk + hh * 32;
    h16_t* dVo = reinterpret_cast<h16_t*>(p.dv) + (int64_t)b * p.Lk * p.lddv + hh * 32;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        store_acc(dK[kb], dKo, p.lddk, key0 + kb * 32 + r, true, 32, h, p.scale / p.premul);
        store_acc(dV[kb], dVo, p.lddv, key0 + kb * 32 + r, true, 32, h, 1.f);
    }
}
// ---- round 5: the body of a FULL workgroup (four 32-key blocks per wave), every instruction re-placed by the issue rules measured
// with tools/micro/gen_mfma_fillers.py (profiles/round5_mfma_fillers.md).  Same algorithm, same LDS map and scratch as the template
// above (which keeps serving the tail groups); what changed is WHERE things issue:
//   * a gap (the instructions between two MFMAs) hides 24 issue cycles: exp 8, cvt_pk 5, mul / add 4.  Every gap carries 21-30.
//   * v_pk_mul_f32 does not overlap an MFMA at all (one per gap: 32 -> 50 cycles): plain v_mul_f32.
//   * an LDS operation costs a lone wave 5 cycles right BEHIND an MFMA and 14-25 in front of the next one; stores go through a
//     path the four SIMDs share (24 cycles per ds_write_b64 when all four waves store): every LDS operation is the FIRST
//     instruction of a gap, at most one store per gap, the step's operand / constant / partial traffic is dealt over the gaps of
//     the four blocks instead of standing in clusters at the step boundary.
//   * the exponentials of block j + 1 start in the last two gaps of block j (its scores are complete by then), so no gap runs
//     short of vector work and none has to take more than two exps.
//   * the dQ product runs ONE block behind (was two): the transposed read-back of dS is issued in gap 10 / gap 1 and consumed by
//     the MFMAs behind gaps 5 / 7 — five gaps of latency cover; 8 registers less.
// Gap k = what follows MFMA k.  MFMAs: 1, 3 score of the next block, 2, 4 its dP (alternating: no product waits for the accumulator of
// the MFMA right in front of it), 5 / 9 dV, 7 / 10 dK, 6 / 8 dQ of the previous block.
template <int ABL = 0>
__device__ __forceinline__ void attn_bwd_sp_body4(const Args& p, char* smem, int hh, int b, int kbase, int t_begin, int t_end) {
    constexpr int NKB = 4;
    float* sL = reinterpret_cast<float*>(smem);   // [2][KT] -lse2
    float* sD = sL + 2 * KT;                      // [2][KT] -delta
    float* sNeg = sL + 4 * KT;                    // [KT]    -inf (drain steps)
    char* sQ = smem + SP_OFF_Q;            // [2][IMG]   Q tiles (128 queries); dO tiles 2 * IMG behind
    char* sPart = smem + SP_OFF_P;         // [2][SP_PART] dQ partials
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    constexpr int dqw = 8 * 32;
    const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * 32;
    const h16_t* dO = reinterpret_cast<const h16_t*>(p.d_o) + (int64_t)b * p.Lq * p.lddo + hh * 32;
    const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * 32;
    const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * 32;
    const float* nl_g = reinterpret_cast<const float*>(p.nl2) + ((int64_t)b * p.H + hh) * p.Lq;
    const float* nd_g = reinterpret_cast<const float*>(p.nd2) + ((int64_t)b * p.H + hh) * p.Lq;
    const int key0 = kbase + wave * 32 * NKB;
    char* sT = smem + SP_OFF_T + wave * IMG;   // this wave's dS images: [NKB][32 keys][32 q]
    const int ntl = t_end - t_begin;       // tiles of this workgroup (>= 1)

    u32x4 kbk[NKB][2], vbk[NKB][2], kd[NKB][2];
    f32x16 dK[NKB], dV[NKB];
    {   // this wave's K rows, then its V rows, through ITS quarter of the (still unused) Q / dO buffers
        char* sS = sQ + wave * IMG;
#pragma unroll
        for (int pc = 0; pc < 2 * NKB; ++pc) dma_piece(sS, K, p.ldk, key0, pc, lane);
        dma_wait_all();
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            uint4 t0[2], t1[2];
            read_rows(t0, sS, kb * 32 + r, h);
            read_tr_nat(t1, sS, kb, lane);
#pragma unroll
            for (int s = 0; s < 2; ++s) { kbk[kb][s] = to_acc(as_u32x4(t0[s])); kd[kb][s] = to_acc(as_u32x4(t1[s])); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int pc = 0; pc < 2 * NKB; ++pc) dma_piece(sS, V, p.ldv, key0, pc, lane);
        dma_wait_all();
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            uint4 t0[2];
            read_rows(t0, sS, kb * 32 + r, h);
#pragma unroll
            for (int s = 0; s < 2; ++s) vbk[kb][s] = to_acc(as_u32x4(t0[s]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            dK[kb] = zero16();
            dV[kb] = zero16();
            asm volatile("; accumulators -> AGPR" : "+a"(dK[kb]), "+a"(dV[kb]));
        }
    }
#pragma unroll
    for (int buf = 0; buf < 2; ++buf)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(sPart + buf * SP_PART + ((wave * 4 + g) * 64 + lane) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < KT) sNeg[tid] = -INFINITY;
    // the wave's dS images start as zeros: the stream's first block reads "the previous block's" dS^T (image 3) before anything wrote it
#pragma unroll
    for (int g = 0; g < IMG / 1024; ++g) *reinterpret_cast<f32x4*>(sT + (g * 64 + lane) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- addresses: everything lane dependent is computed ONCE ----
    const int prow = t_begin * KT + 32 * wave + (lane >> 2), pch = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    const h16_t* gq = Q + (int64_t)prow * p.ldq + pch;
    const h16_t* gdo = dO + (int64_t)prow * p.lddo + pch;
    const float* gst = (wave < 2 ? nl_g : nd_g) + t_begin * KT + (wave & 1) * 64 + (lane & 15) * 4;
    const unsigned lds_q = (unsigned)(size_t)(lds_vptr)sQ + wave * 2048, lds_st = (unsigned)(size_t)(lds_vptr)((wave < 2 ? sL : sD) + (wave & 1) * 64);
    auto dma_tile_at = [&](int tl, int buf) {   // local tile tl -> buffer buf: 5 vector-memory instructions per wave
        const h16_t* a = gq + (int64_t)tl * KT * p.ldq;
        const h16_t* c = gdo + (int64_t)tl * KT * p.lddo;
        const unsigned dq_ = __builtin_amdgcn_readfirstlane(lds_q + buf * IMG), dd_ = dq_ + 2 * IMG;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dq_), "v"(a) : "m0");
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dq_ + 1024), "v"(a + 16 * p.ldq) : "m0");
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dd_), "v"(c) : "m0");
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dd_ + 1024), "v"(c + 16 * p.lddo) : "m0");
        const unsigned ds_ = __builtin_amdgcn_readfirstlane(lds_st + buf * KT * 4);
        const float* e = gst + tl * KT;
        if (lane < 16) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(ds_), "v"(e) : "m0");
    };
    const unsigned o_rows0 = img_off(r, h), o_rows1 = img_off(r, 2 + h);
    unsigned o_trl, o_trh;
    {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h2 = g >> 1;
        const int ch = 2 * (g & 1) + (pp >> 1), inner = 8 * (pp & 1);
        o_trl = img_off(4 * h2 + q, ch) + inner;
        o_trh = img_off(4 * h2 + q + 8, ch) + inner;
    }
    const unsigned o_st = 16 * h;
    // one k-step (s) of a transposed A operand: two ds_read_b64_tr_b16
    auto tr_one = [&](uint4& a, const char* img, int s) {
        const h16x4 lo = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(img + o_trl + 1024 * s));
        const h16x4 hi = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(img + o_trh + 1024 * s));
        a = __builtin_bit_cast(uint4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // halves (two 16-byte reads) of a step's row constants; c = the constant vector's base for this step
    auto load_c2 = [&](f32x16& C, const char* c, int part) {
#pragma unroll
        for (int g = 2 * part; g < 2 * part + 2; ++g) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(c + o_st + 32 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) C[4 * g + e] = a[e];
        }
    };
    auto load_c1 = [&](f32x16& C, const char* c, int g) {   // one 16-byte read: rows 8 g + 4 h .. + 3 of the step
        const f32x4 a = *reinterpret_cast<const f32x4*>(c + o_st + 32 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) C[4 * g + e] = a[e];
    };
    unsigned o_w[4], o_rl, o_rh;
#pragma unroll
    for (int g = 0; g < 4; ++g) o_w[g] = ds_off(r, g, h);
    {
        const int gg = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3, h2 = gg >> 1;
        o_rl = ds_off(8 * h2 + qq, 2 * (gg & 1) + (pp >> 1), pp & 1);
        o_rh = ds_off(8 * h2 + qq + 4, 2 * (gg & 1) + (pp >> 1), pp & 1);
    }
    auto ds_tr_one = [&](uint4& a, const char* img, int s) {   // k-step s of dS^T from a [32 keys][32 q] image
        const h16x4 lo = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(img + o_rl + 1024 * s));
        const h16x4 hi = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(img + o_rh + 1024 * s));
        a = __builtin_bit_cast(uint4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    float* dq_base = p.ws_dq + ((int64_t)b * p.Lq + (int64_t)t_begin * KT) * dqw + hh * 32;   // (uniform)
    const int dq_lane = (8 * wave + 4 * h) * dqw + r;
    const char* pr_base = sPart + (wave * 64 + lane) * 16;
    const int last_step = 4 * ntl - 1;

    sp_barrier();                          // every wave is done with its staging quarter; zeros / -inf are in place
    dma_tile_at(0, 0);
    if (ntl > 1) dma_tile_at(1, 1);
    dma_wait_all();
    sp_barrier();

    // operands of the step whose score / dP products are issued next, the transposed operands of the current step
    uint4 qa[2], doa[2], qt[2], dot[2];
    f32x16 Cl, Cd;
    f32x16 S, dP;              // score - lse / dP - delta of the block that is processed next
    {
        const char* qi = sQ;
        qa[0] = *reinterpret_cast<const uint4*>(qi + o_rows0);
        qa[1] = *reinterpret_cast<const uint4*>(qi + o_rows1);
        doa[0] = *reinterpret_cast<const uint4*>(qi + 2 * IMG + o_rows0);
        doa[1] = *reinterpret_cast<const uint4*>(qi + 2 * IMG + o_rows1);
        load_c2(Cl, reinterpret_cast<const char*>(sL), 0);
        load_c2(Cl, reinterpret_cast<const char*>(sL), 1);
        load_c2(Cd, reinterpret_cast<const char*>(sD), 0);
        load_c2(Cd, reinterpret_cast<const char*>(sD), 1);
        tr_one(qt[0], qi, 0);
        tr_one(qt[1], qi, 1);
        tr_one(dot[0], qi + 2 * IMG, 0);
        tr_one(dot[1], qi + 2 * IMG, 1);
    }
    sp_mfma_c(S, qa[0], kbk[0][0], Cl);
    sp_mfma_v(S, qa[1], kbk[0][1]);
    sp_mfma_c(dP, doa[0], vbk[0][0], Cd);
    sp_mfma_v(dP, doa[1], vbk[0][1]);
    SP_FENCE();
    uint4 dsa[2];                          // dS^T fragments of the previous block
#pragma unroll
    for (int s = 0; s < 2; ++s) dsa[s] = make_uint4(0, 0, 0, 0);
    f32x16 dQp = zero16();
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(S), "+v"(dP), "+v"(dQp));   // the four products above are complete
#define SP4_E(X, i) do { if (!(ABL & 1)) asm volatile("v_exp_f32 %0, %0" : "+v"(X[i])); } while (0)
#define SP4_M(i) do { if (!(ABL & 1)) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(dP[i]) : "v"(S[i])); } while (0)
#define SP4_CP(j) do { if (!(ABL & 1)) asm volatile("v_cvt_pk_" SVOL_H16_ASM "_f32 %0, %1, %2" : "=v"(((j) < 4 ? pw0 : pw1)[(j) & 3]) : "v"(S[2 * (j)]), "v"(S[2 * (j) + 1])); } while (0)
#define SP4_CD(j) do { if (!(ABL & 1)) asm volatile("v_cvt_pk_" SVOL_H16_ASM "_f32 %0, %1, %2" : "=v"(((j) < 4 ? dw0 : dw1)[(j) & 3]) : "v"(dP[2 * (j)]), "v"(dP[2 * (j) + 1])); } while (0)
#define SP4_W(g) do { if (!(ABL & 8)) *reinterpret_cast<uint2*>(img + o_w[g]) = (g) < 2 ? make_uint2(dw0[2 * ((g) & 1)], dw0[2 * ((g) & 1) + 1]) \
                                                                             : make_uint2(dw1[2 * ((g) & 1)], dw1[2 * ((g) & 1) + 1]); } while (0)
#define SP4_MF(stmt) do { SP_FENCE(); if (!(ABL & 2)) { stmt; } SP_FENCE(); } while (0)
#ifdef SP_ABLATE   // lab only: plain stores in place of the atomics — same instruction and vmcnt counts
#define SP4_ATOM(ptr, val) __builtin_nontemporal_store(val, ptr)
#else
#define SP4_ATOM(ptr, val) unsafeAtomicAdd(ptr, val)
#endif
#define SP4_ADD(d, x, y) do { _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) { float o_; asm volatile("v_add_f32 %0, %1, %2" : "=v"(o_) : "v"((x)[e_]), "v"((y)[e_])); (d)[e_] = o_; } } while (0)
    SP4_E(S, 0); SP4_E(S, 1); SP4_E(S, 2); SP4_E(S, 3); SP4_E(S, 4);   // (the loop does these five in the last two gaps of the block before)
    f32x4 red[4], rsum;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) red[w2] = f32x4{0.f, 0.f, 0.f, 0.f};
    rsum = red[0];
    u32x4 pw0, pw1, dw0, dw1;              // P / dS of the current block, packed: the B operands of the dV / dK products

    // one iteration = one 128-query tile = four steps with a compile-time sub-tile; iteration ntl is the DRAIN tile
    for (int tl = 0; tl <= ntl; ++tl) {
        const bool drain = tl == ntl;
        const char* qtile = sQ + ((drain ? ntl - 1 : tl) & 1) * IMG;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
        const int st = 4 * tl + sub;
        if (!(ABL & 16) && sub == 0 && tl >= 1 && tl + 1 < ntl) {   // the other tile buffers were last read in the step before, behind its barrier
            dma_tile_at(tl + 1, (tl + 1) & 1);
            asm volatile("" ::: "memory");   // the atomics below stay BEHIND these five transfers (the counted vmcnt below relies on it)
        }
        // the NEXT step: its operands are fetched during this one (tiles past the end are DRAIN tiles: any resident rows, -inf)
        const int subn = (sub + 1) & 3, tn = tl + (sub + 1 >= 4 ? 1 : 0);
        const bool drain_n = tn >= ntl;
        const int cur_n = (drain_n ? ntl - 1 : tn) & 1;
        const char* qn = sQ + cur_n * IMG + subn * 2048;                      // Q rows of the next step (dO: + 2 IMG)
        const char* cln = drain_n ? reinterpret_cast<const char*>(sNeg) : reinterpret_cast<const char*>(sL) + cur_n * KT * 4 + subn * 128;
        const char* cdn = reinterpret_cast<const char*>(sD) + cur_n * KT * 4 + subn * 128;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int kn = (kb + 1) & 3;         // key block of the NEXT block in the stream
            const int ks = (kb + 3) & 3;         // key block whose dQ product runs here (the previous block of the stream)
            f32x16 Sn, dPn;
            char* img = sT + kb * 2048;
            const char* img_prev = sT + ks * 2048;
            // ---- gap 1
            SP4_MF(sp_mfma_c(Sn, qa[0], kbk[kn][0], Cl));
            if (!(ABL & 8)) ds_tr_one(dsa[1], img_prev, 1);
            if (kb == 1) *reinterpret_cast<f32x4*>(sPart + ((sub - 1) & 1) * SP_PART + ((wave * 4 + 0) * 64 + lane) * 16) = f32x4{dQp[0], dQp[1], dQp[2], dQp[3]};
            if (kb == 2) load_c2(Cl, cln, 0);
            SP_FENCE();
            SP4_E(S, 5); SP4_E(S, 6); SP4_CP(0); SP4_CP(1); SP4_M(0);
            // ---- gap 2
            SP4_MF(sp_mfma_c(dPn, doa[0], vbk[kn][0], Cd));
            if (kb == 0) tr_one(qt[1], qtile + sub * 2048, 1);
            if (kb == 1) *reinterpret_cast<f32x4*>(sPart + ((sub - 1) & 1) * SP_PART + ((wave * 4 + 1) * 64 + lane) * 16) = f32x4{dQp[4], dQp[5], dQp[6], dQp[7]};
            if (kb == 2) load_c2(Cl, cln, 1);
            if (kb == 3) {   // the sum of the partial published by the last barrier -> fp32 image (rows of this workgroup only: see the template)
                float* dst = dq_base + (int64_t)min(max(st - 2, 0), last_step) * 32 * dqw + dq_lane;
                SP4_ATOM(dst, rsum[0]);
                SP4_ATOM(dst + dqw, rsum[1]);
            }
            SP_FENCE();
            SP4_E(S, 7); SP4_M(1); SP4_M(2); SP4_M(3); SP4_CD(0); SP4_CP(2);
            // ---- gap 3
            SP4_MF(sp_mfma_v(Sn, qa[1], kbk[kn][1]));
            if (kb == 1) *reinterpret_cast<f32x4*>(sPart + ((sub - 1) & 1) * SP_PART + ((wave * 4 + 2) * 64 + lane) * 16) = f32x4{dQp[8], dQp[9], dQp[10], dQp[11]};
            if (kb == 2) load_c2(Cd, cdn, 0);   // (the dP product of gap 2 was the old constants' last reader)
            if (kb == 3) {
                float* dst = dq_base + (int64_t)min(max(st - 2, 0), last_step) * 32 * dqw + dq_lane;
                SP4_ATOM(dst + 2 * dqw, rsum[2]);
                SP4_ATOM(dst + 3 * dqw, rsum[3]);
            }
            SP_FENCE();
            SP4_E(S, 8); SP4_E(S, 9); SP4_M(4); SP4_M(5); SP4_CD(1);
            // ---- gap 4
            SP4_MF(sp_mfma_v(dPn, doa[1], vbk[kn][1]));
            SP4_W(0);
            if (kb == 2) load_c1(Cd, cdn, 2);
            SP_FENCE();
            SP4_CP(3); SP4_E(S, 10); SP4_M(6); SP4_M(7); SP4_CD(2);
            // ---- gap 5: dV^T += dO^T P, first k-step (P registers 0..7; the last conversion is five instructions back)
            SP4_MF(sp_mfma_a(dV[kb], dot[0], pw0));
            if (kb == 1) *reinterpret_cast<f32x4*>(sPart + ((sub - 1) & 1) * SP_PART + ((wave * 4 + 3) * 64 + lane) * 16) = f32x4{dQp[12], dQp[13], dQp[14], dQp[15]};
            if (kb == 2) {
                load_c1(Cd, cdn, 3);
                qa[0] = *reinterpret_cast<const uint4*>(qn + o_rows0);
            }
            if (kb == 3) tr_one(dot[0], qn + 2 * IMG, 0);
            SP_FENCE();
            SP4_E(S, 11); SP4_E(S, 12); SP4_CD(3); SP4_CP(4); SP4_M(8);
            // ---- gap 6: dQ += dS K of the previous block, first k-step (the partial restarts with the step's first key block)
            SP4_MF(if (ks == 0) sp_mfma_z(dQp, dsa[0], kd[0][0]); else sp_mfma_v(dQp, dsa[0], kd[ks][0]));
            SP4_W(1);
            if (kb == 2) qa[1] = *reinterpret_cast<const uint4*>(qn + o_rows1);
            SP_FENCE();
            SP4_E(S, 13); SP4_M(9); SP4_M(10); SP4_M(11); SP4_CD(4); SP4_CP(5);
            // ---- gap 7: dK^T += Q^T dS, first k-step (dS registers 0..7: the last conversion is a whole gap back)
            SP4_MF(sp_mfma_a(dK[kb], qt[0], dw0));
            if (kb == 1) {
                red[0] = *reinterpret_cast<const f32x4*>(pr_base + (sub & 1) * SP_PART + 0 * 4 * 64 * 16);
                red[1] = *reinterpret_cast<const f32x4*>(pr_base + (sub & 1) * SP_PART + 1 * 4 * 64 * 16);
            }
            if (kb == 2) {
                doa[0] = *reinterpret_cast<const uint4*>(qn + 2 * IMG + o_rows0);
                doa[1] = *reinterpret_cast<const uint4*>(qn + 2 * IMG + o_rows1);
            }
            if (kb == 3) tr_one(qt[0], qn, 0);
            SP_FENCE();
            SP4_E(S, 14); SP4_E(S, 15); SP4_M(12); SP4_M(13); SP4_CD(5);
            // ---- gap 8
            SP4_MF(sp_mfma_v(dQp, dsa[1], kd[ks][1]));
            SP4_W(2);
            SP_FENCE();
            SP4_CP(6); SP4_CP(7); SP4_M(14); SP4_M(15); SP4_CD(6);
            if (kb == 2) { SP4_ADD(rsum, red[0], red[1]); }   // (asm: hipcc packs a vector add into v_pk_add_f32, which no MFMA gap hides)
            // ---- gap 9: dV, second k-step (P registers 8..15; the last conversion is four instructions back)
            SP4_MF(sp_mfma_a(dV[kb], dot[1], pw1));
            if (kb == 1) {
                red[2] = *reinterpret_cast<const f32x4*>(pr_base + (sub & 1) * SP_PART + 2 * 4 * 64 * 16);
                red[3] = *reinterpret_cast<const f32x4*>(pr_base + (sub & 1) * SP_PART + 3 * 4 * 64 * 16);
            }
            if (kb == 3) tr_one(dot[1], qn + 2 * IMG, 1);
            SP_FENCE();
            SP4_CD(7); SP4_E(Sn, 0); SP4_E(Sn, 1);
            if (kb == 2) { SP4_ADD(rsum, rsum, red[2]); }
            // ---- gap 10: dK, second k-step
            SP4_MF(sp_mfma_a(dK[kb], qt[1], dw1));
            SP4_W(3);
            if (!(ABL & 8)) ds_tr_one(dsa[0], img, 0);
            SP_FENCE();
            SP4_E(Sn, 2); SP4_E(Sn, 3); SP4_E(Sn, 4);
            if (kb == 2) { SP4_ADD(rsum, rsum, red[3]); }
            S = Sn;
            dP = dPn;
        }
        if (ABL & 16) { }
        else if (sub == 2 && tl >= 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        // the partial written in block 1 must be in LDS before the barrier: LDS operations of a wave complete in order and more than
        // eight follow those stores inside the step
        if (ABL & 4) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(8)\n\ts_barrier" ::: "memory");
        }
    }
#undef SP4_E
#undef SP4_M
#undef SP4_CP
#undef SP4_CD
#undef SP4_W
#undef SP4_MF
#undef SP4_ADD
#undef SP4_ATOM
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // the last asm MFMAs' results are read by compiler-generated code below
    h16_t* dKo = reinterpret_cast<h16_t*>(p.dk) + (int64_t)b * p.Lk * p.lddk + hh * 32;
    h16_t* dVo = reinterpret_cast<h16_t*>(p.dv) + (int64_t)b * p.Lk * p.lddv + hh * 32;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        store_acc(dK[kb], dKo, p.lddk, key0 + kb * 32 + r, true, 32, h, p.scale / p.premul);
        store_acc(dV[kb], dVo, p.lddv, key0 + kb * 32 + r, true, 32, h, 1.f);
    }
}
// ---- round 5: the fast forward as a hand-placed stream (same algorithm as attn_fwd_bf16_fast: anchored once, LDS-DMA tiles,
// overflow flag + attn_fwd_bf16_pre behind it).  attn_fwd_bf16_fast is hipcc-scheduled: per 128-key tile it issues, beside 16 MFMAs,
// 64 exp and 32 cvt_pk, 26 v_pk_add_f32 (row sums in register pairs), 31 v_mov (to build those pairs) and 16 s_nop — and a packed
// fp32 instruction does not overlap an MFMA at all (profiles/round5_mfma_fillers.md): ~390 cycles per 32 x 32 block against an issue
// floor of ~264 (4 MFMAs x 8 + 16 exp x 8 + 16 add x 4 + 8 cvt x 5).  Here every MFMA and every vector instruction is an asm statement
// in source order: a block = [V^T fragments + the next block's K rows from LDS] E0-7 | S' k-step 0 | E8-11 A0-7 | S' k-step 1 |
// C0-3 E12-15 | PV k-step 0 | C4-7 A8-15 | PV k-step 1, the next block's scores (S') one block ahead and ITS K rows fetched a block before that —
// across tile boundaries too: three tile buffers, the wait + barrier for tile t + 1 stand in front of tile t's THIRD block and the transfer of tile t + 2 is
// issued right behind that barrier (every wave is then done with tile t - 1, whose buffer it takes).  Plain v_add_f32 row sums in four
// rotating accumulators.  Conversions stand four statements in front of the MFMA that reads them (hipcc pads closer pairs).
__device__ __forceinline__ void fw_mfma_c(f32x16& d, const uint4& a, const uint4& b, const f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, %3" : "=&v"(d) : "v"(as_u32x4(a)), "v"(as_u32x4(b)), "v"(c));
}
__device__ __forceinline__ void fw_mfma_v(f32x16& d, const uint4& a, const uint4& b) {
    asm volatile("v_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, %0" : "+v"(d) : "v"(as_u32x4(a)), "v"(as_u32x4(b)));
}
__device__ __forceinline__ void fw_mfma_p(f32x16& d, const uint4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, %0" : "+v"(d) : "v"(as_u32x4(a)), "v"(b));
}
__global__ __launch_bounds__(256, 2) void attn_fwd_bf16_fast2(Args p) {
    __shared__ __attribute__((aligned(16))) char smem[6 * IMG];   // [3] K tiles | [3] V tiles
    char* sK = smem;
    char* sV = smem + 3 * IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int xt, hh, b;
    block_coords(p, xt, hh, b);
    const int qrow = xt * 128 + wave * 32 + r;
    const bool qvalid = qrow < p.Lq;
    const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * p.dh;
    const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * p.dh;
    const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * p.dh;
    uint4 qb[2];
    load_lane_block(qb, Q, p.ldq, qrow, qvalid, p.dh, h);
    const int nt = p.Lk / KT;     // launcher guarantees Lk % KT == 0 and dh == 32
    dma_tile(sK, K, p.ldk, 0, wave, lane);
    dma_tile(sV, V, p.ldv, 0, wave, lane);
    if (nt > 1) {
        dma_tile(sK + IMG, K, p.ldk, KT, wave, lane);
        dma_tile(sV + IMG, V, p.ldv, KT, wave, lane);
    }
    dma_wait_all();
    __syncthreads();
    float m;                      // the anchor: this query's largest score in key tile 0 (log2 domain)
    {
        float mm = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            uint4 ka[2];
            read_rows(ka, sK, sub * 32 + r, h);
            const f32x16 S0 = mma_first(ka, qb);
#pragma unroll
            for (int i = 0; i < 16; ++i) mm = fmaxf(mm, S0[i]);
        }
        const HalfPair sw = swap_halves(__builtin_bit_cast(unsigned, mm));
        m = fmaxf(__builtin_bit_cast(float, sw.lo), __builtin_bit_cast(float, sw.hi));
    }
    const f32x16 Cm = splat16(-m);
    f32x16 O = zero16();
    float lacc[4] = {0.f, 0.f, 0.f, 0.f};
    // lane-dependent LDS offsets inside a tile image, computed ONCE (a tile / sub-tile adds uniform bytes: 32 rows = 2048)
    const unsigned o_rows0 = img_off(r, h), o_rows1 = img_off(r, 2 + h);
    unsigned o_trl, o_trh;
    {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h2 = g >> 1;
        const int ch = 2 * (g & 1) + (pp >> 1), inner = 8 * (pp & 1);
        o_trl = img_off(4 * h2 + q, ch) + inner;
        o_trh = img_off(4 * h2 + q + 8, ch) + inner;
    }
    f32x16 S;                      // score - anchor of the block that is processed next
    uint4 kn[2];                   // K rows of the block after it (fetched a whole block before the products that read them)
    {
        uint4 ka[2];
        read_rows(ka, sK, r, h);
        read_rows(kn, sK, 32 + r, h);
        // (Cm was written by vector moves just above: a VALU-written operand needs 2 wait states in front of the MFMA that reads it
        // and hipcc pads nothing in front of an asm statement — found by tests/test_isa_hazards.py; inside the loop Cm is invariant)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, %3" : "=&v"(S) : "v"(as_u32x4(ka[0])), "v"(as_u32x4(qb[0])), "v"(Cm));
        fw_mfma_v(S, ka[1], qb[1]);
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(S), "+v"(O));
    }
#define FW_E(i) asm volatile("v_exp_f32 %0, %0" : "+v"(S[i]))
#define FW_A(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(lacc[(i) & 3]) : "v"(S[i]))
#define FW_C(j) asm volatile("v_cvt_pk_" SVOL_H16_ASM "_f32 %0, %1, %2" : "=v"(((j) < 4 ? P0 : P1)[(j) & 3]) : "v"(S[2 * (j)]), "v"(S[2 * (j) + 1]))
#define FW_MF(stmt) do { SP_FENCE(); stmt; SP_FENCE(); } while (0)
    // LDS-DMA sources of this wave's two 16-row pieces of a tile (dma_piece's address arithmetic), kept as running pointers to tile t + 2
    const int prow = 32 * wave + (lane >> 2), pch = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    const h16_t* gk2 = K + ((int64_t)2 * KT + prow) * p.ldk + pch;
    const h16_t* gv2 = V + ((int64_t)2 * KT + prow) * p.ldv + pch;
    const unsigned lds_k = (unsigned)(size_t)(lds_vptr)sK + wave * 2048, lds_v = (unsigned)(size_t)(lds_vptr)sV + wave * 2048;
    int cur = 0;                   // buffer of tile t (t % 3, kept incrementally)
    for (int t = 0; t < nt; ++t) {
        const int nxt = cur == 2 ? 0 : cur + 1, nxt2 = nxt == 2 ? 0 : nxt + 1;
        const char* kimg = sK + cur * IMG;
        const char* vimg = sV + cur * IMG;
        const bool more = t + 1 < nt;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            if (sub == 2 && more) {   // tile t + 1 (issued a whole tile ago) has landed: this block fetches its first K rows
                dma_wait_all();
                __syncthreads();      // ... and every wave is done with tile t - 1: its buffer takes tile t + 2
                if (t + 2 < nt) {
                    const unsigned dk_ = __builtin_amdgcn_readfirstlane(lds_k + nxt2 * IMG), dv_ = __builtin_amdgcn_readfirstlane(lds_v + nxt2 * IMG);
                    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dk_), "v"(gk2) : "m0");
                    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dk_ + 1024), "v"(gk2 + 16 * p.ldk) : "m0");
                    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dv_), "v"(gv2) : "m0");
                    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dv_ + 1024), "v"(gv2 + 16 * p.ldv) : "m0");
                    gk2 += (int64_t)KT * p.ldk;
                    gv2 += (int64_t)KT * p.ldv;
                }
            }
            // (the last block of the last tile computes the scores of a block that does not exist, from stale rows of the next
            // buffer: branch-free, nothing reads them)
            const char* nrow = sub < 2 ? kimg + (sub + 2) * 2048 : sK + nxt * IMG + (sub - 2) * 2048;   // K rows of the block after next
            uint4 va[2];
            f32x16 Sn;
            u32x4 P0, P1;
            SP_FENCE();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {   // (= read_tr(va, vimg, sub, lane))
                const h16x4 lo = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(vimg + sub * 2048 + o_trl + 1024 * ks));
                const h16x4 hi = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(vimg + sub * 2048 + o_trh + 1024 * ks));
                va[ks] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
            SP_FENCE();
            FW_E(0); FW_E(1); FW_E(2); FW_E(3); FW_E(4); FW_E(5); FW_E(6); FW_E(7);
            FW_MF(fw_mfma_c(Sn, kn[0], qb[0], Cm));
            FW_E(8); FW_E(9); FW_E(10); FW_E(11); FW_A(0); FW_A(1); FW_A(2); FW_A(3); FW_A(4); FW_A(5); FW_A(6); FW_A(7);
            FW_MF(fw_mfma_v(Sn, kn[1], qb[1]));
            kn[0] = *reinterpret_cast<const uint4*>(nrow + o_rows0);   // (= read_rows: the rows of the block after next, behind their
            kn[1] = *reinterpret_cast<const uint4*>(nrow + o_rows1);   //  registers' last reader)
            SP_FENCE();
            FW_C(0); FW_C(1); FW_C(2); FW_C(3); FW_E(12); FW_E(13); FW_E(14); FW_E(15);
            FW_MF(fw_mfma_p(O, va[0], P0));
            FW_C(4); FW_C(5); FW_C(6); FW_C(7); FW_A(8); FW_A(9); FW_A(10); FW_A(11); FW_A(12); FW_A(13); FW_A(14); FW_A(15);
            FW_MF(fw_mfma_p(O, va[1], P1));
            S = Sn;
        }
        cur = nxt;
    }
#undef FW_E
#undef FW_A
#undef FW_C
#undef FW_MF
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(O));   // the last product's result is read by compiler-generated code below
    const float l = (lacc[0] + lacc[1]) + (lacc[2] + lacc[3]);
    const float lt = l + __shfl_xor(l, 32, 64);
    const int bad = __syncthreads_or(qvalid && !(lt < SVOL_H16_PSUM_MAX));   // inf / NaN (fp16: any P near 65504): a score left the anchor's range
    if (tid == 0) p.redo[blockIdx.x] = bad ? 1 : 0;
    if (bad) return;              // attn_fwd_bf16_pre (next launch on the stream) recomputes this workgroup
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    h16_t* Oo = reinterpret_cast<h16_t*>(p.out_o) + (int64_t)b * p.Lq * p.ldo + hh * p.dh;
    store_acc(O, Oo, p.ldo, qrow, qvalid, p.dh, h, inv);
    if (qvalid && h == 0) p.lse2[((int64_t)b * p.H + hh) * p.Lq + qrow] = m + __builtin_amdgcn_logf(lt);
}

// workgroup id -> work.  [0, n_main): the full 512-key groups, heads dealt to the XCDs as block_coords does (nxt = full groups per
// head).  [n_main, n_main + 4 B H): the tail groups, four query quarters each, on the same head -> XCD deal.
struct SpWork { int hh, b, kbase, t0, t1, part; bool tail; };
__device__ __forceinline__ SpWork sp_work(const Args& p) {
    SpWork w;
    const int nfull = p.Lk / SP_KEYS, n_main = p.B * p.H * nfull, nt = p.Lq / KT;
    int id = blockIdx.x;
    w.tail = id >= n_main;
    if (w.tail) id -= n_main;
    const int xcd = id & 7, slot = id >> 3, per = w.tail ? 4 : nfull;   // work items per head
    int hl, xt;
    if (p.head_xcd == 2) { const int g = slot / (2 * per), r = slot - g * 2 * per; hl = g * 2 + (r & 1); xt = r >> 1; }
    else { hl = slot / per; xt = slot - hl * per; }
    const int head = p.head_xcd == 2 ? (((hl >> 1) * 8 + xcd) * 2 + (hl & 1)) : hl * 8 + xcd;
    w.hh = head % p.H;
    w.b = head / p.H;
    w.part = w.tail ? xt : 0;
    w.kbase = w.tail ? nfull * SP_KEYS : xt * SP_KEYS;
    w.t0 = w.tail ? xt * nt / 4 : 0;
    w.t1 = w.tail ? (xt + 1) * nt / 4 : nt;
    return w;
}
// tail partial of (head, part): [B*H][4][2][tail keys][32] floats behind the fp32 dQ image
__device__ __forceinline__ float* sp_tail_ptr(const Args& p, const SpWork& w) {
    const int tk = p.Lk % SP_KEYS;
    return p.ws_dq + (int64_t)p.B * p.Lq * p.H * 32 + ((int64_t)(w.b * p.H + w.hh) * 4 + w.part) * 2 * tk * 32;
}
// V9: the round-4 body for the full workgroups too (A/B: SVOL_ATTN_SP_V9=1)
template <int ABL, bool V9 = false>
__device__ __forceinline__ void attn_bwd_sp_dispatch(const Args& p, char* smem) {
    const SpWork w = sp_work(p);
    if (!w.tail) {
        if (V9) attn_bwd_sp_body<4, ABL>(p, smem, w.hh, w.b, w.kbase, w.t0, w.t1, nullptr);
        else attn_bwd_sp_body4<ABL>(p, smem, w.hh, w.b, w.kbase, w.t0, w.t1);
        return;
    }
    float* to = sp_tail_ptr(p, w);
    const int tk = p.Lk % SP_KEYS;         // workgroup-uniform; Lk % 128 == 0
    if (w.t1 <= w.t0) {                    // fewer than four query tiles: this part is empty — its partial still has to exist
        for (int i = threadIdx.x; i < 2 * tk * 32; i += 256) to[i] = 0.f;
        return;
    }
    if (tk == 384) attn_bwd_sp_body<3, ABL>(p, smem, w.hh, w.b, w.kbase, w.t0, w.t1, to);
    else if (tk == 256) attn_bwd_sp_body<2, ABL>(p, smem, w.hh, w.b, w.kbase, w.t0, w.t1, to);
    else attn_bwd_sp_body<1, ABL>(p, smem, w.hh, w.b, w.kbase, w.t0, w.t1, to);
}
#ifdef SP_LAB
template <int ABL>
__global__ __launch_bounds__(256, 1) void attn_bwd_sp_lab(Args p) {
    __shared__ __attribute__((aligned(1024))) char smem[SP_LDS];
    attn_bwd_sp_dispatch<ABL>(p, smem);
}
#endif
__global__ __launch_bounds__(256, 1) void attn_bwd_sp_bf16(Args p) {
    __shared__ __attribute__((aligned(1024))) char smem[SP_LDS];
    attn_bwd_sp_dispatch<0>(p, smem);
}
__global__ __launch_bounds__(256, 1) void attn_bwd_sp_bf16_v9(Args p) {
    __shared__ __attribute__((aligned(1024))) char smem[SP_LDS];
    attn_bwd_sp_dispatch<0, true>(p, smem);
}
// workgroups of the launch above
// blocks of attn_dq_round_bf16: the image (8 elements per thread, rounded up to whole blocks), then the tail partials
static inline unsigned sp_round_grid(int B, int H, int Lq, int Lk) {
    return (unsigned)(((int64_t)B * Lq * H * 32 + 2047) / 2048 + ((int64_t)B * H * 2 * (Lk % SP_KEYS) * 32 + 2047) / 2048);
}
static inline unsigned sp_grid(int B, int H, int Lk) { return (unsigned)(B * H * (Lk / SP_KEYS + (Lk % SP_KEYS ? 4 : 0))); }

}  // namespace

// entry points used by attention.hip's C-ABI functions
// key-split decision shared by forward and backward: only launches whose (query tiles x heads x batch) grid leaves most
// CUs idle and whose key loop is long; the workspace must hold the partials.  Returns ksplit (1 = no split).
static int plan_ksplit(int B, int H, int Lq, int Lk, int dh, int64_t ws_floats, int* tiles_per_split) {
    const int nt = (Lk + KT - 1) / KT;
    const int64_t wgs = (int64_t)((Lq + 127) / 128) * H * B;
    *tiles_per_split = nt;
    if (wgs >= 192 || nt < 8) return 1;
    static const int target = getenv("SVOL_ATTN_KSPLIT_WGS") ? atoi(getenv("SVOL_ATTN_KSPLIT_WGS")) : 256;   // (these launches run on the query stream beside the video half: 256 -> 18.98 / 19.07 ms per step, 512 -> 19.12, 1024 -> 19.28)
    int want = (int)((target + wgs - 1) / wgs);
    if (want > 16) want = 16;
    int tps = (nt + want - 1) / want;
    if (tps < 2) tps = 2;
    const int ks = (nt + tps - 1) / tps;
    const int64_t need = (int64_t)ks * B * Lq * H * dh + (int64_t)ks * B * H * Lq * 2 + (int64_t)B * Lq * H * dh;
    if (ks < 2 || need > ws_floats) return 1;
    *tiles_per_split = tps;
    return ks;
}
// shapes the single-pass backward (attn_bwd_sp_bf16) serves: full 128-row tiles both ways, heads dealt to the XCDs, enough keys for
// the 512-key workgroups to fill the chip (SVOL_ATTN_SP_MIN_LK lowers the bar: tests drive small shapes through it)
// SVOL_DETERMINISTIC=1: no floating-point atomics in the attention backward — the two-pass kernels (dQ by a query-stationary pass)
// instead of the single pass, no key split for launches with few queries (their dQ partials meet through atomics).  Gradients are
// then bit-identical from run to run (tests/test_gpu_ops.py::test_attention_backward_is_bit_reproducible_in_deterministic_mode).
static bool attn_deterministic() {
    static const bool det = getenv("SVOL_DETERMINISTIC") != nullptr;
    return det;
}
static bool sp_shape_ok(int B, int H, int Lq, int Lk, int dh) {
    static const bool no_sp = getenv("SVOL_ATTN_NO_SP") != nullptr || attn_deterministic();
    static const int min_lk = getenv("SVOL_ATTN_SP_MIN_LK") ? atoi(getenv("SVOL_ATTN_SP_MIN_LK")) : 2 * SP_KEYS;
    return !no_sp && dh == 32 && H == 8 && (B * H) % 8 == 0 && Lq % KT == 0 && Lk % KT == 0 && Lk >= min_lk;
}
// the single-pass few-query backward (attn_bwd_fq_bf16): <= 128 queries, the key-split path's fp32 dQ image bound (ksplit > 1: few
// query tiles, many key tiles, workspace large enough), head width 32, no attention dropout; SVOL_ATTN_NO_FEWQ=1: the two-pass kernels
static bool fewq_ok(const Args& p) {
    static const bool off = getenv("SVOL_ATTN_NO_FEWQ") != nullptr;
    return !off && !attn_deterministic() && p.ksplit > 1 && p.Lq <= KT && p.dh == 32 && p.drop_p == 0.f && p.ws_dq != nullptr;
}
static int64_t sp_ws_floats(int B, int H, int Lq, int Lk) { return (int64_t)B * Lq * H * 32 + (int64_t)B * H * 4 * 2 * (Lk % SP_KEYS) * 32; }
// an unmasked, pre-multiplied, dropout-free launch of this shape with this workspace runs the single pass
static bool sp_taken(int B, int H, int Lq, int Lk, int dh, const void* ws, int64_t ws_bytes) {
    static const bool no_head_xcd = getenv("SVOL_ATTN_NO_HEAD_XCD") != nullptr;
    return !no_head_xcd && sp_shape_ok(B, H, Lq, Lk, dh) && ws && ws_bytes >= 4 * sp_ws_floats(B, H, Lq, Lk);
}
// bytes of the fp32 dQ image at the head of the workspace when the single pass serves the shape, else 0 (svol_attn_bwd_sp_image_bytes)
int64_t svol_attn_sp_image_bytes_bf16(int B, int H, int Lq, int Lk, int dh, int64_t ws_bytes) {
    return sp_taken(B, H, Lq, Lk, dh, reinterpret_cast<const void*>(16), ws_bytes) ? (int64_t)B * Lq * H * 32 * 4 : 0;
}
int svol_attn_sp_zero_bf16_launch(float* ws, int64_t ws_bytes, int B, int H, int Lq, int Lk, int dh, hipStream_t s) {
    if (!sp_taken(B, H, Lq, Lk, dh, ws, ws_bytes)) return SVOL_E_UNSUPPORTED;
    const int64_t n4 = (int64_t)B * Lq * H * 32 / 4;
    // few workgroups: the launch runs BESIDE a single-pass kernel whose workgroups own their CUs' register files; a wide grid takes
    // dispatch slots from that kernel's first round (measured: 2048 workgroups cost it 30 us, more than the fill saves)
    static const int wgs = getenv("SVOL_SP_ZERO_WGS") ? atoi(getenv("SVOL_SP_ZERO_WGS")) : 64;
    const int64_t want = (n4 + 255) / 256;
    hipLaunchKernelGGL(attn_sp_zero_image, dim3((unsigned)(want < wgs ? want : (wgs > 0 ? wgs : 1))), dim3(256), 0, s, ws, n4);
    return hipGetLastError() == hipSuccess ? SVOL_OK : SVOL_E_LAUNCH;
}
int64_t svol_attn_ws_floats_bf16(int B, int H, int Lq, int Lk, int dh) {
    int tps;
    const int ks = plan_ksplit(B, H, Lq, Lk, dh, INT64_MAX, &tps);
    if (ks < 2) {
        // key-tile classes of the masked fast kernels (one int per (batch, key tile)), or the redo flags of the unmasked fast
        // forward (one int per workgroup: (batch, head, 128-query tile)) — whichever is larger
        const int64_t cls = (int64_t)B * ((Lk + KT - 1) / KT), redo = (int64_t)B * H * ((Lq + 127) / 128);
        const int64_t sp = sp_shape_ok(B, H, Lq, Lk, dh) ? sp_ws_floats(B, H, Lq, Lk) : 0;   // single-pass backward: fp32 dQ image + tail partials
        const int64_t m = cls > redo ? cls : redo;
        return m > sp ? m : sp;
    }
    return (int64_t)ks * B * Lq * H * dh + (int64_t)ks * B * H * Lq * 2 + (int64_t)B * Lq * H * dh;
}
// masked / ragged launches take the fast kernels when the key-split would not have been chosen anyway (enough query tiles to
// fill the chip): the encoder self-attention of the enc/dec Transformer, video lengths that are not a multiple of 128
static bool pre_masked_ok(int B, int H, int Lq, int Lk, int dh) {
    int tps;
    return plan_ksplit(B, H, Lq, Lk, dh, INT64_MAX, &tps) == 1;
}
static void bind_ws(Args& p, float* ws) {
    p.ws_o = ws;
    p.ws_ml = p.ws_o + (int64_t)p.ksplit * p.B * p.Lq * p.H * p.dh;
    p.ws_dq = p.ws_ml + (int64_t)p.ksplit * p.B * p.H * p.Lq * 2;
}

int svol_attn_fwd_bf16_launch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                              int64_t ldo, float* lse2, const float* kbias, int B, int H, int Lq, int Lk, int dh, float scale,
                              float premul, float* ws, int64_t ws_bytes, float drop_p, uint64_t drop_seed, hipStream_t s) {
    Args p{};
    p.drop_p = drop_p; p.drop_inv = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; p.drop_seed = drop_seed;
    p.q = q; p.k = k; p.v = v; p.out_o = o; p.lse2 = lse2; p.kbias = kbias;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk; p.dh = dh; p.scale = scale; p.premul = premul;
    const bool masked = kbias != nullptr || (Lk % KT) != 0;
    // pre-multiplied q: the fast kernels; masked launches with few queries keep the key-split path below
    static const bool no_pre_masked = getenv("SVOL_ATTN_NO_PRE_MASKED") != nullptr;
    const int ntk = (Lk + KT - 1) / KT;
    const bool pre = premul != 0.f && drop_p == 0.f &&   // (attention dropout lives in the general kernels only)
                     (!masked || (!no_pre_masked && pre_masked_ok(B, H, Lq, Lk, dh) && ws && ws_bytes >= (int64_t)B * ntk * 4));
    p.ksplit = (ws && !pre) ? plan_ksplit(B, H, Lq, Lk, dh, ws_bytes / 4, &p.tiles_per_split) : 1;
    if (p.ksplit == 1) p.tiles_per_split = ntk;
    else bind_ws(p, ws);
    dim3 grid((unsigned)(((Lq + 127) / 128) * p.ksplit), (unsigned)H, (unsigned)B);
    static const bool no_head_xcd = getenv("SVOL_ATTN_NO_HEAD_XCD") != nullptr;
    static const bool head_pair = getenv("SVOL_ATTN_NO_HEAD_PAIR") == nullptr;
    if (pre && !no_head_xcd && (B * H) % 8 == 0) {  // heads dealt to the XCDs (block_coords)
        p.head_xcd = (head_pair && (B * H) % 16 == 0 && H % 2 == 0 && dh == 32) ? 2 : 1;
        p.nxt = (Lq + 127) / 128;
        grid = dim3((unsigned)(B * H * p.nxt));
    }
    if (pre && masked) {
        p.tile_flags = reinterpret_cast<const int*>(ws);
        hipLaunchKernelGGL(attn_tile_flags_bf16, dim3((unsigned)ntk, (unsigned)B), dim3(64), 0, s, kbias, Lk, ntk, reinterpret_cast<int*>(ws));
        hipLaunchKernelGGL(attn_fwd_bf16_pre_masked, grid, dim3(256), 0, s, p);
    } else if (pre) {
        static const bool no_fast = getenv("SVOL_ATTN_NO_FAST_FWD") != nullptr;
        const int64_t nwg = (int64_t)grid.x * grid.y * grid.z;
        if (!no_fast && p.head_xcd && dh == 32 && ws && ws_bytes >= nwg * 4) {
            p.redo = reinterpret_cast<int*>(ws);
            static const bool fwd_v1 = getenv("SVOL_ATTN_FWD_V1") != nullptr;   // round 2's hipcc-scheduled fast forward (A/B)
            if (fwd_v1) hipLaunchKernelGGL(attn_fwd_bf16_fast, grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL(attn_fwd_bf16_fast2, grid, dim3(256), 0, s, p);
        }
        hipLaunchKernelGGL(attn_fwd_bf16_pre, grid, dim3(256), 0, s, p);  // all workgroups, or (p.redo) only the flagged ones
    } else if (masked) hipLaunchKernelGGL(attn_fwd_bf16<true>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(attn_fwd_bf16<false>, grid, dim3(256), 0, s, p);
    if (p.ksplit > 1) {
        const int64_t total = (int64_t)B * Lq * H;
        hipLaunchKernelGGL(attn_combine_bf16, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? SVOL_OK : SVOL_E_LAUNCH;
}

int svol_attn_bwd_bf16_launch(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                              const void* o, int64_t ldo, const void* d_o, int64_t lddo, const float* lse2, float* delta,
                              const float* kbias, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, int B,
                              int H, int Lq, int Lk, int dh, float scale, float premul, float* ws, int64_t ws_bytes,
                              float drop_p, uint64_t drop_seed, int flags, void* ev_prep, hipStream_t s) {
    // ev_prep: recorded behind the launches that must precede work on the caller's OTHER workspace (single pass: behind the prep
    // kernel, i.e. in front of the long key-stationary kernel; every other path: behind the last launch)
    struct PrepEvent {
        void* ev; hipStream_t s; bool done = false;
        void fire() { if (ev && !done) (void)hipEventRecord(static_cast<hipEvent_t>(ev), s); done = true; }
        ~PrepEvent() { fire(); }
    } prep_ev{ev_prep, s};
    Args p{};
    p.drop_p = drop_p; p.drop_inv = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; p.drop_seed = drop_seed;
    p.q = q; p.k = k; p.v = v; p.o = o; p.d_o = d_o; p.lse2 = const_cast<float*>(lse2); p.delta = delta; p.kbias = kbias;
    p.dq = dq; p.dk = dk; p.dv = dv;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk; p.dh = dh; p.scale = scale; p.premul = premul;
    const int64_t total = (int64_t)B * Lq * H;
    const bool masked = kbias != nullptr || (Lk % KT) != 0;
    static const bool no_pre_masked = getenv("SVOL_ATTN_NO_PRE_MASKED") != nullptr;
    const int ntk = (Lk + KT - 1) / KT;
    const bool pre = premul != 0.f && drop_p == 0.f &&
                     (!masked || (!no_pre_masked && pre_masked_ok(B, H, Lq, Lk, dh) && ws && ws_bytes >= (int64_t)B * ntk * 4));
    p.ksplit = (ws && !pre && !attn_deterministic()) ? plan_ksplit(B, H, Lq, Lk, dh, ws_bytes / 4, &p.tiles_per_split) : 1;
    if (p.ksplit == 1) p.tiles_per_split = ntk;
    else bind_ws(p, ws);
    dim3 gd((unsigned)((total + 255) / 256));
    dim3 gq((unsigned)(((Lq + 127) / 128) * p.ksplit), (unsigned)H, (unsigned)B);
    dim3 gk((unsigned)((Lk + 127) / 128), (unsigned)H, (unsigned)B);
    if (!pre) hipLaunchKernelGGL(attn_delta_bf16, gd, dim3(256), 0, s, p);  // (the fast dQ kernel computes delta in its prologue)
    if (pre) {
        static const bool no_head_xcd = getenv("SVOL_ATTN_NO_HEAD_XCD") != nullptr;
        Args pq = p, pk = p;
        dim3 gq2((unsigned)((Lq + 255) / 256), (unsigned)H, (unsigned)B);
        dim3 gk2 = gk;
        if (!no_head_xcd && (B * H) % 8 == 0) {  // heads dealt to the XCDs (block_coords)
            static const bool head_pair = getenv("SVOL_ATTN_NO_HEAD_PAIR") == nullptr;
            pq.head_xcd = pk.head_xcd = (head_pair && (B * H) % 16 == 0 && H % 2 == 0 && dh == 32) ? 2 : 1;
            pq.nxt = (Lq + 255) / 256;
            pk.nxt = (Lk + 127) / 128;
            pq.tail_last = (Lq % 256 >= 1 && Lq % 256 <= 128 && pq.nxt > 1) ? 1 : 0;
            gq2 = dim3((unsigned)(B * H * pq.nxt));
            gk2 = dim3((unsigned)(B * H * pk.nxt));
        }
        if (masked) {
            pq.tile_flags = pk.tile_flags = reinterpret_cast<const int*>(ws);
            hipLaunchKernelGGL(attn_tile_flags_bf16, dim3((unsigned)ntk, (unsigned)B), dim3(64), 0, s, kbias, Lk, ntk, reinterpret_cast<int*>(ws));
            hipLaunchKernelGGL(attn_bwd_dq_bf16_pre_masked, gq2, dim3(256), 0, s, pq);
            hipLaunchKernelGGL(attn_bwd_dkdv_bf16_pre_masked, gk2, dim3(256), 0, s, pk);
        } else {
            if (pq.head_xcd && sp_taken(B, H, Lq, Lk, dh, ws, ws_bytes)) {
                // single pass: row constants + zeroed fp32 dQ image, the key-stationary kernel, rounding of dQ
                Args ps = pq;
                const int64_t n = (int64_t)B * H * Lq;
                ps.ws_dq = ws;
                ps.nl2 = reinterpret_cast<unsigned*>(delta + n);     // here: plain fp32 -lse2
                ps.nd2 = reinterpret_cast<unsigned*>(delta + 2 * n);  //       plain fp32 -delta
                if (flags & SVOL_ATTN_DQ_PREZEROED) hipLaunchKernelGGL(attn_bwd_sp_prep_bf16<false>, dim3((unsigned)((int64_t)B * Lq / 32)), dim3(256), 0, s, ps);
                else hipLaunchKernelGGL(attn_bwd_sp_prep_bf16<true>, dim3((unsigned)((int64_t)B * Lq / 32)), dim3(256), 0, s, ps);
                prep_ev.fire();
                static const bool sp_v9 = getenv("SVOL_ATTN_SP_V9") != nullptr;   // round 4's placement of the full workgroups' stream (A/B)
                if (sp_v9) hipLaunchKernelGGL(attn_bwd_sp_bf16_v9, dim3(sp_grid(B, H, Lk)), dim3(256), 0, s, ps);
                else hipLaunchKernelGGL(attn_bwd_sp_bf16, dim3(sp_grid(B, H, Lk)), dim3(256), 0, s, ps);
                hipLaunchKernelGGL(attn_dq_round_bf16, dim3(sp_round_grid(B, H, Lq, Lk)), dim3(256), 0, s, ps);
                return hipGetLastError() == hipSuccess ? SVOL_OK : SVOL_E_LAUNCH;
            }
            static const bool no_dq_rot = getenv("SVOL_ATTN_NO_DQ_ROT") != nullptr;
            pq.dq_rot = no_dq_rot ? 0 : 1;
            // delta is a 3 x [B,H,Lq] scratch: fp32 delta | -lse2 pairs | -delta pairs (the last two for the DMA dK/dV kernel)
            static const bool no_dma = getenv("SVOL_ATTN_NO_DKDV_DMA") != nullptr;
            const bool dma = !no_dma && dh == 32 && Lq % KT == 0;
            if (dma) {
                const int64_t n = (int64_t)B * H * Lq;
                pq.nl2 = pk.nl2 = reinterpret_cast<unsigned*>(delta + n);
                pq.nd2 = pk.nd2 = reinterpret_cast<unsigned*>(delta + 2 * n);
            }
            if (pq.dq_rot && dh == 32) hipLaunchKernelGGL(attn_bwd_dq_bf16_rot, gq2, dim3(256), 0, s, pq);
            else hipLaunchKernelGGL(attn_bwd_dq_bf16_pre, gq2, dim3(256), 0, s, pq);
            if (dma) hipLaunchKernelGGL(attn_bwd_dkdv_bf16_pre_dma, gk2, dim3(256), 0, s, pk);
            else hipLaunchKernelGGL(attn_bwd_dkdv_bf16_pre, gk2, dim3(256), 0, s, pk);
        }
    } else if (fewq_ok(p)) {
        // few queries against many keys (the query -> video cross attention): ONE key-stationary pass (attn_bwd_fq_bf16); the delta launch
        // above has zeroed the fp32 dQ image (ksplit > 1), attn_dq_finish_bf16 scales and rounds it
        Args pf = p;
        const int want = max(1, 512 / (B * H));                 // ~512 workgroups
        const int chunks = min(ntk, want);
        pf.tiles_per_split = (ntk + chunks - 1) / chunks;
        const dim3 gf((unsigned)((ntk + pf.tiles_per_split - 1) / pf.tiles_per_split), (unsigned)H, (unsigned)B);
        if (kbias) hipLaunchKernelGGL(attn_bwd_fq_bf16<true>, gf, dim3(256), 0, s, pf);
        else hipLaunchKernelGGL(attn_bwd_fq_bf16<false>, gf, dim3(256), 0, s, pf);
        hipLaunchKernelGGL(attn_dq_finish_bf16, gd, dim3(256), 0, s, p);
    } else if (masked) {
        hipLaunchKernelGGL(attn_bwd_dq_bf16<true>, gq, dim3(256), 0, s, p);
        if (p.ksplit > 1) hipLaunchKernelGGL(attn_dq_finish_bf16, gd, dim3(256), 0, s, p);
        hipLaunchKernelGGL(attn_bwd_dkdv_bf16<true>, gk, dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL(attn_bwd_dq_bf16<false>, gq, dim3(256), 0, s, p);
        if (p.ksplit > 1) hipLaunchKernelGGL(attn_dq_finish_bf16, gd, dim3(256), 0, s, p);
        hipLaunchKernelGGL(attn_bwd_dkdv_bf16<false>, gk, dim3(256), 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? SVOL_OK : SVOL_E_LAUNCH;
}
