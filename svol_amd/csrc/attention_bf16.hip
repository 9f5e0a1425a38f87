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
__device__ __forceinline__ float attn_drop(uint64_t seed, uint64_t rowbase, int key, float p, float inv) {
    return dropout_scale(seed, rowbase + (uint64_t)key, p, inv);
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
            const uint64_t rb = (((uint64_t)b * p.H + hh) * p.Lq + (qvalid ? qrow : 0)) * (uint64_t)p.Lk;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        S[sub][4 * g + e] *= attn_drop(p.drop_seed, rb, t * KT + sub * 32 + 8 * g + 4 * h + e, p.drop_p, p.drop_inv);
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
                const uint64_t rb = (((uint64_t)b * p.H + hh) * p.Lq + (qvalid ? qrow : 0)) * (uint64_t)p.Lk;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        dP[4 * g + e] *= attn_drop(p.drop_seed, rb, t * KT + sub * 32 + 8 * g + 4 * h + e, p.drop_p, p.drop_inv);
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
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int qi = min(t * KT + sub * 32 + 8 * g + 4 * h + e, p.Lq - 1);
                        MS[4 * g + e] = attn_drop(p.drop_seed, (((uint64_t)b * p.H + hh) * p.Lq + qi) * (uint64_t)p.Lk, kvalid ? krow : 0,
                                                  p.drop_p, p.drop_inv);
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
#pragma unroll
    for (int u = 0; u < 2; ++u) store_acc(dQ[u], dQo, p.lddq, qrow[u], qvalid[u], p.dh, h, p.scale);
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
#pragma unroll
    for (int u = 0; u < 2; ++u) store_acc(dQ[u], dQo, p.lddq, qrow[u], qvalid[u], p.dh, h, p.scale);
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
constexpr int SP_LDS = 4 * IMG + 4 * IMG + 2 * SP_PART + 4 * KT * 4;

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

// before the single pass: delta = rowsum(dO * O); the row constants as plain fp32 vectors nl = -lse2, nd = -delta ([B,H,Lq], the
// loop DMAs 64 of them per wave instruction); the fp32 dQ image zeroed.  One thread per (row, head), 32 rows x H heads per block;
// the [B,H,Lq] side is read / written through LDS so that both sides of the transposition are coalesced.
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
        float* z = p.ws_dq + row * (p.H * 32) + hh * 32;
#pragma unroll
        for (int i = 0; i < 32; i += 4) *reinterpret_cast<f32x4*>(z + i) = f32x4{0.f, 0.f, 0.f, 0.f};
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
// after it: dq (16-bit) = scale * fp32 image
__global__ __launch_bounds__(256) void attn_dq_round_bf16(Args p) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;   // 8 consecutive columns of one row
    const int64_t W = (int64_t)p.H * 32, total = (int64_t)p.B * p.Lq * W;
    if (i >= total) return;
    const int64_t row = i / W, c = i % W;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p.ws_dq + i), b = *reinterpret_cast<const f32x4*>(p.ws_dq + i + 4);
    h16x8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (h16_t)(a[e] * p.scale); v[4 + e] = (h16_t)(b[e] * p.scale); }
    *reinterpret_cast<h16x8*>(reinterpret_cast<h16_t*>(p.dq) + row * p.lddq + c) = v;
}

// Every MFMA of the single-pass kernel is inline asm: at one wave per SIMD (512 registers) hipcc selects the AGPR-destination form
// for the builtin, so score / dP tiles that the VALU exponentiates would pay a v_accvgpr_read per element.  Register classes by
// constraint: results the VALU touches "v", the dK / dV accumulators and the loop-invariant K / V fragments "a".
// HAZARDS are ours inside and behind an asm statement (hipcc pads nothing): s_nop 1 in front = VALU-written operand -> MFMA read;
// an MFMA result must not be read by a non-MFMA instruction for 12 wait states — the loop below keeps every such consumer dozens of
// instructions behind its producer, and tests/test_isa_hazards.py checks the distance in the emitted code.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ u32x4 to_acc(u32x4 r) {
    asm volatile("; operand -> AGPR" : "+a"(r));
    return r;
}
__device__ __forceinline__ u32x4 as_u32x4(const uint4& v) { return u32x4{v.x, v.y, v.z, v.w}; }
// D = A B + C, two k-steps, B fragments in AGPRs, D and C different VGPR ranges
__device__ __forceinline__ f32x16 sp_mma_c(const uint4 (&a)[2], const u32x4 (&b)[2], const f32x16& c0) {
    f32x16 acc;
    asm("s_nop 1\n\tv_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, %5\n\tv_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %3, %4, %0"
        : "=&v"(acc)
        : "v"(as_u32x4(a[0])), "a"(b[0]), "v"(as_u32x4(a[1])), "a"(b[1]), "v"(c0));
    return acc;
}
// acc (AGPRs) += A B, two k-steps, both operands in VGPRs (B fresh from v_cvt_pk)
__device__ __forceinline__ void sp_mma_acc(f32x16& acc, const uint4 (&a)[2], const uint4 (&b)[2]) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %3, %4, %0"
                 : "+a"(acc)
                 : "v"(as_u32x4(a[0])), "v"(as_u32x4(b[0])), "v"(as_u32x4(a[1])), "v"(as_u32x4(b[1])));
}
// dQ partial (VGPRs): FIRST = A B (C is the inline constant 0), then += A B; B (K fragments) in AGPRs
template <bool FIRST>
__device__ __forceinline__ void sp_mma_dq(f32x16& acc, const uint4 (&a)[2], const u32x4 (&b)[2]) {
    if (FIRST)
        asm volatile("v_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, 0\n\tv_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %3, %4, %0"
                     : "=&v"(acc)
                     : "v"(as_u32x4(a[0])), "a"(b[0]), "v"(as_u32x4(a[1])), "a"(b[1]));
    else
        asm volatile("v_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %3, %4, %0"
                     : "+v"(acc)
                     : "v"(as_u32x4(a[0])), "a"(b[0]), "v"(as_u32x4(a[1])), "a"(b[1]));
}

template <bool LIVE>
__device__ __forceinline__ void attn_bwd_sp_body(const Args& p, char* smem, int xt, int hh, int b) {
    // every LDS-DMA destination (row constants, Q / dO tiles, the prologue's K / V staging) sits in the first 64 KiB: the existing
    // kernels never put an M0 base above 0xFFFF, and nothing here depends on how many bits of M0 the transfer honours
    float* sL = reinterpret_cast<float*>(smem);   // [2][KT] -lse2
    float* sD = sL + 2 * KT;                      // [2][KT] -delta
    char* sQ = smem + 4 * KT * 4;          // [2][IMG]   Q tiles (128 queries)
    char* sdO = sQ + 2 * IMG;              // [2][IMG]   dO tiles
    char* sTall = sQ + 4 * IMG;            // [4 waves][4 blocks][32 keys][32 q] dS images
    char* sPart = sQ + 8 * IMG;            // [2][SP_PART] dQ partials
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * 32;
    const h16_t* dO = reinterpret_cast<const h16_t*>(p.d_o) + (int64_t)b * p.Lq * p.lddo + hh * 32;
    const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * 32;
    const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * 32;
    const float* nl_g = reinterpret_cast<const float*>(p.nl2) + ((int64_t)b * p.H + hh) * p.Lq;
    const float* nd_g = reinterpret_cast<const float*>(p.nd2) + ((int64_t)b * p.H + hh) * p.Lq;
    const int dqw = p.H * 32;
    float* dq32 = p.ws_dq + (int64_t)b * p.Lq * dqw + hh * 32;
    const int key0 = xt * SP_KEYS + wave * SP_WKEYS;
    char* sT = sTall + wave * IMG;

    u32x4 kbk[SP_KB][2], vbk[SP_KB][2], kd[SP_KB][2];
    f32x16 dK[SP_KB], dV[SP_KB];
    if (LIVE) {   // this wave's K tile, then its V tile, through ITS quarter of the (still unused) Q / dO buffers
        char* sS = sQ + wave * IMG;
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) dma_piece(sS, K, p.ldk, key0, pc, lane);
        dma_wait_all();
#pragma unroll
        for (int kb = 0; kb < SP_KB; ++kb) {
            uint4 t0[2], t1[2];
            read_rows(t0, sS, kb * 32 + r, h);
            read_tr_nat(t1, sS, kb, lane);
#pragma unroll
            for (int s = 0; s < 2; ++s) { kbk[kb][s] = to_acc(as_u32x4(t0[s])); kd[kb][s] = to_acc(as_u32x4(t1[s])); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) dma_piece(sS, V, p.ldv, key0, pc, lane);
        dma_wait_all();
#pragma unroll
        for (int kb = 0; kb < SP_KB; ++kb) {
            uint4 t0[2];
            read_rows(t0, sS, kb * 32 + r, h);
#pragma unroll
            for (int s = 0; s < 2; ++s) vbk[kb][s] = to_acc(as_u32x4(t0[s]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int kb = 0; kb < SP_KB; ++kb) {
            dK[kb] = zero16();
            dV[kb] = zero16();
            asm volatile("; accumulators -> AGPR" : "+a"(dK[kb]), "+a"(dV[kb]));
        }
    }
    // both partial slots start as zeros: a wave without keys never writes its own, and step 0 "reduces" an empty buffer (every
    // step then issues the same four atomics, which keeps the counted waits below uniform)
#pragma unroll
    for (int buf = 0; buf < 2; ++buf)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(sPart + buf * SP_PART + ((wave * 4 + g) * 64 + lane) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-query constants of a 128-query tile: wave 0 / 1 -> -lse of queries 0..63 / 64..127, wave 2 / 3 -> -delta
    auto dma_stats = [&](int buf, int row0) {
        const float* src = (wave < 2 ? nl_g : nd_g) + row0 + (wave & 1) * 64 + (lane & 15) * 4;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_vptr)((wave < 2 ? sL : sD) + buf * KT + (wave & 1) * 64));
        if (lane < 16) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "m0");
    };
    const int nt = p.Lq / KT, nsteps = p.Lq / 32;   // launcher: Lq % 128 == 0
    sp_barrier();                          // every wave is done with its staging quarter
    dma_tile_async(sQ, Q, p.ldq, 0, wave, lane);
    dma_tile_async(sdO, dO, p.lddo, 0, wave, lane);
    dma_stats(0, 0);
    dma_wait_all();
    sp_barrier();

    // the sum over the four waves of one row group (rows 8 wave + 4 h + e of the step's 32 queries) of the partials in `buf`,
    // added to the fp32 image: lanes 0..31 / 32..63 = the 128 contiguous bytes of two rows
    const int dq_lane = (8 * wave + 4 * h) * dqw + r;
    auto reduce_step = [&](int buf, int qbase) {
        const char* pb = sPart + buf * SP_PART + (wave * 64 + lane) * 16;
        f32x4 acc = *reinterpret_cast<const f32x4*>(pb);
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) acc += *reinterpret_cast<const f32x4*>(pb + w2 * 4 * 64 * 16);
        float* dst = dq32 + (int64_t)qbase * dqw + dq_lane;
#pragma unroll
        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dst + e * dqw, acc[e]);
    };

    for (int st = 0; st < nsteps; ++st) {
        const int sub = st & 3, t = st >> 2, cur = t & 1;
        if (sub == 0 && t + 1 < nt) {      // the other tile buffers were last read in step st - 1, behind that step's barrier
            dma_tile_async(sQ + (cur ^ 1) * IMG, Q, p.ldq, (t + 1) * KT, wave, lane);
            dma_tile_async(sdO + (cur ^ 1) * IMG, dO, p.lddo, (t + 1) * KT, wave, lane);
            dma_stats(cur ^ 1, (t + 1) * KT);
            asm volatile("" ::: "memory");   // the atomics below stay BEHIND these five transfers (the counted vmcnt at the tile's end relies on it)
        }
        if (LIVE) {
            const char* qimg = sQ + cur * IMG;
            const char* doimg = sdO + cur * IMG;
            uint4 qa[2], doa[2], qt[2], dot[2];
            read_rows(qa, qimg, sub * 32 + r, h);
            read_rows(doa, doimg, sub * 32 + r, h);
            const f32x16 Cl = rows16(sL + cur * KT, sub, h), Cd = rows16(sD + cur * KT, sub, h);
            read_tr(qt, qimg, sub, lane);
            read_tr(dot, doimg, sub, lane);
            f32x16 dQp;
            f32x16 S = sp_mma_c(qa, kbk[0], Cl), dP = sp_mma_c(doa, vbk[0], Cd);
#pragma unroll
            for (int kb = 0; kb < SP_KB; ++kb) {
                f32x16 Sn, dPn;
                if (kb + 1 < SP_KB) {      // the next block's products go out ahead of this block's vector work
                    Sn = sp_mma_c(qa, kbk[kb + 1], Cl);
                    dPn = sp_mma_c(doa, vbk[kb + 1], Cd);
                }
                __builtin_amdgcn_sched_barrier(0);   // S / dP of THIS block are read below: at least four MFMAs behind their own
#pragma unroll
                for (int i = 0; i < 16; ++i) S[i] = __builtin_amdgcn_exp2f(S[i]);
                uint4 pk[2];
                pack16(pk, S);
                sp_mma_acc(dV[kb], dot, pk);          // dV^T += dO^T P
#pragma unroll
                for (int i = 0; i < 16; ++i) S[i] *= dP[i];
                pack16(pk, S);
                sp_mma_acc(dK[kb], qt, pk);           // dK^T += Q^T dS
                char* img = sT + kb * 2048;
                *reinterpret_cast<uint2*>(img + ds_off(r, 0, h)) = make_uint2(pk[0].x, pk[0].y);
                *reinterpret_cast<uint2*>(img + ds_off(r, 1, h)) = make_uint2(pk[0].z, pk[0].w);
                *reinterpret_cast<uint2*>(img + ds_off(r, 2, h)) = make_uint2(pk[1].x, pk[1].y);
                *reinterpret_cast<uint2*>(img + ds_off(r, 3, h)) = make_uint2(pk[1].z, pk[1].w);
                if (kb > 0) {                          // dQ += dS K of the PREVIOUS block: its image has long been written
                    uint4 a[2];
                    read_ds_tr(a, sT + (kb - 1) * 2048, lane);
                    if (kb == 1) sp_mma_dq<true>(dQp, a, kd[0]);
                    else sp_mma_dq<false>(dQp, a, kd[kb - 1]);
                }
                if (kb + 1 < SP_KB) { S = Sn; dP = dPn; }
            }
            {
                uint4 a[2];
                read_ds_tr(a, sT + (SP_KB - 1) * 2048, lane);
                sp_mma_dq<false>(dQp, a, kd[SP_KB - 1]);
            }
            // the previous step's partials (published by the last barrier) leave as atomics while the last products finish: the
            // partial below is read from MFMA results by ds_write (MFMA -> non-MFMA read: tests/test_isa_hazards.py)
            __builtin_amdgcn_sched_barrier(0);
            reduce_step((st - 1) & 1, st > 0 ? (st - 1) * 32 : 0);
            __builtin_amdgcn_sched_barrier(0);
            char* pw = sPart + (st & 1) * SP_PART + (wave * 4 * 64 + lane) * 16;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(pw + g * 64 * 16) = f32x4{dQp[4 * g], dQp[4 * g + 1], dQp[4 * g + 2], dQp[4 * g + 3]};
        } else {
            reduce_step((st - 1) & 1, st > 0 ? (st - 1) * 32 : 0);
        }
        // tile t + 1 (5 LDS-DMA instructions, issued at the top of this tile) is older than this tile's 16 atomics
        if (sub == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        sp_barrier();
    }
    reduce_step((nsteps - 1) & 1, (nsteps - 1) * 32);
    if (LIVE) {
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // the last asm MFMAs' results are read by compiler-generated code below
        h16_t* dKo = reinterpret_cast<h16_t*>(p.dk) + (int64_t)b * p.Lk * p.lddk + hh * 32;
        h16_t* dVo = reinterpret_cast<h16_t*>(p.dv) + (int64_t)b * p.Lk * p.lddv + hh * 32;
#pragma unroll
        for (int kb = 0; kb < SP_KB; ++kb) {
            store_acc(dK[kb], dKo, p.lddk, key0 + kb * 32 + r, true, 32, h, p.scale / p.premul);
            store_acc(dV[kb], dVo, p.lddv, key0 + kb * 32 + r, true, 32, h, 1.f);
        }
    }
}
__global__ __launch_bounds__(256, 1) void attn_bwd_sp_bf16(Args p) {
    __shared__ __attribute__((aligned(1024))) char smem[SP_LDS];
    int xt, hh, b;
    block_coords(p, xt, hh, b);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // wave-uniform (Lk % 128 == 0): the last workgroup of a head may own fewer than 512 keys; its key-less waves only help with
    // staging, the reduction and the barriers (same number of barriers and vector-memory instructions on both paths)
    if (xt * SP_KEYS + wave * SP_WKEYS < p.Lk) attn_bwd_sp_body<true>(p, smem, xt, hh, b);
    else attn_bwd_sp_body<false>(p, smem, xt, hh, b);
}

}  // namespace

// entry points used by attention.hip's C-ABI functions
// key-split decision shared by forward and backward: only launches whose (query tiles x heads x batch) grid leaves most
// CUs idle and whose key loop is long; the workspace must hold the partials.  Returns ksplit (1 = no split).
static int plan_ksplit(int B, int H, int Lq, int Lk, int dh, int64_t ws_floats, int* tiles_per_split) {
    const int nt = (Lk + KT - 1) / KT;
    const int64_t wgs = (int64_t)((Lq + 127) / 128) * H * B;
    *tiles_per_split = nt;
    if (wgs >= 192 || nt < 8) return 1;
    int want = (int)((512 + wgs - 1) / wgs);
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
static bool sp_shape_ok(int B, int H, int Lq, int Lk, int dh) {
    static const bool no_sp = getenv("SVOL_ATTN_NO_SP") != nullptr;
    static const int min_lk = getenv("SVOL_ATTN_SP_MIN_LK") ? atoi(getenv("SVOL_ATTN_SP_MIN_LK")) : 2 * SP_KEYS;
    return !no_sp && dh == 32 && H <= 8 && (B * H) % 8 == 0 && Lq % KT == 0 && Lk % KT == 0 && Lk >= min_lk;
}
int64_t svol_attn_ws_floats_bf16(int B, int H, int Lq, int Lk, int dh) {
    int tps;
    const int ks = plan_ksplit(B, H, Lq, Lk, dh, INT64_MAX, &tps);
    if (ks < 2) {
        // key-tile classes of the masked fast kernels (one int per (batch, key tile)), or the redo flags of the unmasked fast
        // forward (one int per workgroup: (batch, head, 128-query tile)) — whichever is larger
        const int64_t cls = (int64_t)B * ((Lk + KT - 1) / KT), redo = (int64_t)B * H * ((Lq + 127) / 128);
        const int64_t sp = sp_shape_ok(B, H, Lq, Lk, dh) ? (int64_t)B * Lq * H * dh : 0;   // fp32 dQ image of the single-pass backward
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
            hipLaunchKernelGGL(attn_fwd_bf16_fast, grid, dim3(256), 0, s, p);
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
                              float drop_p, uint64_t drop_seed, hipStream_t s) {
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
    p.ksplit = (ws && !pre) ? plan_ksplit(B, H, Lq, Lk, dh, ws_bytes / 4, &p.tiles_per_split) : 1;
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
            if (pq.head_xcd && sp_shape_ok(B, H, Lq, Lk, dh) && ws && ws_bytes >= (int64_t)B * Lq * H * dh * 4) {
                // single pass: row constants + zeroed fp32 dQ image, the key-stationary kernel, rounding of dQ
                Args ps = pq;
                const int64_t n = (int64_t)B * H * Lq;
                ps.ws_dq = ws;
                ps.nl2 = reinterpret_cast<unsigned*>(delta + n);     // here: plain fp32 -lse2
                ps.nd2 = reinterpret_cast<unsigned*>(delta + 2 * n);  //       plain fp32 -delta
                ps.nxt = (Lk + SP_KEYS - 1) / SP_KEYS;
                ps.tail_last = (Lk % SP_KEYS != 0 && ps.nxt > 1) ? 1 : 0;
                hipLaunchKernelGGL(attn_bwd_sp_prep_bf16, dim3((unsigned)((int64_t)B * Lq / 32)), dim3(256), 0, s, ps);
                hipLaunchKernelGGL(attn_bwd_sp_bf16, dim3((unsigned)(B * H * ps.nxt)), dim3(256), 0, s, ps);
                hipLaunchKernelGGL(attn_dq_round_bf16, dim3((unsigned)(((int64_t)B * Lq * H * 32 / 8 + 255) / 256)), dim3(256), 0, s, ps);
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
