// bf16 NT GEMM fast path (gfx950):  C = epilogue(A[M,K] * B[N,K]^T),  K % 64 == 0.
//
// Differences from the generic kernel in gemm.hip (which stays for f32 and odd shapes), all driven by
// the rocprofv3 profile of the first version (190 TFLOP/s aggregate, K = 256 for most launches so the
// per-tile prologue / epilogue dominated, 2-byte scattered stores for 205 MB outputs):
//
//  * operand tiles go HBM -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction, no
//    VGPR staging, hardware bounds check zero-fills ragged M / N edges); the LDS image keeps the XOR
//    swizzle of the generic kernel by permuting the per-lane SOURCE chunk (the LDS side of an LDS-DMA is
//    lane-linear);
//  * two LDS stages, ONE barrier per 64-deep K step: the next stage's DMA is issued right after the
//    barrier and flies under the 32 MFMAs of the current stage;
//  * operand roles are swapped (weights feed the MFMA A operand) so each lane ends up with 4 CONSECUTIVE
//    output columns per accumulator; the epilogue goes through LDS and writes whole 128-byte row
//    segments (16 lanes x 4 columns), with bias / activation / residual / pre-activation applied on the
//    coalesced side;
//  * fused backward epilogue: out = acc * gelu'(pre) plus per-column sums (the bias gradient), which
//    removes a 205 MB read-modify-write pass and a column-sum pass per MLP.
#include <cstdlib>

#include "common.h"

namespace {

struct FastArgs {
    const h16_t* A; const h16_t* B; void* C;
    const float* bias; const float* colscale; h16_t* pre; const void* res; const h16_t* aux; float* colsum;
    int64_t lda, ldb, ldc, ldp, ldr, ldaux;
    int M, N, K, act, epi;
    // implicit-GEMM convolution (CONV instantiation): A is an NHWC activation [n, cH, cW, cC]; row m of the GEMM is output pixel
    // (n, ho, wo) and column k = (ky*ckw + kx)*cC + c is gathered on the fly — the im2col matrix never exists
    int cH, cW, cC, ckw, cstride, cpad, cHo, cWo;
    int64_t a_bytes;
    int kwrap;  // A's contraction index is k % kwrap (kwrap == K: plain).  Split weights: C = A [W_hi | W_lo]^T reads [A | A] in place
};

constexpr int EP_STRIDE = 272;       // bytes per staged accumulator row (64 floats + 16 pad)

// LDS operand image: 128 rows of BK bf16 (BK*2 bytes), 16-byte chunks permuted so that the ds_read_b128
// fragment reads of a 16x16x32 MFMA (lanes: row = lane&15, chunk = lane>>4) are bank-conflict free.
template <int BK> __device__ __forceinline__ int slot_of(int row, int ch) {
    if constexpr (BK == 64) {
        return ch ^ (row & 7);
    } else {
        const int rh = (row >> 2) & 3;
        return ch ^ ((0x78 >> (2 * rh)) & 3);  // g(rh) = {0,2,3,1}: the 16 lanes of every ds_read_b128 group hit 16 distinct slots
    }
}
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <typename TC> struct Out4;
template <> struct Out4<float> {
    static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void store(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Out4<h16_t> {
    static __device__ __forceinline__ f32x4 load(const h16_t* p) {
        const h16x4 t = *reinterpret_cast<const h16x4*>(p);
        return f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
    }
    static __device__ __forceinline__ void store(h16_t* p, const f32x4& v) {
        h16x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = (h16_t)v[e];
        *reinterpret_cast<h16x4*>(p) = t;
    }
};

template <typename TC, int BK, bool CONV = false>
__device__ __forceinline__ void gemm_nt_bf16_body(const FastArgs& p) {
    constexpr int RB = BK * 2;                 // bytes per image row
    constexpr int CPR = RB / 16;               // chunks per row
    constexpr int STAGE = 128 * RB;            // one operand tile
    constexpr int RPI = 1024 / RB;             // rows per 1-KiB wave DMA instruction
    constexpr int NI = STAGE / 1024 / 4;       // DMA instructions per wave per operand per stage
    constexpr int EPB = 4 * 16 * EP_STRIDE;    // epilogue staging bytes (4 waves x 16 rows)
    constexpr int SMEM = (4 * STAGE > EPB) ? 4 * STAGE : EPB;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int bm = blockIdx.y * 128, bn = blockIdx.x * 128;

    // buffer descriptors rooted at this workgroup's tile rows: out-of-range rows read as zero
    const int rowsA = min(128, p.M - bm), rowsB = min(128, p.N - bn);
    const __amdgpu_buffer_rsrc_t rA = CONV
        ? __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000)   // the whole activation tensor
        : __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)bm * p.lda), 0, (int)(((int64_t)rowsA * p.lda) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.B + (int64_t)bn * p.ldb), 0, (int)(((int64_t)rowsB * p.ldb) * 2), 0x00020000);
    // the LDS side of an LDS-DMA is lane-linear (slot lane%CPR of row lane/CPR): permute the SOURCE chunk instead.
    // slot_of is an involution in ch for a fixed row, so the chunk stored in slot s is slot_of(row, s).
    int voffA[NI], voffB[NI];
    int cvh[NI], cvw[NI], cvn[NI];  // CONV: input row / column of the window origin, image row base; cvn < 0: row past M
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = (wave * NI + i) * RPI + lane / CPR;
        const int lch = slot_of<BK>(row, lane % CPR);
        voffA[i] = CONV ? lch * 16 : (int)(((int64_t)row * p.lda + lch * 8) * 2);
        voffB[i] = (int)(((int64_t)row * p.ldb + lch * 8) * 2);
        if (CONV) {
            const int m = bm + row;
            const int wo = m % p.cWo, t_ = m / p.cWo;
            const int ho = t_ % p.cHo, n = t_ / p.cHo;
            cvh[i] = ho * p.cstride - p.cpad;
            cvw[i] = wo * p.cstride - p.cpad;
            cvn[i] = m < p.M ? n * p.cH : -(1 << 28);
        }
    }
    auto issue = [&](int stage, int k0) {
        char* sa = smem + stage * STAGE;
        char* sb = smem + (2 + stage) * STAGE;
        int ky = 0, kx = 0, c0 = 0;
        if (CONV) {  // a K step never straddles two (ky, kx) segments: cC % BK == 0
            const int seg = k0 / p.cC;
            c0 = k0 - seg * p.cC;
            ky = seg / p.ckw;
            kx = seg - ky * p.ckw;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (CONV) {
                const int hi = cvh[i] + ky, wi = cvw[i] + kx;
                const bool ok = cvn[i] >= 0 && (unsigned)hi < (unsigned)p.cH && (unsigned)wi < (unsigned)p.cW;
                // padding / rows past M: an offset past the buffer's size reads as zero (hardware bounds check)
                const int off = ok ? ((((cvn[i] + hi) * p.cW + wi) * p.cC + c0) * 2 + voffA[i]) : 0x7ffffff0;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_ptr)(sa + (wave * NI + i) * 1024), 16, off, 0, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_ptr)(sa + (wave * NI + i) * 1024), 16, voffA[i],
                                                         (k0 >= p.kwrap ? k0 - p.kwrap : k0) * 2, 0, 0);
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_ptr)(sb + (wave * NI + i) * 1024), 16, voffB[i], k0 * 2, 0, 0);
        }
    };

    f32x4 acc[4][4];  // [nt][mt]: rows <-> n = nt*16 + fq*4 + r, cols <-> m = mt*16 + fr
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    const int nk = p.K / BK;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // waits for this stage's DMA (vmcnt(0)) and for everyone to be done with the other stage
        if (kt + 1 < nk) issue((kt + 1) & 1, (kt + 1) * BK);
        const char* sa = smem + (kt & 1) * STAGE;
        const char* sb = smem + (2 + (kt & 1)) * STAGE;
#pragma unroll
        for (int g = 0; g < BK / 32; ++g) {
            uint4 mf[4], nf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int rowM = wm * 64 + t * 16 + fr;
                mf[t] = *reinterpret_cast<const uint4*>(sa + rowM * RB + (slot_of<BK>(rowM, 4 * g + fq) << 4));
                const int rowN = wn * 64 + t * 16 + fr;
                nf[t] = *reinterpret_cast<const uint4*>(sb + rowN * RB + (slot_of<BK>(rowN, 4 * g + fq) << 4));
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = SVOL_MFMA_16x16x32_H16(__builtin_bit_cast(h16x8, nf[nt]),
                                                                          __builtin_bit_cast(h16x8, mf[mt]), acc[nt][mt], 0, 0, 0);
        }
    }

    // ---- epilogue through LDS: four passes of 16 rows per wave -------------------------------------
    __syncthreads();
    char* st = smem + wave * (16 * EP_STRIDE);
    const int cq = lane & 15, rq = lane >> 4;           // coalesced side: 4 columns cq*4.., rows rq, rq+4, ...
    const int ncol = bn + wn * 64 + cq * 4;
    const bool colv = ncol + 4 <= p.N;                  // fast vector path for this lane's 4 columns
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bias4[e] = (ncol + e < p.N) ? p.bias[ncol + e] : 0.f;
    }
    f32x4 scale4 = {1.f, 1.f, 1.f, 1.f};
    if (p.colscale) {
#pragma unroll
        for (int e = 0; e < 4; ++e) scale4[e] = (ncol + e < p.N) ? p.colscale[ncol + e] : 1.f;
    }
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    TC* C = reinterpret_cast<TC*>(p.C);
    const TC* R = reinterpret_cast<const TC*>(p.res);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            *reinterpret_cast<f32x4*>(st + fr * EP_STRIDE + (nt * 16 + fq * 4) * 4) = acc[nt][pass];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rl = it * 4 + rq;
            const int m = bm + wm * 64 + pass * 16 + rl;
            f32x4 v = *reinterpret_cast<const f32x4*>(st + rl * EP_STRIDE + cq * 16);
            if (m < p.M) {
                if (p.epi == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (v[e] + bias4[e]) * scale4[e];
                    if (p.pre) {
                        f32x4 sv = v;
                        if (p.act == SVOL_ACT_GELU_D) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) sv[e] = dgelu_fast(v[e]);
                        }
                        if (colv) Out4<h16_t>::store(p.pre + (int64_t)m * p.ldp + ncol, sv);
                        else
                            for (int e = 0; e < 4; ++e)
                                if (ncol + e < p.N) p.pre[(int64_t)m * p.ldp + ncol + e] = (h16_t)sv[e];
                    }
                    if (p.act == SVOL_ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    } else if (act_is_gelu(p.act)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
                    } else if (p.act == SVOL_ACT_SIGMOID) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
                    }
                    if (R) {
                        if (colv) {
                            const f32x4 rr = Out4<TC>::load(R + (int64_t)m * p.ldr + ncol);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += rr[e];
                        } else {
                            for (int e = 0; e < 4; ++e)
                                if (ncol + e < p.N) v[e] += to_f32(R[(int64_t)m * p.ldr + ncol + e]);
                        }
                    }
                    if (p.act == SVOL_ACT_RELU_RES) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                } else {  // epi 1: v = acc * gelu'(aux), column sums of v
                    if (colv) {
                        const f32x4 a = Out4<h16_t>::load(p.aux + (int64_t)m * p.ldaux + ncol);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= dact_fast(a[e], p.act);
                    } else {
                        for (int e = 0; e < 4; ++e)
                            v[e] = (ncol + e < p.N) ? v[e] * dact_fast((float)p.aux[(int64_t)m * p.ldaux + ncol + e], p.act) : 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) csum[e] += v[e];
                }
                if (colv) Out4<TC>::store(C + (int64_t)m * p.ldc + ncol, v);
                else
                    for (int e = 0; e < 4; ++e)
                        if (ncol + e < p.N) C[(int64_t)m * p.ldc + ncol + e] = from_f32<TC>(v[e]);
            }
        }
        __syncthreads();
    }
    if (p.epi == 1 && p.colsum) {
        // lanes rq = 0..3 hold partial sums of the same 4 columns: fold, then combine the two m-waves in LDS
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            csum[e] += __shfl_xor(csum[e], 16, 64);
            csum[e] += __shfl_xor(csum[e], 32, 64);
        }
        float* red = reinterpret_cast<float*>(smem);  // [2 wm][128 cols]
        if (rq == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wm * 128 + wn * 64 + cq * 4 + e] = csum[e];
        }
        __syncthreads();
        if (tid < 128 && bn + tid < p.N) atomicAdd(p.colsum + bn + tid, red[tid] + red[128 + tid]);
    }
}

// concrete kernels (one per output type x K-step)
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_f32_k64(FastArgs p) { gemm_nt_bf16_body<float, 64>(p); }
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_b16_k64(FastArgs p) { gemm_nt_bf16_body<h16_t, 64>(p); }
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_f32_k32(FastArgs p) { gemm_nt_bf16_body<float, 32>(p); }
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_b16_k32(FastArgs p) { gemm_nt_bf16_body<h16_t, 32>(p); }
__global__ __launch_bounds__(256, 2) void conv_nhwc_bf16_k64(FastArgs p) { gemm_nt_bf16_body<h16_t, 64, true>(p); }
__global__ __launch_bounds__(256, 2) void conv_nhwc_bf16_k32(FastArgs p) { gemm_nt_bf16_body<h16_t, 32, true>(p); }

// ---------------------------------------------------------------------------------------------------
// Skinny-M variant (the query stream: M = B * num_queries = 800 rows at the benchmark size).  A 128x128
// tiling leaves 14 workgroups on 256 CUs, each walking the whole K loop on its own (54 us for K = 2048).
// Here one workgroup owns a 32 x 64 output tile and its four waves split K between them (intra-workgroup
// split-K: each wave streams its K/4 slice of A and W straight from global memory into MFMA operands, 64
// elements = 128 contiguous bytes per row and batch, no LDS staging, next batch in flight under the current
// one), the partial tiles meet in LDS and all 256 threads run the usual epilogue on 8 consecutive columns each.
constexpr int SK_BM = 32, SK_BN = 64, SK_LD = SK_BN + 4;

template <typename TC>
__device__ __forceinline__ void gemm_nt_bf16_skinny_body(const FastArgs& p) {
    __shared__ __attribute__((aligned(16))) float red[4][SK_BM][SK_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int bm = blockIdx.y * SK_BM, bn = blockIdx.x * SK_BN;
    const int kslice = p.K >> 2;
    const int nb = kslice >> 6;  // batches of 64
    const bool va = bm + r < p.M, vb0 = bn + r < p.N, vb1 = bn + 32 + r < p.N;
    // the contraction index may be permuted freely as long as both operands agree: lane (r, h) takes the 32
    // consecutive elements [h*32, h*32+32) of each 64-element batch, 8 per MFMA
    const h16_t* pa = p.A + (int64_t)(bm + r) * p.lda + (wave * kslice) % p.kwrap + h * 32;  // (kwrap % kslice == 0: launcher)
    const h16_t* pb0 = p.B + (int64_t)(bn + r) * p.ldb + wave * kslice + h * 32;
    const h16_t* pb1 = pb0 + (int64_t)32 * p.ldb;
    const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
    uint4 a[4], b0[4], b1[4], na[4], nb0[4], nb1[4];
    auto load = [&](uint4 (&xa)[4], uint4 (&xb0)[4], uint4 (&xb1)[4], int bt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xa[i] = va ? *reinterpret_cast<const uint4*>(pa + bt * 64 + i * 8) : z4;
            xb0[i] = vb0 ? *reinterpret_cast<const uint4*>(pb0 + bt * 64 + i * 8) : z4;
            xb1[i] = vb1 ? *reinterpret_cast<const uint4*>(pb1 + bt * 64 + i * 8) : z4;
        }
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    load(a, b0, b1, 0);
    for (int bt = 0; bt < nb; ++bt) {
        if (bt + 1 < nb) load(na, nb0, nb1, bt + 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc0 = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, a[i]), __builtin_bit_cast(h16x8, b0[i]), acc0, 0, 0, 0);
            acc1 = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, a[i]), __builtin_bit_cast(h16x8, b1[i]), acc1, 0, 0, 0);
        }
        if (bt + 1 < nb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = na[i]; b0[i] = nb0[i]; b1[i] = nb1[i]; }
        }
    }
    // D[m][n]: register 4g+e of lane (n = r, h) is row m = 8g + 4h + e
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[wave][8 * g + 4 * h + e][r] = acc0[4 * g + e];
            red[wave][8 * g + 4 * h + e][32 + r] = acc1[4 * g + e];
        }
    __syncthreads();
    const int row = tid >> 3, c0 = (tid & 7) * 8;
    const int m = bm + row, n0 = bn + c0;
    float v[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        f32x4 t = *reinterpret_cast<const f32x4*>(&red[0][row][c0 + 4 * q]);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(&red[w][row][c0 + 4 * q]);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] += u[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] = t[e];
    }
    const bool mv = m < p.M;  // N % 64 == 0 (launcher), so every column of the tile exists
    TC* C = reinterpret_cast<TC*>(p.C);
    if (p.epi == 0) {
        if (mv) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (p.bias) v[e] += p.bias[n0 + e];
                if (p.colscale) v[e] *= p.colscale[n0 + e];
            }
            if (p.pre) {
                float sv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) sv[e] = pre_save_fast(v[e], p.act);
                Out4<h16_t>::store(p.pre + (int64_t)m * p.ldp + n0, f32x4{sv[0], sv[1], sv[2], sv[3]});
                Out4<h16_t>::store(p.pre + (int64_t)m * p.ldp + n0 + 4, f32x4{sv[4], sv[5], sv[6], sv[7]});
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (p.act == SVOL_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
                else if (act_is_gelu(p.act)) v[e] = gelu_fast(v[e]);
                else if (p.act == SVOL_ACT_SIGMOID) v[e] = 1.f / (1.f + __expf(-v[e]));
            }
            if (p.res) {
                const TC* R = reinterpret_cast<const TC*>(p.res) + (int64_t)m * p.ldr + n0;
                const f32x4 r0 = Out4<TC>::load(R), r1 = Out4<TC>::load(R + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
            }
            if (p.act == SVOL_ACT_RELU_RES) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
        }
    } else {  // epi 1: v = acc * gelu'(aux), column sums of v
        if (mv) {
            const h16_t* X = p.aux + (int64_t)m * p.ldaux + n0;
            const f32x4 x0 = Out4<h16_t>::load(X), x1 = Out4<h16_t>::load(X + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] *= dact_fast(x0[e], p.act); v[4 + e] *= dact_fast(x1[e], p.act); }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        if (p.colsum) {
            __syncthreads();  // everyone has read the partial tiles
#pragma unroll
            for (int e = 0; e < 8; ++e) red[0][row][c0 + e] = v[e];
            __syncthreads();
            if (tid < SK_BN) {
                float sacc = 0.f;
#pragma unroll 8
                for (int i = 0; i < SK_BM; ++i) sacc += red[0][i][tid];
                atomicAdd(p.colsum + bn + tid, sacc);
            }
        }
    }
    if (mv) {
        Out4<TC>::store(C + (int64_t)m * p.ldc + n0, f32x4{v[0], v[1], v[2], v[3]});
        Out4<TC>::store(C + (int64_t)m * p.ldc + n0 + 4, f32x4{v[4], v[5], v[6], v[7]});
    }
}
__global__ __launch_bounds__(256) void gemm_nt_bf16_skinny_f32(FastArgs p) { gemm_nt_bf16_skinny_body<float>(p); }
__global__ __launch_bounds__(256) void gemm_nt_bf16_skinny_b16(FastArgs p) { gemm_nt_bf16_skinny_body<h16_t>(p); }

}  // namespace

int svol_gemm_ws_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias, int act,
                      void* pre, int64_t ldp, const void* res, int64_t ldr, int out_f32, const void* aux, int64_t ldaux,
                      float* colsum, int epi, const float* colscale, int64_t M, int64_t N, int64_t K, hipStream_t s);

int svol_gemm_n256_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias, int act,
                        void* pre, const void* res, int64_t ldr, int out_f32, int epi, const float* colscale, int64_t M,
                        int64_t N, int64_t K, int64_t kwrap, hipStream_t s);

// launcher used by gemm.hip's C-ABI entry points.  Returns SVOL_E_UNSUPPORTED when the shape does not
// qualify (the caller then uses the generic kernel).
int svol_gemm_nt_bf16_fast(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias,
                           int act, void* pre, int64_t ldp, const void* res, int64_t ldr, int out_f32, const void* aux,
                           int64_t ldaux, float* colsum, int epi, const float* colscale, int64_t M, int64_t N, int64_t K,
                           int64_t kwrap, hipStream_t s) {
    // kwrap != K: the split-weight product C = A [W_hi | W_lo]^T, K = 2 * kwrap, A has kwrap columns (svol_gemm_nt_split)
    if (kwrap != K && (2 * kwrap != K || kwrap % 32)) return SVOL_E_UNSUPPORTED;
    if (K % 32 || lda % 8 || ldb % 8 || !aligned16(A) || !aligned16(B)) return SVOL_E_UNSUPPORTED;
    // vector epilogue needs 8-byte (bf16) / 16-byte (f32) aligned rows; otherwise the scalar tail path is used per lane
    if ((int64_t)128 * lda * 2 >= (1ll << 31) || (int64_t)128 * ldb * 2 >= (1ll << 31)) return SVOL_E_UNSUPPORTED;
    const int celt = out_f32 ? 4 : 2;
    if ((ldc * celt) % (4 * celt) || (reinterpret_cast<uintptr_t>(C) % (4 * celt))) return SVOL_E_UNSUPPORTED;
    if (res && ((ldr * celt) % (4 * celt) || (reinterpret_cast<uintptr_t>(res) % (4 * celt)))) return SVOL_E_UNSUPPORTED;
    if (pre && (ldp % 4 || (reinterpret_cast<uintptr_t>(pre) % 8))) return SVOL_E_UNSUPPORTED;
    if (aux && (ldaux % 4 || (reinterpret_cast<uintptr_t>(aux) % 8))) return SVOL_E_UNSUPPORTED;
    FastArgs p{(const h16_t*)A, (const h16_t*)B, C, bias, colscale, (h16_t*)pre, res, (const h16_t*)aux, colsum,
               lda, ldb, ldc, ldp, ldr, ldaux, (int)M, (int)N, (int)K, act, epi};
    p.kwrap = (int)kwrap;
    if (kwrap == K) {   // K = 256, tall M: weight-stationary kernel (gemm_ws_bf16.hip)
        const int rc = svol_gemm_ws_bf16(A, lda, B, ldb, C, ldc, bias, act, pre, ldp, res, ldr, out_f32, aux, ldaux, colsum, epi,
                                         colscale, M, N, K, s);
        if (rc != SVOL_E_UNSUPPORTED) return rc;
    }
    {   // N = 256, deep K, tall M: full-width tiles (gemm_n256_bf16.hip)
        const int rc = svol_gemm_n256_bf16(A, lda, B, ldb, C, ldc, bias, act, pre, res, ldr, out_f32, epi, colscale, M, N, K, kwrap, s);
        if (rc != SVOL_E_UNSUPPORTED) return rc;
    }
    static const int skinny_max = getenv("SVOL_GEMM_SKINNY_M") ? atoi(getenv("SVOL_GEMM_SKINNY_M")) : 2048;
    if (M <= skinny_max && K % 256 == 0 && N % SK_BN == 0 && kwrap % (K / 4) == 0) {
        dim3 g((unsigned)(N / SK_BN), (unsigned)((M + SK_BM - 1) / SK_BM));
        if (out_f32) hipLaunchKernelGGL(gemm_nt_bf16_skinny_f32, g, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(gemm_nt_bf16_skinny_b16, g, dim3(256), 0, s, p);
        return hipGetLastError() == hipSuccess ? SVOL_OK : SVOL_E_LAUNCH;
    }
    dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128));
    if (grid.y > 65535u) return SVOL_E_UNSUPPORTED;
    // K-step 32 keeps the two-stage ring at 32 KiB per workgroup (4-5 workgroups per CU hide the DMA latency of the
    // short K = 256 loops); deep-K launches use 64-deep steps (half the barriers)
    static const int force_bk = getenv("SVOL_GEMM_BK") ? atoi(getenv("SVOL_GEMM_BK")) : 0;
    const bool bk64 = force_bk ? (force_bk == 64) : (K % 64 == 0 && K >= 1024);
    if (bk64 && K % 64 == 0 && kwrap % 64 == 0) {
        if (out_f32) hipLaunchKernelGGL(gemm_nt_bf16_f32_k64, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(gemm_nt_bf16_b16_k64, grid, dim3(256), 0, s, p);
    } else {
        if (out_f32) hipLaunchKernelGGL(gemm_nt_bf16_f32_k32, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(gemm_nt_bf16_b16_k32, grid, dim3(256), 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? SVOL_OK : SVOL_E_LAUNCH;
}

#ifndef SVOL_H16_FP16   // (the ResNet extractor is a bf16 / fp32 path)
// Implicit-GEMM convolution on NHWC bf16 activations: y[(n,ho,wo), co] = act(sum_{ky,kx,c} x[n, ho*s-p+ky,
// wo*s-p+kx, c] * w[co, (ky,kx,c)] + bias[co] (+ residual)), the 128x128 tile kernel above with the A operand gathered by the
// LDS-DMA loads themselves (padding = out-of-range buffer offsets, which read as zero).  C % 32 == 0.
extern "C" int svol_conv_nhwc(const void* x, const void* w, int64_t ldw, void* y, const float* bias, int act, const void* residual,
                              int64_t N, int64_t H, int64_t W, int64_t C, int64_t Cout, int64_t kh, int64_t kw, int64_t stride,
                              int64_t pad, int dtype, void* stream) {
    if (!x || !w || !y || N < 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0)
        return SVOL_E_INVALID;
    if (dtype != SVOL_BF16) return SVOL_E_UNSUPPORTED;
    const int64_t Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return SVOL_E_INVALID;
    const int64_t M = N * Ho * Wo, K = kh * kw * C;
    if (M == 0) return SVOL_OK;
    if (C % 32 || ldw < K || ldw % 8 || Cout % 4 || !aligned16(x) || !aligned16(w) || !aligned16(y)) return SVOL_E_UNSUPPORTED;
    if (residual && !aligned16(residual)) return SVOL_E_UNSUPPORTED;
    const int64_t a_bytes = N * H * W * C * 2;
    if (a_bytes >= 0x7ffffff0ll || M >= (1ll << 31) || (int64_t)128 * ldw * 2 >= (1ll << 31)) return SVOL_E_UNSUPPORTED;
    if (act != SVOL_ACT_NONE && act != SVOL_ACT_RELU && act != SVOL_ACT_RELU_RES) return SVOL_E_UNSUPPORTED;
    FastArgs p{(const h16_t*)x, (const h16_t*)w, y, bias, nullptr, nullptr, residual, nullptr, nullptr,
               0, ldw, Cout, 0, Cout, 0, (int)M, (int)Cout, (int)K, act, 0,
               (int)H, (int)W, (int)C, (int)kw, (int)stride, (int)pad, (int)Ho, (int)Wo, a_bytes, (int)K};
    dim3 grid((unsigned)((Cout + 127) / 128), (unsigned)((M + 127) / 128));
    if (grid.y > 65535u) return SVOL_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (C % 64 == 0 && K >= 1024) hipLaunchKernelGGL(conv_nhwc_bf16_k64, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_nhwc_bf16_k32, grid, dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? SVOL_OK : SVOL_E_LAUNCH;
}
#endif
