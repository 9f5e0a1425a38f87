// bf16 NT GEMM, K = 256, weight-stationary (gfx950):  C[M,N] = epilogue(A[M,256] * W[N,256]^T).
//
// Every projection of the d = 256 model has K = 256 and M = B*L = 50k rows: the tiled kernel of
// gemm_bf16.hip re-stages a 64 KiB weight tile through LDS for every 128 rows, spends as many LDS-DMA and
// ds_read instructions on W as on A, and pays a pipeline fill + a 4-pass LDS epilogue per 8 K-steps — it
// reaches ~2.2 TB/s of algorithmic traffic on these HBM-bound shapes.  Here:
//
//  * a workgroup owns 256 output columns (64 per wave) and keeps ITS slice of W in registers for its whole
//    life: 4 column tiles x 8 K-steps of MFMA A-operand fragments = 128 VGPRs per lane, loaded once;
//  * the activation rows stream through a ring of 16-row slabs (8 KiB) in LDS, filled by LDS-DMA, three
//    slabs in flight behind a counted s_waitcnt vmcnt (loads, LDS-DMA and the epilogue's stores retire in
//    issue order), one raw barrier per slab; a slab's fragments are 8 ds_read_b128 per wave for 32 MFMAs;
//  * operands are swapped (W is the MFMA A operand) and W's rows are permuted when the fragments are
//    loaded, so that a lane ends up with 8 CONSECUTIVE output columns of one row per pair of accumulators:
//    the epilogue runs straight out of the accumulators (bias, column scale, GELU, pre-activation copy,
//    16-byte stores), no LDS round trip;
//  * operands the epilogue needs per element (the fp32 residual stream, the saved pre-activation of the
//    GELU backward) ride the same ring as extra slab planes, so nothing is loaded "late" (a late load would
//    force every older DMA to retire first).
#include <cstdlib>

#include "common.h"

namespace {

struct WsArgs {
    const h16_t* A; const h16_t* W; void* C;
    const float* bias; const float* colscale; h16_t* pre; const void* res; const h16_t* aux; float* colsum;
    int64_t lda, ldw, ldc, ldp, ldr, ldaux;
    int M, N, act, tiles_per_wg, nchunk;
};

typedef __attribute__((address_space(3))) void* lds_void_ptr;

constexpr int WS_K = 256;
constexpr int TR = 16;                 // rows per slab
constexpr int A_PLANE = TR * WS_K * 2;  // 8 KiB

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void ws_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void ws_wait_vm_dyn(int n) {
    switch (n) {
#define C_(x) case x: ws_wait_vm<x>(); break;
        C_(0) C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16)
        C_(17) C_(18) C_(19) C_(20) C_(21) C_(22) C_(23) C_(24) C_(25) C_(26) C_(27) C_(28) C_(29) C_(30) C_(31) C_(32)
#undef C_
        default: ws_wait_vm<32>(); break;  // never more than asked for: 32 <= n here
    }
}

// column (within the wave's 64) of MFMA output row i of accumulator tile t: lane fq then holds, for its
// row, columns (t>>1)*32 + fq*8 + [0,8) in the accumulator pair (t & ~1, t | 1)
__device__ __forceinline__ int ws_col(int t, int i) { return (t >> 1) * 32 + (i >> 2) * 8 + (t & 1) * 4 + (i & 3); }

// MODE 0: C (bf16) = act((acc + bias) * colscale) [+ pre-activation copy]
// MODE 1: C (f32)  = acc + bias + residual(f32)            (the fp32 residual stream; extra plane: residual slab)
// MODE 2: C (bf16) = acc * gelu'(aux), column sums          (extra plane: aux slab)
template <int MODE> struct WsCfg;
template <> struct WsCfg<0> { static constexpr int SLOT = A_PLANE, STG = 4, ND = 2; };
template <> struct WsCfg<1> { static constexpr int SLOT = A_PLANE + TR * 256 * 4, STG = 3, ND = 6; };
template <> struct WsCfg<2> { static constexpr int SLOT = 2 * A_PLANE, STG = 4, ND = 4; };

template <int MODE>
__device__ __forceinline__ void gemm_ws_body(const WsArgs& p) {
    constexpr int SLOT = WsCfg<MODE>::SLOT, STG = WsCfg<MODE>::STG, ND = WsCfg<MODE>::ND;
    __shared__ __attribute__((aligned(1024))) char smem[STG * SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    // 1-D grid, row chunk fastest: workgroup ids go round-robin over the 8 XCDs, so with nchunk % 8 == 0 the column
    // groups that share a row chunk (and re-read its A slabs) sit on ONE XCD and meet in its L2
    // (rocprofv3 FETCH_SIZE of fc1, 8 column groups: 197 MB with the column group fastest = every re-read from memory)
    const int chunk_id = blockIdx.x % p.nchunk, cg = blockIdx.x / p.nchunk;
    const int cg0 = cg * 256;                  // first column of this workgroup
    const int c0 = cg0 + wave * 64;            // first column of this wave (N % 64 == 0)
    const bool active = c0 < p.N;              // idle waves still move their share of the slabs
    const int tile0 = chunk_id * p.tiles_per_wg;
    const int row_base = tile0 * TR;
    const int rows_here = min(p.tiles_per_wg * TR, p.M - row_base);
    const int ntile = (rows_here + TR - 1) / TR;
    // epilogue stores per slab issued by this wave (an idle wave issues none)
    const int ns = !active ? 0 : (MODE == 0 ? (p.pre ? 4 : 2) : (MODE == 1 ? 4 : 2));

    // ---- slab DMA geometry -------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.A + (int64_t)row_base * p.lda), 0, (int)((((int64_t)rows_here - 1) * p.lda + WS_K) * 2), 0x00020000);
    // A plane: 16 rows x 32 chunks(16 B); instruction i covers rows 2i, 2i+1; physical chunk s holds source chunk s ^ row
    int voffA[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 2 + (lane >> 5), s = lane & 31;
        voffA[j] = (int)(((int64_t)row * p.lda + ((s ^ row) & 31) * 8) * 2);
    }
    __amdgpu_buffer_rsrc_t rX = rA;
    int voffX[4] = {0, 0, 0, 0};
    if constexpr (MODE == 1) {  // residual plane: 16 rows x 64 chunks (f32, this workgroup's 256 columns); 1 row per instruction
        const float* R = reinterpret_cast<const float*>(p.res) + (int64_t)row_base * p.ldr + cg0;
        const int cols = min(256, p.N - cg0);
        rX = __builtin_amdgcn_make_buffer_rsrc((void*)R, 0, (int)((((int64_t)rows_here - 1) * p.ldr + cols) * 4), 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wave * 4 + j, s = lane;
            voffX[j] = (int)(((int64_t)row * p.ldr + (((s & 48) | ((s ^ row) & 15))) * 4) * 4);
        }
    } else if constexpr (MODE == 2) {  // aux plane: like the A plane (bf16, 256 columns)
        const h16_t* X = p.aux + (int64_t)row_base * p.ldaux + cg0;
        const int cols = min(256, p.N - cg0);
        rX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)((((int64_t)rows_here - 1) * p.ldaux + cols) * 2), 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (wave * 2 + j) * 2 + (lane >> 5), s = lane & 31;
            voffX[j] = (int)(((int64_t)row * p.ldaux + ((s ^ row) & 31) * 8) * 2);
        }
    }
    auto issue = [&](int slot, int tile) {
        char* sl = smem + slot * SLOT;
        const int soA = (int)((int64_t)tile * TR * p.lda * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_ptr)(sl + (wave * 2 + j) * 1024), 16, voffA[j], soA, 0, 0);
        if constexpr (MODE == 1) {
            const int so = (int)((int64_t)tile * TR * p.ldr * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void_ptr)(sl + A_PLANE + (wave * 4 + j) * 1024), 16, voffX[j], so, 0, 0);
        } else if constexpr (MODE == 2) {
            const int so = (int)((int64_t)tile * TR * p.ldaux * 2);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void_ptr)(sl + A_PLANE + (wave * 2 + j) * 1024), 16, voffX[j], so, 0, 0);
        }
    };

    // ---- this wave's slice of W, as MFMA A-operand fragments, for the whole kernel ---------------------
    uint4 wf[4][8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const h16_t* wrow = p.W + (int64_t)(c0 + ws_col(t, fr)) * p.ldw + fq * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wf[t][ks] = active ? *reinterpret_cast<const uint4*>(wrow + ks * 32) : make_uint4(0, 0, 0, 0);
    }
    // per-lane epilogue constants: columns cbase(hf) + [0,8), hf = 0,1
    float bias8[2][8], scale8[2][8];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = c0 + hf * 32 + fq * 8 + e;
            bias8[hf][e] = (active && p.bias) ? p.bias[n] : 0.f;
            scale8[hf][e] = (MODE == 0 && active && p.colscale) ? p.colscale[n] : 1.f;
        }
    float csum[2][8];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[hf][e] = 0.f;

    // fragment read addresses inside a slab: row fr, logical chunk ks*4 + fq  ->  physical (chunk ^ fr)
    const unsigned lds0 = (unsigned)(size_t)(lds_void_ptr)smem;
    const unsigned aoff = (unsigned)(fr * 512 + ((fq ^ fr) & 3) * 16);  // + ((ks*4) ^ (fr & 12)) * 16 per K-step (below)

#pragma unroll
    for (int s = 0; s < STG - 1; ++s)
        if (s < ntile) issue(s, s);

    for (int j = 0; j < ntile; ++j) {
        // slab j is complete once at most the younger operations of this wave are outstanding
        ws_wait_vm_dyn(ND * min(STG - 2, ntile - 1 - j) + ns * min(j, STG - 1));
        __builtin_amdgcn_s_barrier();
        if (j + STG - 1 < ntile) issue((j + STG - 1) % STG, j + STG - 1);
        const unsigned sl = lds0 + (unsigned)((j % STG) * SLOT);
        // A fragments (MFMA B operand): inline asm so that hipcc does not drain the ring (it cannot tell which slot
        // an LDS-DMA wrote and would put s_waitcnt vmcnt(0) in front of any LDS read it can see)
        uint4 af[8];
        {
            const unsigned b = sl + aoff;
            const unsigned x = (unsigned)((fr & 12) * 16);  // chunk bits 2..3 of the swizzle
            asm volatile(
                "ds_read_b128 %0, %8\n\t"
                "ds_read_b128 %1, %9\n\t"
                "ds_read_b128 %2, %10\n\t"
                "ds_read_b128 %3, %11\n\t"
                "ds_read_b128 %4, %8 offset:256\n\t"
                "ds_read_b128 %5, %9 offset:256\n\t"
                "ds_read_b128 %6, %10 offset:256\n\t"
                "ds_read_b128 %7, %11 offset:256\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(af[0]), "=&v"(af[1]), "=&v"(af[2]), "=&v"(af[3]), "=&v"(af[4]), "=&v"(af[5]), "=&v"(af[6]), "=&v"(af[7])
                : "v"(b + ((0u * 64) ^ x)), "v"(b + ((1u * 64) ^ x)), "v"(b + ((2u * 64) ^ x)), "v"(b + ((3u * 64) ^ x))
                : "memory");
        }
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = SVOL_MFMA_16x16x32_H16(__builtin_bit_cast(h16x8, wf[t][ks]),
                                                                 __builtin_bit_cast(h16x8, af[ks]), acc[t], 0, 0, 0);
        // ---- epilogue from the accumulators: this lane's row, two runs of 8 consecutive columns ----------
        const int m = row_base + j * TR + fr;
        float v[2][8];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[hf][e] = acc[2 * hf + (e >> 2)][e & 3];
        if constexpr (MODE == 0) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[hf][e] = (v[hf][e] + bias8[hf][e]) * scale8[hf][e];
            if (active && m < p.M) {
                if (p.pre) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        h16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (h16_t)pre_save_fast(v[hf][e], p.act);
                        *reinterpret_cast<h16x8*>(p.pre + (int64_t)m * p.ldp + c0 + hf * 32 + fq * 8) = o;
                    }
                }
                h16_t* C = reinterpret_cast<h16_t*>(p.C);
                if (act_is_gelu(p.act)) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[hf][e] = gelu_fast(v[hf][e]);
                } else if (p.act == SVOL_ACT_RELU) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[hf][e] = fmaxf(v[hf][e], 0.f);
                }
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    h16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (h16_t)v[hf][e];
                    *reinterpret_cast<h16x8*>(C + (int64_t)m * p.ldc + c0 + hf * 32 + fq * 8) = o;
                }
            }
        } else if constexpr (MODE == 1) {
            // residual slab: row fr, f32 chunks (wave*16 + hf*8 + fq*2 + {0,1}), low 4 chunk bits swizzled by the row
            f32x4 rr[4];
            {
                const unsigned rb = sl + A_PLANE + (unsigned)(fr * 1024 + wave * 256);
                const unsigned q0 = (unsigned)(((fq * 2) ^ fr) & 15), q1 = (unsigned)(((fq * 2 + 1) ^ fr) & 15);
                const unsigned q2 = (unsigned)(((8 + fq * 2) ^ fr) & 15), q3 = (unsigned)(((8 + fq * 2 + 1) ^ fr) & 15);
                asm volatile(
                    "ds_read_b128 %0, %4\n\t"
                    "ds_read_b128 %1, %5\n\t"
                    "ds_read_b128 %2, %6\n\t"
                    "ds_read_b128 %3, %7\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(rr[0]), "=&v"(rr[1]), "=&v"(rr[2]), "=&v"(rr[3])
                    : "v"(rb + q0 * 16), "v"(rb + q1 * 16), "v"(rb + q2 * 16), "v"(rb + q3 * 16)
                    : "memory");
            }
            if (active && m < p.M) {
                float* C = reinterpret_cast<float*>(p.C);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = v[hf][4 * q + e] + bias8[hf][4 * q + e] + rr[2 * hf + q][e];
                        *reinterpret_cast<f32x4*>(C + (int64_t)m * p.ldc + c0 + hf * 32 + fq * 8 + 4 * q) = o;
                    }
            }
        } else {
            // aux slab (bf16): row fr, chunk (wave*8 + hf*4 + fq), swizzled like the A plane
            uint4 ax[2];
            {
                const unsigned xb = sl + A_PLANE + (unsigned)(fr * 512);
                const unsigned q0 = (unsigned)(((wave * 8 + fq) ^ fr) & 31), q1 = (unsigned)(((wave * 8 + 4 + fq) ^ fr) & 31);
                asm volatile(
                    "ds_read_b128 %0, %2\n\t"
                    "ds_read_b128 %1, %3\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(ax[0]), "=&v"(ax[1])
                    : "v"(xb + q0 * 16), "v"(xb + q1 * 16)
                    : "memory");
            }
            const bool ok = active && m < p.M;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const h16x8 a8 = __builtin_bit_cast(h16x8, ax[hf]);
                h16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float d = v[hf][e] * dact_fast((float)a8[e], p.act);  // rows past M: acc = 0 and aux = 0 (zero-filled slabs)
                    d = ok ? d : 0.f;
                    csum[hf][e] += d;
                    o[e] = (h16_t)d;
                }
                if (ok) *reinterpret_cast<h16x8*>(reinterpret_cast<h16_t*>(p.C) + (int64_t)m * p.ldc + c0 + hf * 32 + fq * 8) = o;
            }
        }
    }
    if constexpr (MODE == 2) {
        if (p.colsum) {
            // the 16 lanes fr = 0..15 of a fq group hold partial sums of the same 16 columns
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float s = csum[hf][e];
                    s += __shfl_xor(s, 1, 64);
                    s += __shfl_xor(s, 2, 64);
                    s += __shfl_xor(s, 4, 64);
                    s += __shfl_xor(s, 8, 64);
                    if (fr == 0 && active) atomicAdd(p.colsum + c0 + hf * 32 + fq * 8 + e, s);
                }
        }
    }
}

__global__ __launch_bounds__(256, 2) void gemm_ws_bf16_m0(WsArgs p) { gemm_ws_body<0>(p); }
__global__ __launch_bounds__(256, 2) void gemm_ws_bf16_m1(WsArgs p) { gemm_ws_body<1>(p); }
__global__ __launch_bounds__(256, 2) void gemm_ws_bf16_m2(WsArgs p) { gemm_ws_body<2>(p); }

// ---- software-pipelined body (bf16 outputs: MODE 0 and MODE 2) ---------------------------------------------------
// The plain loop above runs a slab's stages back to back in every wave — wait, barrier, ds_read (full LDS latency exposed),
// 32 MFMAs, the epilogue's VALU, the stores — and with W holding 128 VGPRs only two waves share a SIMD, so the stage
// times ADD (stage ablation on MI355X, fc1 shape: 79 us = 29 skeleton + 8 ds_read + ~20 MFMA + ~25 epilogue/stores).
// Here slab j+1's fragment reads are issued as soon as slab j's MFMAs are, and slab j's epilogue is ordinary VALU code
// in the same basic block as slab j+1's MFMAs (the activation, the epilogue flags are template parameters: no branches),
// so it fills the LDS latency and the matrix pipe's shadow.  Everything slab j's epilogue needs from LDS (MODE 2: the
// saved pre-activation) is read with the fragments, because sync(j) hands slab j's slot back to the DMA ring.

// A THIRD compiler trap (round 6).  buffer_store_dwordx4 reads its 16 bytes of data per lane over several cycles, a quarter of the wave
// at a time; a VALU instruction that OVERWRITES one of the four data registers right behind the store can win the race for the last
// lanes of the later dwords.  LLVM pads this hazard only for stores WITHOUT an SGPR soffset (GCNHazardRecognizer::createsVALUHazard);
// with one — every epilogue store here: the slab advances the scalar offset — hipcc (ROCm 7.2) emitted
//     buffer_store_dwordx4 v[0:3], v212, s[0:3], s4 offen ; v_pk_mul_f32 v[0:1], v[8:9], v[8:9]
// and on gfx950 the `pre` output carried the low halves of fp32 squares in element 2 of rows 12-15 of a slab (lanes 48-63, dword 1):
// NaN / +-65 440 in ~1 element of 5 000, sporadic, `out` untouched.  Round 1-5's epilogues never put a writer of the data registers
// directly behind a store (allocation luck); a shorter GELU epilogue tried in round 6 did (that epilogue itself measured +-0 in the
// step and was not kept).  The data registers are kept alive across two wait states here.
__device__ __forceinline__ void store_b128_guarded(u32x4_t v, __amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voffset, soffset, 0);
    asm volatile("s_nop 1" ::"v"(v));
}

template <int MODE, int ACT, bool PRE, bool SCALE, int VPM>
__device__ __forceinline__ void gemm_wsp_body(const WsArgs& p) {
    static_assert(MODE == 0 || MODE == 2, "bf16 outputs only");
    constexpr int SLOT = WsCfg<MODE>::SLOT, STG = 4, ND = WsCfg<MODE>::ND;
    constexpr int NS = MODE == 0 ? (PRE ? 4 : 2) : 2;
    __shared__ __attribute__((aligned(1024))) char smem[STG * SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int chunk_id = blockIdx.x % p.nchunk, cg = blockIdx.x / p.nchunk;
    const int cg0 = cg * 256;
    const int c0 = cg0 + wave * 64;
    const bool active = c0 < p.N;
    const int tile0 = chunk_id * p.tiles_per_wg;
    const int row_base = tile0 * TR;
    const int rows_here = min(p.tiles_per_wg * TR, p.M - row_base);
    const int ntile = (rows_here + TR - 1) / TR;

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.A + (int64_t)row_base * p.lda), 0, (int)((((int64_t)rows_here - 1) * p.lda + WS_K) * 2), 0x00020000);
    int voffA[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 2 + (lane >> 5), s = lane & 31;
        voffA[j] = (int)(((int64_t)row * p.lda + ((s ^ row) & 31) * 8) * 2);
    }
    __amdgpu_buffer_rsrc_t rX = rA;
    int voffX[2] = {0, 0};
    if constexpr (MODE == 2) {
        const h16_t* X = p.aux + (int64_t)row_base * p.ldaux + cg0;
        const int cols = min(256, p.N - cg0);
        rX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)((((int64_t)rows_here - 1) * p.ldaux + cols) * 2), 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (wave * 2 + j) * 2 + (lane >> 5), s = lane & 31;
            voffX[j] = (int)(((int64_t)row * p.ldaux + ((s ^ row) & 31) * 8) * 2);
        }
    }
    auto issue = [&](int slot, int tile) {
        char* sl = smem + slot * SLOT;
        const int soA = (int)((int64_t)tile * TR * p.lda * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_ptr)(sl + (wave * 2 + j) * 1024), 16, voffA[j], soA, 0, 0);
        if constexpr (MODE == 2) {
            const int so = (int)((int64_t)tile * TR * p.ldaux * 2);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void_ptr)(sl + A_PLANE + (wave * 2 + j) * 1024), 16, voffX[j], so, 0, 0);
        }
    };

    uint4 wf[4][8];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const h16_t* wrow = p.W + (int64_t)(c0 + ws_col(t, fr)) * p.ldw + fq * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wf[t][ks] = active ? *reinterpret_cast<const uint4*>(wrow + ks * 32) : make_uint4(0, 0, 0, 0);
    }
    float bias8[2][8], scale8[2][8];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = c0 + hf * 32 + fq * 8 + e;
            bias8[hf][e] = (MODE == 0 && active && p.bias) ? p.bias[n] : 0.f;
            scale8[hf][e] = (MODE == 0 && SCALE && active && p.colscale) ? p.colscale[n] : 1.f;
        }
    float csum[2][8];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[hf][e] = 0.f;

    const unsigned lds0 = (unsigned)(size_t)(lds_void_ptr)smem;
    const unsigned aoff = (unsigned)(fr * 512 + ((fq ^ fr) & 3) * 16);
    const unsigned ax_x = (unsigned)((fr & 12) * 16);
    const unsigned xq0 = (unsigned)(A_PLANE + fr * 512 + (((wave * 8 + fq) ^ fr) & 31) * 16);
    const unsigned xq1 = (unsigned)(A_PLANE + fr * 512 + (((wave * 8 + 4 + fq) ^ fr) & 31) * 16);
    // outputs go through buffer stores: the range check drops rows past M (and everything of an idle wave: 0 records), so
    // the epilogue has no branch — one basic block with the MFMAs — and every wave issues the same number of stores;
    // per-lane offset constant, the slab advances the scalar offset (no address arithmetic in the loop)
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<h16_t*>(p.C) + (int64_t)row_base * p.ldc), 0,
        active ? (int)((((int64_t)rows_here - 1) * p.ldc + p.N) * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(PRE ? p.pre + (int64_t)row_base * p.ldp : nullptr), 0,
        (PRE && active) ? (int)((((int64_t)rows_here - 1) * p.ldp + p.N) * 2) : 0, 0x00020000);
    const int voffC = (int)(((int64_t)fr * p.ldc + c0 + fq * 8) * 2), voffP = (int)(((int64_t)fr * p.ldp + c0 + fq * 8) * 2);
    const int cstep = (int)(TR * p.ldc * 2), pstep = (int)(TR * p.ldp * 2);

    u32x4_t af[8], ax[2] = {u32x4_t{0, 0, 0, 0}, u32x4_t{0, 0, 0, 0}};
    // fragment (and MODE 2: aux) reads of the slab in `slot`, NOT waited for
    auto read_frags = [&](int slot) {
        const unsigned sl = lds0 + (unsigned)(slot * SLOT);
        const unsigned b = sl + aoff;
        asm volatile(
            "ds_read_b128 %0, %8\n\t"
            "ds_read_b128 %1, %9\n\t"
            "ds_read_b128 %2, %10\n\t"
            "ds_read_b128 %3, %11\n\t"
            "ds_read_b128 %4, %8 offset:256\n\t"
            "ds_read_b128 %5, %9 offset:256\n\t"
            "ds_read_b128 %6, %10 offset:256\n\t"
            "ds_read_b128 %7, %11 offset:256"
            : "=&v"(af[0]), "=&v"(af[1]), "=&v"(af[2]), "=&v"(af[3]), "=&v"(af[4]), "=&v"(af[5]), "=&v"(af[6]), "=&v"(af[7])
            : "v"(b + ((0u * 64) ^ ax_x)), "v"(b + ((1u * 64) ^ ax_x)), "v"(b + ((2u * 64) ^ ax_x)), "v"(b + ((3u * 64) ^ ax_x))
            : "memory");
        if constexpr (MODE == 2) {
            asm volatile(
                "ds_read_b128 %0, %2\n\t"
                "ds_read_b128 %1, %3"
                : "=&v"(ax[0]), "=&v"(ax[1])
                : "v"(sl + xq0), "v"(sl + xq1)
                : "memory");
        }
    };
    auto wait_frags = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]), "+v"(af[6]), "+v"(af[7]),
                       "+v"(ax[0]), "+v"(ax[1]));
    };
    auto mfma_slab = [&](f32x4 (&acc)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = SVOL_MFMA_16x16x32_H16(__builtin_bit_cast(h16x8, wf[t][ks]),
                                                                 __builtin_bit_cast(h16x8, af[ks]), acc[t], 0, 0, 0);
    };
    // epilogue of slab j from its accumulators (and, MODE 2, its aux fragment)
    auto epilogue = [&](int j, const f32x4 (&acc)[4], const u32x4_t (&axj)[2]) {
        float v[2][8];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[hf][e] = acc[2 * hf + (e >> 2)][e & 3];
        if constexpr (MODE == 0) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[hf][e] += bias8[hf][e];
                    if constexpr (SCALE) v[hf][e] *= scale8[hf][e];
                }
            if constexpr (PRE) {
                h16x8 o[2];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[hf][e] = (h16_t)pre_save_fast(v[hf][e], ACT);
                store_b128_guarded(__builtin_bit_cast(u32x4_t, o[0]), rP, voffP, j * pstep);
                store_b128_guarded(__builtin_bit_cast(u32x4_t, o[1]), rP, voffP + 64, j * pstep);
            }
            h16x8 o[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = v[hf][e];
                    if constexpr (ACT == SVOL_ACT_GELU || ACT == SVOL_ACT_GELU_D) y = gelu_fast(y);
                    if constexpr (ACT == SVOL_ACT_RELU) y = fmaxf(y, 0.f);
                    o[hf][e] = (h16_t)y;
                }
            store_b128_guarded(__builtin_bit_cast(u32x4_t, o[0]), rC, voffC, j * cstep);
            store_b128_guarded(__builtin_bit_cast(u32x4_t, o[1]), rC, voffC + 64, j * cstep);
        } else {
            h16x8 o[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const h16x8 a8 = __builtin_bit_cast(h16x8, axj[hf]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    // rows past M and idle waves: acc = 0 (zero-filled slabs, zero W) and a finite derivative: d = 0
                    const float d = v[hf][e] * dact_fast((float)a8[e], ACT);
                    csum[hf][e] += d;
                    o[hf][e] = (h16_t)d;
                }
            }
            store_b128_guarded(__builtin_bit_cast(u32x4_t, o[0]), rC, voffC, j * cstep);
            store_b128_guarded(__builtin_bit_cast(u32x4_t, o[1]), rC, voffC + 64, j * cstep);
        }
    };

#pragma unroll
    for (int s = 0; s < STG; ++s)
        if (s < ntile) issue(s, s);
    ws_wait_vm_dyn(ND * min(STG - 1, ntile - 1));
    __builtin_amdgcn_s_barrier();
    read_frags(0);
    wait_frags();
    f32x4 acc[4];
    mfma_slab(acc);
    for (int j = 0; j + 1 < ntile; ++j) {
        // ---- sync(j): slab j+1 has landed everywhere, slab j's slot is free --------------------------------------
        // operations of this wave younger than slab j+1's DMA: the DMAs of slabs j+2 .. j+STG-1 and the stores of the
        // epilogues j+1-STG .. j-1
        const int nd = min(STG - 2, ntile - 2 - j), nst = min(j, STG - 1);
        if (nd == STG - 2 && nst == STG - 1) ws_wait_vm<ND * (STG - 2) + NS * (STG - 1)>();
        else ws_wait_vm_dyn(ND * nd + NS * nst);
        __builtin_amdgcn_s_barrier();
        if (j + STG < ntile) issue(j % STG, j + STG);
        f32x4 accp[4];
        u32x4_t axp[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) accp[t] = acc[t];
        axp[0] = ax[0];
        axp[1] = ax[1];
        read_frags((j + 1) % STG);
        wait_frags();
        mfma_slab(acc);
        epilogue(j, accp, axp);
        // hipcc would issue the 32 MFMAs back to back and the epilogue after them; spread the epilogue's VALU work over
        // the matrix pipe's shadow instead (a 16x16x32 MFMA occupies the pipe for 16 cycles, its issue takes 4)
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        }
    }
    {
        u32x4_t axp[2] = {ax[0], ax[1]};
        epilogue(ntile - 1, acc, axp);
    }
    if constexpr (MODE == 2) {
        if (p.colsum) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float s = csum[hf][e];
                    s += __shfl_xor(s, 1, 64);
                    s += __shfl_xor(s, 2, 64);
                    s += __shfl_xor(s, 4, 64);
                    s += __shfl_xor(s, 8, 64);
                    if (fr == 0 && active) atomicAdd(p.colsum + c0 + hf * 32 + fq * 8 + e, s);
                }
        }
    }
}

// VPM: VALU instructions scheduled behind each MFMA (about the epilogue's VALU count / 32)
#define WSP_KERNEL(name, MODE, ACT, PRE, SCALE, VPM) \
    __global__ __launch_bounds__(256, 2) void name(WsArgs p) { gemm_wsp_body<MODE, ACT, PRE, SCALE, VPM>(p); }
WSP_KERNEL(gemm_wsp_bf16_none, 0, SVOL_ACT_NONE, false, false, 2)
WSP_KERNEL(gemm_wsp_bf16_none_scale, 0, SVOL_ACT_NONE, false, true, 2)
WSP_KERNEL(gemm_wsp_bf16_gelu, 0, SVOL_ACT_GELU, false, false, 7)
WSP_KERNEL(gemm_wsp_bf16_gelu_pre, 0, SVOL_ACT_GELU, true, false, 8)
WSP_KERNEL(gemm_wsp_bf16_gelu_dpre, 0, SVOL_ACT_GELU_D, true, false, 9)
WSP_KERNEL(gemm_wsp_bf16_relu, 0, SVOL_ACT_RELU, false, false, 2)
WSP_KERNEL(gemm_wsp_bf16_dgelu, 2, SVOL_ACT_GELU, false, false, 9)
WSP_KERNEL(gemm_wsp_bf16_dmul, 2, SVOL_ACT_GELU_D, false, false, 3)
WSP_KERNEL(gemm_wsp_bf16_drelu, 2, SVOL_ACT_RELU, false, false, 3)
#undef WSP_KERNEL

}  // namespace

// launcher used by gemm_bf16.hip's fast-path dispatcher.  Returns SVOL_E_UNSUPPORTED when the call does not fit.
int svol_gemm_ws_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias, int act,
                      void* pre, int64_t ldp, const void* res, int64_t ldr, int out_f32, const void* aux, int64_t ldaux,
                      float* colsum, int epi, const float* colscale, int64_t M, int64_t N, int64_t K, hipStream_t s) {
    static const bool off = getenv("SVOL_GEMM_NO_WS") != nullptr;
    if (off || K != WS_K || N % 64 || M < 4096) return SVOL_E_UNSUPPORTED;
    int mode;
    if (epi == 1) {
        if (!aux || out_f32) return SVOL_E_UNSUPPORTED;
        mode = 2;
    } else if (out_f32) {
        if (!res || act != SVOL_ACT_NONE || pre || colscale) return SVOL_E_UNSUPPORTED;
        mode = 1;
    } else {
        if (res || (act != SVOL_ACT_NONE && act != SVOL_ACT_GELU && act != SVOL_ACT_RELU && act != SVOL_ACT_GELU_D)) return SVOL_E_UNSUPPORTED;
        if (act == SVOL_ACT_GELU_D && !pre) return SVOL_E_INVALID;
        mode = 0;
    }
    auto al16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) == 0; };
    if (lda % 8 || ldw % 8 || !al16(A) || !al16(W) || !al16(C)) return SVOL_E_UNSUPPORTED;
    if (mode == 1 ? (ldc % 4 || ldr % 4 || !al16(res)) : (ldc % 8 != 0)) return SVOL_E_UNSUPPORTED;
    if (pre && (ldp % 8 || !al16(pre))) return SVOL_E_UNSUPPORTED;
    if (mode == 2 && (ldaux % 8 || !al16(aux))) return SVOL_E_UNSUPPORTED;
    const int64_t ntile = (M + TR - 1) / TR;
    const int64_t ncg = (N + 255) / 256;
    // slabs per workgroup: every workgroup first reads its 128 KiB slice of W, so few long-lived workgroups win as
    // long as they fill the chip (measured, M = 50176: N = 256 best at ~250 workgroups, N = 2048 at ~512)
    static const int force_tpw = getenv("SVOL_WS_TPW") ? atoi(getenv("SVOL_WS_TPW")) : 0;
    int64_t tpw = force_tpw ? force_tpw : (ntile * ncg + (ncg <= 2 ? 255 : 511)) / (ncg <= 2 ? 256 : 512);
    if (!force_tpw) {
        if (tpw < 4) tpw = 4;
        if (tpw > 64) tpw = 64;
    }
    const int64_t ldmax = lda > ldr * 2 ? lda : ldr * 2;
    int64_t ldbig = ldmax > ldaux ? ldmax : ldaux;  // 32-bit buffer offsets inside one workgroup's rows (loads and stores)
    if (ldc > ldbig) ldbig = ldc;
    if (pre && ldp > ldbig) ldbig = ldp;
    if (tpw * TR * ldbig * 2 >= (1ll << 31)) return SVOL_E_UNSUPPORTED;
    int64_t nchunk = (ntile + tpw - 1) / tpw;
    if (ncg > 1 && nchunk > 8 && nchunk % 8) {  // a multiple of the 8 XCDs when column groups share rows
        const int64_t n8 = (nchunk + 7) / 8 * 8;
        const int64_t t8 = (ntile + n8 - 1) / n8;
        if (t8 >= 1 && (ntile + t8 - 1) / t8 == n8) { tpw = t8; nchunk = n8; }
    }
    if (nchunk * ncg > (1ll << 30)) return SVOL_E_UNSUPPORTED;
    WsArgs p{(const h16_t*)A, (const h16_t*)W, C, bias, colscale, (h16_t*)pre, res, (const h16_t*)aux, colsum,
             lda, ldw, ldc, ldp, ldr, ldaux, (int)M, (int)N, act, (int)tpw, (int)nchunk};
    dim3 grid((unsigned)(ncg * nchunk));
    static const bool no_pipe = getenv("SVOL_WS_NO_PIPE") != nullptr;
    void (*kp)(WsArgs) = nullptr;
    if (!no_pipe && mode == 0) {
        if (act == SVOL_ACT_NONE && !pre) kp = colscale ? gemm_wsp_bf16_none_scale : gemm_wsp_bf16_none;
        else if (act == SVOL_ACT_GELU && !colscale) kp = pre ? gemm_wsp_bf16_gelu_pre : gemm_wsp_bf16_gelu;
        else if (act == SVOL_ACT_GELU_D && !colscale) kp = gemm_wsp_bf16_gelu_dpre;
        else if (act == SVOL_ACT_RELU && !pre && !colscale) kp = gemm_wsp_bf16_relu;
    } else if (!no_pipe && mode == 2) {
        kp = act == SVOL_ACT_RELU ? gemm_wsp_bf16_drelu : (act == SVOL_ACT_GELU_D ? gemm_wsp_bf16_dmul : gemm_wsp_bf16_dgelu);
    }
    if (kp) hipLaunchKernelGGL(kp, grid, dim3(256), 0, s, p);
    else if (mode == 0) hipLaunchKernelGGL(gemm_ws_bf16_m0, grid, dim3(256), 0, s, p);
    else if (mode == 1) hipLaunchKernelGGL(gemm_ws_bf16_m1, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(gemm_ws_bf16_m2, grid, dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? SVOL_OK : SVOL_E_LAUNCH;
}
