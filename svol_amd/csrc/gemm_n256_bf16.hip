// bf16 NT GEMM with N a multiple of 256 and deep K (fc2: K = 2048, the MLP / q|k dX products: K = 2048 / 512; the ViT
// extractor's projections: K = 768 / 3072, N = 768 ... 3072), gfx950.
//
// With a 128x128 tiling these launches stream A twice (two column tiles) and re-stage a W tile per 128 rows:
// 812 MB through the L2 -> LDS path for fc2 at B = 8, which is what bounds them (~9.5 TB/s measured), on a
// two-stage ring that drains at every barrier.  Here a workgroup owns 128 rows x ALL 256 columns:
//
//  * A is streamed exactly once, W once per 128 rows: 597 MB for fc2;
//  * 32-deep K slabs (A 8 KiB + W 16 KiB) through a 3-slot LDS ring filled by LDS-DMA, one slab always in flight
//    across the raw barrier (counted s_waitcnt vmcnt), fragment reads in inline asm with immediate slot offsets
//    (the loop is unrolled over the ring), 12 ds_read_b128 per wave for 32 MFMAs;
//  * operands swapped and W's rows permuted in its LDS image (as in gemm_ws_bf16.hip) so that a lane owns 8
//    consecutive output columns: bias / fp32 residual / 16-byte stores straight from the accumulators.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

struct N256Args {
    const h16_t* A; const h16_t* W; void* C;
    const float* bias; const float* res;
    int64_t lda, ldw, ldc, ldr;
    int M, K, act, nrt;  // nrt: row tiles per column group (padded to a multiple of 8 when there are several groups)
    int kwt;             // A's contraction index wraps every kwt slabs (kwt = K / BK: no wrap).  Split weights: C = A [W_hi | W_lo]^T
                         // reads the K-concatenated [A | A] without materialising it
};

typedef __attribute__((address_space(3))) void* lds_void_ptr;

constexpr int BM = 128, BN = 256, BK = 32, STG = 3;
constexpr int A_ST = BM * BK * 2;   // 8 KiB
constexpr int W_ST = BN * BK * 2;   // 16 KiB
constexpr int SLOT = A_ST + W_ST;   // 24 KiB

// 64-byte rows, four 16-byte chunks: chunk permutation that makes the 16 lanes of a fragment read hit 16 distinct slots
__device__ __forceinline__ int slot32(int row, int ch) { return ch ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3); }
__device__ __forceinline__ int wcol(int nt, int i) { return (nt >> 1) * 32 + (i >> 2) * 8 + (nt & 1) * 4 + (i & 3); }

template <int N> __device__ __forceinline__ void n256_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int OFF>
__device__ __forceinline__ void read_frags(uint4 (&af)[4], uint4 (&wf)[8], const unsigned (&aa)[4], const unsigned (&wa)[8]) {
    asm volatile(
        "ds_read_b128 %0, %12 offset:%c24\n\t"
        "ds_read_b128 %1, %13 offset:%c24\n\t"
        "ds_read_b128 %2, %14 offset:%c24\n\t"
        "ds_read_b128 %3, %15 offset:%c24\n\t"
        "ds_read_b128 %4, %16 offset:%c25\n\t"
        "ds_read_b128 %5, %17 offset:%c25\n\t"
        "ds_read_b128 %6, %18 offset:%c25\n\t"
        "ds_read_b128 %7, %19 offset:%c25\n\t"
        "ds_read_b128 %8, %20 offset:%c25\n\t"
        "ds_read_b128 %9, %21 offset:%c25\n\t"
        "ds_read_b128 %10, %22 offset:%c25\n\t"
        "ds_read_b128 %11, %23 offset:%c25\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(af[0]), "=&v"(af[1]), "=&v"(af[2]), "=&v"(af[3]), "=&v"(wf[0]), "=&v"(wf[1]), "=&v"(wf[2]), "=&v"(wf[3]),
          "=&v"(wf[4]), "=&v"(wf[5]), "=&v"(wf[6]), "=&v"(wf[7])
        : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(wa[0]), "v"(wa[1]), "v"(wa[2]), "v"(wa[3]), "v"(wa[4]), "v"(wa[5]),
          "v"(wa[6]), "v"(wa[7]), "n"(OFF), "n"(OFF + A_ST)
        : "memory");
}

template <bool OUT_F32>
__device__ __forceinline__ void gemm_n256_body(const N256Args& p) {
    __shared__ __attribute__((aligned(1024))) char smem[STG * SLOT];  // 72 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    // 1-D grid, row tile fastest: the column groups of one row tile land on the same XCD (ids differ by a multiple of 8)
    // and share its L2 for the A rows they all read
    const int rt = blockIdx.x % p.nrt, cg = blockIdx.x / p.nrt;
    const int bm = rt * BM;
    if (bm >= p.M) return;  // padding workgroups of the XCD-aligned grid
    const int rows = min(BM, p.M - bm);
    const int bn = cg * BN;

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.A + (int64_t)bm * p.lda), 0, (int)((((int64_t)rows - 1) * p.lda + p.kwt * BK) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rW =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)bn * p.ldw), 0, (int)((((int64_t)BN - 1) * p.ldw + p.K) * 2), 0x00020000);
    // DMA: one instruction = 16 image rows x 4 chunks.  A: instructions 2w, 2w+1; W: 4w .. 4w+3.
    // image row rho of W holds global row  (rho/128)*128 + wcol((rho%128)/16, rho%16)
    int voffA[2], voffW[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 16 + (lane >> 2);
        voffA[j] = (int)(((int64_t)row * p.lda + slot32(row, lane & 3) * 8) * 2);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rho = (wave * 4 + j) * 16 + (lane >> 2);
        const int n = (rho >> 7) * 128 + wcol((rho & 127) >> 4, rho & 15);
        voffW[j] = (int)(((int64_t)n * p.ldw + slot32(rho, lane & 3) * 8) * 2);
    }
    auto issue = [&](int slot, int kt) {
        char* sl = smem + slot * SLOT;
        const int so = kt * BK * 2;
        const int soA = (kt >= p.kwt ? kt - p.kwt : kt) * BK * 2;   // (at most one wrap: K <= 2 * kwt * BK)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_ptr)(sl + (wave * 2 + j) * 1024), 16, voffA[j], soA, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_void_ptr)(sl + A_ST + (wave * 4 + j) * 1024), 16, voffW[j], so, 0, 0);
    };

    // fragment addresses (slot 0): A rows wm*64 + mt*16 + fr, W image rows wn*128 + nt*16 + fr, chunk fq
    const unsigned lds0 = (unsigned)(size_t)(lds_void_ptr)smem;
    unsigned aa[4], wa[8];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int row = wm * 64 + mt * 16 + fr;
        aa[mt] = lds0 + (unsigned)(row * 64 + (slot32(row, fq) << 4));
    }
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const int rho = wn * 128 + nt * 16 + fr;
        wa[nt] = lds0 + (unsigned)(rho * 64 + (slot32(rho, fq) << 4));
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    auto step = [&](auto slot_tag, int kt) {
        constexpr int S = decltype(slot_tag)::value;
        if (kt + 1 < nk) n256_wait_vm<6>(); else n256_wait_vm<0>();  // slab kt has landed; slab kt+1 may still fly
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue((S + 2) % STG, kt + 2);
        uint4 af[4], wf[8];
        read_frags<S * SLOT>(af, wf, aa, wa);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[nt][mt] = SVOL_MFMA_16x16x32_H16(__builtin_bit_cast(h16x8, wf[nt]),
                                                                      __builtin_bit_cast(h16x8, af[mt]), acc[nt][mt], 0, 0, 0);
    };
    int kt = 0;
    for (; kt + 3 <= nk; kt += 3) {
        step(std::integral_constant<int, 0>{}, kt);
        step(std::integral_constant<int, 1>{}, kt + 1);
        step(std::integral_constant<int, 2>{}, kt + 2);
    }
    if (kt < nk) step(std::integral_constant<int, 0>{}, kt++);
    if (kt < nk) step(std::integral_constant<int, 1>{}, kt++);

    // ---- epilogue from the accumulators: lane (fr = row, fq): columns wn*128 + pr*32 + fq*8 + [0,8) ----------------
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = bm + wm * 64 + mt * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const int n0 = bn + wn * 128 + pr * 32 + fq * 8;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[2 * pr + (e >> 2)][mt][e & 3];
            if (p.bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += p.bias[n0 + e];
            }
            if constexpr (OUT_F32) {
                float* C = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n0;
                if (p.res) {
                    const float* R = p.res + (int64_t)m * p.ldr + n0;
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(R), r1 = *reinterpret_cast<const f32x4*>(R + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
                }
                *reinterpret_cast<f32x4*>(C) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(C + 4) = f32x4{v[4], v[5], v[6], v[7]};
            } else {
                if (act_is_gelu(p.act)) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_fast(v[e]);
                }
                if (p.act == SVOL_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                h16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (h16_t)v[e];
                *reinterpret_cast<h16x8*>(reinterpret_cast<h16_t*>(p.C) + (int64_t)m * p.ldc + n0) = o;
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void gemm_n256_bf16_f32(N256Args p) { gemm_n256_body<true>(p); }
__global__ __launch_bounds__(256, 2) void gemm_n256_bf16_b16(N256Args p) { gemm_n256_body<false>(p); }

}  // namespace

// launcher used by gemm_bf16.hip's fast-path dispatcher.  Returns SVOL_E_UNSUPPORTED when the call does not fit.
int svol_gemm_n256_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias, int act,
                        void* pre, const void* res, int64_t ldr, int out_f32, int epi, const float* colscale, int64_t M,
                        int64_t N, int64_t K, int64_t kwrap, hipStream_t s) {
    static const bool off = getenv("SVOL_GEMM_NO_N256") != nullptr;
    if (off || N % BN || K % BK || K < 512 || M < 4096) return SVOL_E_UNSUPPORTED;
    if (epi != 0 || pre || colscale) return SVOL_E_UNSUPPORTED;
    if (kwrap != K && (kwrap % BK || 2 * kwrap != K)) return SVOL_E_UNSUPPORTED;
    if (act != SVOL_ACT_NONE && ((act != SVOL_ACT_GELU && act != SVOL_ACT_RELU) || out_f32)) return SVOL_E_UNSUPPORTED;   // (GELU_D: needs `pre`, not here)
    if (res && !out_f32) return SVOL_E_UNSUPPORTED;
    auto al16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) == 0; };
    if (lda % 8 || ldw % 8 || !al16(A) || !al16(W) || !al16(C)) return SVOL_E_UNSUPPORTED;
    if (out_f32 ? (ldc % 4 || (res && (ldr % 4 || !al16(res)))) : (ldc % 8 != 0)) return SVOL_E_UNSUPPORTED;
    if ((int64_t)BM * lda * 2 >= (1ll << 31) || (int64_t)BN * ldw * 2 >= (1ll << 31)) return SVOL_E_UNSUPPORTED;
    const int64_t ncg = N / BN;
    int64_t nrt = (M + BM - 1) / BM;
    if (ncg > 1) nrt = (nrt + 7) / 8 * 8;
    if (nrt * ncg > (1ll << 30)) return SVOL_E_UNSUPPORTED;
    N256Args p{(const h16_t*)A, (const h16_t*)W, C, bias, (const float*)res, lda, ldw, ldc, ldr, (int)M, (int)K, act, (int)nrt,
               (int)(kwrap / BK)};
    dim3 grid((unsigned)(nrt * ncg));
    if (out_f32) hipLaunchKernelGGL(gemm_n256_bf16_f32, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(gemm_n256_bf16_b16, grid, dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? SVOL_OK : SVOL_E_LAUNCH;
}
