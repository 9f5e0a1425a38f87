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
#include "common.h"

namespace {

struct FastArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    const float* bias; bf16_t* pre; const void* res; const bf16_t* aux; float* colsum;
    int64_t lda, ldb, ldc, ldp, ldr, ldaux;
    int M, N, K, act, epi;
};

constexpr int STAGE = 16384;         // one 128 x 64 bf16 operand tile
constexpr int EP_STRIDE = 272;       // bytes per staged accumulator row (64 floats + 16 pad)
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <typename TC> struct Out4;
template <> struct Out4<float> {
    static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void store(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Out4<bf16_t> {
    static __device__ __forceinline__ f32x4 load(const bf16_t* p) {
        const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
        return f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
    }
    static __device__ __forceinline__ void store(bf16_t* p, const f32x4& v) {
        bf16x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = (bf16_t)v[e];
        *reinterpret_cast<bf16x4*>(p) = t;
    }
};

template <typename TC>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_kernel(FastArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int bm = blockIdx.y * 128, bn = blockIdx.x * 128;

    // buffer descriptors rooted at this workgroup's tile rows: out-of-range rows read as zero
    const int rowsA = min(128, p.M - bm), rowsB = min(128, p.N - bn);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.A + (int64_t)bm * p.lda), 0, (int)(((int64_t)rowsA * p.lda) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.B + (int64_t)bn * p.ldb), 0, (int)(((int64_t)rowsB * p.ldb) * 2), 0x00020000);
    // lane -> (row within an 8-row group, source chunk): LDS slot (lane&7) of row (lane>>3) holds chunk slot^row
    const int lrow = lane >> 3, lch = (lane & 7) ^ lrow;
    int voffA[4], voffB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + lrow;
        voffA[i] = (int)(((int64_t)row * p.lda + lch * 8) * 2);
        voffB[i] = (int)(((int64_t)row * p.ldb + lch * 8) * 2);
    }
    auto issue = [&](int stage, int k0) {
        char* sa = smem + stage * STAGE;
        char* sb = smem + (2 + stage) * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_ptr)(sa + (wave * 4 + i) * 1024), 16, voffA[i], k0 * 2, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_ptr)(sb + (wave * 4 + i) * 1024), 16, voffB[i], k0 * 2, 0, 0);
        }
    };

    f32x4 acc[4][4];  // [nt][mt]: rows <-> n = nt*16 + fq*4 + r, cols <-> m = mt*16 + fr
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    const int nk = p.K / 64;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // waits for this stage's DMA (vmcnt(0)) and for everyone to be done with the other stage
        if (kt + 1 < nk) issue((kt + 1) & 1, (kt + 1) * 64);
        const char* sa = smem + (kt & 1) * STAGE;
        const char* sb = smem + (2 + (kt & 1)) * STAGE;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            uint4 mf[4], nf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int rowM = wm * 64 + t * 16 + fr;
                mf[t] = *reinterpret_cast<const uint4*>(sa + rowM * 128 + (((4 * g + fq) ^ (rowM & 7)) << 4));
                const int rowN = wn * 64 + t * 16 + fr;
                nf[t] = *reinterpret_cast<const uint4*>(sb + rowN * 128 + (((4 * g + fq) ^ (rowN & 7)) << 4));
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, nf[nt]),
                                                                          __builtin_bit_cast(bf16x8, mf[mt]), acc[nt][mt], 0, 0, 0);
        }
    }

    // ---- epilogue through LDS: two passes of 32 rows per wave --------------------------------------
    __syncthreads();
    char* st = smem + wave * (32 * EP_STRIDE);
    const int cq = lane & 15, rq = lane >> 4;           // coalesced side: 4 columns cq*4.., rows rq, rq+4, ...
    const int ncol = bn + wn * 64 + cq * 4;
    const bool colv = ncol + 4 <= p.N;                  // fast vector path for this lane's 4 columns
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bias4[e] = (ncol + e < p.N) ? p.bias[ncol + e] : 0.f;
    }
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    TC* C = reinterpret_cast<TC*>(p.C);
    const TC* R = reinterpret_cast<const TC*>(p.res);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int ml = 0; ml < 2; ++ml) {
            const int mt = pass * 2 + ml;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *reinterpret_cast<f32x4*>(st + (ml * 16 + fr) * EP_STRIDE + (nt * 16 + fq * 4) * 4) = acc[nt][mt];
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rl = it * 4 + rq;
            const int m = bm + wm * 64 + pass * 32 + rl;
            f32x4 v = *reinterpret_cast<const f32x4*>(st + rl * EP_STRIDE + cq * 16);
            if (m < p.M) {
                if (p.epi == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bias4[e];
                    if (p.pre) {
                        if (colv) Out4<bf16_t>::store(p.pre + (int64_t)m * p.ldp + ncol, v);
                        else
                            for (int e = 0; e < 4; ++e)
                                if (ncol + e < p.N) p.pre[(int64_t)m * p.ldp + ncol + e] = (bf16_t)v[e];
                    }
                    if (p.act == SVOL_ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    } else if (p.act == SVOL_ACT_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
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
                } else {  // epi 1: v = acc * gelu'(aux), column sums of v
                    if (colv) {
                        const f32x4 a = Out4<bf16_t>::load(p.aux + (int64_t)m * p.ldaux + ncol);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= dgelu_f(a[e]);
                    } else {
                        for (int e = 0; e < 4; ++e)
                            v[e] = (ncol + e < p.N) ? v[e] * dgelu_f((float)p.aux[(int64_t)m * p.ldaux + ncol + e]) : 0.f;
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

}  // namespace

// launcher used by gemm.hip's C-ABI entry points.  Returns SVOL_E_UNSUPPORTED when the shape does not
// qualify (the caller then uses the generic kernel).
int svol_gemm_nt_bf16_fast(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias,
                           int act, void* pre, int64_t ldp, const void* res, int64_t ldr, int out_f32, const void* aux,
                           int64_t ldaux, float* colsum, int epi, int64_t M, int64_t N, int64_t K, hipStream_t s) {
    if (K % 64 || lda % 8 || ldb % 8 || !aligned16(A) || !aligned16(B)) return SVOL_E_UNSUPPORTED;
    // vector epilogue needs 8-byte (bf16) / 16-byte (f32) aligned rows; otherwise the scalar tail path is used per lane
    if ((int64_t)128 * lda * 2 >= (1ll << 31) || (int64_t)128 * ldb * 2 >= (1ll << 31)) return SVOL_E_UNSUPPORTED;
    const int celt = out_f32 ? 4 : 2;
    if ((ldc * celt) % (4 * celt) || (reinterpret_cast<uintptr_t>(C) % (4 * celt))) return SVOL_E_UNSUPPORTED;
    if (res && ((ldr * celt) % (4 * celt) || (reinterpret_cast<uintptr_t>(res) % (4 * celt)))) return SVOL_E_UNSUPPORTED;
    if (pre && (ldp % 4 || (reinterpret_cast<uintptr_t>(pre) % 8))) return SVOL_E_UNSUPPORTED;
    if (aux && (ldaux % 4 || (reinterpret_cast<uintptr_t>(aux) % 8))) return SVOL_E_UNSUPPORTED;
    FastArgs p{(const bf16_t*)A, (const bf16_t*)B, C, bias, (bf16_t*)pre, res, (const bf16_t*)aux, colsum,
               lda, ldb, ldc, ldp, ldr, ldaux, (int)M, (int)N, (int)K, act, epi};
    dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128));
    if (grid.y > 65535u) return SVOL_E_UNSUPPORTED;
    if (out_f32) hipLaunchKernelGGL(gemm_nt_bf16_kernel<float>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(gemm_nt_bf16_kernel<bf16_t>, grid, dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? SVOL_OK : SVOL_E_LAUNCH;
}
