// GEMM kernels of libsvol_hip (gfx950 / CDNA4, wave64, MFMA).
//
//   svol_gemm_nt : C = act(A * B^T + bias) + residual      (nn.Linear forward, dX with pre-transposed W)
//   svol_gemm_tn : dW += A^T * B                           (weight gradient, contraction over rows)
//   svol_colsum  : bias gradient
//   svol_cast / svol_cast_transpose / svol_act_bwd : dtype + activation plumbing
//
// Tiling: 128x128 output tile per 256-thread workgroup (2x2 waves, 64x64 per wave = 4x4 MFMA 16x16
// tiles).  Operand tiles live in LDS as 128-byte rows of eight 16-byte chunks, XOR-swizzled
// (chunk ^ (row & 7)) so the ds_read_b128 fragment reads of a 16x16 MFMA are bank-conflict free
// (MI355X guide: ds_read_b128 banks = (addr/4) % 64, 16-lane groups).  One template serves both
// element types: bf16 feeds v_mfma_f32_16x16x32_bf16 (one MFMA per chunk-group of 4 chunks), f32
// feeds v_mfma_f32_16x16x4_f32 (four MFMAs per chunk group, element e of every lane's chunk) —
// the contraction index permutation is identical for A and B so the sum is unchanged.
#include <cstdlib>

#include "common.h"

namespace {

template <typename T> struct Tr;
template <> struct Tr<bf16_t> { static constexpr int EPC = 8; };  // elements per 16-byte chunk
template <> struct Tr<f16_t> { static constexpr int EPC = 8; };
template <> struct Tr<float> { static constexpr int EPC = 4; };

__device__ __forceinline__ void mma_chunk(f32x4& acc, const uint4& a, const uint4& b, bf16_t) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc,
                                                  0, 0, 0);
}
__device__ __forceinline__ void mma_chunk(f32x4& acc, const uint4& a, const uint4& b, f16_t) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_chunk(f32x4& acc, const uint4& a, const uint4& b, float) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bf[e], acc, 0, 0, 0);
}

struct NtArgs {
    const void* A; const void* A2; const void* B; void* C;
    const float* bias; const float* colscale; void* pre; const void* res;
    int64_t lda, ldb, ldc, ldr, ldp, n_split;
    int M, N, K, act;
    int kwrap;  // A's contraction index is k % kwrap (kwrap == K: plain; svol_gemm_nt_split: K = 2 * kwrap)
};

// TC = element type of C and of the residual (T, or float for the fp32 residual stream)
// TILE = 128 (four waves of 64x64) or 64 (four waves of 32x32): the small tile for launches whose 128-tiling would leave most
// CUs idle (the fp32 class / box heads: 4800 x 256 outputs = 76 workgroups -> 300)
template <typename T, typename TC, int TILE = 128>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(NtArgs p) {
    constexpr int WT = TILE / 2;        // rows / columns per wave
    constexpr int FT = WT / 16;         // 16x16 fragments per wave and direction
    constexpr int LI = TILE * 8 / 256;  // 16-byte chunks per thread and operand per K step
    constexpr int EPC = Tr<T>::EPC;
    constexpr int BK = 8 * EPC;
    __shared__ __attribute__((aligned(16))) char smem[2 * TILE * 128];
    char* sA = smem;
    char* sB = smem + TILE * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bm = blockIdx.y * TILE, bn = blockIdx.x * TILE;
    const T* A = reinterpret_cast<const T*>((p.A2 != nullptr && bn >= p.n_split) ? p.A2 : p.A);
    const T* B = reinterpret_cast<const T*>(p.B);

    uint4 ra[LI], rb[LI];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < LI; ++i) {
            const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
            const int kk = k0 + ch * EPC;
            const int gm = bm + row, gn = bn + row;
            ra[i] = (gm < p.M && kk < p.K) ? *reinterpret_cast<const uint4*>(A + (int64_t)gm * p.lda + (kk >= p.kwrap ? kk - p.kwrap : kk))
                                           : make_uint4(0, 0, 0, 0);
            rb[i] = (gn < p.N && kk < p.K) ? *reinterpret_cast<const uint4*>(B + (int64_t)gn * p.ldb + kk)
                                           : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < LI; ++i) {
            const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
            const int off = row * 128 + ((ch ^ (row & 7)) << 4);
            *reinterpret_cast<uint4*>(sA + off) = ra[i];
            *reinterpret_cast<uint4*>(sB + off) = rb[i];
        }
    };

    f32x4 acc[FT][FT];
#pragma unroll
    for (int i = 0; i < FT; ++i)
#pragma unroll
        for (int j = 0; j < FT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    load_tile(0);
    for (int k0 = 0; k0 < p.K; k0 += BK) {
        store_tile();
        __syncthreads();
        if (k0 + BK < p.K) load_tile(k0 + BK);  // next tile's HBM latency hides under the MFMAs
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            uint4 af[FT], bf[FT];
#pragma unroll
            for (int t = 0; t < FT; ++t) {
                const int rowA = wm * WT + t * 16 + fr;
                af[t] = *reinterpret_cast<const uint4*>(sA + rowA * 128 + (((4 * g + fq) ^ (rowA & 7)) << 4));
                const int rowB = wn * WT + t * 16 + fr;
                bf[t] = *reinterpret_cast<const uint4*>(sB + rowB * 128 + (((4 * g + fq) ^ (rowB & 7)) << 4));
            }
#pragma unroll
            for (int mt = 0; mt < FT; ++mt)
#pragma unroll
                for (int nt = 0; nt < FT; ++nt) mma_chunk(acc[mt][nt], af[mt], bf[nt], T());
        }
        __syncthreads();
    }

    // epilogue: acc[mt][nt][r] -> C[m0 + mt*16 + fq*4 + r][n0 + nt*16 + fr]
    TC* C = reinterpret_cast<TC*>(p.C);
    T* pre = reinterpret_cast<T*>(p.pre);
    const TC* res = reinterpret_cast<const TC*>(p.res);
#pragma unroll
    for (int mt = 0; mt < FT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < FT; ++nt) {
            const int n = bn + wn * WT + nt * 16 + fr;
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
            const float cs = p.colscale ? p.colscale[n] : 1.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = bm + wm * WT + mt * 16 + fq * 4 + r;
                if (m >= p.M) continue;
                float v = (acc[mt][nt][r] + bv) * cs;
                if (pre) pre[(int64_t)m * p.ldp + n] = from_f32<T>(p.act == SVOL_ACT_GELU_D ? dgelu_f(v) : v);
                if (p.act == SVOL_ACT_RELU) v = fmaxf(v, 0.f);
                else if (act_is_gelu(p.act)) v = gelu_f(v);
                else if (p.act == SVOL_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                if (res) v += to_f32(res[(int64_t)m * p.ldr + n]);
                if (p.act == SVOL_ACT_RELU_RES) v = fmaxf(v, 0.f);
                C[(int64_t)m * p.ldc + n] = from_f32<TC>(v);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// TN: C[n,k] += sum_m A[m,n] * B[m,k].  Contraction tiles [CT rows of m][128 cols] sit in LDS
// row-major (as they are in HBM, so the loads stay coalesced); the MFMA operands need the
// contraction index along the fragment's k, i.e. a TRANSPOSED read:
//   bf16: ds_read_b64_tr_b16 (gfx950 hardware transpose read, 4 rows x 16 cols per 16-lane group);
//         rows padded to 288 B so the 8 rows a 32-lane half touches land on distinct banks.  The
//         fragment's k index 8g+j is mapped to tile row 4g + (j&3) + 16(j>>2) for BOTH operands
//         (any common permutation of the contraction index leaves the sum unchanged), which makes
//         each half-wave read 8 consecutive rows.
//   f32 : v_mfma_f32_16x16x4_f32 takes one element per lane, k = lane>>4: a plain ds_read_b32 of
//         row (k), column (lane&15); rows padded to 576 B so the two rows of a half-wave differ
//         by 16 banks.
// ---------------------------------------------------------------------------
struct TnArgs {
    const void* A; const void* B; float* C; float* colsum;
    int64_t lda, ldb, ldc;
    int Mc, N, K, m_chunk, ktiles, splits;
};

template <typename T> struct TnCfg;
template <> struct TnCfg<bf16_t> { static constexpr int CT = 64, STRIDE = 288, CPR = 16; };  // chunks per row
template <> struct TnCfg<f16_t> { static constexpr int CT = 64, STRIDE = 288, CPR = 16; };
template <> struct TnCfg<float> { static constexpr int CT = 32, STRIDE = 576, CPR = 32; };

template <typename T>
__device__ __forceinline__ void gemm_tn_body(const TnArgs& p, const int bid) {
    constexpr int EPC = Tr<T>::EPC;
    constexpr int CT = TnCfg<T>::CT, S = TnCfg<T>::STRIDE, CPR = TnCfg<T>::CPR;
    __shared__ __attribute__((aligned(16))) char smem[2 * CT * S];
    char* sA = smem;
    char* sB = smem + CT * S;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    // 1-D grid, split index fastest: consecutive workgroup ids go to consecutive XCDs, so with splits % 8 == 0 all
    // output tiles of one row chunk run on the same XCD and share its L2 for the re-read A / B rows
    const int split = bid % p.splits, tile = bid / p.splits;
    const int kt_ = tile % p.ktiles, nt_ = tile / p.ktiles;
    const int n0 = nt_ * 128, k0 = kt_ * 128;
    const int m_begin = split * p.m_chunk;
    const int m_end = min(p.Mc, m_begin + p.m_chunk);
    const T* A = reinterpret_cast<const T*>(p.A);
    const T* B = reinterpret_cast<const T*>(p.B);

    uint4 ra[4], rb[4];
    auto load_tile = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i, row = c / CPR, ch = c % CPR;
            const int gm = m0 + row;
            const int an = n0 + ch * EPC, bk = k0 + ch * EPC;
            ra[i] = (gm < m_end && an < p.N) ? *reinterpret_cast<const uint4*>(A + (int64_t)gm * p.lda + an)
                                             : make_uint4(0, 0, 0, 0);
            rb[i] = (gm < m_end && bk < p.K) ? *reinterpret_cast<const uint4*>(B + (int64_t)gm * p.ldb + bk)
                                             : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i, row = c / CPR, ch = c % CPR;
            *reinterpret_cast<uint4*>(sA + row * S + ch * 16) = ra[i];
            *reinterpret_cast<uint4*>(sB + row * S + ch * 16) = rb[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    // optional column sums of A (= bias gradient) on the matrix pipe: A^T * ones, only in the k-tile-0 workgroups'
    // wk == 0 waves (every output column of the extra tile holds the same sum)
    const bool do_cs = p.colsum != nullptr && kt_ == 0 && wk == 0;
    f32x4 cs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    load_tile(m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += CT) {
        store_tile();
        __syncthreads();
        if (m0 + CT < m_end) load_tile(m0 + CT);
        if constexpr (sizeof(T) == 2) {
            typedef H16<T> HT;
            typedef typename HT::lds_v4_ptr lds_v4_ptr;
            typedef typename HT::v8 v8;
            typedef typename HT::v4 v4;
            const int q = fr >> 2, pp = fr & 3;
#pragma unroll
            for (int ks = 0; ks < CT / 32; ++ks) {
                const int row = ks * 32 + 4 * fq + q;
                v8 af[4], bfr[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int colA = wn * 64 + t * 16 + 4 * pp, colB = wk * 64 + t * 16 + 4 * pp;
                    v4 a0 = HT::read_tr((lds_v4_ptr)(sA + row * S + colA * 2));
                    v4 a1 = HT::read_tr((lds_v4_ptr)(sA + (row + 16) * S + colA * 2));
                    v4 b0 = HT::read_tr((lds_v4_ptr)(sB + row * S + colB * 2));
                    v4 b1 = HT::read_tr((lds_v4_ptr)(sB + (row + 16) * S + colB * 2));
                    af[t] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                    bfr[t] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) acc[nt][kt] = HT::mfma16(af[nt], bfr[kt], acc[nt][kt]);
                if (do_cs) {
                    v8 ones;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ones[j] = (T)1.0f;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) cs[nt] = HT::mfma16(af[nt], ones, cs[nt]);
                }
            }
        } else {
#pragma unroll 2
            for (int ks = 0; ks < CT / 4; ++ks) {
                const int row = ks * 4 + fq;
                float af[4], bfr[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    af[t] = *reinterpret_cast<const float*>(sA + row * S + (wn * 64 + t * 16 + fr) * 4);
                    bfr[t] = *reinterpret_cast<const float*>(sB + row * S + (wk * 64 + t * 16 + fr) * 4);
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
                        acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[nt], bfr[kt], acc[nt][kt], 0, 0, 0);
                if (do_cs) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) cs[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[nt], 1.0f, cs[nt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // C[n][k] += acc: row (output n) = nt*16 + fq*4 + r, col (output k) = kt*16 + fr
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int k = k0 + wk * 64 + kt * 16 + fr;
            if (k >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + nt * 16 + fq * 4 + r;
                if (n < p.N) atomicAdd(p.C + (int64_t)n * p.ldc + k, acc[nt][kt][r]);
            }
        }
    if (do_cs && fr == 0) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + nt * 16 + fq * 4 + r;
                if (n < p.N) atomicAdd(p.colsum + n, cs[nt][r]);
            }
    }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnArgs p) { gemm_tn_body<T>(p, (int)blockIdx.x); }
// several problems in one launch (svol_gemm_tn_grouped): see gemm_tn_bf16.hip
struct TnGroupArgsG {
    int n;
    int begin[SVOL_TN_GROUP_MAX + 1];
    TnArgs a[SVOL_TN_GROUP_MAX];
};
template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_tn_grouped_kernel(TnGroupArgsG g) {
    int i = 0;
#pragma unroll
    for (int j = 1; j < SVOL_TN_GROUP_MAX; ++j)
        if (j < g.n && (int)blockIdx.x >= g.begin[j]) i = j;
    gemm_tn_body<T>(g.a[i], (int)blockIdx.x - g.begin[i]);
}

// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* X, int64_t ldx, float* out, int M, int N, int rows_per_block, float* det_part) {
    // thread t owns column blockIdx.x*256 + t; rows [blockIdx.y*rpb, ...)
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
    float s = 0.f;
    for (int m = m0; m < m1; ++m) s += to_f32(X[(int64_t)m * ldx + n]);
    if (det_part) det_part[(int64_t)blockIdx.y * N + n] = s;   // deterministic mode: row block's partial, folded in index order (common.h)
    else atomicAdd(out + n, s);
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* src, TD* dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = from_f32<TD>(to_f32(src[i]));
}

// fp32 -> bf16, 8 elements per thread (two 16-byte loads, one 16-byte store); n % 8 == 0 and 16-byte aligned pointers
__global__ __launch_bounds__(256) void cast_f32_bf16_vec8_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + i * 8), b = *reinterpret_cast<const f32x4*>(src + i * 8 + 4);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)a[e]; o[4 + e] = (bf16_t)b[e]; }
    *reinterpret_cast<bf16x8*>(dst + i * 8) = o;
}

template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* src, T* dst, T* dstT, int R, int C) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, c = c0 + tx;
        float v = (r < R && c < C) ? src[(int64_t)r * C + c] : 0.f;
        tile[ty + 8 * j][tx] = v;
        if (dst && r < R && c < C) dst[(int64_t)r * C + c] = from_f32<T>(v);
    }
    __syncthreads();
    if (dstT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + ty + 8 * j, r = r0 + tx;
            if (r < R && c < C) dstT[(int64_t)c * R + r] = from_f32<T>(tile[tx][ty + 8 * j]);
        }
    }
}

// all of a model's weights in ONE launch: block -> (descriptor, 32x32 tile) through a prefix table
struct CastDesc {
    const float* src; void* dst; void* dstT; void* dstS;
    int R, C, tiles_c, tile_begin;
};
// split copy [R, 2C] = [hi | lo]: hi = bf16(w), lo = bf16(w - hi) — 16 mantissa bits of the fp32 master weight in two bf16 operands
__device__ __forceinline__ void store_split(void* dstS, int r, int c, int C, float v) {
    bf16_t* d = reinterpret_cast<bf16_t*>(dstS) + (int64_t)r * 2 * C + c;
    const bf16_t hi = (bf16_t)v;
    d[0] = hi;
    d[C] = (bf16_t)(v - (float)hi);
}
template <typename T>
__global__ __launch_bounds__(256) void cast_split_kernel(const float* src, int64_t lds, void* dstS, int R, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)R * C) return;
    const int r = (int)(i / C), c = (int)(i % C);
    store_split(dstS, r, c, C, src[(int64_t)r * lds + c]);
}
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_multi_kernel(const CastDesc* __restrict__ descs, int n_desc) {
    __shared__ float tile[32][33];
    __shared__ int which;
    if (threadIdx.x == 0) {  // binary search: last descriptor with tile_begin <= blockIdx.x
        int lo = 0, hi = n_desc - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (descs[mid].tile_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        which = lo;
    }
    __syncthreads();
    const CastDesc d = descs[which];
    const int t = blockIdx.x - d.tile_begin;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = (t / d.tiles_c) * 32, c0 = (t % d.tiles_c) * 32;
    T* dst = reinterpret_cast<T*>(d.dst);
    T* dstT = reinterpret_cast<T*>(d.dstT);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, c = c0 + tx;
        float v = (r < d.R && c < d.C) ? d.src[(int64_t)r * d.C + c] : 0.f;
        tile[ty + 8 * j][tx] = v;
        if (dst && r < d.R && c < d.C) dst[(int64_t)r * d.C + c] = from_f32<T>(v);
        if (d.dstS && r < d.R && c < d.C) store_split(d.dstS, r, c, d.C, v);
    }
    __syncthreads();
    if (dstT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + ty + 8 * j, r = r0 + tx;
            if (r < d.R && c < d.C) dstT[(int64_t)c * d.R + r] = from_f32<T>(tile[tx][ty + 8 * j]);
        }
    }
}

template <typename T>
__global__ void act_bwd_kernel(const T* dy, const T* aux, T* dpre, int act, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float g = to_f32(dy[i]), a = to_f32(aux[i]);
        float d;
        if (act == SVOL_ACT_RELU) d = a > 0.f ? g : 0.f;
        else if (act == SVOL_ACT_GELU) d = g * dgelu_f(a);
        else if (act == SVOL_ACT_GELU_D) d = g * a;
        else if (act == SVOL_ACT_SIGMOID) d = g * a * (1.f - a);
        else d = g;
        dpre[i] = from_f32<T>(d);
    }
}

// ---------------------------------------------------------------------------------------------------
// Skinny-M fp32 NT GEMM (the object-query stream of the bf16 model runs in exact fp32: M = B * num_queries = 800 rows).
// The generic 64x64 kernel gives such launches 52 workgroups that each walk the K loop alone behind a barrier pair per
// 32-deep step (30 us for 800 x 256 x 256).  Same idea as gemm_nt_bf16_skinny (gemm_bf16.hip): one workgroup owns a
// 32 x 32 output tile, its four waves split K between them and stream their K/4 slices of A and W straight from global
// memory (L2-resident at these sizes) into v_mfma_f32_32x32x2_f32 operands — 16 bytes per lane and row, no LDS staging,
// the next 32-deep batch in flight under the current one — and the four partial tiles meet in LDS for the epilogue.
// The contraction index is permuted (lane half h of chunk c, element e <-> k = 8c + 4h + e) identically for both operands.
struct SkArgs {
    const float* A; const float* B; float* C;
    const float* bias; const float* colscale; float* pre; const float* res; const float* aux; float* colsum;
    int64_t lda, ldb, ldc, ldp, ldr, ldaux;
    int M, N, K, act, epi;
};
constexpr int SKF_T = 32, SKF_LD = SKF_T + 4;

__global__ __launch_bounds__(256) void gemm_nt_f32_skinny(SkArgs p) {
    __shared__ __attribute__((aligned(16))) float red[4][SKF_T][SKF_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int bm = blockIdx.y * SKF_T, bn = blockIdx.x * SKF_T;
    const int kslice = p.K >> 2;
    const int nb = kslice >> 5;  // batches of 32
    const bool va = bm + r < p.M, vb = bn + r < p.N;
    const float* pa = p.A + (int64_t)(bm + r) * p.lda + wave * kslice + h * 4;
    const float* pb = p.B + (int64_t)(bn + r) * p.ldb + wave * kslice + h * 4;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a[4], b[4], na[4], nb_[4];
    auto load = [&](f32x4 (&xa)[4], f32x4 (&xb)[4], int bt) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            xa[c] = va ? *reinterpret_cast<const f32x4*>(pa + bt * 32 + c * 8) : z4;
            xb[c] = vb ? *reinterpret_cast<const f32x4*>(pb + bt * 32 + c * 8) : z4;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    load(a, b, 0);
    for (int bt = 0; bt < nb; ++bt) {
        if (bt + 1 < nb) load(na, nb_, bt + 1);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][e], b[c][e], acc, 0, 0, 0);
        if (bt + 1 < nb) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { a[c] = na[c]; b[c] = nb_[c]; }
        }
    }
    // D[m][n]: register 4g+e of lane (n = r, h) is row m = 8g + 4h + e
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][8 * g + 4 * h + e][r] = acc[4 * g + e];
    __syncthreads();
    const int row = tid >> 3, c0 = (tid & 7) * 4;
    const int m = bm + row, n0 = bn + c0;
    f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][row][c0]);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(&red[w][row][c0]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += u[e];
    }
    const bool mv = m < p.M;  // N % 32 == 0 (launcher): every column of the tile exists
    if (p.epi == 0) {
        if (mv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (p.bias) v[e] += p.bias[n0 + e];
                if (p.colscale) v[e] *= p.colscale[n0 + e];
            }
            if (p.pre) {
                f32x4 sv = v;
                if (p.act == SVOL_ACT_GELU_D) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) sv[e] = dgelu_f(v[e]);
                }
                *reinterpret_cast<f32x4*>(p.pre + (int64_t)m * p.ldp + n0) = sv;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (p.act == SVOL_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
                else if (act_is_gelu(p.act)) v[e] = gelu_f(v[e]);
                else if (p.act == SVOL_ACT_SIGMOID) v[e] = 1.f / (1.f + __expf(-v[e]));
            }
            if (p.res) {
                const f32x4 rr = *reinterpret_cast<const f32x4*>(p.res + (int64_t)m * p.ldr + n0);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += rr[e];
            }
            if (p.act == SVOL_ACT_RELU_RES) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
        }
    } else {  // epi 1: v = acc * act'(aux), column sums of v (fused MLP backward step)
        if (mv) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(p.aux + (int64_t)m * p.ldaux + n0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= p.act == SVOL_ACT_RELU ? (x[e] > 0.f ? 1.f : 0.f) : (p.act == SVOL_ACT_GELU_D ? x[e] : dgelu_f(x[e]));
        } else {
            v = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (p.colsum) {
            __syncthreads();  // everyone has read the partial tiles
            *reinterpret_cast<f32x4*>(&red[0][row][c0]) = v;
            __syncthreads();
            if (tid < SKF_T) {
                float sacc = 0.f;
#pragma unroll 8
                for (int i = 0; i < SKF_T; ++i) sacc += red[0][i][tid];
                atomicAdd(p.colsum + bn + tid, sacc);
            }
        }
    }
    if (mv) *reinterpret_cast<f32x4*>(p.C + (int64_t)m * p.ldc + n0) = v;
}

// the shapes the skinny fp32 kernel takes: few rows, K a multiple of 128 (four waves x 32-deep batches), N a multiple of 32,
// 16-byte aligned rows everywhere
inline bool skinny_f32_ok(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, const void* A, const void* B,
                          const void* C) {
    static const bool off = getenv("SVOL_GEMM_NO_F32_SKINNY") != nullptr;
    static const int64_t max_m = getenv("SVOL_F32_SKINNY_M") ? atoll(getenv("SVOL_F32_SKINNY_M")) : 2048;   // (8192 would put the fp32 heads of the bf16 mode, M = 4800, on this kernel: -0.07 ms per step — but also the fp32 mode's video GEMMs at B = 1, whose summation order the golden assignments are pinned on)
    return !off && M <= max_m && K % 128 == 0 && N % 32 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && aligned16(A) &&
           aligned16(B) && aligned16(C) && (M + 31) / 32 <= 65535;
}

inline int grid_1d(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

// bf16 fast path (gemm_bf16.hip)
int svol_gemm_tn_bf16_fast(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* colsum,
                           int64_t Mc, int64_t N, int64_t K, hipStream_t s);
int svol_gemm_nt_bf16_fast(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias,
                           int act, void* pre, int64_t ldp, const void* res, int64_t ldr, int out_f32, const void* aux,
                           int64_t ldaux, float* colsum, int epi, const float* colscale, int64_t M, int64_t N, int64_t K,
                           int64_t kwrap, hipStream_t s);

int svol_gemm_tn_f16_fast(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* colsum,
                          int64_t Mc, int64_t N, int64_t K, hipStream_t s);
int svol_gemm_tn_bf16_grouped(const svol_tn_problem* pr, int n, hipStream_t s);
int svol_gemm_tn_f16_grouped(const svol_tn_problem* pr, int n, hipStream_t s);
int svol_gemm_nt_f16_fast(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias,
                          int act, void* pre, int64_t ldp, const void* res, int64_t ldr, int out_f32, const void* aux,
                          int64_t ldaux, float* colsum, int epi, const float* colscale, int64_t M, int64_t N, int64_t K,
                          int64_t kwrap, hipStream_t s);

// split plan of the generic TN kernel (shared by svol_gemm_tn and svol_gemm_tn_grouped)
static int tn_generic_plan(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* colsum, int64_t Mc,
                           int64_t N, int64_t K, int dtype, TnArgs& out, int64_t& wgs) {
    const int epc = svol_is16(dtype) ? 8 : 4;
    if (N % epc || K % epc || lda % epc || ldb % epc) return SVOL_E_UNSUPPORTED;
    if (!aligned16(A) || !aligned16(B)) return SVOL_E_INVALID;
    if (Mc > (1 << 30) || N > (1 << 30) || K > (1 << 30)) return SVOL_E_UNSUPPORTED;
    const int ct = svol_is16(dtype) ? 64 : 32;
    const int64_t tiles = ((N + 127) / 128) * ((K + 127) / 128);
    // split the contraction so that ~TARGET workgroups exist, chunk a multiple of CT
    // (measured on MI355X: the fp32 atomics of the final accumulation dominate small outputs, so few, long-running
    //  workgroups win there: 256x256 outputs 0.065 ms at 1024 workgroups vs 0.032 ms at 256)
    static const int force_wgs = getenv("SVOL_TN_WGS") ? atoi(getenv("SVOL_TN_WGS")) : 0;
    // fp32 with few rows (the query stream's weight gradients, Mc = 800): every split adds a full tile of fp32 atomics — 13 splits of the
    // [256, 2048] MLP gradients were 6.8 M atomics per problem, which is what their 87 us were (atomic rate ~128 per clock, tools/micro/
    // atomic_rate) — so few splits: ~64 workgroups per problem
    // (round 4, in the step: 64 -> 18.75-18.92 ms, 128 -> 18.94-18.97, 32 -> 18.82, 256 -> 18.90: these launches run beside the video half)
    static const int small_wgs = getenv("SVOL_TN_SMALL_WGS") ? atoi(getenv("SVOL_TN_SMALL_WGS")) : 64;
    const int target_wgs = force_wgs ? force_wgs : ((dtype == SVOL_F32 && Mc <= 4096) ? small_wgs : (tiles <= 8 ? 256 : 512));
    int64_t want = (target_wgs + tiles - 1) / tiles;
    if (svol_deterministic()) want = 1;   // no contraction split: one adder per output element
    int64_t chunk = (Mc + want - 1) / want;
    chunk = ((chunk + ct - 1) / ct) * ct;
    // (fp32 with few rows — the query stream's weight gradients, Mc = 800: the 4 x CT floor would leave 28 workgroups, each
    //  walking 128 rows of fp32 MFMAs; one CT-row slab per workgroup gives 100 and the atomic volume stays small)
    const int64_t floor_rows = (dtype == SVOL_F32 && Mc <= 4096) ? ct : 4 * ct;
    if (chunk < floor_rows) chunk = floor_rows;
    int64_t splits = (Mc + chunk - 1) / chunk;
    if (splits > 8 && splits % 8) {  // keep the split count a multiple of the 8 XCDs when the rows allow it
        const int64_t s8 = (splits + 7) / 8 * 8;
        int64_t c8 = (Mc + s8 - 1) / s8;
        c8 = ((c8 + ct - 1) / ct) * ct;
        if (c8 >= floor_rows && (Mc + c8 - 1) / c8 == s8) { chunk = c8; splits = s8; }
    }
    const int64_t ktiles = (K + 127) / 128;
    if (splits * tiles > (1ll << 30)) return SVOL_E_UNSUPPORTED;
    out = TnArgs{A, B, C, colsum, lda, ldb, ldc, (int)Mc, (int)N, (int)K, (int)chunk, (int)ktiles, (int)splits};
    wgs = splits * tiles;
    return SVOL_OK;
}

extern "C" {

int svol_abi_version(void) { return 7; }

const char* svol_strerror(int code) {
    switch (code) {
        case SVOL_OK: return "ok";
        case SVOL_E_INVALID: return "invalid argument (null pointer, bad size or misaligned buffer)";
        case SVOL_E_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
        case SVOL_E_LAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}

static int gemm_nt_impl(const void* A, int64_t lda, const void* A2, int64_t n_split, const void* B, int64_t ldb, void* C,
                        int64_t ldc, const float* bias, const float* colscale, int act, void* pre_act_out, int64_t ldp,
                        const void* residual, int64_t ldr, int out_f32, int64_t M, int64_t N, int64_t K, int64_t kwrap, int dtype,
                        void* stream) {
    if (!A || !B || !C || M < 0 || N < 0 || K <= 0) return SVOL_E_INVALID;
    if (act == SVOL_ACT_GELU_D && !pre_act_out) return SVOL_E_INVALID;   // the derivative is what this variant exists to save
    if (M == 0 || N == 0) return SVOL_OK;
    const int epc = svol_is16(dtype) ? 8 : 4;
    if (!svol_is16(dtype) && dtype != SVOL_F32) return SVOL_E_INVALID;
    if (K % epc || lda % epc || ldb % epc) return SVOL_E_UNSUPPORTED;
    if (!aligned16(A) || !aligned16(B) || (A2 && !aligned16(A2))) return SVOL_E_INVALID;
    if (A2 && (n_split % 128)) return SVOL_E_UNSUPPORTED;
    if (M > (1 << 30) || N > (1 << 30) || K > (1 << 30)) return SVOL_E_UNSUPPORTED;
    NtArgs p{A, A2, B, C, bias, colscale, pre_act_out, residual, lda, ldb, ldc, ldr, ldp, n_split, (int)M, (int)N, (int)K, act, (int)kwrap};
    dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128));
    if (grid.y > 65535u) return SVOL_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (svol_is16(dtype) && !A2) {
        const int rc = (dtype == SVOL_BF16 ? svol_gemm_nt_bf16_fast : svol_gemm_nt_f16_fast)(
            A, lda, B, ldb, C, ldc, bias, act, pre_act_out, ldp, residual, ldr, out_f32, nullptr, 0, nullptr, 0, colscale, M, N, K, kwrap, s);
        if (rc != SVOL_E_UNSUPPORTED) return rc;
    }
    if (dtype == SVOL_F32 && !A2 && kwrap == K && skinny_f32_ok(M, N, K, lda, ldb, ldc, A, B, C) && (!residual || (ldr % 4 == 0 && aligned16(residual))) &&
        (!pre_act_out || (ldp % 4 == 0 && aligned16(pre_act_out)))) {
        SkArgs q{(const float*)A, (const float*)B, (float*)C, bias, colscale, (float*)pre_act_out, (const float*)residual, nullptr,
                 nullptr, lda, ldb, ldc, ldp, ldr, 0, (int)M, (int)N, (int)K, act, 0};
        hipLaunchKernelGGL(gemm_nt_f32_skinny, dim3((unsigned)(N / 32), (unsigned)((M + 31) / 32)), dim3(256), 0, s, q);
        SVOL_CHECK_LAUNCH();
        return SVOL_OK;
    }
    // 64x64 tiles when the 128x128 tiling would leave most of the 256 CUs idle (the fp32 heads: M = 4800, N = 256 -> 76 workgroups)
    const bool small = (int64_t)grid.x * grid.y < 256 && (M + 63) / 64 <= 65535;
    const dim3 g64((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
    if (dtype == SVOL_BF16) {
        if (out_f32) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, float>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, p);
    } else if (dtype == SVOL_F16) {
        if (out_f32) hipLaunchKernelGGL((gemm_nt_kernel<f16_t, float>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((gemm_nt_kernel<f16_t, f16_t>), grid, dim3(256), 0, s, p);
    } else if (small) {
        hipLaunchKernelGGL((gemm_nt_kernel<float, float, 64>), g64, dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL((gemm_nt_kernel<float, float>), grid, dim3(256), 0, s, p);
    }
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_gemm_nt(const void* A, int64_t lda, const void* A2, int64_t n_split, const void* B, int64_t ldb, void* C,
                 int64_t ldc, const float* bias, const float* colscale, int act, void* pre_act_out, int64_t ldp,
                 const void* residual, int64_t ldr, int out_f32, int64_t M, int64_t N, int64_t K, int dtype, void* stream) {
    return gemm_nt_impl(A, lda, A2, n_split, B, ldb, C, ldc, bias, colscale, act, pre_act_out, ldp, residual, ldr, out_f32, M, N, K, K,
                        dtype, stream);
}

// C = A (W_hi + W_lo)^T + bias as ONE K-concatenated product [A | A] [W_hi | W_lo]^T: the kernels wrap A's contraction index
int svol_gemm_nt_split(const void* A, int64_t lda, const void* W_hilo, int64_t ldw, void* C, int64_t ldc, const float* bias,
                       int64_t M, int64_t N, int64_t K, void* stream) {
    if (K <= 0 || K % 8) return SVOL_E_UNSUPPORTED;
    return gemm_nt_impl(A, lda, nullptr, 0, W_hilo, ldw, C, ldc, bias, nullptr, SVOL_ACT_NONE, nullptr, 0, nullptr, 0, 0, M, N, 2 * K, K,
                        SVOL_BF16, stream);
}

int svol_gemm_nt_dact(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* aux,
                      int64_t ldaux, int act, float* colsum, int64_t M, int64_t N, int64_t K, int dtype, void* stream) {
    if (!A || !B || !C || !aux || M < 0 || N < 0 || K <= 0) return SVOL_E_INVALID;
    if (act != SVOL_ACT_GELU && act != SVOL_ACT_RELU && act != SVOL_ACT_GELU_D) return SVOL_E_INVALID;
    if (M == 0 || N == 0) return SVOL_OK;
    if (colsum && svol_deterministic()) {   // the fused column sums meet through atomics in arrival order: a single-adder pass instead
        const int rc = svol_gemm_nt_dact(A, lda, B, ldb, C, ldc, aux, ldaux, act, nullptr, M, N, K, dtype, stream);
        return rc ? rc : svol_colsum(C, ldc, colsum, M, N, dtype, stream);
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (svol_is16(dtype)) {
        const int rc = (dtype == SVOL_BF16 ? svol_gemm_nt_bf16_fast : svol_gemm_nt_f16_fast)(
            A, lda, B, ldb, C, ldc, nullptr, act, nullptr, 0, nullptr, 0, 0, aux, ldaux, colsum, 1, nullptr, M, N, K, K, s);
        if (rc != SVOL_E_UNSUPPORTED) return rc;
    }
    if (dtype == SVOL_F32 && skinny_f32_ok(M, N, K, lda, ldb, ldc, A, B, C) && ldaux % 4 == 0 && aligned16(aux)) {
        SkArgs q{(const float*)A, (const float*)B, (float*)C, nullptr, nullptr, nullptr, nullptr, (const float*)aux, colsum,
                 lda, ldb, ldc, 0, 0, ldaux, (int)M, (int)N, (int)K, act, 1};
        hipLaunchKernelGGL(gemm_nt_f32_skinny, dim3((unsigned)(N / 32), (unsigned)((M + 31) / 32)), dim3(256), 0, s, q);
        SVOL_CHECK_LAUNCH();
        return SVOL_OK;
    }
    // generic composition (f32 / odd shapes): GEMM, then dpre = dh * act'(aux) in place, then column sums
    if (ldc != N || ldaux != N) return SVOL_E_UNSUPPORTED;
    int rc = svol_gemm_nt(A, lda, nullptr, 0, B, ldb, C, ldc, nullptr, nullptr, SVOL_ACT_NONE, nullptr, 0, nullptr, 0, 0, M, N, K,
                          dtype, stream);
    if (rc) return rc;
    rc = svol_act_bwd(C, aux, C, act, M * N, dtype, stream);
    if (rc) return rc;
    if (colsum) rc = svol_colsum(C, ldc, colsum, M, N, dtype, stream);
    return rc;
}

int svol_gemm_nt_dgelu(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* pre,
                       int64_t ldp, float* colsum, int64_t M, int64_t N, int64_t K, int dtype, void* stream) {
    return svol_gemm_nt_dact(A, lda, B, ldb, C, ldc, pre, ldp, SVOL_ACT_GELU, colsum, M, N, K, dtype, stream);
}

int svol_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* colsum, int64_t Mc,
                 int64_t N, int64_t K, int dtype, void* stream) {
    if (!A || !B || !C || Mc < 0 || N <= 0 || K <= 0) return SVOL_E_INVALID;
    if (Mc == 0) return SVOL_OK;
    if (!svol_is16(dtype) && dtype != SVOL_F32) return SVOL_E_INVALID;
    static const bool no_fast_tn = getenv("SVOL_TN_GENERIC") != nullptr;
    if (svol_is16(dtype) && !no_fast_tn) {
        const int rc = (dtype == SVOL_BF16 ? svol_gemm_tn_bf16_fast : svol_gemm_tn_f16_fast)(A, lda, B, ldb, C, ldc, colsum, Mc, N, K,
                                                                                              reinterpret_cast<hipStream_t>(stream));
        if (rc != SVOL_E_UNSUPPORTED) return rc;
    }
    TnArgs p{};
    int64_t wgs = 0;
    const int prc = tn_generic_plan(A, lda, B, ldb, C, ldc, colsum, Mc, N, K, dtype, p, wgs);
    if (prc) return prc;
    const int64_t splits = 1, tiles = wgs;
    dim3 grid((unsigned)(splits * tiles));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SVOL_BF16) hipLaunchKernelGGL(gemm_tn_kernel<bf16_t>, grid, dim3(256), 0, s, p);
    else if (dtype == SVOL_F16) hipLaunchKernelGGL(gemm_tn_kernel<f16_t>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(gemm_tn_kernel<float>, grid, dim3(256), 0, s, p);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_gemm_tn_grouped(const svol_tn_problem* pr, int32_t n, int dtype, void* stream) {
    if (!pr || n < 0) return SVOL_E_INVALID;
    if (!svol_is16(dtype) && dtype != SVOL_F32) return SVOL_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    static const bool no_group = getenv("SVOL_TN_NO_GROUP") != nullptr;
    for (int32_t i0 = 0; i0 < n; i0 += SVOL_TN_GROUP_MAX) {
        const int m = (int)((n - i0 < SVOL_TN_GROUP_MAX) ? n - i0 : SVOL_TN_GROUP_MAX);
        int rc = SVOL_E_UNSUPPORTED;
        bool ok = !no_group && m > 1;
        for (int i = 0; ok && i < m; ++i) ok = pr[i0 + i].A && pr[i0 + i].B && pr[i0 + i].C && pr[i0 + i].Mc > 0 && pr[i0 + i].N > 0 && pr[i0 + i].K > 0;
        if (ok && svol_is16(dtype)) {
            rc = (dtype == SVOL_BF16 ? svol_gemm_tn_bf16_grouped : svol_gemm_tn_f16_grouped)(pr + i0, m, s);
        } else if (ok) {
            TnGroupArgsG g{};
            g.n = m;
            int64_t tot = 0;
            rc = SVOL_OK;
            for (int i = 0; i < m && rc == SVOL_OK; ++i) {
                int64_t wgs = 0;
                const svol_tn_problem& q = pr[i0 + i];
                rc = tn_generic_plan(q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.colsum, q.Mc, q.N, q.K, dtype, g.a[i], wgs);
                g.begin[i] = (int)tot;
                tot += wgs;
            }
            if (rc == SVOL_OK && tot <= (1ll << 30)) {
                for (int i = m; i <= SVOL_TN_GROUP_MAX; ++i) g.begin[i] = (int)tot;
                hipLaunchKernelGGL(gemm_tn_grouped_kernel<float>, dim3((unsigned)tot), dim3(256), 0, s, g);
                SVOL_CHECK_LAUNCH();
            } else {
                rc = SVOL_E_UNSUPPORTED;
            }
        }
        if (rc == SVOL_E_UNSUPPORTED) {   // one by one (odd shapes, a single problem, SVOL_TN_NO_GROUP)
            for (int i = 0; i < m; ++i) {
                const svol_tn_problem& q = pr[i0 + i];
                const int r1 = svol_gemm_tn(q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.colsum, q.Mc, q.N, q.K, dtype, stream);
                if (r1) return r1;
            }
        } else if (rc) {
            return rc;
        }
    }
    return SVOL_OK;
}

int svol_colsum(const void* X, int64_t ldx, float* out, int64_t M, int64_t N, int dtype, void* stream) {
    if (!X || !out || M < 0 || N <= 0) return SVOL_E_INVALID;
    if (M == 0) return SVOL_OK;
    if (M > (1 << 30) || N > (1 << 30)) return SVOL_E_UNSUPPORTED;
    const int colblocks = (int)((N + 255) / 256);
    int rowblocks = (int)((M + 255) / 256);
    int maxrb = 2048 / colblocks;
    if (maxrb < 1) maxrb = 1;
    if (rowblocks > maxrb) rowblocks = maxrb;
    const int rpb = (int)((M + rowblocks - 1) / rowblocks);
    dim3 grid(colblocks, (unsigned)((M + rpb - 1) / rpb));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool det_mode = svol_deterministic();
    DetScratch det(det_mode ? (size_t)grid.y * (size_t)N : 0, s);
    if (det_mode && !det.p) return SVOL_E_LAUNCH;
    if (dtype == SVOL_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)X, ldx, out, (int)M, (int)N, rpb, det.p);
    else if (dtype == SVOL_F16)
        hipLaunchKernelGGL(colsum_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)X, ldx, out, (int)M, (int)N, rpb, det.p);
    else if (dtype == SVOL_F32)
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)X, ldx, out, (int)M, (int)N, rpb, det.p);
    else return SVOL_E_INVALID;
    if (det_mode) det_fold(det.p, (int)grid.y, N, out, N, s);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_cast(const void* src, int dtype_src, void* dst, int dtype_dst, int64_t n, void* stream) {
    if (!src || !dst || n < 0) return SVOL_E_INVALID;
    if (n == 0) return SVOL_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int g = grid_1d(n, 256);
    if (dtype_src == SVOL_F32 && dtype_dst == SVOL_BF16 && n % 8 == 0 && aligned16(src) && aligned16(dst) && n / 8 / 256 < (1ll << 31))
        hipLaunchKernelGGL(cast_f32_bf16_vec8_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, s, (const float*)src,
                           (bf16_t*)dst, n / 8);
    else {
#define SVOL_CAST(TS, TD) hipLaunchKernelGGL((cast_kernel<TS, TD>), dim3(g), dim3(256), 0, s, (const TS*)src, (TD*)dst, n)
#define SVOL_CAST_FROM(TS)                                      \
    do {                                                        \
        if (dtype_dst == SVOL_F32) SVOL_CAST(TS, float);        \
        else if (dtype_dst == SVOL_BF16) SVOL_CAST(TS, bf16_t); \
        else if (dtype_dst == SVOL_F16) SVOL_CAST(TS, f16_t);   \
        else return SVOL_E_INVALID;                             \
    } while (0)
        if (dtype_src == SVOL_F32) SVOL_CAST_FROM(float);
        else if (dtype_src == SVOL_BF16) SVOL_CAST_FROM(bf16_t);
        else if (dtype_src == SVOL_F16) SVOL_CAST_FROM(f16_t);
        else return SVOL_E_INVALID;
#undef SVOL_CAST_FROM
#undef SVOL_CAST
    }
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_cast_transpose(const float* src, void* dst, void* dstT, int dtype, int64_t R, int64_t C, void* stream) {
    if (!src || R <= 0 || C <= 0 || (!dst && !dstT)) return SVOL_E_INVALID;
    if (R > (1 << 30) || C > (1 << 30)) return SVOL_E_UNSUPPORTED;
    dim3 grid((unsigned)((C + 31) / 32), (unsigned)((R + 31) / 32));
    if (grid.y > 65535u) return SVOL_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SVOL_BF16)
        hipLaunchKernelGGL(cast_transpose_kernel<bf16_t>, grid, dim3(256), 0, s, src, (bf16_t*)dst, (bf16_t*)dstT, (int)R, (int)C);
    else if (dtype == SVOL_F16)
        hipLaunchKernelGGL(cast_transpose_kernel<f16_t>, grid, dim3(256), 0, s, src, (f16_t*)dst, (f16_t*)dstT, (int)R, (int)C);
    else if (dtype == SVOL_F32)
        hipLaunchKernelGGL(cast_transpose_kernel<float>, grid, dim3(256), 0, s, src, (float*)dst, (float*)dstT, (int)R, (int)C);
    else return SVOL_E_INVALID;
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_cast_split(const float* src, int64_t ld_src, void* dst_hilo, int64_t R, int64_t C, void* stream) {
    if (!src || !dst_hilo || R <= 0 || C <= 0 || ld_src < C) return SVOL_E_INVALID;
    if (R * C > (1ll << 38)) return SVOL_E_UNSUPPORTED;
    hipLaunchKernelGGL(cast_split_kernel<bf16_t>, dim3((unsigned)((R * C + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       src, ld_src, dst_hilo, (int)R, (int)C);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_cast_transpose_multi(const void* descs, int32_t n_desc, int64_t total_tiles, int dtype, void* stream) {
    static_assert(sizeof(CastDesc) == 48, "descriptor layout is part of the ABI (include/svol_hip.h)");
    if (!descs || n_desc <= 0 || total_tiles <= 0) return SVOL_E_INVALID;
    if (total_tiles > (1ll << 30)) return SVOL_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const CastDesc* d = reinterpret_cast<const CastDesc*>(descs);
    if (dtype == SVOL_BF16) hipLaunchKernelGGL(cast_transpose_multi_kernel<bf16_t>, dim3((unsigned)total_tiles), dim3(256), 0, s, d, n_desc);
    else if (dtype == SVOL_F16) hipLaunchKernelGGL(cast_transpose_multi_kernel<f16_t>, dim3((unsigned)total_tiles), dim3(256), 0, s, d, n_desc);
    else if (dtype == SVOL_F32) hipLaunchKernelGGL(cast_transpose_multi_kernel<float>, dim3((unsigned)total_tiles), dim3(256), 0, s, d, n_desc);
    else return SVOL_E_INVALID;
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_act_bwd(const void* dy, const void* aux, void* dpre, int act, int64_t n, int dtype, void* stream) {
    if (!dy || !aux || !dpre || n < 0) return SVOL_E_INVALID;
    if (n == 0) return SVOL_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int g = grid_1d(n, 256);
    if (dtype == SVOL_BF16)
        hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)aux, (bf16_t*)dpre, act, n);
    else if (dtype == SVOL_F16)
        hipLaunchKernelGGL(act_bwd_kernel<f16_t>, dim3(g), dim3(256), 0, s, (const f16_t*)dy, (const f16_t*)aux, (f16_t*)dpre, act, n);
    else if (dtype == SVOL_F32)
        hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)dy, (const float*)aux, (float*)dpre, act, n);
    else return SVOL_E_INVALID;
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

}  // extern "C"
