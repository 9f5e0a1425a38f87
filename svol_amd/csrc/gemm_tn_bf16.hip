// bf16 TN GEMM fast path (gfx950): dW[N,K] (fp32) += A[Mc,N]^T * B[Mc,K]  — the weight gradients.
//
// The first TN kernel (gemm.hip, kept for f32 and odd shapes) staged one 64-row slab ahead through
// registers; with 16 MFMAs per wave per 32 rows the slab's compute is ~0.2 us while an HBM load takes
// ~2 us, so every workgroup idled on its single outstanding slab (rocprofv3: 32 us for the 51 MB of a
// d x d projection gradient = 1.6 TB/s).  Here the row slabs go HBM -> LDS by LDS-DMA into a 4-slot ring
// and THREE slabs stay in flight across the (raw) barriers behind a counted s_waitcnt vmcnt:
//
//  * slab = 32 rows x 128 columns of A and of B (8 KiB each), 16 one-KiB DMA instructions per slab, 4 per wave;
//  * the LDS image is row-major as in memory (both operands are contracted along rows, so the MFMA
//    fragments are transposed reads, ds_read_b64_tr_b16); 256-byte rows cannot be padded under a
//    lane-linear DMA, so the 32-byte granules of row r are XOR-permuted by (r & 7) on the SOURCE side,
//    which keeps every half-wave of a transposed read on 8 distinct granules;
//  * ragged edges: the buffer descriptor ends at the last valid element of the chunk (reads past it
//    return 0); columns >= N / >= K of a partial tile read finite-or-not garbage that only ever reaches
//    output elements that are masked at the final atomics;
//  * split-M over workgroups with fp32 atomics as before; the bias gradient (column sums of A) rides
//    the matrix pipe (A^T * ones) in the k-tile-0 workgroups.
#include <cstdlib>

#include "common.h"

namespace {

struct TnFastArgs {
    const h16_t* A; const h16_t* B; float* C; float* colsum;
    int64_t lda, ldb, ldc;
    int Mc, N, K, m_chunk, ktiles, splits;
    int split_major;   // workgroup -> work mapping, see tn_dma_body
};

// Implicit-GEMM weight gradient of a convolution (round 5, the trainable ResNet): B is not a matrix in memory but the im2col view of an NHWC
// activation x [n, cH, cW, cC] — row m = output pixel (n, oy, ox), column k = (ky*ckw + kx)*cC + c reads x[n, oy*s - p + ky, ox*s - p + kx, c]
// (zero outside the image and for k >= Kreal): the B half of a slab is gathered by the LDS-DMA loads themselves, the [M, kh*kw*C] matrix
// (0.9 GB for a layer-1 convolution at 256 frames) never exists.
struct TnConvGeom {
    int cH, cW, cC, cHo, cWo, ckw, cstride, cpad, Kreal, x_bytes;
};

constexpr int CT = 32;            // rows per slab
constexpr int STG = 4;            // ring slots
constexpr int OPB = CT * 256;     // bytes per operand slab
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef __attribute__((address_space(3))) h16x4* lds_bf16x4_ptr;

// physical byte offset of element (row, col) inside a slab image
__device__ __forceinline__ int phys(int row, int col) {
    return row * 256 + ((((col >> 4) ^ (row & 7))) << 5) + (col & 15) * 2;
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool CONV = false>
__device__ __forceinline__ void tn_dma_body(const TnFastArgs& p, const int bid, const TnConvGeom* gp = nullptr) {
    __shared__ __attribute__((aligned(1024))) char smem[STG * 2 * OPB];  // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;
    // Workgroup -> (row chunk, 128-column tile of A, 128-column tile of B).  The tiles that share an A panel (same rows, same A
    // columns, different B columns) read the SAME A bytes: they must meet in one L2 at the same time.  Consecutive workgroup ids
    // go round-robin over the 8 XCDs, so an item (A tile, row chunk) takes id % 8 = its XCD and its ktiles B tiles follow each
    // other 8 ids apart.  (Round 3 walked all row chunks of one output tile first: every A panel came from HBM ktiles times —
    // 947 MiB moved for 613 MiB of operands in the video half's grouped launch.)
    // Round 6, split_major (splits a multiple of 8): ALL output tiles of one row chunk on one XCD — id % 8 = chunk % 8, the chunk's
    // ntiles x ktiles workgroups 8 ids apart — so the B panel is shared as well (it was fetched once per A tile: the video half's
    // MLP pair moved 884 MB per launch for ~480 MB of operands, profiles/round6_hbm_traffic.txt).
    int kt_, nt_, split;
    if (p.split_major) {
        const int ntiles = (p.N + 127) / 128, T = ntiles * p.ktiles;
        const int xcd = bid & 7, j = bid >> 3, grp = j / T, tile = j - grp * T;
        split = grp * 8 + xcd;
        if (split >= p.splits) return;
        nt_ = tile / p.ktiles;
        kt_ = tile - nt_ * p.ktiles;
    } else {
        const int per8 = 8 * p.ktiles, g8 = bid / per8, r8 = bid - g8 * per8;
        const int item = g8 * 8 + (r8 & 7);
        kt_ = r8 >> 3;
        const int nitems = p.splits * ((p.N + 127) / 128);
        if (item >= nitems) return;               // (the last group of 8 is padded)
        split = item % p.splits;
        nt_ = item / p.splits;
    }
    const int n0 = nt_ * 128, k0 = kt_ * 128;
    const int m_begin = split * p.m_chunk;
    const int m_end = min(p.Mc, m_begin + p.m_chunk);
    const int rows = m_end - m_begin;
    if (rows <= 0) return;
    const int nst = (rows + CT - 1) / CT;

    const int colsA = min(128, p.N - n0), colsB = min(128, p.K - k0);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.A + (int64_t)m_begin * p.lda + n0), 0, (int)((((int64_t)rows - 1) * p.lda + colsA) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = CONV
        ? __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, gp->x_bytes, 0x00020000)   // the whole activation tensor
        : __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (int64_t)m_begin * p.ldb + k0), 0, (int)((((int64_t)rows - 1) * p.ldb + colsB) * 2),
                                            0x00020000);
    // DMA geometry: instruction i of an operand covers slab rows 4i..4i+3 (1 KiB); wave w issues i = 2w, 2w+1.
    // lane -> (row 4i + lane/16, physical 16-byte slot lane%16); the slot holds source chunk ((slot/2) ^ (row&7))*2 + slot%2
    int voffA[2], voffB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave * 2 + j) * 4 + (lane >> 4);
        const int s = lane & 15;
        const int c = ((((s >> 1) ^ (r & 7))) << 1) | (s & 1);
        voffA[j] = (int)(((int64_t)r * p.lda + c * 8) * 2);
        voffB[j] = (int)(((int64_t)r * p.ldb + c * 8) * 2);
    }
    // CONV: this lane's chunk of 8 columns is one tap (ky, kx) and 8 channels from ch0 (cC % 8 == 0: a chunk never straddles taps) —
    // constants of the workgroup's column tile; its two rows' output pixels (n, oy, ox) advance by CT rows per slab (issue() is called
    // with stage = 0, 1, 2, ... in order)
    int cv_n[2] = {0, 0}, cv_oy[2] = {0, 0}, cv_ox[2] = {0, 0}, cv_m[2] = {0, 0};
    if constexpr (CONV) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (wave * 2 + j) * 4 + (lane >> 4);
            const int m = m_begin + r;
            cv_m[j] = m;
            const int hw = gp->cHo * gp->cWo;
            cv_n[j] = m / hw;
            const int rem = m - cv_n[j] * hw;
            cv_oy[j] = rem / gp->cWo;
            cv_ox[j] = rem - cv_oy[j] * gp->cWo;
        }
    }
    int cv_kyj[2] = {0, 0}, cv_kxj[2] = {0, 0}, cv_chj[2] = {0, 0};
    bool cv_colj[2] = {false, false};
    if constexpr (CONV) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {   // (the source chunk depends on the row through the XOR permutation: per instruction)
            const int r = (wave * 2 + j) * 4 + (lane >> 4);
            const int s = lane & 15;
            const int c = ((((s >> 1) ^ (r & 7))) << 1) | (s & 1);
            const int k = k0 + c * 8;
            cv_colj[j] = k < gp->Kreal;
            const int tap = k / gp->cC;
            cv_chj[j] = k - tap * gp->cC;
            cv_kyj[j] = tap / gp->ckw;
            cv_kxj[j] = tap - cv_kyj[j] * gp->ckw;
        }
    }
    auto issue = [&](int slot, int stage) {
        char* sa = smem + slot * (2 * OPB);
        char* sb = sa + OPB;
        const int soA = (int)((int64_t)stage * CT * p.lda * 2), soB = (int)((int64_t)stage * CT * p.ldb * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_ptr)(sa + (wave * 2 + j) * 1024), 16, voffA[j], soA, 0, 0);
            if constexpr (CONV) {
                const int iy = cv_oy[j] * gp->cstride - gp->cpad + cv_kyj[j], ix = cv_ox[j] * gp->cstride - gp->cpad + cv_kxj[j];
                const bool ok = cv_colj[j] && cv_m[j] < m_end && (unsigned)iy < (unsigned)gp->cH && (unsigned)ix < (unsigned)gp->cW;
                // padding, rows past the chunk, padded columns: an offset past the buffer reads as zero (hardware bounds check)
                const int off = ok ? (((cv_n[j] * gp->cH + iy) * gp->cW + ix) * gp->cC + cv_chj[j]) * 2 : 0x7ffffff0;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_ptr)(sb + (wave * 2 + j) * 1024), 16, off, 0, 0, 0);
                // the next slab's pixel of this row: CT rows further
                cv_m[j] += CT;
                cv_ox[j] += CT;
                while (cv_ox[j] >= gp->cWo) {
                    cv_ox[j] -= gp->cWo;
                    if (++cv_oy[j] == gp->cHo) { cv_oy[j] = 0; ++cv_n[j]; }
                }
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_ptr)(sb + (wave * 2 + j) * 1024), 16, voffB[j], soB, 0, 0);
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    const bool do_cs = p.colsum != nullptr && kt_ == 0 && wk == 0;
    f32x4 cs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    h16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (h16_t)1.0f;

    // fragment addresses: transposed read of rows 4*fq + q (+16), 4 columns at 4*pp of 16-column tile t
    const int q = fr >> 2, pp = fr & 3;
    const int row0 = 4 * fq + q;
    int offA[4], offB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        offA[t] = phys(row0, wn * 64 + t * 16 + 4 * pp);
        offB[t] = OPB + phys(row0, wk * 64 + t * 16 + 4 * pp);
    }

#pragma unroll
    for (int s = 0; s < STG - 1; ++s)
        if (s < nst) issue(s, s);
    for (int t = 0; t < nst; ++t) {
        // slab t has landed once at most the younger slabs' DMAs (4 instructions each) are outstanding
        const int younger = min(STG - 2, nst - 1 - t);
        if (younger >= 2) wait_vm<8>();
        else if (younger == 1) wait_vm<4>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();  // everyone's part of slab t is visible; everyone is done reading slab t-1
        if (t + STG - 1 < nst) issue((t + STG - 1) % STG, t + STG - 1);
        // the fragment reads are inline asm on purpose: hipcc cannot tell which ring slot an LDS-DMA wrote, so any
        // LDS read it can see gets an s_waitcnt vmcnt(0) in front of it, which would drain the slabs in flight
        const unsigned sl = (unsigned)(size_t)(lds_void_ptr)smem + (unsigned)((t % STG) * (2 * OPB));
        h16x4 a0[4], a1[4], b0[4], b1[4];
        asm volatile(
            "ds_read_b64_tr_b16 %0, %16\n\t"
            "ds_read_b64_tr_b16 %1, %16 offset:4096\n\t"
            "ds_read_b64_tr_b16 %2, %17\n\t"
            "ds_read_b64_tr_b16 %3, %17 offset:4096\n\t"
            "ds_read_b64_tr_b16 %4, %18\n\t"
            "ds_read_b64_tr_b16 %5, %18 offset:4096\n\t"
            "ds_read_b64_tr_b16 %6, %19\n\t"
            "ds_read_b64_tr_b16 %7, %19 offset:4096\n\t"
            "ds_read_b64_tr_b16 %8, %20\n\t"
            "ds_read_b64_tr_b16 %9, %20 offset:4096\n\t"
            "ds_read_b64_tr_b16 %10, %21\n\t"
            "ds_read_b64_tr_b16 %11, %21 offset:4096\n\t"
            "ds_read_b64_tr_b16 %12, %22\n\t"
            "ds_read_b64_tr_b16 %13, %22 offset:4096\n\t"
            "ds_read_b64_tr_b16 %14, %23\n\t"
            "ds_read_b64_tr_b16 %15, %23 offset:4096\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(a0[0]), "=&v"(a1[0]), "=&v"(a0[1]), "=&v"(a1[1]), "=&v"(a0[2]), "=&v"(a1[2]), "=&v"(a0[3]), "=&v"(a1[3]),
              "=&v"(b0[0]), "=&v"(b1[0]), "=&v"(b0[1]), "=&v"(b1[1]), "=&v"(b0[2]), "=&v"(b1[2]), "=&v"(b0[3]), "=&v"(b1[3])
            : "v"(sl + offA[0]), "v"(sl + offA[1]), "v"(sl + offA[2]), "v"(sl + offA[3]), "v"(sl + offB[0]),
              "v"(sl + offB[1]), "v"(sl + offB[2]), "v"(sl + offB[3])
            : "memory");
        h16x8 af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i] = __builtin_shufflevector(a0[i], a1[i], 0, 1, 2, 3, 4, 5, 6, 7);
            bfr[i] = __builtin_shufflevector(b0[i], b1[i], 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
                acc[nt][kt] = SVOL_MFMA_16x16x32_H16(af[nt], bfr[kt], acc[nt][kt], 0, 0, 0);
        if (do_cs) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) cs[nt] = SVOL_MFMA_16x16x32_H16(af[nt], ones, cs[nt], 0, 0, 0);
        }
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
__global__ __launch_bounds__(256, 2) void gemm_tn_bf16_dma(TnFastArgs p) { tn_dma_body(p, (int)blockIdx.x); }
struct TnConvArgs { TnFastArgs t; TnConvGeom g; };
__global__ __launch_bounds__(256, 2) void gemm_tn_bf16_conv(TnConvArgs p) { tn_dma_body<true>(p.t, (int)blockIdx.x, &p.g); }

// Several weight gradients of one backward block in ONE launch (92 weight-gradient launches per step): the
// problems ride in the kernel arguments, a workgroup finds its problem by a scan over at most SVOL_TN_GROUP_MAX prefix sums.
struct TnGroupArgs {
    int n;
    int begin[SVOL_TN_GROUP_MAX + 1];
    TnFastArgs a[SVOL_TN_GROUP_MAX];
};
__global__ __launch_bounds__(256, 2) void gemm_tn_bf16_dma_grouped(TnGroupArgs g) {
    int i = 0;
#pragma unroll
    for (int j = 1; j < SVOL_TN_GROUP_MAX; ++j)
        if (j < g.n && (int)blockIdx.x >= g.begin[j]) i = j;
    tn_dma_body(g.a[i], (int)blockIdx.x - g.begin[i]);
}

// split plan of one problem (shared by the single and the grouped launcher); false = the shape does not qualify
bool tn_plan(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* colsum, int64_t Mc, int64_t N,
             int64_t K, TnFastArgs& out, int64_t& wgs, int wgs_override = 0) {
    if (N % 8 || K % 8 || lda % 8 || ldb % 8 || !aligned16(A) || !aligned16(B)) return false;
    if (Mc < 1 || Mc > (1 << 30) || N > (1 << 30) || K > (1 << 30)) return false;
    const int64_t tiles = ((N + 127) / 128) * ((K + 127) / 128);
    // split the contraction: every split ends in fp32 atomics over its whole output tile, and the kernel shares the chip with the
    // backward's critical path (it runs on the weight-gradient stream): ~128 workgroups per problem.  Round 4, same box, ms per step
    // of bench.py against this target: 32 -> 20.1, 64 -> 19.0, 96 -> 18.9, **128 -> 18.6-18.8**, 160 -> 18.9, 256 / 512 by tile count
    // (rounds 2-3: the best for the kernel ALONE, 146 us against ~200 for the MLP pair) -> 19.1, 512 -> 19.5, 1024 -> 20.9.  A
    // multiple of the 8 XCDs so that the tiles of one row chunk share an L2.  SVOL_TN_WGS / SVOL_TN_WGS2 (problems with more
    // than 8 output tiles) override.
    static const int force_wgs = getenv("SVOL_TN_WGS") ? atoi(getenv("SVOL_TN_WGS")) : 0;
    static const int force_wgs2 = getenv("SVOL_TN_WGS2") ? atoi(getenv("SVOL_TN_WGS2")) : force_wgs;
    const int target_wgs = wgs_override ? wgs_override : (tiles <= 8 ? (force_wgs ? force_wgs : 128) : (force_wgs2 ? force_wgs2 : 128));
    int64_t want = (target_wgs + tiles - 1) / tiles;
    if (want > 8) want = want / 8 * 8;
    // split-major mapping (tn_dma_body): needs a multiple of 8 row chunks; taken when that costs at most 2 x the target
    static const int split_major_on = getenv("SVOL_TN_SPLIT_MAJOR") ? atoi(getenv("SVOL_TN_SPLIT_MAJOR")) : 1;
    bool split_major = false;
    if (split_major_on && !wgs_override && 8 * tiles <= 2 * target_wgs) {
        if (want < 8) want = 8;
        split_major = true;
    }
    if (svol_deterministic()) { want = 1; split_major = false; }   // no contraction split: one adder per output element
    int64_t chunk = (Mc + want - 1) / want;
    chunk = ((chunk + CT - 1) / CT) * CT;
    if (chunk < 4 * CT) chunk = 4 * CT;
    const int64_t splits = (Mc + chunk - 1) / chunk;
    if (splits * tiles > (1ll << 28)) return false;
    if (splits < 8) split_major = false;       // (short problems: the row chunks have a minimum length)
    const int64_t ldmax = lda > ldb ? lda : ldb;
    if (chunk * ldmax * 2 >= (1ll << 31)) return false;  // 32-bit buffer offsets
    out = TnFastArgs{(const h16_t*)A, (const h16_t*)B, C, colsum, lda, ldb, ldc, (int)Mc, (int)N, (int)K, (int)chunk,
                     (int)((K + 127) / 128), (int)splits, split_major ? 1 : 0};
    if (split_major) wgs = ((splits + 7) / 8) * 8 * tiles;
    else wgs = ((splits * ((N + 127) / 128) + 7) / 8) * 8 * ((K + 127) / 128);   // items padded to whole groups of 8 (tn_dma_body)
    return true;
}

}  // namespace

// grouped launcher used by gemm.hip's svol_gemm_tn_grouped: all problems or none (SVOL_E_UNSUPPORTED)
int svol_gemm_tn_bf16_grouped(const svol_tn_problem* pr, int n, hipStream_t s) {
    if (n < 1 || n > SVOL_TN_GROUP_MAX) return SVOL_E_UNSUPPORTED;
    TnGroupArgs g{};
    g.n = n;
    int64_t tot = 0;
    for (int i = 0; i < n; ++i) {
        int64_t wgs = 0;
        if (!tn_plan(pr[i].A, pr[i].lda, pr[i].B, pr[i].ldb, pr[i].C, pr[i].ldc, pr[i].colsum, pr[i].Mc, pr[i].N, pr[i].K, g.a[i], wgs))
            return SVOL_E_UNSUPPORTED;
        g.begin[i] = (int)tot;
        tot += wgs;
    }
    for (int i = n; i <= SVOL_TN_GROUP_MAX; ++i) g.begin[i] = (int)tot;
    if (tot > (1ll << 30)) return SVOL_E_UNSUPPORTED;
    hipLaunchKernelGGL(gemm_tn_bf16_dma_grouped, dim3((unsigned)tot), dim3(256), 0, s, g);
    return hipGetLastError() == hipSuccess ? SVOL_OK : SVOL_E_LAUNCH;
}

// dWp[Cout, Kp] (fp32, caller zeroes) += dz[M, Cout]^T * im2col(x)[M, Kp] with the im2col view gathered in the kernel (TnConvGeom)
int svol_conv_wgrad_bf16_fast(const void* dz, const void* x, float* dwp, int64_t N, int64_t H, int64_t W, int64_t C, int64_t Cout, int64_t kh,
                              int64_t kw, int64_t stride, int64_t pad, int64_t Kp, hipStream_t stream) {
    const int64_t Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return SVOL_E_INVALID;
    const int64_t M = N * Ho * Wo, K = kh * kw * C;
    if (C % 8 || Cout % 8 || Kp % 8 || Kp < K || M > (1ll << 30) || N * H * W * C * 2 >= (1ll << 31) - 64) return SVOL_E_UNSUPPORTED;
    TnConvArgs a{};
    int64_t wgs = 0;
    // the convolutions' weight gradients are most of the backbone's backward, not a side dish beside an attention kernel: fill the chip
    // (SVOL_CONV_WGRAD_WGS: lab override; measured in profiles/round5_resnet_train_kernel_stats.txt)
    static const int conv_wgs = getenv("SVOL_CONV_WGRAD_WGS") ? atoi(getenv("SVOL_CONV_WGRAD_WGS")) : 256;   // 128 -> 34.2, 256 -> 29.6, 512 -> 29.9, 1024 -> 30.6 ms per step
    if (!tn_plan(dz, Cout, x, 8, dwp, Kp, nullptr, M, Cout, Kp, a.t, wgs, conv_wgs)) return SVOL_E_UNSUPPORTED;   // (ldb is not used by the gather)
    a.g = TnConvGeom{(int)H, (int)W, (int)C, (int)Ho, (int)Wo, (int)kw, (int)stride, (int)pad, (int)K, (int)(N * H * W * C * 2)};
    hipLaunchKernelGGL(gemm_tn_bf16_conv, dim3((unsigned)wgs), dim3(256), 0, stream, a);
    return hipGetLastError() == hipSuccess ? SVOL_OK : SVOL_E_LAUNCH;
}

// launcher used by gemm.hip's svol_gemm_tn.  Returns SVOL_E_UNSUPPORTED when the shape does not qualify.
int svol_gemm_tn_bf16_fast(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* colsum,
                           int64_t Mc, int64_t N, int64_t K, hipStream_t s) {
    TnFastArgs p{};
    int64_t wgs = 0;
    if (!tn_plan(A, lda, B, ldb, C, ldc, colsum, Mc, N, K, p, wgs)) return SVOL_E_UNSUPPORTED;
    hipLaunchKernelGGL(gemm_tn_bf16_dma, dim3((unsigned)wgs), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? SVOL_OK : SVOL_E_LAUNCH;
}
