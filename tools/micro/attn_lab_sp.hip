// Single-pass attention BACKWARD laboratory (development tool, round 4): attn_bwd_sp_bf16 (key-stationary, one pass, dQ by fp32
// atomics) against the two-pass production kernels at the benchmark launch shape (B 8, H 8, L 6272, d_h 32, bf16, pre-scaled q):
// gradients compared, both paths timed in interleaved rounds in one process, the three launches of the single pass also one by one.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-inline-asm -o tools/micro/attn_lab_sp tools/micro/attn_lab_sp.hip
//   run:   tools/micro/attn_lab_sp [rounds] [B] [L]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define SP_LAB 1
#include "../../svol_amd/csrc/attention_bf16.hip"

namespace {
__global__ void split_stats_sp8(unsigned* nl, unsigned* nd, int64_t n) {   // fp32 row constants -> (hi, lo) pairs, in place
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { nl[i] = split_bf16x2(__builtin_bit_cast(float, nl[i])); nd[i] = split_bf16x2(__builtin_bit_cast(float, nd[i])); }
}
// REJECTED VARIANT, kept here for the record (profiles/round4_attention_lab.md): 1.12-1.20 ms against 0.95-1.02 of the 4-wave kernel.
// ---- the same single pass with EIGHT waves (two per SIMD) and 64 keys per wave -------------------------------------------------------
// A lone wave issues one vector instruction per 4 cycles (MI355X_MICROARCH: "one wave alone: 4"), two co-resident waves one per 2:
// with 48 vector instructions (16 of them v_exp) per 10 MFMAs the 4-wave kernel above is bound by exactly that (measured 3400 cycles
// per step against an issue sum of 2200, stamps in tools/micro/attn_lab_sp).  Here a workgroup is 512 threads; a wave owns two
// 32-key blocks (64 accumulator registers for dK^T / dV^T, 48 for its K / V fragments) and everything lives in 256 registers, so the
// per-query constants cannot stay in registers as initial accumulators (2 x 16): they ride the matrix pipe as (hi, lo) pairs —
// one more 16-deep MFMA per product, as in attn_bwd_dkdv_pre_body — 12 MFMAs per block.  Plain builtins: hipcc schedules and pads.
constexpr int S8_PART = 8 * 4 * 64 * 16;    // one partial buffer: [wave][row group][lane] x 16 bytes
constexpr int S8_OFF_Q = 4 * KT * 4;        // LDS map: [2][KT] -lse2 pairs | [2][KT] -delta pairs | Q tiles | dO tiles | dS images | partials
constexpr int S8_OFF_T = S8_OFF_Q + 4 * IMG;
constexpr int S8_OFF_P = S8_OFF_T + 8 * 2 * 2048;
constexpr int S8_LDS = S8_OFF_P + 2 * S8_PART;

template <bool LIVE>
__device__ __forceinline__ void attn_bwd_sp8_body(const Args& p, char* smem, int xt, int hh, int b) {
    unsigned* sL = reinterpret_cast<unsigned*>(smem);   // [2][KT] -lse2 as (hi, lo) pairs
    unsigned* sD = sL + 2 * KT;                         // [2][KT] -delta as (hi, lo) pairs
    char* sQ = smem + S8_OFF_Q;            // [2][IMG] Q tiles (128 queries); dO tiles 2 * IMG behind
    char* sPart = smem + S8_OFF_P;         // [2][S8_PART] dQ partials
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + (int64_t)b * p.Lq * p.ldq + hh * 32;
    const h16_t* dO = reinterpret_cast<const h16_t*>(p.d_o) + (int64_t)b * p.Lq * p.lddo + hh * 32;
    const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + (int64_t)b * p.Lk * p.ldk + hh * 32;
    const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + (int64_t)b * p.Lk * p.ldv + hh * 32;
    const unsigned* nl_g = p.nl2 + ((int64_t)b * p.H + hh) * p.Lq;
    const unsigned* nd_g = p.nd2 + ((int64_t)b * p.H + hh) * p.Lq;
    const int dqw = p.H * 32;
    const int key0 = xt * SP_KEYS + wave * 64;
    char* sT = smem + S8_OFF_T + wave * 4096;   // this wave's dS images: [2][32 keys][32 q]

    uint4 kbk[2][2], vbk[2][2], kd[2][2];
    f32x16 dK[2], dV[2];
    if (LIVE) {   // this wave's 64 K rows, then its V rows, through ITS eighth of the (still unused) Q / dO buffers
        char* sS = sQ + wave * 4096;
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) dma_piece(sS, K, p.ldk, key0, pc, lane);
        dma_wait_all();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            read_rows(kbk[kb], sS, kb * 32 + r, h);
            read_tr_nat(kd[kb], sS, kb, lane);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) dma_piece(sS, V, p.ldv, key0, pc, lane);
        dma_wait_all();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) read_rows(vbk[kb], sS, kb * 32 + r, h);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) { dK[kb] = zero16(); dV[kb] = zero16(); }
    }
    // both partial slots start as zeros: a wave without keys never writes its own, and step 0 "reduces" an empty buffer
#pragma unroll
    for (int buf = 0; buf < 2; ++buf)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(sPart + buf * S8_PART + ((wave * 4 + g) * 64 + lane) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = p.Lq / KT, nsteps = p.Lq / 32;   // launcher: Lq % 128 == 0
    // LDS-DMA: per tile this wave moves ONE 16-row piece of Q, one of dO and 32 of the 256 row-constant pairs (3 instructions)
    const int prow = 16 * wave + (lane >> 2), pch = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    const h16_t* gq = Q + (int64_t)prow * p.ldq + pch;
    const h16_t* gdo = dO + (int64_t)prow * p.lddo + pch;
    const unsigned* gst = (wave < 4 ? nl_g : nd_g) + (wave & 3) * 32 + (lane & 7) * 4;
    const unsigned lds_q = (unsigned)(size_t)(lds_vptr)sQ + wave * 1024, lds_st = (unsigned)(size_t)(lds_vptr)((wave < 4 ? sL : sD) + (wave & 3) * 32);
    auto dma_tile_at = [&](int t, int buf) {
        const h16_t* a = gq + (int64_t)t * KT * p.ldq;
        const h16_t* c = gdo + (int64_t)t * KT * p.lddo;
        const unsigned dq_ = __builtin_amdgcn_readfirstlane(lds_q + buf * IMG);
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dq_), "v"(a) : "m0");
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dq_ + 2 * IMG), "v"(c) : "m0");
        const unsigned ds_ = __builtin_amdgcn_readfirstlane(lds_st + buf * KT * 4);
        const unsigned* e = gst + t * KT;
        if (lane < 8) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(ds_), "v"(e) : "m0");
    };
    // lane-constant byte offsets (a step adds one scalar): see attn_bwd_sp_body
    const unsigned o_rows0 = img_off(r, h), o_rows1 = img_off(r, 2 + h);
    unsigned o_trl, o_trh, o_w[4], o_rl, o_rh;
    {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h2 = g >> 1;
        const int ch = 2 * (g & 1) + (pp >> 1), inner = 8 * (pp & 1);
        o_trl = img_off(4 * h2 + q, ch) + inner;
        o_trh = img_off(4 * h2 + q + 8, ch) + inner;
        o_rl = ds_off(8 * h2 + q, ch, pp & 1);
        o_rh = ds_off(8 * h2 + q + 4, ch, pp & 1);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) o_w[g] = ds_off(r, g, h);
    auto tr_pair = [&](uint4 (&a)[2], const char* img, unsigned ol, unsigned oh) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const h16x4 lo = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(img + ol + 1024 * s));
            const h16x4 hi = SVOL_DS_READ_TR16_H16((lds_bf16x4_ptr)(img + oh + 1024 * s));
            a[s] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    // the previous step's eight partials: this wave sums two rows of row group wave / 2 (rows 8 (wave / 2) + 4 h + 2 (wave & 1) + e) and
    // adds them to the fp32 image: lanes 0..31 / 32..63 = the 128 contiguous bytes of two rows
    float* dq_ptr = p.ws_dq + (int64_t)b * p.Lq * dqw + hh * 32 + (8 * (wave >> 1) + 4 * h + 2 * (wave & 1)) * dqw + r;
    const char* pr_base = sPart + ((wave >> 1) * 64 + lane) * 16 + (wave & 1) * 8;
    auto reduce_step = [&](int buf, int qstep) {
        f32x2 acc = *reinterpret_cast<const f32x2*>(pr_base + buf * S8_PART);
#pragma unroll
        for (int w2 = 1; w2 < 8; ++w2) acc += *reinterpret_cast<const f32x2*>(pr_base + buf * S8_PART + w2 * 4 * 64 * 16);
        float* dst = dq_ptr + (int64_t)qstep * 32 * dqw;
        unsafeAtomicAdd(dst, acc[0]);
        unsafeAtomicAdd(dst + dqw, acc[1]);
    };
    const uint4 ones = h == 0 ? make_uint4(SVOL_H16_ONE2, 0u, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);  // [1, 1, 0...]

    __syncthreads();                       // every wave is done with its staging eighth; the zeros are in place
    dma_tile_at(0, 0);
    dma_wait_all();
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int sub = st & 3, t = st >> 2, cur = t & 1;
        if (sub == 0 && t + 1 < nt) {      // the other tile buffers were last read in step st - 1, behind that step's barrier
            dma_tile_at(t + 1, cur ^ 1);
            asm volatile("" ::: "memory");   // the atomics below stay BEHIND these three transfers (counted vmcnt at the tile's end)
        }
        reduce_step((st - 1) & 1, st > 0 ? st - 1 : 0);
        if (LIVE) {
            const char* qimg = sQ + cur * IMG + sub * 2048;
            uint4 qa[2], doa[2], qt[2], dot[2];
            qa[0] = *reinterpret_cast<const uint4*>(qimg + o_rows0);
            qa[1] = *reinterpret_cast<const uint4*>(qimg + o_rows1);
            doa[0] = *reinterpret_cast<const uint4*>(qimg + 2 * IMG + o_rows0);
            doa[1] = *reinterpret_cast<const uint4*>(qimg + 2 * IMG + o_rows1);
            // every lane loads its query's pair; for the h = 1 lanes (k = 8..15) `ones` is zero and the pair is finite
            const uint4 el = make_uint4(sL[cur * KT + sub * 32 + r], 0u, 0u, 0u), ed = make_uint4(sD[cur * KT + sub * 32 + r], 0u, 0u, 0u);
            tr_pair(qt, qimg, o_trl, o_trh);
            tr_pair(dot, qimg + 2 * IMG, o_trl, o_trh);
            f32x16 dQp;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 S = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, el), __builtin_bit_cast(h16x8, ones), zero16(), 0, 0, 0);
                S = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, qa[0]), __builtin_bit_cast(h16x8, kbk[kb][0]), S, 0, 0, 0);
                S = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, qa[1]), __builtin_bit_cast(h16x8, kbk[kb][1]), S, 0, 0, 0);   // score - lse[q]
                f32x16 dP = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, ed), __builtin_bit_cast(h16x8, ones), zero16(), 0, 0, 0);
                dP = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, doa[0]), __builtin_bit_cast(h16x8, vbk[kb][0]), dP, 0, 0, 0);
                dP = SVOL_MFMA_32x32x16_H16(__builtin_bit_cast(h16x8, doa[1]), __builtin_bit_cast(h16x8, vbk[kb][1]), dP, 0, 0, 0); // dO V^T - delta[q]
#pragma unroll
                for (int i = 0; i < 16; ++i) S[i] = __builtin_amdgcn_exp2f(S[i]);
                mma_second(dV[kb], dot, S);           // dV^T += dO^T P
#pragma unroll
                for (int i = 0; i < 16; i += 2) {     // v_pk_mul_f32 on adjacent accumulator registers
                    const f32x2 m = f32x2{S[i], S[i + 1]} * f32x2{dP[i], dP[i + 1]};
                    S[i] = m[0];
                    S[i + 1] = m[1];
                }
                uint4 pk[2];
                pack16(pk, S);
                mma_packed(dK[kb], qt, pk);           // dK^T += Q^T dS
                char* img = sT + kb * 2048;
                *reinterpret_cast<uint2*>(img + o_w[0]) = make_uint2(pk[0].x, pk[0].y);
                *reinterpret_cast<uint2*>(img + o_w[1]) = make_uint2(pk[0].z, pk[0].w);
                *reinterpret_cast<uint2*>(img + o_w[2]) = make_uint2(pk[1].x, pk[1].y);
                *reinterpret_cast<uint2*>(img + o_w[3]) = make_uint2(pk[1].z, pk[1].w);
                uint4 dsa[2];
                tr_pair(dsa, img, o_rl, o_rh);        // the block's dS, transposed
                if (kb == 0) dQp = zero16();
                mma_packed(dQp, dsa, kd[kb]);         // dQ += dS K
            }
            char* pwr = sPart + (st & 1) * S8_PART + (wave * 4 * 64 + lane) * 16;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(pwr + g * 64 * 16) = f32x4{dQp[4 * g], dQp[4 * g + 1], dQp[4 * g + 2], dQp[4 * g + 3]};
        }
        // tile t + 1 (3 LDS-DMA instructions, issued at the top of this tile) is older than this tile's 8 atomics
        if (sub == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        sp_barrier();
    }
    reduce_step((nsteps - 1) & 1, nsteps - 1);
    if (LIVE) {
        h16_t* dKo = reinterpret_cast<h16_t*>(p.dk) + (int64_t)b * p.Lk * p.lddk + hh * 32;
        h16_t* dVo = reinterpret_cast<h16_t*>(p.dv) + (int64_t)b * p.Lk * p.lddv + hh * 32;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            store_acc(dK[kb], dKo, p.lddk, key0 + kb * 32 + r, true, 32, h, p.scale / p.premul);
            store_acc(dV[kb], dVo, p.lddv, key0 + kb * 32 + r, true, 32, h, 1.f);
        }
    }
}
__global__ __launch_bounds__(512, 2) void attn_bwd_sp8_bf16(Args p) {
    __shared__ __attribute__((aligned(1024))) char smem[S8_LDS];
    int xt, hh, b;
    block_coords(p, xt, hh, b);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // wave-uniform (Lk % 128 == 0): the last workgroup of a head may own fewer than 512 keys; its key-less waves only help with
    // staging, the reduction and the barriers (same number of barriers and vector-memory instructions on both paths)
    if (xt * SP_KEYS + wave * 64 < p.Lk) attn_bwd_sp8_body<true>(p, smem, xt, hh, b);
    else attn_bwd_sp8_body<false>(p, smem, xt, hh, b);
}

}  // namespace

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

static unsigned short f2bf(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short b) {
    unsigned u = (unsigned)b << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int rounds = argc > 1 ? atoi(argv[1]) : 7;
    const int B = argc > 2 ? atoi(argv[2]) : 8, H = 8, L = argc > 3 ? atoi(argv[3]) : 6272, dh = 32, d = H * dh;
    const float scale = 1.f / sqrtf((float)dh), premul = 1.4426950408889634f * scale;
    const size_t n = (size_t)B * L * 3 * d, no = (size_t)B * L * d;
    std::vector<unsigned short> hq(n), hdo(no);
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (size_t i = 0; i < n; ++i) {
        const int col = (int)(i % (3 * d));
        float x = nd(rng) * (col < 2 * d ? 1.5f : 1.0f);
        if (col < d) x *= premul;
        hq[i] = f2bf(x);
    }
    for (size_t i = 0; i < no; ++i) hdo[i] = f2bf(nd(rng));
    unsigned short *dqkv, *dout, *ddo, *dgrad, *dref;
    float *dlse, *ddelta, *dws;
    const int64_t ws_bytes = 4 * svol_attn_ws_floats_bf16(B, H, L, L, dh);
    printf("B %d H %d L %d: workspace %.1f MB\n", B, H, L, ws_bytes / 1e6);
    CK(hipMalloc(&dqkv, n * 2));
    CK(hipMalloc(&dout, no * 2));
    CK(hipMalloc(&ddo, no * 2));
    CK(hipMalloc(&dgrad, n * 2));
    CK(hipMalloc(&dref, n * 2));
    CK(hipMalloc(&dlse, (size_t)B * H * L * 4));
    CK(hipMalloc(&ddelta, (size_t)3 * B * H * L * 4));
    CK(hipMalloc(&dws, (size_t)ws_bytes));
    CK(hipMemcpy(dqkv, hq.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(ddo, hdo.data(), no * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dgrad, 0xFF, n * 2));
    CK(hipMemset(dref, 0xFF, n * 2));

    if (svol_attn_fwd_bf16_launch(dqkv, 3 * d, dqkv + d, 3 * d, dqkv + 2 * d, 3 * d, dout, d, dlse, nullptr, B, H, L, L, dh, scale, premul,
                                  dws, ws_bytes, 0.f, 0, 0) != SVOL_OK) { printf("forward failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    auto bwd = [&](unsigned short* g, float* ws, int64_t wsb) {
        return svol_attn_bwd_bf16_launch(dqkv, 3 * d, dqkv + d, 3 * d, dqkv + 2 * d, 3 * d, dout, d, ddo, d, dlse, ddelta, nullptr, g, 3 * d,
                                         g + d, 3 * d, g + 2 * d, 3 * d, B, H, L, L, dh, scale, premul, ws, wsb, 0.f, 0, 0);
    };
    if (bwd(dref, nullptr, 0) != SVOL_OK) { printf("two-pass backward failed\n"); return 1; }   // no scratch -> the two-pass kernels
    CK(hipDeviceSynchronize());
    if (bwd(dgrad, dws, ws_bytes) != SVOL_OK) { printf("single-pass backward failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<unsigned short> href(n), hg(n);
    CK(hipMemcpy(href.data(), dref, n * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hg.data(), dgrad, n * 2, hipMemcpyDeviceToHost));
    const char* names[3] = {"dq", "dk", "dv"};
    for (int part = 0; part < 3; ++part) {
        double e = 0, m = 0, se = 0, sr = 0;
        size_t bad = 0;
        for (size_t row = 0; row < (size_t)B * L; ++row)
            for (int c = part * d; c < (part + 1) * d; ++c) {
                const size_t j = row * 3 * d + c;
                const double a = bf2f(hg[j]), r = bf2f(href[j]);
                if (!(fabs(a - r) <= 1e30)) ++bad;
                e = std::max(e, fabs(a - r));
                m = std::max(m, fabs(r));
                se += (a - r) * (a - r);
                sr += r * r;
            }
        printf("check %s: max|diff| %.3e  max|ref| %.3f  rel L2 %.3e  non-finite %zu\n", names[part], e, m, sqrt(se / std::max(sr, 1e-300)), bad);
    }

    auto compare = [&](const char* tag) {
        CK(hipMemcpy(hg.data(), dgrad, n * 2, hipMemcpyDeviceToHost));
        for (int part = 0; part < 3; ++part) {
            double e = 0, m = 0, se = 0, sr = 0;
            size_t bad = 0;
            for (size_t row = 0; row < (size_t)B * L; ++row)
                for (int c = part * d; c < (part + 1) * d; ++c) {
                    const size_t j = row * 3 * d + c;
                    const double a = bf2f(hg[j]), r = bf2f(href[j]);
                    if (!(fabs(a - r) <= 1e30)) ++bad;
                    e = std::max(e, fabs(a - r));
                    m = std::max(m, fabs(r));
                    se += (a - r) * (a - r);
                    sr += r * r;
                }
            printf("check %s %s: max|diff| %.3e  max|ref| %.3f  rel L2 %.3e  non-finite %zu\n", tag, names[part], e, m, sqrt(se / std::max(sr, 1e-300)), bad);
        }
        return 0;
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // the single pass's three launches one by one (arguments as the launcher builds them)
    Args ps{};
    ps.q = dqkv; ps.k = dqkv + d; ps.v = dqkv + 2 * d; ps.o = dout; ps.d_o = ddo; ps.lse2 = dlse; ps.delta = ddelta;
    ps.dq = dgrad; ps.dk = dgrad + d; ps.dv = dgrad + 2 * d;
    ps.ldq = ps.ldk = ps.ldv = 3 * d; ps.ldo = d; ps.lddo = d; ps.lddq = ps.lddk = ps.lddv = 3 * d;
    ps.B = B; ps.H = H; ps.Lq = L; ps.Lk = L; ps.dh = dh; ps.scale = scale; ps.premul = premul; ps.ksplit = 1;
    ps.head_xcd = (B * H) % 16 == 0 ? 2 : 1;
    ps.ws_dq = dws;
    ps.nl2 = reinterpret_cast<unsigned*>(ddelta + (size_t)B * H * L);
    ps.nd2 = reinterpret_cast<unsigned*>(ddelta + (size_t)2 * B * H * L);
    ps.nxt = (L + SP_KEYS - 1) / SP_KEYS;
    ps.tail_last = (L % SP_KEYS != 0 && ps.nxt > 1) ? 1 : 0;
    const dim3 gprep((unsigned)((int64_t)B * L / 32)), gsp(sp_grid(B, H, L)), grnd(sp_round_grid(B, H, L, L)), g8((unsigned)(B * H * ps.nxt));
    // the eight-wave variant: same three launches, row constants as (hi, lo) pairs
    Args p8 = ps;
    p8.dq_rot = 1;
    CK(hipMemset(dgrad, 0xFF, n * 2));
    const int64_t nst = (int64_t)B * H * L;
    hipLaunchKernelGGL(attn_bwd_sp_prep_bf16, gprep, dim3(256), 0, 0, p8);
    hipLaunchKernelGGL(split_stats_sp8, dim3((unsigned)((nst + 255) / 256)), dim3(256), 0, 0, p8.nl2, p8.nd2, nst);
    hipLaunchKernelGGL(attn_bwd_sp8_bf16, g8, dim3(512), 0, 0, p8);
    hipLaunchKernelGGL(attn_dq_round_bf16, grnd, dim3(256), 0, 0, p8);
    CK(hipDeviceSynchronize());
    compare("sp8");
    std::vector<float> t2, t1, tp, tm, tr, t8;
    for (int r = 0; r < rounds; ++r) {
        float t;
        CK(hipEventRecord(e0));
        bwd(dref, nullptr, 0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r > 0) t2.push_back(t);
        CK(hipEventRecord(e0));
        bwd(dgrad, dws, ws_bytes);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r > 0) t1.push_back(t);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(attn_bwd_sp_prep_bf16, gprep, dim3(256), 0, 0, ps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r > 0) tp.push_back(t);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(attn_bwd_sp_bf16, gsp, dim3(256), 0, 0, ps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r > 0) tm.push_back(t);
        hipLaunchKernelGGL(attn_bwd_sp_prep_bf16, gprep, dim3(256), 0, 0, p8);
        hipLaunchKernelGGL(split_stats_sp8, dim3((unsigned)((nst + 255) / 256)), dim3(256), 0, 0, p8.nl2, p8.nd2, nst);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(attn_bwd_sp8_bf16, g8, dim3(512), 0, 0, p8);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r > 0) t8.push_back(t);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(attn_dq_round_bf16, grnd, dim3(256), 0, 0, ps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r > 0) tr.push_back(t);
    }
    auto rep = [](const char* name, std::vector<float>& v) {
        std::sort(v.begin(), v.end());
        printf("%-44s median %.4f ms  min %.4f ms\n", name, v[v.size() / 2], v[0]);
    };
    rep("two-pass backward (dQ rot + dK/dV dma)", t2);
    rep("single-pass backward (3 launches)", t1);
    rep("  prep (delta, row constants, zero dQ32)", tp);
    rep("  attn_bwd_sp_bf16", tm);
    rep("  attn_dq_round_bf16", tr);
    rep("  attn_bwd_sp8_bf16 (8 waves, 2 per SIMD)", t8);
    if (getenv("SP_ABLATIONS")) {
        struct AV { const char* name; void (*k)(Args); };
        const AV av[] = {{"lab copy (ABL 0)", attn_bwd_sp_lab<0>}, {"no vector fillers", attn_bwd_sp_lab<1>}, {"no MFMAs", attn_bwd_sp_lab<2>},
                         {"no step barrier", attn_bwd_sp_lab<4>}, {"no dS LDS round trip", attn_bwd_sp_lab<8>}, {"no tile DMA / waits", attn_bwd_sp_lab<16>},
                         {"no fillers, no MFMAs", attn_bwd_sp_lab<3>}, {"only skeleton (31)", attn_bwd_sp_lab<31>}, {"no fillers/MFMA/LDS (11)", attn_bwd_sp_lab<11>}};
        for (const AV& v : av) {
            std::vector<float> tt;
            for (int r = 0; r < 4; ++r) {
                float t;
                hipLaunchKernelGGL(attn_bwd_sp_prep_bf16, gprep, dim3(256), 0, 0, ps);
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(v.k, gsp, dim3(256), 0, 0, ps);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&t, e0, e1));
                if (r > 0) tt.push_back(t);
            }
            std::sort(tt.begin(), tt.end());
            printf("ablation %-32s median %.4f ms\n", v.name, tt[tt.size() / 2]);
        }
    }
    const double blocks = (double)B * H * (L / 32.0) * (L / 32.0);
    std::sort(tm.begin(), tm.end());
    printf("attn_bwd_sp_bf16: %.0f cycles per 32x32 block per SIMD at 2.0 GHz (1024 SIMDs), atomics %.2f GB -> %.2f TB/s\n",
           tm[tm.size() / 2] * 1e-3 * 2.0e9 / (blocks / 1024.0), (double)B * H * ps.nxt * (L / 32.0) * 4096 / 1e9,
           (double)B * H * ps.nxt * (L / 32.0) * 4096 / 1e12 / (tm[tm.size() / 2] * 1e-3));
    return 0;
}
