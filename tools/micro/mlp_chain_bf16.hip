// Two chained NT GEMMs through the MLP's wide hidden layer in ONE kernel (gfx950): the [M, F] intermediate is produced in registers,
// written to HBM once (the backward / the weight gradients need it) and consumed from registers — it is never read back.
//
//   T = X Wa^T            X [M, 256] 16-bit, Wa [F, 256]
//   U = f(T)              mode 0 (forward):  U = gelu(T + ba) -> hid,  gelu'(T + ba) -> aux_out      (cross_modal_transformer.py:142,
//                         mode 1 (backward): U = T * aux_in   -> hid                                   the MLP of the video half)
//   Y = U Wb^T            Wb [256, F];  mode 0: Y (fp32) = . + bb + res32 ; mode 1: Y 16-bit
//
// Before (round 3): two launches per direction — the first wrote U (and, forward, gelu'), the second read U back: 196 MiB per
// tensor at M = 50176, F = 2048, per layer and direction, in kernels that are HBM-bound.
//
// Shape of the kernel (flash-attention-like, the hidden units play the keys):
//  * a workgroup = 4 waves x 64 rows; a wave keeps its rows' X fragments (64 x 256: 128 VGPRs) and its Y accumulators (64 x 256 fp32:
//    all 256 AGPRs) for its whole life and walks the F hidden units in chunks of 64;
//  * a chunk's weights — Wa rows [64 x 256] and Wb columns [256 x 64], 64 KiB — come from L2 by LDS-DMA into a 2-slot ring shared by
//    the four waves (one raw barrier per chunk, counted vmcnt); every 16-byte fragment read from LDS feeds TWO MFMAs (two row tiles);
//  * LAB STATUS (round 4, profiles/round4_mlp_chain_lab.md): results identical to the two-launch form (forward bit for bit), but no
//    faster inside the step, so svol_amd/csrc/blocks.hip leaves it off by default (SVOL_MLP_CHAIN_BWD=1 / SVOL_MLP_CHAIN_FWD=1).  The
//    `abl` argument (SVOL_CHAIN_ABL, a bit mask) is a timing aid of that lab: 1 = no hidden stores, 2 = no aux loads, 8 / 16 = other
//    store cache policies, 32 = stores into L2-resident lines, 64 = default-policy stores; its results are NOT valid outputs.
//  * both products are computed TRANSPOSED (hidden x rows, outputs x rows) with the weight rows as the MFMA A operand in a permuted
//    row order (perm32), so that a lane's 16 accumulator values are 2 x 8 CONSECUTIVE hidden units (outputs) of ONE row: the packed
//    result is at once the next MFMA's B operand and a 16-byte global store, and aux / residual / bias arrive as 16-byte loads.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

struct ChainArgs {
    const h16_t* X; const h16_t* Wa; const h16_t* Wb; h16_t* hid; const h16_t* aux_in; h16_t* aux_out; void* Y;
    const float* ba; const float* bb; const float* res;
    int64_t ldx, ldh, lda, ldy, ldr;
    int M, F, abl;
};

constexpr int HC = 32;                 // hidden units per chunk
constexpr int WA_B = HC * 512;         // Wa rows of a chunk: 32 x 256 x 2 B = 16 KiB
constexpr int WB_B = 256 * HC * 2;     // Wb columns of a chunk: 256 x 32 x 2 B = 16 KiB
constexpr int WSLOT = WA_B + WB_B;     // 32 KiB
constexpr int NSLOT = 3;               // weight ring AND aux ring depth: a chunk's operands are fetched two chunks ahead
constexpr int ASLOT = 64 * HC * 2;     // a wave's aux tile of a chunk: 64 rows x 32 hidden x 2 B = 4 KiB
constexpr int AUX0 = NSLOT * WSLOT;    // aux rings (mode 1) / bias vector (mode 0) behind the weight ring
constexpr int CH_LDS = AUX0 + 4 * NSLOT * ASLOT;   // 96 + 48 = 144 KiB: one workgroup per CU

// MFMA row m of a 32-row A tile holds weight row perm32(m): accumulator registers 0..7 of lane group g are then rows 8g .. 8g+7
// and registers 8..15 rows 16+8g .. 16+8g+7 of the tile (register r = 4q + e sits at MFMA row 8q + 4g + e)
__device__ __forceinline__ int perm32(int m) { return 16 * ((m >> 4) & 1) + 8 * ((m >> 2) & 1) + 4 * ((m >> 3) & 1) + (m & 3); }

// every MFMA is inline asm: at one wave per SIMD hipcc otherwise picks the AGPR-destination form for everything and moves the
// accumulators between the register halves (profiles/round4_attention_lab.md); hipcc pads nothing behind an asm statement, the
// waits between a result and its first non-MFMA reader are placed by hand (chain_settle)
__device__ __forceinline__ void mfma_v0(f32x16& d, const u32x4_t& a, const u32x4_t& b) {
    asm volatile("v_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x16& d, const u32x4_t& a, const u32x4_t& b) {
    asm volatile("v_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_a(f32x16& d, const u32x4_t& a, const u32x4_t& b) {
    asm volatile("v_mfma_f32_32x32x16_" SVOL_H16_ASM " %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}
// 8-pass MFMA result -> VALU reader: passes + 4 wait states (20 under a 16-pass model); ties the two tiles to the statement
__device__ __forceinline__ void chain_settle(f32x16& a, f32x16& b) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b)); }

template <int OFF>
__device__ __forceinline__ void lds_read4(u32x4_t (&f)[4], unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
    asm volatile(
        "ds_read_b128 %0, %4 offset:%c8\n\t"
        "ds_read_b128 %1, %5 offset:%c8\n\t"
        "ds_read_b128 %2, %6 offset:%c8\n\t"
        "ds_read_b128 %3, %7 offset:%c8"
        : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3])
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(OFF)
        : "memory");
}
template <int OFF0, int OFF1>
__device__ __forceinline__ void lds_read2x2(u32x4_t (&f)[4], unsigned a0, unsigned a1) {
    asm volatile(
        "ds_read_b128 %0, %4 offset:%c6\n\t"
        "ds_read_b128 %1, %5 offset:%c6\n\t"
        "ds_read_b128 %2, %4 offset:%c7\n\t"
        "ds_read_b128 %3, %5 offset:%c7"
        : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3])
        : "v"(a0), "v"(a1), "n"(OFF0), "n"(OFF1)
        : "memory");
}
// wait until at most N LDS reads are outstanding; names the fragments (and whatever else must be complete before the MFMAs that
// follow: a VALU-written operand needs 2 wait states in front of the MFMA that reads it)
template <int N> __device__ __forceinline__ void lds_wait(u32x4_t (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)\n\ts_nop 1" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N));
}
template <int N> __device__ __forceinline__ void lds_wait_ops(u32x4_t (&f)[4], u32x4_t (&u)[2][2]) {
    asm volatile("s_waitcnt lgkmcnt(%8)\n\ts_nop 1"
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(u[0][0]), "+v"(u[0][1]), "+v"(u[1][0]), "+v"(u[1][1])
                 : "n"(N));
}
template <int N> __device__ __forceinline__ void chain_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ u32x4_t pack8(const float (&v)[8]) {
    u32x4_t o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = __builtin_bit_cast(unsigned, cvt_pk_h16(v[2 * i], v[2 * i + 1]));
    return o;
}
__device__ __forceinline__ void unpack8(const u32x4_t& p, float (&v)[8]) {
    const h16x8 h = __builtin_bit_cast(h16x8, p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
}

template <int MODE>
__device__ __forceinline__ void mlp_chain_body(const ChainArgs& p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, g = lane >> 5;
    const int row_base = blockIdx.x * 256;
    const int rows_here = min(256, p.M - row_base);
    const int nch = p.F / HC;
    const int rowl = wave * 64 + n;   // this lane's row of row tile 0 (row tile 1: + 32)

    // ---- buffer descriptors: rows past M read zeros and drop their stores ----------------------------------------------
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.X + (int64_t)row_base * p.ldx), 0, (int)((((int64_t)rows_here - 1) * p.ldx + 256) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.hid + (int64_t)row_base * p.ldh), 0, (int)((((int64_t)rows_here - 1) * p.ldh + p.F) * 2), 0x00020000);
    const h16_t* auxp = MODE == 0 ? p.aux_out : p.aux_in;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(auxp + (int64_t)row_base * p.lda), 0, (int)((((int64_t)rows_here - 1) * p.lda + p.F) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rWa = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wa, 0, (int)((int64_t)p.F * 512), 0x00020000);
    const __amdgpu_buffer_rsrc_t rWb = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wb, 0, (int)((int64_t)256 * p.F * 2), 0x00020000);

    // ---- operand rings: a chunk = 32 one-KiB weight DMA instructions (8 per wave) + (mode 1) 4 aux instructions per wave -----------
    // Wa image : row j (512 B) keeps granule c at (c & 16) | ((c ^ j) & 15);                   instruction i = rows 2i, 2i+1
    // Wb image : row o (64 B)  keeps granule c at c ^ ((o >> 2) & 3);                          instruction i = rows 16i .. 16i+15
    // aux image: row r (64 B) of the wave's 64 keeps granule c at c ^ ((r >> 2) & 3);          instruction k = rows 16k .. 16k+15
    auto issue = [&](int c, int slot) {
        char* buf = smem + slot * WSLOT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = wave * 4 + j;
            const int ra = 2 * i + (lane >> 5), pa = lane & 31;
            const int ca = (pa & 16) | ((pa ^ ra) & 15);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rWa, (lds_void_ptr)(buf + i * 1024), 16, (ra * 256 + ca * 8) * 2, c * (HC * 512), 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = wave * 4 + j;
            const int rb = 16 * i + (lane >> 2), pb = lane & 3;
            const int cb = pb ^ ((rb >> 2) & 3);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rWb, (lds_void_ptr)(buf + WA_B + i * 1024), 16, (rb * p.F + cb * 8) * 2, c * (HC * 2), 0,
                                                     0);
        }
        if constexpr (MODE == 1) {
            char* ab = smem + AUX0 + (wave * NSLOT + slot) * ASLOT;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 16 * k + (lane >> 2), pc = lane & 3;
                const int cc = pc ^ ((r >> 2) & 3);
                if (!(p.abl & 2))
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_ptr)(ab + k * 1024), 16,
                                                             ((wave * 64 + r) * (int)p.lda + cc * 8) * 2, c * (HC * 2), 0, 0);
            }
        }
    };
    issue(0, 0);
    if (nch > 1) issue(1, 1);

    // ---- this wave's rows of X: MFMA B operand (k = the 256 inputs, 16 per step), both row tiles --------------------------
    u32x4_t xf[2][16];
    {
        const int vx = (rowl * (int)p.ldx + 8 * g) * 2;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int s = 0; s < 16; ++s)
                xf[rt][s] = __builtin_amdgcn_raw_buffer_load_b128(rX, vx + rt * 32 * (int)p.ldx * 2, s * 32, 0);
    }
    const int vh = (rowl * (int)p.ldh + 8 * g) * 2, vh1 = vh + 32 * (int)p.ldh * 2;
    const int va = (rowl * (int)p.lda + 8 * g) * 2, va1 = va + 32 * (int)p.lda * 2;
    const unsigned lds0 = (unsigned)(size_t)(lds_void_ptr)smem;
    if constexpr (MODE == 0) {   // the first layer's bias, once, behind the weight ring (read back 16 floats per chunk)
        float* sb = reinterpret_cast<float*>(smem + AUX0);
        for (int i = tid; i < p.F; i += 256) sb[i] = p.ba[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the first chunk's barrier publishes it)
    }

    // ---- LDS fragment addresses (slot 0; advanced per chunk) -----------------------------------------------------------
    const int pm = perm32(n);
    unsigned a1[8], a2[2], ax4[4];
#pragma unroll
    for (int s = 0; s < 8; ++s) a1[s] = lds0 + pm * 512 + (((2 * s + g) ^ pm) & 15) * 16;                 // + (s>>3)*256
#pragma unroll
    for (int q = 0; q < 2; ++q) a2[q] = lds0 + WA_B + pm * 64 + (((2 * q + g) ^ (pm >> 2)) & 3) * 16;       // q = s2 ; + it*2048
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // q = 2*rt + h: row 32 rt + n of the wave's aux tile, granule 2h + g
        const int r = 32 * (q >> 1) + n;
        ax4[q] = lds0 + AUX0 + wave * NSLOT * ASLOT + r * 64 + (((2 * (q & 1) + g) ^ (r >> 2)) & 3) * 16;
    }
    const unsigned ab0 = lds0 + AUX0 + 8 * g * 4;   // mode 0: bias of hidden units 8g .. 8g+7 (+ chunk, + 16 h)

    f32x16 acc[2][8];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][it][r] = 0.f;
            asm volatile("; acc -> AGPR" : "+a"(acc[rt][it]));
        }

    int slot = 0;
    for (int c = 0; c < nch; ++c) {
        // chunk c's operands have landed (this wave's share; the barrier makes it everyone's).  They were issued two chunks ago;
        // younger: that chunk's stores and everything of the chunk in between — a store has two chunks to complete before a wait
        // reaches it (vmcnt is one in-order counter: with a one-chunk window the loads waited for the stores, 197 us against 122 / 133
        // with either stream alone, profiles/round4_mlp_chain_lab.md)
        constexpr int PER = (MODE == 1 ? 16 : 16), ST = (MODE == 1 ? 4 : 8);
        if (c == 0) chain_wait_vm<0>();
        else if (c + 1 < nch) chain_wait_vm<ST + PER>();
        else chain_wait_vm<2 * ST>();   // (the chunk before the last one fetched nothing)
        __builtin_amdgcn_s_barrier();
        if (c + 2 < nch) issue(c + 2, slot == 0 ? 2 : slot - 1);
        const int so = c * HC * 2;   // byte offset of this chunk's first hidden unit in a row

        // ---- T^T tile (32 hidden x 2 x 32 rows) ---------------------------------------------------------------------
        f32x16 d0, d1;
        u32x4_t fa[4], fb[4], ax[2][2];
#define CH_A1(buf_, k_) \
    lds_read4<((k_) >> 1) * 256>(buf_, a1[(4 * (k_)) & 7], a1[(4 * (k_) + 1) & 7], a1[(4 * (k_) + 2) & 7], a1[(4 * (k_) + 3) & 7])
        CH_A1(fa, 0);
        CH_A1(fb, 1);
        if constexpr (MODE == 0) {   // T starts from the bias: accumulator registers 8h .. 8h+7 = hidden units 16 h + 8 g + [0, 8)
            const unsigned bo = ab0 + (unsigned)(c * HC * 4);
            f32x4 b[4];
            asm volatile(
                "ds_read_b128 %0, %4\n\t"
                "ds_read_b128 %1, %4 offset:16\n\t"
                "ds_read_b128 %2, %4 offset:64\n\t"
                "ds_read_b128 %3, %4 offset:80\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
                : "v"(bo)
                : "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r) { d0[r] = b[r >> 2][r & 3]; d1[r] = b[r >> 2][r & 3]; }
            asm volatile("s_nop 1" : "+v"(d0), "+v"(d1));
            lds_wait<0>(fa);
            mfma_v(d0, fa[0], xf[0][0]);
            mfma_v(d1, fa[0], xf[1][0]);
        } else {
            lds_wait<4>(fa);
            mfma_v0(d0, fa[0], xf[0][0]);
            mfma_v0(d1, fa[0], xf[1][0]);
        }
#pragma unroll
        for (int i = 1; i < 4; ++i) { mfma_v(d0, fa[i], xf[0][i]); mfma_v(d1, fa[i], xf[1][i]); }
        CH_A1(fa, 2);
        lds_wait<4>(fb);
#pragma unroll
        for (int i = 0; i < 4; ++i) { mfma_v(d0, fb[i], xf[0][4 + i]); mfma_v(d1, fb[i], xf[1][4 + i]); }
        CH_A1(fb, 3);
        lds_wait<4>(fa);
#pragma unroll
        for (int i = 0; i < 4; ++i) { mfma_v(d0, fa[i], xf[0][8 + i]); mfma_v(d1, fa[i], xf[1][8 + i]); }
        // mode 1: this chunk's aux tile and the first Wb fragments of the second product, their latency under the last MFMAs / f.
        // (Mode 1 only: the registers of a read in flight must not be touched before its wait, and hipcc, which sees an ordinary
        // asm output, is free to copy or spill them when f needs registers — the forward's GELU does, found by the parity check)
        if constexpr (MODE == 1) {
            asm volatile(
                "ds_read_b128 %0, %4\n\t"
                "ds_read_b128 %1, %5\n\t"
                "ds_read_b128 %2, %6\n\t"
                "ds_read_b128 %3, %7"
                : "=&v"(ax[0][0]), "=&v"(ax[0][1]), "=&v"(ax[1][0]), "=&v"(ax[1][1])
                : "v"(ax4[0]), "v"(ax4[1]), "v"(ax4[2]), "v"(ax4[3])
                : "memory");
            lds_read2x2<0, 2048>(fa, a2[0], a2[1]);
            lds_wait<8>(fb);
        } else {
            lds_wait<0>(fb);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { mfma_v(d0, fb[i], xf[0][12 + i]); mfma_v(d1, fb[i], xf[1][12 + i]); }
#undef CH_A1
        if constexpr (MODE == 1)
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ax[0][0]), "+v"(ax[0][1]), "+v"(ax[1][0]), "+v"(ax[1][1]));
        chain_settle(d0, d1);

        // ---- U = f(T): per row tile two groups of 8 consecutive hidden units ---------------------------------------------
        u32x4_t ub[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const f32x16& d = rt ? d1 : d0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = d[8 * h + r];
                if constexpr (MODE == 0) {
                    float dv[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float x = v[r];
                        float Phi, E;
                        gelu_parts(x, Phi, E);
                        v[r] = x * Phi;
                        dv[r] = fmaf(x * 0.39894228040143267794f, E, Phi);
                    }
                    if (!(p.abl & 1)) __builtin_amdgcn_raw_buffer_store_b128(pack8(dv), rA, rt ? va1 : va, so + 32 * h, 0);
                } else {
                    float a8[8];
                    unpack8(ax[rt][h], a8);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] *= a8[r];
                }
                ub[rt][h] = pack8(v);
            }
        }
#define CH_ST(AUXB, SO)                                                             \
    __builtin_amdgcn_raw_buffer_store_b128(ub[0][0], rH, vh, (SO), AUXB);           \
    __builtin_amdgcn_raw_buffer_store_b128(ub[0][1], rH, vh, (SO) + 32, AUXB);      \
    __builtin_amdgcn_raw_buffer_store_b128(ub[1][0], rH, vh1, (SO), AUXB);          \
    __builtin_amdgcn_raw_buffer_store_b128(ub[1][1], rH, vh1, (SO) + 32, AUXB);
        if (p.abl & 32) { CH_ST(0, (c & 3) * 64) }   // lab: every chunk into the first 128 hidden columns (L2-resident lines)
        else if (p.abl & 8) { CH_ST(1, so) }
        else if (p.abl & 16) { CH_ST(3, so) }
        else if (p.abl & 64) { CH_ST(0, so) }
        else if (!(p.abl & 1)) { CH_ST(2, so) }   // streaming (nt) stores: 197 us against 248 with the default policy (M = 50176, backward, alone)
#undef CH_ST

        // ---- Y^T += Wb rows x U^T: 8 output tiles x 2 k-steps x 2 row tiles ---------------------------------------------
        // a group = two output tiles (k-steps 0, 1 each); groups alternate between fa and fb
#define CH_A2(buf_, j_) lds_read2x2<(2 * (j_)) * 2048, (2 * (j_) + 1) * 2048>(buf_, a2[0], a2[1])
#define CH_M2(buf_, j_)                                  \
    mfma_a(acc[0][2 * (j_)], buf_[0], ub[0][0]);         \
    mfma_a(acc[1][2 * (j_)], buf_[0], ub[1][0]);         \
    mfma_a(acc[0][2 * (j_)], buf_[1], ub[0][1]);         \
    mfma_a(acc[1][2 * (j_)], buf_[1], ub[1][1]);         \
    mfma_a(acc[0][2 * (j_) + 1], buf_[2], ub[0][0]);     \
    mfma_a(acc[1][2 * (j_) + 1], buf_[2], ub[1][0]);     \
    mfma_a(acc[0][2 * (j_) + 1], buf_[3], ub[0][1]);     \
    mfma_a(acc[1][2 * (j_) + 1], buf_[3], ub[1][1]);
        if constexpr (MODE == 0) lds_read2x2<0, 2048>(fa, a2[0], a2[1]);
        CH_A2(fb, 1);
        lds_wait_ops<4>(fa, ub);
        CH_M2(fa, 0)
        CH_A2(fa, 2);
        lds_wait<4>(fb);
        CH_M2(fb, 1)
        CH_A2(fb, 3);
        lds_wait<4>(fa);
        CH_M2(fa, 2)
        lds_wait<0>(fb);
        CH_M2(fb, 3)
#undef CH_A2
#undef CH_M2
        // next chunk's operands live in the next slot
        const bool wrap = slot == NSLOT - 1;
        const unsigned dw = wrap ? (unsigned)(-(NSLOT - 1) * WSLOT) : (unsigned)WSLOT;
        const unsigned da = wrap ? (unsigned)(-(NSLOT - 1) * ASLOT) : (unsigned)ASLOT;
#pragma unroll
        for (int s = 0; s < 8; ++s) a1[s] += dw;
#pragma unroll
        for (int q = 0; q < 2; ++q) a2[q] += dw;
#pragma unroll
        for (int q = 0; q < 4; ++q) ax4[q] += da;
        slot = wrap ? 0 : slot + 1;
    }
    // last MFMAs -> accumulator reads
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");

    // ---- epilogue: a lane holds, per output tile, outputs 32 it + 16 h + 8 g + [0, 8) of its row ------------------------------
    if constexpr (MODE == 1) {
        const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(reinterpret_cast<h16_t*>(p.Y) + (int64_t)row_base * p.ldy), 0, (int)((((int64_t)rows_here - 1) * p.ldy + 256) * 2),
            0x00020000);
        const int vy = (rowl * (int)p.ldy + 8 * g) * 2;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int it = 0; it < 8; ++it)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = acc[rt][it][8 * h + r];
                    __builtin_amdgcn_raw_buffer_store_b128(pack8(v), rY, vy + rt * 32 * (int)p.ldy * 2, (32 * it + 16 * h) * 2, 0);
                }
    } else {
        const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(reinterpret_cast<float*>(p.Y) + (int64_t)row_base * p.ldy), 0, (int)((((int64_t)rows_here - 1) * p.ldy + 256) * 4),
            0x00020000);
        const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.res + (int64_t)row_base * p.ldr), 0, (int)((((int64_t)rows_here - 1) * p.ldr + 256) * 4), 0x00020000);
        const int vy = (rowl * (int)p.ldy + 8 * g) * 4, vr = (rowl * (int)p.ldr + 8 * g) * 4;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int it = 0; it < 8; ++it)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int col = 32 * it + 16 * h;
                    const float* bp = p.bb + col + 8 * g;
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
                    // (the loaded vectors are re-typed WHOLE: hipcc 7.2 turns a per-element bit_cast of a b128 buffer load into a
                    // one-dword load splatted over the four lanes)
                    const f32x4 r0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rR, vr + rt * 32 * (int)p.ldr * 4, col * 4, 0));
                    const f32x4 r1 =
                        __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rR, vr + rt * 32 * (int)p.ldr * 4, col * 4 + 16, 0));
                    f32x4 o0, o1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o0[r] = acc[rt][it][8 * h + r] + b0[r] + r0[r];
                        o1[r] = acc[rt][it][8 * h + 4 + r] + b1[r] + r1[r];
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o0), rY, vy + rt * 32 * (int)p.ldy * 4, col * 4, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o1), rY, vy + rt * 32 * (int)p.ldy * 4, col * 4 + 16, 0);
                }
    }
}

__global__ __launch_bounds__(256, 1) void mlp_chain_fwd_bf16(ChainArgs p) { mlp_chain_body<0>(p); }
__global__ __launch_bounds__(256, 1) void mlp_chain_bwd_bf16(ChainArgs p) { mlp_chain_body<1>(p); }

}  // namespace

// launcher behind svol_mlp_chain (gemm.hip).  SVOL_E_UNSUPPORTED when the shape does not qualify; dry: checks only.
int svol_mlp_chain_bf16(const void* X, int64_t ldx, const void* Wa, const void* Wb, void* hid, int64_t ldh, const void* aux_in,
                        void* aux_out, int64_t lda, void* Y, int64_t ldy, const float* ba, const float* bb, const float* res, int64_t ldr,
                        int mode, int64_t M, int64_t F, hipStream_t s, int dry) {
    static const bool off = getenv("SVOL_NO_MLP_CHAIN") != nullptr;
    if (off || M < 1 || F < HC || F % HC || F * 4 > 4 * NSLOT * ASLOT) return SVOL_E_UNSUPPORTED;   // (mode 0 keeps the bias in LDS)
    if (ldx % 8 || ldh % 8 || lda % 8 || (mode == 0 ? (ldy % 4 || ldr % 4) : (ldy % 8 != 0))) return SVOL_E_UNSUPPORTED;
    if (!aligned16(X) || !aligned16(Wa) || !aligned16(Wb) || !aligned16(hid) || !aligned16(Y)) return SVOL_E_UNSUPPORTED;
    if (mode == 0 ? (!aux_out || !aligned16(aux_out) || !ba || !bb || !res || !aligned16(res) || !aligned16(ba) || !aligned16(bb))
                  : (!aux_in || !aligned16(aux_in)))
        return SVOL_E_UNSUPPORTED;
    int64_t ldbig = ldx > ldh ? ldx : ldh;
    if (lda > ldbig) ldbig = lda;
    if (2 * ldy > ldbig) ldbig = 2 * ldy;
    if (2 * ldr > ldbig) ldbig = 2 * ldr;
    if (256 * ldbig * 2 + F * 2 >= (1ll << 31) || 256 * F * 2 >= (1ll << 31)) return SVOL_E_UNSUPPORTED;   // 32-bit buffer offsets
    if (dry) return SVOL_OK;
    // the lab's ablation mask (results INVALID) is honoured only together with SVOL_LAB=1: a stray SVOL_CHAIN_ABL alone is refused
    // instead of silently skipping stores; both are read once
    static const int abl = getenv("SVOL_CHAIN_ABL") ? atoi(getenv("SVOL_CHAIN_ABL")) : 0;
    static const bool lab = getenv("SVOL_LAB") != nullptr;
    if (abl && !lab) return SVOL_E_INVALID;
    ChainArgs p{(const h16_t*)X, (const h16_t*)Wa, (const h16_t*)Wb, (h16_t*)hid, (const h16_t*)aux_in, (h16_t*)aux_out, Y, ba, bb, res,
                ldx, ldh, lda, ldy, ldr, (int)M, (int)F, abl};
    const unsigned grid = (unsigned)((M + 255) / 256);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_chain_fwd_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_chain_bwd_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS);
        attr_done = true;
    }
    if (mode == 0) hipLaunchKernelGGL(mlp_chain_fwd_bf16, dim3(grid), dim3(256), CH_LDS, s, p);
    else hipLaunchKernelGGL(mlp_chain_bwd_bf16, dim3(grid), dim3(256), CH_LDS, s, p);
    return hipGetLastError() == hipSuccess ? SVOL_OK : SVOL_E_LAUNCH;
}
