// ViT-B/16 feature extractor pieces (Hugging Face ViTModel semantics, inference only):
//
//   svol_patchify        pixel_values [n,C,H,W] fp32 -> patch rows [n*P, C*p*p] (the im2col of the stride-p conv, in the
//                        conv weight's own (c, ky, kx) order) so that the patch embedding is ONE MFMA GEMM
//   svol_vit_embed       tokens = [cls; patch_proj] + position embeddings, fp32 residual stream + compute-dtype copy
//   svol_attn_small_fwd  softmax(Q K^T / sqrt(d_h)) V for SHORT sequences (L <= 256: 197 tokens per image), d_h = 32 or
//                        64, bf16.  One workgroup per (image, head): K and V sit in LDS once (row-major images with
//                        XOR-swizzled 16-byte chunks, V read back through ds_read_b64_tr_b16), a wave owns 32 queries
//                        at a time with the WHOLE score row of its query in registers (<= 8 accumulator tiles), so the
//                        softmax is a plain two-pass one — no running maximum, no rescale; swapped products keep the
//                        statistics lane-local and feed P to the second MFMA straight from the accumulators, as in
//                        attention_bf16.hip.  Attention is 4 % of ViT-B's FLOPs; the GEMMs (gemm_bf16.hip) carry it.
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

// ---- patchify -------------------------------------------------------------------------------------------------
// one thread per (patch, c, ky): 16-float contiguous read (p = 16 -> kx run), p outputs
template <typename T>
__global__ void patchify_kernel(const float* __restrict__ pix, T* __restrict__ out, int n, int C, int H, int W, int p) {
    const int gw = W / p, gh = H / p;
    const int64_t total = (int64_t)n * gh * gw * C * p;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ky = (int)(i % p);
    const int c = (int)((i / p) % C);
    const int64_t patch = i / ((int64_t)p * C);
    const int px = (int)(patch % gw), py = (int)((patch / gw) % gh);
    const int64_t img = patch / ((int64_t)gw * gh);
    const float* src = pix + ((img * C + c) * H + (py * p + ky)) * (int64_t)W + px * p;
    T* dst = out + patch * ((int64_t)C * p * p) + ((int64_t)c * p + ky) * p;
    for (int kx = 0; kx < p; ++kx) dst[kx] = from_f32<T>(src[kx]);
}

// ---- cls + position embeddings ----------------------------------------------------------------------------------
template <typename T>
__global__ void vit_embed_kernel(const float* __restrict__ proj, const float* __restrict__ cls, const float* __restrict__ pos,
                                 float* __restrict__ x32, T* __restrict__ x, int64_t n, int P, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 4 channels
    const int64_t total = n * (P + 1) * (D / 4);
    if (i >= total) return;
    const int c = (int)(i % (D / 4)) * 4;
    const int64_t row = i / (D / 4);
    const int tok = (int)(row % (P + 1));
    const int64_t img = row / (P + 1);
    const f32x4 pe = *reinterpret_cast<const f32x4*>(pos + (int64_t)tok * D + c);
    const f32x4 src = tok == 0 ? *reinterpret_cast<const f32x4*>(cls + c)
                               : *reinterpret_cast<const f32x4*>(proj + (img * P + tok - 1) * D + c);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = src[e] + pe[e];
    *reinterpret_cast<f32x4*>(x32 + row * D + c) = v;
    if (x) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[row * D + c + e] = from_f32<T>(v[e]);
    }
}

// ---- short-sequence attention -------------------------------------------------------------------------------------
struct SmallArgs {
    const bf16_t *q, *k, *v;
    bf16_t* o;
    int64_t ldq, ldk, ldv, ldo;
    int H, L, nkb;  // nkb = ceil(L / 32) <= 8
    float scale_log2e;
};
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

// image: row-major, DH bf16 per row (DH*2 bytes = DH/8 chunks of 16 bytes), chunk index XOR-ed with the row
template <int DH> __device__ __forceinline__ int img_off(int row, int ch) {
    constexpr int CPR = DH / 8;
    return row * (DH * 2) + (((ch ^ (row >> (DH == 32 ? 2 : 0))) & (CPR - 1)) << 4);
}

template <int DH>
__global__ __launch_bounds__(256, 2) void attn_small_kernel(SmallArgs p) {
    constexpr int CPR = DH / 8, KS = DH / 16, DB = DH / 32;
    constexpr int LMAX = 256;
    __shared__ __attribute__((aligned(16))) char smem[2 * LMAX * DH * 2];
    char* sK = smem;
    char* sV = smem + LMAX * DH * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int seq = blockIdx.y, hh = blockIdx.x;
    const bf16_t* Q = p.q + (int64_t)seq * p.L * p.ldq + hh * DH;
    const bf16_t* K = p.k + (int64_t)seq * p.L * p.ldk + hh * DH;
    const bf16_t* V = p.v + (int64_t)seq * p.L * p.ldv + hh * DH;
    const int Lp = p.nkb * 32;
    // stage K and V (zero rows past L)
    for (int c = tid; c < Lp * CPR; c += 256) {
        const int row = c / CPR, ch = c % CPR;
        const uint4 z = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(sK + img_off<DH>(row, ch)) = row < p.L ? *reinterpret_cast<const uint4*>(K + (int64_t)row * p.ldk + ch * 8) : z;
        *reinterpret_cast<uint4*>(sV + img_off<DH>(row, ch)) = row < p.L ? *reinterpret_cast<const uint4*>(V + (int64_t)row * p.ldv + ch * 8) : z;
    }
    __syncthreads();
    for (int qb = wave; qb < p.nkb; qb += 4) {
        const int qrow = qb * 32 + r;
        const bool qvalid = qrow < p.L;
        uint4 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[ks] = qvalid ? *reinterpret_cast<const uint4*>(Q + (int64_t)qrow * p.ldq + ks * 16 + h * 8) : make_uint4(0, 0, 0, 0);
        // scores of this lane's query against every key: S[kb][i] <-> key kb*32 + 8*(i/4) + 4*h + i%4
        f32x16 S[8];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            if (kb < p.nkb) {
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const uint4 kf = *reinterpret_cast<const uint4*>(sK + img_off<DH>(kb * 32 + r, ks * 2 + h));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[ks]), acc, 0, 0, 0);
                }
                S[kb] = acc;
            }
        }
        // mask keys >= L (only the last block can hold them), row maximum over registers then across the two half-waves
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            if (kb < p.nkb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kb * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
                    if (key >= p.L) S[kb][i] = -INFINITY;
                    m = fmaxf(m, S[kb][i]);
                }
            }
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float mc = -m * p.scale_log2e;
        float l = 0.f;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            if (kb < p.nkb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    S[kb][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kb][i], p.scale_log2e, mc));
                    l += S[kb][i];
                }
            }
        }
        l += __shfl_xor(l, 32, 64);
        // O^T[d][q] += V^T[d][keys] * P^T[keys][q]
        f32x16 O[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) O[db][i] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            if (kb < p.nkb) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 pb;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pb[j] = (bf16_t)S[kb][8 * s + j];
#pragma unroll
                    for (int db = 0; db < DB; ++db) {
                        // transposed fragment of V: lane (r -> column d = db*32 + r); per 16-lane group, lane 4q+pp supplies
                        // the address of key row q (of 4), columns 4pp..4pp+3 of its 16-column group
                        const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3, hb = g >> 1;
                        const int col = db * 32 + 16 * (g & 1) + 4 * pp;  // first of 4 columns this lane addresses
                        const int r1 = kb * 32 + 16 * s + 4 * hb + q4;
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (lds_bf16x4_ptr)(sV + img_off<DH>(r1, col >> 3) + (col & 7) * 2));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (lds_bf16x4_ptr)(sV + img_off<DH>(r1 + 8, col >> 3) + (col & 7) * 2));
                        const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        O[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, O[db], 0, 0, 0);
                    }
                }
            }
        }
        if (qvalid) {
            const float inv = 1.f / l;
            bf16_t* out = p.o + ((int64_t)seq * p.L + qrow) * p.ldo + hh * DH;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 v4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] = (bf16_t)(O[db][4 * g + e] * inv);
                    *reinterpret_cast<bf16x4*>(out + db * 32 + 8 * g + 4 * h) = v4;
                }
        }
    }
}

}  // namespace

extern "C" {

int svol_patchify(const float* pixel_values, void* out, int64_t n, int64_t C, int64_t H, int64_t W, int64_t p, int dtype,
                  void* stream) {
    if (!pixel_values || !out || n <= 0 || C <= 0 || H <= 0 || W <= 0 || p <= 0) return SVOL_E_INVALID;
    if (H % p || W % p) return SVOL_E_UNSUPPORTED;
    const int64_t total = n * (H / p) * (W / p) * C * p;
    if (total > (1ll << 40)) return SVOL_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == SVOL_BF16) hipLaunchKernelGGL(patchify_kernel<bf16_t>, grid, dim3(256), 0, s, pixel_values, (bf16_t*)out, (int)n, (int)C, (int)H, (int)W, (int)p);
    else if (dtype == SVOL_F32) hipLaunchKernelGGL(patchify_kernel<float>, grid, dim3(256), 0, s, pixel_values, (float*)out, (int)n, (int)C, (int)H, (int)W, (int)p);
    else return SVOL_E_INVALID;
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_vit_embed(const float* patch_proj, const float* cls_token, const float* pos_embed, float* x32, void* x, int64_t n,
                   int64_t P, int64_t D, int dtype, void* stream) {
    if (!patch_proj || !cls_token || !pos_embed || !x32 || n <= 0 || P <= 0 || D <= 0) return SVOL_E_INVALID;
    if (D % 4) return SVOL_E_UNSUPPORTED;
    const int64_t total = n * (P + 1) * (D / 4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == SVOL_BF16) hipLaunchKernelGGL(vit_embed_kernel<bf16_t>, grid, dim3(256), 0, s, patch_proj, cls_token, pos_embed, x32, (bf16_t*)x, n, (int)P, (int)D);
    else if (dtype == SVOL_F32) hipLaunchKernelGGL(vit_embed_kernel<float>, grid, dim3(256), 0, s, patch_proj, cls_token, pos_embed, x32, (float*)x, n, (int)P, (int)D);
    else return SVOL_E_INVALID;
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

int svol_attn_small_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo,
                        int64_t n_seq, int64_t H, int64_t L, int64_t dh, float scale, int dtype, void* stream) {
    if (!q || !k || !v || !o || n_seq <= 0 || H <= 0 || L <= 0) return SVOL_E_INVALID;
    if (dtype != SVOL_BF16 || (dh != 32 && dh != 64) || L > 256 || n_seq > 65535) return SVOL_E_UNSUPPORTED;
    if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4 || !aligned16(q) || !aligned16(k) || !aligned16(v) || (reinterpret_cast<uintptr_t>(o) & 7))
        return SVOL_E_UNSUPPORTED;
    SmallArgs p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, ldq, ldk, ldv, ldo, (int)H, (int)L,
                (int)((L + 31) / 32), scale * LOG2E};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)H, (unsigned)n_seq);
    if (dh == 64) hipLaunchKernelGGL(attn_small_kernel<64>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(attn_small_kernel<32>, grid, dim3(256), 0, s, p);
    SVOL_CHECK_LAUNCH();
    return SVOL_OK;
}

}  // extern "C"
