// Single-pass attention BACKWARD laboratory, round 5: the re-placed stream of the full workgroups (attn_bwd_sp_bf16: body4) against
// round 4's placement (attn_bwd_sp_bf16_v9) and the two-pass kernels at the benchmark launch shape (B 8, H 8, L 6272, d_h 32, bf16,
// pre-scaled q): gradients compared against the two-pass kernels, kernels timed in interleaved rounds in one process.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-inline-asm -o tools/micro/attn_lab_sp5 tools/micro/attn_lab_sp5.hip
//   run:   tools/micro/attn_lab_sp5 [rounds] [B] [L]        SP_ABLATIONS=1: timing-only ablations of the new body (results invalid)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define SP_LAB 1
#include "../../svol_amd/csrc/attention_bf16.hip"

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

    if (getenv("SP_ROWS")) {   // dq error of batch 0, head 0 per 32-query step (debugging aid)
        for (int stp = 0; stp < L / 32 && stp < 48; ++stp) {
            double se = 0, sr = 0;
            for (int q = 0; q < 32; ++q)
                for (int c = 0; c < 32; ++c) {
                    const size_t j = (size_t)(stp * 32 + q) * 3 * d + c;
                    const double a = bf2f(hg[j]), r = bf2f(href[j]);
                    se += (a - r) * (a - r);
                    sr += r * r;
                }
            printf("  step %2d: rel %.3f |ref| %.3f\n", stp, sqrt(se / std::max(sr, 1e-300)), sqrt(sr));
        }
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
    const dim3 gprep((unsigned)((int64_t)B * L / 32)), gsp(sp_grid(B, H, L)), grnd(sp_round_grid(B, H, L, L));
    // round 4's placement, same three launches: gradients against the two-pass kernels
    CK(hipMemset(dgrad, 0xFF, n * 2));
    hipLaunchKernelGGL(attn_bwd_sp_prep_bf16, gprep, dim3(256), 0, 0, ps);
    hipLaunchKernelGGL(attn_bwd_sp_bf16_v9, gsp, dim3(256), 0, 0, ps);
    hipLaunchKernelGGL(attn_dq_round_bf16, grnd, dim3(256), 0, 0, ps);
    CK(hipDeviceSynchronize());
    compare("v9");
    std::vector<float> t2, t1, tp, tm, tr, t9;
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
        hipLaunchKernelGGL(attn_bwd_sp_prep_bf16, gprep, dim3(256), 0, 0, ps);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(attn_bwd_sp_bf16_v9, gsp, dim3(256), 0, 0, ps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r > 0) t9.push_back(t);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(attn_dq_round_bf16, grnd, dim3(256), 0, 0, ps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r > 0) tr.push_back(t);
    }
    auto rep = [](const char* name, std::vector<float>& v) {
        std::sort(v.begin(), v.end());
        printf("%-52s median %.4f ms  min %.4f ms\n", name, v[v.size() / 2], v[0]);
    };
    rep("two-pass backward (dQ rot + dK/dV dma)", t2);
    rep("single-pass backward (3 launches)", t1);
    rep("  prep (delta, row constants, zero dQ32)", tp);
    rep("  attn_bwd_sp_bf16 (round 5 placement)", tm);
    rep("  attn_bwd_sp_bf16_v9 (round 4 placement)", t9);
    rep("  attn_dq_round_bf16", tr);
    {
        const double blocks = (double)B * H * ((double)L / 32) * ((double)L / 32), per_simd = blocks / 1024.0;
        printf("  -> %.0f 32x32 blocks per SIMD: round 5 %.0f ns = ~%.0f cycles per block at 2.08 GHz; round 4 %.0f\n", per_simd,
               tm[tm.size() / 2] * 1e6 / per_simd, tm[tm.size() / 2] * 1e6 / per_simd * 2.08, t9[t9.size() / 2] * 1e6 / per_simd * 2.08);
    }
    if (getenv("SP_ABLATIONS")) {
        struct AV { const char* name; void (*k)(Args); };
        const AV av[] = {{"lab copy (ABL 0)", attn_bwd_sp_lab<0>}, {"no vector fillers", attn_bwd_sp_lab<1>}, {"no MFMAs", attn_bwd_sp_lab<2>},
                         {"no step barrier", attn_bwd_sp_lab<4>}, {"no dS LDS round trip", attn_bwd_sp_lab<8>}, {"no tile DMA / waits", attn_bwd_sp_lab<16>},
                         {"no fillers, no MFMAs", attn_bwd_sp_lab<3>}};
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
            printf("  ablation %-28s median %.4f ms\n", v.name, tt[tt.size() / 2]);
        }
    }
    return 0;
}
