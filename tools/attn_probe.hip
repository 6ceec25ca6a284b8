// Developer probe (round 5): k_attn_short (round 2-4 form) against k_attn_short2 on the 0.6B / 4B attention shapes, stand-alone:
// bitwise comparison of everything the launch writes (xb, its int8 + scales, the key-cache row) and the launch period of a
// hipGraph chain of the kernel at a fixed position; with -DQ3_DEV also the in-kernel timelines of workgroup 3.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DQ3_DEV -o tools/attn_probe tools/attn_probe.hip
#include "../qwen3-rs_amd/csrc/q3_kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <chrono>
using namespace q3;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

static double gauss() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307 * v); }

int main(int argc, char** argv) {
    const int n_heads = argc > 1 ? atoi(argv[1]) : 16, n_kv = argc > 2 ? atoi(argv[2]) : 8, HD = 128, S = 1024;
    const size_t kvd = (size_t)n_kv * HD;
    srand(7);
    std::vector<float> hq(n_heads * HD), hk(kvd), hkc(S * kvd), hvc(S * kvd), hqw(HD), hkw(HD), hrope((size_t)S * HD);
    for (auto& v : hq) v = (float)gauss() * 3.0f;
    for (auto& v : hk) v = (float)gauss() * 3.0f;
    for (auto& v : hkc) v = (float)gauss();
    for (auto& v : hvc) v = (float)gauss();
    for (auto& v : hqw) v = 1.0f + 0.1f * (float)gauss();
    for (auto& v : hkw) v = 1.0f + 0.1f * (float)gauss();
    for (int p = 0; p < S; ++p) for (int i = 0; i < HD / 2; ++i) { const float fr = powf(1e6f, -(float)i / (HD / 2)), ang = (float)p * fr; hrope[(size_t)p * HD + 2 * i] = cosf(ang); hrope[(size_t)p * HD + 2 * i + 1] = sinf(ang); }
    float *dq, *dk, *dkc[2], *dvc, *dqw, *dkw, *drope, *dxb[2], *dxbs[2]; int8_t* dxbq[2]; State* dst; unsigned long long* dstamps;
    CK(hipMalloc(&dq, 4 * hq.size())); CK(hipMalloc(&dk, 4 * hk.size())); CK(hipMalloc(&dvc, 4 * hvc.size()));
    CK(hipMalloc(&dqw, 4 * HD)); CK(hipMalloc(&dkw, 4 * HD)); CK(hipMalloc(&drope, 4 * hrope.size())); CK(hipMalloc(&dst, sizeof(State))); CK(hipMalloc(&dstamps, 8 * 16));
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&dkc[i], 4 * hkc.size())); CK(hipMalloc(&dxb[i], 4 * n_heads * HD)); CK(hipMalloc(&dxbs[i], 4 * n_heads * HD / 64)); CK(hipMalloc(&dxbq[i], n_heads * HD)); }
    CK(hipMemcpy(dq, hq.data(), 4 * hq.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dk, hk.data(), 4 * hk.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dvc, hvc.data(), 4 * hvc.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dqw, hqw.data(), 4 * HD, hipMemcpyHostToDevice));
    CK(hipMemcpy(dkw, hkw.data(), 4 * HD, hipMemcpyHostToDevice)); CK(hipMemcpy(drope, hrope.data(), 4 * hrope.size(), hipMemcpyHostToDevice));
    const size_t smem2 = attn_short2_smem_bytes(HD, kS2MaxT);
    CK(hipFuncSetAttribute((const void*)k_attn_short2<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto args = [&](int which) {
        AttnArgs a; memset(&a, 0, sizeof(a));
        a.q = dq; a.key_cache = dkc[which]; a.k_raw = dk; a.value_cache = dvc; a.q_norm_w = dqw; a.k_norm_w = dkw; a.rope = drope; a.xb = dxb[which];
        a.st = dst; a.pos_override = -1; attn_set_heads(a, n_heads, n_kv); a.hd = HD; a.seq_len = S; a.strict = 1;
        a.xbq = dxbq[which]; a.xbs = dxbs[which]; a.xb_group = 64; a.stamps = dstamps;
        return a;
    };
    auto launch = [&](int which) {
        const AttnArgs a = args(which);
        if (which == 0) hipLaunchKernelGGL(k_attn_short<128>, dim3(n_heads), dim3(kWG), 0, s, a);
        else hipLaunchKernelGGL(k_attn_short2<128>, dim3(n_heads, getenv("NOVSPLIT") ? 1 : 2), dim3(kS2Threads), smem2, s, a);
    };
    int bad_total = 0;
    for (int pos : {0, 1, 7, 8, 9, 16, 17, 31, 32, 33, 63, 64, 65, 70, 100, 127, 128, 129, 135, 160, 161, 200, 255}) {
        State st; memset(&st, 0, sizeof(st)); st.pos = pos; st.token = 5;
        CK(hipMemcpy(dst, &st, sizeof(st), hipMemcpyHostToDevice));
        double period[2]; unsigned long long stamps[2][16];
        for (int which = 0; which < 2; ++which) {
            CK(hipMemcpy(dkc[which], hkc.data(), 4 * hkc.size(), hipMemcpyHostToDevice));
            CK(hipMemset(dstamps, 0, 8 * 16));
            launch(which); CK(hipStreamSynchronize(s));
            CK(hipMemcpy(stamps[which], dstamps, 128, hipMemcpyDeviceToHost));
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < 100; ++i) launch(which);
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            double best = 1e30;
            for (int rep = 0; rep < 5; ++rep) {
                auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 1000.0;
                if (us < best) best = us;
            }
            period[which] = best;
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        std::vector<float> xb0(n_heads * HD), xb1(n_heads * HD), s0(n_heads * HD / 64), s1(n_heads * HD / 64), k0(kvd), k1(kvd);
        std::vector<int8_t> q0(n_heads * HD), q1(n_heads * HD);
        CK(hipMemcpy(xb0.data(), dxb[0], 4 * xb0.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(xb1.data(), dxb[1], 4 * xb1.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(s0.data(), dxbs[0], 4 * s0.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(s1.data(), dxbs[1], 4 * s1.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(q0.data(), dxbq[0], q0.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(q1.data(), dxbq[1], q1.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(k0.data(), dkc[0] + (size_t)pos * kvd, 4 * kvd, hipMemcpyDeviceToHost)); CK(hipMemcpy(k1.data(), dkc[1] + (size_t)pos * kvd, 4 * kvd, hipMemcpyDeviceToHost));
        const int bad = (memcmp(xb0.data(), xb1.data(), 4 * xb0.size()) != 0) + (memcmp(s0.data(), s1.data(), 4 * s0.size()) != 0) +
                        (memcmp(q0.data(), q1.data(), q0.size()) != 0) + (memcmp(k0.data(), k1.data(), 4 * kvd) != 0);
        bad_total += bad;
        printf("pos %3d  %s  period old %.2f us  new %.2f us", pos, bad ? "MISMATCH" : "bit-identical", period[0], period[1]);
        for (int which = 0; which < 2; ++which) {
            const unsigned long long* t = stamps[which];
            if (t[0]) printf("  | %s: issued %llu norm %llu staged %llu scores %llu softmax %llu vsum %llu", which ? "new" : "old", t[1] - t[0], t[2] - t[0], t[3] - t[0], t[4] - t[0], t[5] - t[0], t[6] - t[0]);
            if (which && t[8]) printf(" [stager: pos %llu issued %llu K %llu V %llu; out: sum %llu vchain %llu]", t[8] - t[0], t[9] - t[0], t[11] - t[0], t[10] - t[0], t[13] - t[0], t[14] - t[0]);
        }
        printf("\n");
    }
    printf(bad_total ? "FAILED: %d mismatching outputs\n" : "all outputs bit-identical (%d)\n", bad_total);
    return bad_total != 0;
}
