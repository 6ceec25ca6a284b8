// Feasibility probe for the batched W8A8 group-quant GEMM (B streams x d rows x n contraction) on gfx950:
//   out[b][i] = sum_g ((f32)(sum_k xq[b][g*64+k] * w[i][g*64+k]) * ws[i][g]) * xs[b][g]     (g ascending, tensor.rs:23-62)
// One v_mfma_i32_16x16x64_i8 yields the exact i32 group dots of 16 rows x 16 streams; each lane then owns 4
// (row, stream) outputs and accumulates their group terms itself, in order -> bit-identical to the per-stream GEMV.
// Checks the operand layout against a CPU reference and reports achieved weight bandwidth.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int G = 64;
// wave task: 16 rows (row block rb) x 16 streams (stream block sb); walks all n/64 groups in order.
// KS groups are loaded per step (KS*64 bytes per row), two steps in flight.
template <int KS>
__global__ __launch_bounds__(256) void k_bgemm(const int8_t* __restrict__ w, const float* __restrict__ ws,
                                               const int8_t* __restrict__ xq, const float* __restrict__ xs,
                                               float* __restrict__ out, int n, int d, int B) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nsb = B / 16;
    const int task = blockIdx.x * 4 + wave;            // (row block, stream block): sb fastest so siblings share weights
    const int rb = task / nsb, sb = task % nsb;
    if (rb * 16 >= d) return;
    const int ng = n / G;
    const int i = lane & 15, kq = lane >> 4;           // A: row rb*16+i, k bytes [16kq,+16) of the group; B: stream sb*16+i
    const int8_t* wrow = w + (size_t)(rb * 16 + i) * n + 16 * kq;
    const int8_t* xrow = xq + (size_t)(sb * 16 + i) * n + 16 * kq;
    const float* xsrow = xs + (size_t)(sb * 16 + i) * ng;
    const float* wsr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wsr[r] = ws + (size_t)(rb * 16 + 4 * kq + r) * ng;   // C rows of this lane
    float acc[4] = {-0.0f, -0.0f, -0.0f, -0.0f};
    static_assert(KS % 4 == 0, "scales are fetched as float4 per 4 groups");
    struct Step { v4i a[KS], b[KS]; v4f ws[4][KS / 4], xs[KS / 4]; };
    Step s0, s1;
    auto load = [&](Step& s, int g0) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int g = min(g0 + k, ng - 1);
            s.a[k] = __builtin_nontemporal_load((const v4i*)(wrow + (size_t)g * G));
            s.b[k] = *(const v4i*)(xrow + (size_t)g * G);
        }
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
            const int g = min(g0 + 4 * q, ng - 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) s.ws[r][q] = __builtin_nontemporal_load((const v4f*)(wsr[r] + g));
            s.xs[q] = *(const v4f*)(xsrow + g);
        }
    };
    auto compute = [&](const Step& s, int g0) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int g = g0 + k;
            if (g < ng) {
                const v4i c = __builtin_amdgcn_mfma_i32_16x16x64_i8(s.a[k], s.b[k], (v4i){0, 0, 0, 0}, 0, 0, 0);
                const float xsc = s.xs[k / 4][k % 4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = (float)c[r] * s.ws[r][k / 4][k % 4];
                    t = t * xsc;
                    acc[r] = acc[r] + t;
                }
            }
        }
    };
    load(s0, 0);
    for (int g0 = 0; g0 < ng; g0 += 2 * KS) {
        load(s1, g0 + KS);
        compute(s0, g0);
        load(s0, g0 + 2 * KS);
        compute(s1, g0 + KS);
    }
    // C layout: c[r] = C[row 4*kq + r][col i]  ->  out[stream sb*16 + i][row rb*16 + 4kq + r]
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(size_t)(sb * 16 + i) * d + rb * 16 + 4 * kq + r] = acc[r];
}


// variant 2: one wave = 16 rows x (NSB x 16) streams: the weight fragment and its scales are loaded once and fed to
// NSB MFMAs; 4*NSB accumulators per lane.
template <int KS, int NSB>
__global__ __launch_bounds__(256) void k_bgemm2(const int8_t* __restrict__ w, const float* __restrict__ ws,
                                                const int8_t* __restrict__ xq, const float* __restrict__ xs,
                                                float* __restrict__ out, int n, int d, int B) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rb = blockIdx.x * 4 + wave;
    if (rb * 16 >= d) return;
    const int ng = n / G;
    const int i = lane & 15, kq = lane >> 4;
    const int8_t* wrow = w + (size_t)(rb * 16 + i) * n + 16 * kq;
    const float* wsr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wsr[r] = ws + (size_t)(rb * 16 + 4 * kq + r) * ng;
    float acc[NSB][4];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[sb][r] = -0.0f;
    struct Step { v4i a[KS], b[NSB][KS]; v4f ws[4][KS / 4], xs[NSB][KS / 4]; };
    Step s0, s1;
    auto load = [&](Step& s, int g0) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int g = min(g0 + k, ng - 1);
            s.a[k] = __builtin_nontemporal_load((const v4i*)(wrow + (size_t)g * G));
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) s.b[sb][k] = *(const v4i*)(xq + (size_t)(sb * 16 + i) * n + 16 * kq + (size_t)g * G);
        }
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
            const int g = min(g0 + 4 * q, ng - 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) s.ws[r][q] = __builtin_nontemporal_load((const v4f*)(wsr[r] + g));
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) s.xs[sb][q] = *(const v4f*)(xs + (size_t)(sb * 16 + i) * ng + g);
        }
    };
    auto compute = [&](const Step& s, int g0) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            if (g0 + k < ng) {
#pragma unroll
                for (int sb = 0; sb < NSB; ++sb) {
                    const v4i c = __builtin_amdgcn_mfma_i32_16x16x64_i8(s.a[k], s.b[sb][k], (v4i){0, 0, 0, 0}, 0, 0, 0);
                    const float xsc = s.xs[sb][k / 4][k % 4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t = (float)c[r] * s.ws[r][k / 4][k % 4];
                        t = t * xsc;
                        acc[sb][r] = acc[sb][r] + t;
                    }
                }
            }
        }
    };
    load(s0, 0);
    for (int g0 = 0; g0 < ng; g0 += 2 * KS) {
        load(s1, g0 + KS);
        compute(s0, g0);
        load(s0, g0 + 2 * KS);
        compute(s1, g0 + KS);
    }
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)(sb * 16 + i) * d + rb * 16 + 4 * kq + r] = acc[sb][r];
}

static void cpu_ref(const std::vector<int8_t>& w, const std::vector<float>& ws, const std::vector<int8_t>& xq,
                    const std::vector<float>& xs, std::vector<float>& out, int n, int d, int B) {
    const int ng = n / G;
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < d; ++i) {
            float acc = -0.0f;
            for (int g = 0; g < ng; ++g) {
                int dot = 0;
                for (int k = 0; k < G; ++k) dot += (int)xq[(size_t)b * n + g * G + k] * (int)w[(size_t)i * n + g * G + k];
                float t = (float)dot * ws[(size_t)i * ng + g];
                t = t * xs[(size_t)b * ng + g];
                acc = acc + t;
            }
            out[(size_t)b * d + i] = acc;
        }
}

template <int KS, int VAR = 1>
static void run(int n, int d, int B, bool check) {
    const int ng = n / G;
    const size_t copies = check ? 1 : std::min<size_t>(8, (600ull << 20) / ((size_t)n * d) + 1);
    std::vector<int8_t> hw((size_t)n * d), hx((size_t)B * n);
    std::vector<float> hws((size_t)d * ng), hxs((size_t)B * ng), href((size_t)B * d), hout((size_t)B * d);
    srand(1);
    for (auto& v : hw) v = (int8_t)(rand() % 255 - 127);
    for (auto& v : hx) v = (int8_t)(rand() % 255 - 127);
    for (auto& v : hws) v = (rand() % 1000) * 1e-5f;
    for (auto& v : hxs) v = (rand() % 1000) * 1e-3f;
    int8_t *dw, *dx; float *dws, *dxs, *dout;
    CK(hipMalloc(&dw, hw.size() * copies)); CK(hipMalloc(&dx, hx.size())); CK(hipMalloc(&dws, 4 * hws.size() * copies));
    CK(hipMalloc(&dxs, 4 * hxs.size())); CK(hipMalloc(&dout, 4 * hout.size()));
    for (size_t c = 0; c < copies; ++c) {
        CK(hipMemcpy(dw + c * hw.size(), hw.data(), hw.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dws + c * hws.size(), hws.data(), 4 * hws.size(), hipMemcpyHostToDevice));
    }
    CK(hipMemcpy(dx, hx.data(), hx.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dxs, hxs.data(), 4 * hxs.size(), hipMemcpyHostToDevice));
    const int tasks = VAR == 2 ? (d / 16) : (d / 16) * (B / 16);
    const int grid = (tasks + 3) / 4;
    auto launch = [&](int i) { if (VAR == 2) { hipLaunchKernelGGL((k_bgemm2<KS, 2>), grid, 256, 0, 0, dw + (i % copies) * hw.size(), dws + (i % copies) * hws.size(), dx, dxs, dout, n, d, B); return; } hipLaunchKernelGGL(k_bgemm<KS>, grid, 256, 0, 0, dw + (i % copies) * hw.size(), dws + (i % copies) * hws.size(), dx, dxs, dout, n, d, B); };
    launch(0);
    CK(hipDeviceSynchronize());
    if (check) {
        CK(hipMemcpy(hout.data(), dout, 4 * hout.size(), hipMemcpyDeviceToHost));
        cpu_ref(hw, hws, hx, hxs, href, n, d, B);
        size_t bad = 0;
        for (size_t k = 0; k < hout.size(); ++k) bad += memcmp(&hout[k], &href[k], 4) != 0;
        printf("check var=%d n=%d d=%d B=%d KS=%d: %zu / %zu outputs differ bitwise\n", VAR, n, d, B, KS, bad, hout.size());
    } else {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 30;
        CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) launch(i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)n * d * (1.0 + 4.0 / G);
        printf("var=%d n=%5d d=%6d B=%d KS=%d grid=%d: %.2f us  %.2f TB/s (weights)  %.0f tok-rows/us\n", VAR, n, d, B, KS, grid, ms * 1e3 / reps,
               bytes / (ms * 1e-3 / reps) / 1e12, 0.0);
    }
    hipFree(dw); hipFree(dx); hipFree(dws); hipFree(dxs); hipFree(dout);
}

int main() {
    run<4>(256, 64, 32, true);
    run<4>(1024, 48, 16, true);
    run<8>(2048, 32, 32, true);
    run<4, 2>(2048, 64, 32, true);
    for (auto nd : {std::pair<int,int>{4096, 12288}, {4096, 4096}, {12288, 4096}, {4096, 151936}, {1024, 151936}}) {
        run<4>(nd.first, nd.second, 32, false);
        run<4, 2>(nd.first, nd.second, 32, false);
        run<8, 2>(nd.first, nd.second, 32, false);
    }
    return 0;
}
