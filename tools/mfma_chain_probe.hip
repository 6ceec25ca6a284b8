// Developer probe: issue-to-issue latency of a DEPENDENT accumulate chain -- v_add_f32 against v_mfma_f32_4x4x1 (x * 1.0 + C is the
// same correctly rounded add, profiles/r04_mfma_f32_probe.txt) and v_mfma_f32_16x16x1 / 32x32x1 -- one wave and four waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/mfma_chain_probe tools/mfma_chain_probe.hip && tools/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int N = 1024;
__global__ void k_chain(float* out, unsigned long long* cyc, const float* in, int mode) {
    const int lane = threadIdx.x & 63;
    float xs[8];
    for (int i = 0; i < 8; ++i) xs[i] = in[(threadIdx.x + 64 * i) & 1023];
    const float one = (lane & 3) == 0 ? 1.0f : 0.0f;           // B = (1, 0, 0, 0) inside every 4-lane group: only column 0 accumulates
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float res = 0.0f;
    if (mode == 0) {
        float a = in[lane];
        for (int i = 0; i < N; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a = a + xs[u];
        }
        res = a;
    } else if (mode == 1) {
        v4f c = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < N; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) c = __builtin_amdgcn_mfma_f32_4x4x1f32(xs[u], one, c, 0, 0, 0);
        }
        res = c.x + c.y + c.z + c.w;
    } else if (mode == 2) {
        v16f c;
        for (int k = 0; k < 16; ++k) c[k] = 0.0f;
        for (int i = 0; i < N; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) c = __builtin_amdgcn_mfma_f32_16x16x1f32(xs[u], one, c, 0, 0, 0);
        }
        res = c[0] + c[5];
    } else {
        float a = in[lane], b = in[lane + 64];                 // two independent add chains interleaved
        for (int i = 0; i < N; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { a = a + xs[u]; b = b + xs[7 - u]; }
        }
        res = a + b;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = res;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float *din, *dout; unsigned long long* dcyc;
    CK(hipMalloc(&din, 4 * 2048)); CK(hipMalloc(&dout, 4 * 1024)); CK(hipMalloc(&dcyc, 8));
    float h[2048]; for (int i = 0; i < 2048; ++i) h[i] = 1.0f + i * 1e-3f;
    CK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
    const char* names[] = {"v_add_f32 chain", "v_mfma_f32_4x4x1 chain (C = D)", "v_mfma_f32_16x16x1 chain (C = D)", "two interleaved v_add_f32 chains (per pair)"};
    for (int threads : {64, 256, 1024}) {
        for (int mode = 0; mode < 4; ++mode) {
            unsigned long long best = ~0ull;
            for (int rep = 0; rep < 10; ++rep) {
                hipLaunchKernelGGL(k_chain, 1, threads, 0, 0, dout, dcyc, din, mode); CK(hipDeviceSynchronize());
                unsigned long long c; CK(hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost)); if (c < best) best = c;
            }
            printf("%4d threads  %-46s %6.2f ticks per step\n", threads, names[mode], (double)best / N);
        }
    }
    return 0;
}
