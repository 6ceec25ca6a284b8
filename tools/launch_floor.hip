// Micro-benchmark: period of a dependent chain of small kernels on one stream (eager vs hipGraph).
// Calibrates what a kernel boundary costs on this box so kernel-level numbers can be read against it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_empty() {}
__global__ void k_touch(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = p[1] + 1.0f; }
struct Big { float* p; int pad[60]; };
__global__ void k_bigarg(Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) b.p[0] = b.p[1] + (float)b.pad[59]; }
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ w, float* out, int n4_per_block) {
    extern __shared__ float sm[];
    float acc = 0.f;
    const float4* base = w + (size_t)blockIdx.x * n4_per_block;
    for (int i = threadIdx.x; i < n4_per_block; i += 256) { float4 v = base[i]; acc += v.x + v.y + v.z + v.w; }
    sm[threadIdx.x] = acc; __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = sm[0] + sm[64];
}

template <class F> double run_chain(hipStream_t s, int n, int reps, bool graph, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    if (graph) {
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < n; ++i) launch(i);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    } else { for (int i = 0; i < n; ++i) launch(i); CK(hipStreamSynchronize(s)); }
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) { if (graph) CK(hipGraphLaunch(ge, s)); else for (int i = 0; i < n; ++i) launch(i); }
    CK(hipStreamSynchronize(s));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return us / reps / n;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float* d; CK(hipMalloc(&d, 1 << 20));
    const size_t wbytes = 512ull << 20; float4* w; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 0, wbytes));
    Big b; b.p = d; for (int i = 0; i < 60; ++i) b.pad[i] = i;
    const int N = 142, R = 50;
    for (int graph = 0; graph < 2; ++graph) {
        printf("--- %s, chain of %d kernels, us per kernel\n", graph ? "hipGraph" : "eager", N);
        printf("empty<<<1,64>>>            %.2f\n", run_chain(s, N, R, graph, [&](int) { hipLaunchKernelGGL(k_empty, 1, 64, 0, s); }));
        printf("empty<<<256,256>>>         %.2f\n", run_chain(s, N, R, graph, [&](int) { hipLaunchKernelGGL(k_empty, 256, 256, 0, s); }));
        printf("empty<<<256,256,16KB lds>>> %.2f\n", run_chain(s, N, R, graph, [&](int) { hipLaunchKernelGGL(k_empty, 256, 256, 16384, s); }));
        printf("empty<<<1024,256>>>        %.2f\n", run_chain(s, N, R, graph, [&](int) { hipLaunchKernelGGL(k_empty, 1024, 256, 0, s); }));
        printf("touch<<<1,64>>>            %.2f\n", run_chain(s, N, R, graph, [&](int) { hipLaunchKernelGGL(k_touch, 1, 64, 0, s, d); }));
        printf("bigarg(248B)<<<256,256>>>  %.2f\n", run_chain(s, N, R, graph, [&](int) { hipLaunchKernelGGL(k_bigarg, 256, 256, 0, s, b); }));
        for (int mb : {1, 4, 16, 64}) {
            const int n4 = (mb << 20) / 16 / 256;   // float4 per block, 256 blocks
            printf("stream %2d MB <<<256,256>>>  %.2f  (distinct slices per launch)\n", mb,
                   run_chain(s, N, R, graph, [&](int i) { hipLaunchKernelGGL(k_stream, 256, 256, 1024, s, w + (size_t)(i % 8) * (mb << 20) / 16, d + 1024, n4); }));
        }
    }
    return 0;
}
