// Developer probe (round 5): what the kernel-argument round trip at kernel entry costs per dependent launch, and whether
// gfx950's kernarg preload (user SGPRs filled at wave launch; -mllvm -amdgpu-kernarg-preload-count=N) removes it.
// A chain of dependent launches, each the skeleton of the engine's layer kernels: every workgroup reads the whole 4 KB
// activation the previous launch wrote, reduces it, and writes its own 4 floats of the next activation.
//   hipcc --offload-arch=gfx950 -O3 -o tools/kernarg_probe tools/kernarg_probe.hip
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=8 -o tools/kernarg_probe_pl tools/kernarg_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
struct Args { const float* in; float* out; int n; int pad[61]; };     // 264 bytes like GemvArgs
__device__ __forceinline__ void body(const float* in, float* out, unsigned long long* t, int extra) {
    __shared__ float red[4];
    const float4 v = ((const float4*)in)[threadIdx.x];
    float s = (v.x + v.y) + (v.z + v.w);
    for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < 4) out[blockIdx.x * 4 + threadIdx.x] = (red[0] + red[1] + red[2] + red[3]) * 1e-3f + threadIdx.x + extra;
    if (t != nullptr && blockIdx.x == 0 && threadIdx.x == 0) t[0] = __builtin_amdgcn_s_memrealtime();
}
__global__ __launch_bounds__(256) void k_ptr(const float* in, float* out, int n) { body(in, out, nullptr, n); }
__global__ __launch_bounds__(256) void k_struct(const Args a) { body(a.in, a.out, nullptr, a.pad[60]); }
// hybrid: the hot pointers as leading scalar arguments (preloadable), the rest of the 264-byte block behind them
__global__ __launch_bounds__(256) void k_hybrid(const float* in, float* out, const Args a) { body(in, out, nullptr, a.pad[60]); }

template <class F> double run_chain(hipStream_t s, int n, int reps, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) launch(i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps / n;
        if (us < best) best = us;
    }
    return best;
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float* d[2]; CK(hipMalloc(&d[0], 4096)); CK(hipMalloc(&d[1], 4096)); CK(hipMemset(d[0], 0, 4096)); CK(hipMemset(d[1], 0, 4096));
    const int N = 140, R = 40;
    for (int grid : {16, 256, 512}) {
        const double a = run_chain(s, N, R, [&](int i) { hipLaunchKernelGGL(k_ptr, grid, 256, 0, s, (const float*)d[i & 1], d[(i + 1) & 1], i); });
        Args b; b.n = 0; for (int k = 0; k < 61; ++k) b.pad[k] = k;
        const double c = run_chain(s, N, R, [&](int i) { b.in = d[i & 1]; b.out = d[(i + 1) & 1]; hipLaunchKernelGGL(k_struct, grid, 256, 0, s, b); });
        const double h = run_chain(s, N, R, [&](int i) { b.in = d[i & 1]; b.out = d[(i + 1) & 1]; hipLaunchKernelGGL(k_hybrid, grid, 256, 0, s, (const float*)d[i & 1], d[(i + 1) & 1], b); });
        printf("grid %3d x 256: pointer args %.3f us per hop   264-byte struct %.3f   pointers + struct %.3f\n", grid, a, c, h);
    }
    return 0;
}
