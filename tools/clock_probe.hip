// Calibration: cost of a dependent v_add_f32 chain / LDS read chain for ONE wave, in s_memtime ticks and ns.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_addchain(float* out, unsigned long long* ticks, int n, float a) {
    float s = out[0];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 8) { s = s + a; s = s + a; s = s + a; s = s + a; s = s + a; s = s + a; s = s + a; s = s + a; }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
__global__ void k_ldschain(float* out, unsigned long long* ticks, int n) {
    __shared__ int idx[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) idx[i] = (i + 17) & 1023;
    __syncthreads();
    int p = threadIdx.x & 1023;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) p = idx[p];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = (float)p;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
__global__ void k_empty() {}
int main() {
    float* d; unsigned long long* t; CK(hipMalloc(&d, 4096)); CK(hipMalloc(&t, 8 * 1024)); CK(hipMemset(d, 0, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n = 1 << 20;
    for (int rep = 0; rep < 3; ++rep) {
        for (int blocks : {1, 256}) {
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_addchain, blocks, 64, 0, 0, d, t, n, 1e-9f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); unsigned long long h; CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
            printf("addchain blocks=%3d: %.3f ms, %llu ticks -> %.3f ticks/add, %.3f ns/add, %.3f ticks/ns\n", blocks, ms, h, (double)h / n, ms * 1e6 / n, h / (ms * 1e6));
        }
    }
    const int m = 1 << 16;
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_ldschain, 1, 64, 0, 0, d, t, m); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); unsigned long long h; CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
    printf("ldschain (1 wave): %.3f ms, %.1f ticks/read, %.1f ns/read\n", ms, (double)h / m, ms * 1e6 / m);
    // short bursts separated by idle gaps, like the decode loop: does the clock drop?
    for (int gap_us : {0, 50, 500}) {
        double tot = 0; 
        for (int i = 0; i < 20; ++i) {
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_addchain, 16, 64, 0, 0, d, t, 1 << 14, 1e-9f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
            if (gap_us) { struct timespec ts = {0, gap_us * 1000}; nanosleep(&ts, nullptr); }
        }
        CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
        printf("burst 16 blocks x 16k adds, host gap %3d us: %.2f us per launch, %.3f ticks/add\n", gap_us, tot / 20 * 1e3, (double)h / (1 << 14));
    }
    return 0;
}
