// Calibration of the LDS access patterns the attention kernels use (gfx950), in shader cycles per term, for 1..N waves of
// one workgroup:  (a) V accumulation, one b32 value + one broadcast b32 probability per term, 16-term batches (k_attn);
// (b) the same with float4 operands (k_attn_out); (c) the score dot: K row float4 + broadcast query float4 per 4 terms;
// (d) independent back-to-back ds_read_b32 / ds_read_b128 (issue throughput).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int HD = 128, T = 128, KLD = HD + 4;

__global__ void k_vsum_b32(float* out, unsigned long long* ticks, int reps) {
    __shared__ float vbuf[T * HD];
    __shared__ float w[T];
    for (int i = threadIdx.x; i < T * HD; i += blockDim.x) vbuf[i] = 1e-3f * (i & 255);
    for (int i = threadIdx.x; i < T; i += blockDim.x) w[i] = 1e-2f;
    __syncthreads();
    const float* v = vbuf + (threadIdx.x & (HD - 1));
    float o = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r)
        for (int t = 0; t < T; t += 16) {
            float vv[16], ww[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { vv[u] = v[((t + u + r) & (T - 1)) * HD]; ww[u] = w[(t + u + r) & (T - 1)]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) { const float p = ww[u] * vv[u]; o = o + p; }
        }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = o;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
__global__ void k_vsum_b128(float* out, unsigned long long* ticks, int reps) {
    __shared__ __attribute__((aligned(16))) float vt[HD * (T + 4)];     // transposed [element][T + 4]
    __shared__ __attribute__((aligned(16))) float w[T];
    for (int i = threadIdx.x; i < HD * (T + 4); i += blockDim.x) vt[i] = 1e-3f * (i & 255);
    for (int i = threadIdx.x; i < T; i += blockDim.x) w[i] = 1e-2f;
    __syncthreads();
    const v4f* v = (const v4f*)(vt + (threadIdx.x & (HD - 1)) * (T + 4));
    const v4f* w4 = (const v4f*)w;
    float o = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r)
        for (int q = 0; q < T / 4; q += 4) {
            v4f vv[4], ww[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { vv[u] = v[(q + u + r) & (T / 4 - 1)]; ww[u] = w4[(q + u + r) & (T / 4 - 1)]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float p = ww[u].x * vv[u].x; o = o + p;
                p = ww[u].y * vv[u].y; o = o + p;
                p = ww[u].z * vv[u].z; o = o + p;
                p = ww[u].w * vv[u].w; o = o + p;
            }
        }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = o;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
__global__ void k_dot(float* out, unsigned long long* ticks, int reps) {
    __shared__ __attribute__((aligned(16))) float kbuf[T * KLD];
    __shared__ __attribute__((aligned(16))) float q_s[HD];
    for (int i = threadIdx.x; i < T * KLD; i += blockDim.x) kbuf[i] = 1e-3f * (i & 127);
    for (int i = threadIdx.x; i < HD; i += blockDim.x) q_s[i] = 1e-2f;
    __syncthreads();
    const v4f* q4 = (const v4f*)q_s;
    float acc = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        const v4f* k4 = (const v4f*)(kbuf + ((threadIdx.x + r) & (T - 1)) * KLD);
        float dot = -0.0f;
        for (int i = 0; i < HD / 4; i += 16) {
            v4f kk[16], qq[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { kk[u] = k4[i + u]; qq[u] = q4[i + u]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                float p = qq[u].x * kk[u].x; dot = dot + p;
                p = qq[u].y * kk[u].y; dot = dot + p;
                p = qq[u].z * kk[u].z; dot = dot + p;
                p = qq[u].w * kk[u].w; dot = dot + p;
            }
        }
        acc += dot;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
template <int W>
__global__ void k_reads(float* out, unsigned long long* ticks, int reps) {
    __shared__ __attribute__((aligned(16))) float buf[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) buf[i] = 1e-3f * (i & 255);
    __syncthreads();
    float acc = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (W == 1) {
            float x[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) x[u] = buf[((threadIdx.x & 63) + 64 * u + 7 * r) & 16383];
#pragma unroll
            for (int u = 0; u < 32; ++u) acc += x[u];
        } else {
            v4f x[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) x[u] = ((const v4f*)buf)[((threadIdx.x & 63) + 64 * u + 7 * r) & 4095];
#pragma unroll
            for (int u = 0; u < 32; ++u) acc += x[u].x + x[u].w;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
int main() {
    float* d; unsigned long long* t; CK(hipMalloc(&d, 4096 * 4)); CK(hipMalloc(&t, 64));
    const int reps = 256;
    for (int threads : {64, 128, 256}) {
        unsigned long long h;
        hipLaunchKernelGGL(k_vsum_b32, 1, threads, 0, 0, d, t, reps); CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
        printf("threads %3d  V-sum b32 operands   : %6.2f cycles/term\n", threads, (double)h / (reps * T));
        hipLaunchKernelGGL(k_vsum_b128, 1, threads, 0, 0, d, t, reps); CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
        printf("threads %3d  V-sum float4 operands: %6.2f cycles/term\n", threads, (double)h / (reps * T));
        hipLaunchKernelGGL(k_dot, 1, threads, 0, 0, d, t, reps); CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
        printf("threads %3d  score dot (K + q float4): %6.2f cycles/term\n", threads, (double)h / (reps * HD));
        hipLaunchKernelGGL(k_reads<1>, 1, threads, 0, 0, d, t, reps); CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
        printf("threads %3d  32 independent ds_read_b32 + sum : %6.2f cycles/read\n", threads, (double)h / (reps * 32));
        hipLaunchKernelGGL(k_reads<4>, 1, threads, 0, 0, d, t, reps); CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
        printf("threads %3d  32 independent ds_read_b128 + sum: %6.2f cycles/read\n", threads, (double)h / (reps * 32));
    }
    return 0;
}
