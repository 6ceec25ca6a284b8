// mall_probe.hip -- does a buffer streamed by one kernel stay in the 256 MiB Infinity Cache (MALL) for the next one?
// For sizes S: (a) cold read of S bytes (evicted first by streaming 1 GiB of other data), (b) the same read again right
// after a "prefetch" pass over the same bytes by a small kernel (plain loads / nt loads), (c) concurrent prefetch on a
// second stream while a latency-bound foreground chain runs.  Build: hipcc --offload-arch=gfx950 -O3 -o mall_probe mall_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void k_read(const v4i* __restrict__ p, size_t n16, int* sink) {
    v4i acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        v4i a, b, c, d;
        if (NT) { a = __builtin_nontemporal_load(p + i); b = __builtin_nontemporal_load(p + i + stride); c = __builtin_nontemporal_load(p + i + 2 * stride); d = __builtin_nontemporal_load(p + i + 3 * stride); }
        else { a = p[i]; b = p[i + stride]; c = p[i + 2 * stride]; d = p[i + 3 * stride]; }
        acc += a + b + c + d;
    }
    for (; i < n16; i += stride) acc += NT ? __builtin_nontemporal_load(p + i) : p[i];
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678) *sink = 1;
}

int main() {
    const size_t big = 1ull << 30;
    char *buf, *evict; int* sink;
    CK(hipMalloc(&buf, big)); CK(hipMalloc(&evict, big)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, big)); CK(hipMemset(evict, 2, big));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](auto fn) { CK(hipEventRecord(e0, 0)); fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f; };
    auto flush = [&]() { k_read<0><<<2048, 256>>>((const v4i*)evict, big / 16, sink); CK(hipDeviceSynchronize()); };
    const size_t sizes[] = {4u << 20, 16u << 20, 32u << 20, 64u << 20, 128u << 20, 192u << 20, 256u << 20, 512u << 20};
    printf("%8s %10s %10s %10s %10s %10s   (us; GB/s in parentheses)\n", "MB", "cold_nt", "warm_nt", "warm_plain", "cold_plain", "nt_after_nt");
    for (size_t S : sizes) {
        const size_t n16 = S / 16;
        const int grid = 2048;
        float r[5];
        flush(); r[0] = timed([&] { k_read<1><<<grid, 256>>>((const v4i*)buf, n16, sink); });
        flush(); k_read<0><<<grid, 256>>>((const v4i*)buf, n16, sink); CK(hipDeviceSynchronize());
        r[1] = timed([&] { k_read<1><<<grid, 256>>>((const v4i*)buf, n16, sink); });
        flush(); k_read<0><<<grid, 256>>>((const v4i*)buf, n16, sink); CK(hipDeviceSynchronize());
        r[2] = timed([&] { k_read<0><<<grid, 256>>>((const v4i*)buf, n16, sink); });
        flush(); r[3] = timed([&] { k_read<0><<<grid, 256>>>((const v4i*)buf, n16, sink); });
        flush(); k_read<1><<<grid, 256>>>((const v4i*)buf, n16, sink); CK(hipDeviceSynchronize());
        r[4] = timed([&] { k_read<1><<<grid, 256>>>((const v4i*)buf, n16, sink); });
        printf("%8zu", S >> 20);
        for (int k = 0; k < 5; ++k) printf(" %6.1f(%5.0f)", r[k], S / (r[k] * 1e-6) / 1e9);
        printf("\n");
    }
    // few-CU prefetcher: how fast can 16 / 32 / 64 workgroups pull data into the cache?
    for (int g : {8, 16, 32, 64, 128}) {
        flush();
        const size_t S = 64u << 20;
        float t = timed([&] { k_read<0><<<g, 256>>>((const v4i*)buf, S / 16, sink); });
        flush(); k_read<0><<<g, 256>>>((const v4i*)buf, S / 16, sink); CK(hipDeviceSynchronize());
        float t2 = timed([&] { k_read<1><<<2048, 256>>>((const v4i*)buf, S / 16, sink); });
        printf("prefetch 64 MB with %3d workgroups: %.1f us (%.0f GB/s); full-chip nt read afterwards %.1f us (%.0f GB/s)\n", g, t, S / (t * 1e-6) / 1e9, t2, S / (t2 * 1e-6) / 1e9);
    }
    return 0;
}
