// early_consumer_probe.hip -- VERDICT r2 item 1(d): is a dependent GEMV hop cheaper when the consumer kernel is already
// resident (weights in registers) and waits for a device-side signal, instead of waiting for a kernel boundary?
//
// Chain of N "layer GEMV" hops of the 0.6B shape (256 workgroups x 256 threads; each wave streams one 2 KiB weight row,
// reads the 4 KiB activation the previous hop produced, writes one float per row -- the Wo launch of the engine):
//   A  plain dependent launches on one stream (hipGraph), the engine's form;
//   B  hops alternate between two streams (graph branches); hop i is launched when hop i-2 finished, requests its
//      weight tile first, THEN waits until all 256 workgroups of hop i-1 have arrived on a device-scope counter
//      (producer: plain stores -> __syncthreads -> one-lane agent release fence -> relaxed atomic add; consumer: one lane
//      polls relaxed with s_sleep, one agent acquire fence, __syncthreads, plain loads: the guide's recipe);
//   C  as B with write-through (sc1) payload stores and no release fence.
// Every spin is bounded (a timed-out hop sets an error word and carries on), results are checked against A.
// Build: hipcc --offload-arch=gfx950 -O3 -o early_consumer_probe early_consumer_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kRows = 1024, kN = 2048, kWGs = 256;

template <int MODE>   // 0 plain boundary, 1 resident + release/acquire, 2 resident + write-through payload
__global__ __launch_bounds__(256) void k_hop(const v4i* __restrict__ w, const float* xin, float* xout, unsigned* cnt_prev, unsigned* cnt_mine,
                                             unsigned* err) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    // weights first: 2 KiB per row = two 1 KiB wave loads
    const v4i* wr = w + (size_t)row * (kN / 16);
    v4i w0 = __builtin_nontemporal_load(wr + lane), w1 = __builtin_nontemporal_load(wr + 64 + lane);
    if (MODE != 0 && cnt_prev != nullptr) {
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(cnt_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)kWGs) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 4000u) { atomicAdd(err, 1u); break; }      // ~100 us: a consumer whose producer is not running gives up
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    // activation: 2048 "int8" = 512 floats reinterpreted (the probe only needs the dependency and the bytes)
    const v4i* x4 = (const v4i*)xin;
    v4i x0, x1;
    if (MODE == 2) {
        x0 = __builtin_nontemporal_load(x4 + lane); x1 = __builtin_nontemporal_load(x4 + 64 + lane);
    } else { x0 = x4[lane]; x1 = x4[64 + lane]; }
    int d = 0;
    d = __builtin_amdgcn_sdot4(w0.x, x0.x, d, false); d = __builtin_amdgcn_sdot4(w0.y, x0.y, d, false);
    d = __builtin_amdgcn_sdot4(w0.z, x0.z, d, false); d = __builtin_amdgcn_sdot4(w0.w, x0.w, d, false);
    d = __builtin_amdgcn_sdot4(w1.x, x1.x, d, false); d = __builtin_amdgcn_sdot4(w1.y, x1.y, d, false);
    d = __builtin_amdgcn_sdot4(w1.z, x1.z, d, false); d = __builtin_amdgcn_sdot4(w1.w, x1.w, d, false);
    for (int m = 1; m < 64; m <<= 1) d += __shfl_xor(d, m);
    // each row writes 2 floats so that the 1024 rows refill the 2048-byte activation of the next hop (as 512 floats: rows < 256 write)
    if (lane == 0) {
        const float v = (float)(d & 0xff) * 0.0078125f + 1.0f;
        if (MODE == 2) { __hip_atomic_store(xout + row, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else xout[row] = v;
    }
    if (MODE != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (MODE == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __hip_atomic_fetch_add(cnt_mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main() {
    const int hops = 140, reps = 20;
    v4i* w; float *xa, *xb; unsigned *cnt, *err;
    const size_t wbytes = (size_t)kRows * kN;
    CK(hipMalloc(&w, wbytes * 8)); CK(hipMemset(w, 3, wbytes * 8));      // 8 distinct weight sets cycled (nothing stays cached)
    CK(hipMalloc(&xa, 4 * 2048)); CK(hipMalloc(&xb, 4 * 2048)); CK(hipMalloc(&cnt, 4 * (hops + 1))); CK(hipMalloc(&err, 4));
    CK(hipMemset(err, 0, 4));
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<hipEvent_t> ev(hops);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::vector<float> ref(2048), got(2048);
    const char* names[3] = {"A  dependent launches (kernel boundary)            ", "B  resident consumer, release fence + counter      ", "C  resident consumer, write-through payload + ctr  "};
    auto enqueue = [&](int mode) {
        CK(hipMemsetAsync(cnt, 0, 4 * (hops + 1), s0));
        CK(hipMemsetAsync(xa, 0, 4 * 2048, s0));
        if (mode != 0) { CK(hipEventRecord(ev[0], s0)); CK(hipStreamWaitEvent(s1, ev[0], 0)); }
        for (int i = 0; i < hops; ++i) {
            hipStream_t st = (mode == 0 || (i & 1) == 0) ? s0 : s1;
            const v4i* wi = w + (size_t)(i % 8) * (wbytes / 16);
            float* in = (i & 1) ? xb : xa; float* out = (i & 1) ? xa : xb;
            if (mode == 0) hipLaunchKernelGGL(k_hop<0>, dim3(kWGs), dim3(256), 0, st, wi, in, out, nullptr, nullptr, err);
            else if (mode == 1) hipLaunchKernelGGL(k_hop<1>, dim3(kWGs), dim3(256), 0, st, wi, in, out, i ? cnt + i - 1 : nullptr, cnt + i, err);
            else hipLaunchKernelGGL(k_hop<2>, dim3(kWGs), dim3(256), 0, st, wi, in, out, i ? cnt + i - 1 : nullptr, cnt + i, err);
        }
        if (mode != 0) { CK(hipEventRecord(ev[1], s1)); CK(hipStreamWaitEvent(s0, ev[1], 0)); }
    };
    for (int use_graph = 1; use_graph >= 0; --use_graph)
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipMemset(err, 0, 4));
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        if (use_graph) {
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
            enqueue(mode);
            CK(hipStreamEndCapture(s0, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        auto run = [&]() { if (use_graph) CK(hipGraphLaunch(ge, s0)); else enqueue(mode); };
        run();
        CK(hipStreamSynchronize(s0));
        unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        if (herr) {      // the branches did not run side by side: every consumer waited out its bound -- no timing worth reporting
            printf("%s [%s]  consumers timed out %u times in ONE chain: the two branches were not co-scheduled\n", names[mode], use_graph ? "graph" : "eager", herr);
            fflush(stdout);
            if (ge) { CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); }
            continue;
        }
        CK(hipEventRecord(e0, s0));
        for (int r = 0; r < reps; ++r) run();
        CK(hipEventRecord(e1, s0));
        CK(hipEventSynchronize(e1));
        CK(hipStreamSynchronize(s1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(got.data(), (hops & 1) ? xb : xa, 4 * 2048, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        if (mode == 0) ref = got;
        int diff = 0; for (int i = 0; i < 1024; ++i) diff += got[i] != ref[i];
        printf("%s [%s] %7.2f us per hop   (%d hops; consumer timeouts %u; outputs differing from A: %d)\n", names[mode], use_graph ? "graph" : "eager",
               ms * 1e3f / reps / hops, hops, herr, diff);
        fflush(stdout);
        if (ge) { CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); }
    }
    return 0;
}
