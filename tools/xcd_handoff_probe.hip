// xcd_handoff_probe.hip -- VERDICT r3 item 4(a): what does a per-kv-head, XCD-local hand-off inside ONE launch cost, against the
// kernel boundary it would replace (QKV -> attention of the 0.6B shape: 8 kv-head groups of 512 rows, 256 workgroups)?
//
// A "layer hop" = producer phase (256 workgroups x 512 threads; workgroup b streams 16 weight rows of 1 KiB and writes 16
// floats of its group's 512-float vector) followed by a consumer phase (2 workgroups per group read the group's 512 floats
// and write 2 x 128 floats).  Group of a workgroup = its XCC id (s_getreg HW_REG_XCC_ID), work inside a group handed out by a
// per-group ticket, so the protocol does not DEPEND on the block -> XCD placement for correctness of the probe's bookkeeping
// (the data path below does: it is only valid when producers and consumers of a group share an L2).
//   A  two launches per hop (producer kernel, consumer kernel): the engine's form, boundary between them
//   B  one launch per hop: consumers = the first two workgroups of a group to finish their rows; they wait for the
//      group's arrival counter (32 arrivals), XCD-local protocol: producers plain stores -> s_waitcnt vmcnt(0) -> L2 atomic
//      (workgroup scope: performed in the XCD's L2, no write-back); consumers poll with sc1 loads (L1 bypass, L2-served),
//      read the vector with sc1 loads
//   C  as B with the agent-scope protocol of the guide (release fence + relaxed agent atomic; acquire fence; plain loads)
// Chain of NH hops captured in a hipGraph; reports us per hop and the placement census.
// Build: hipcc --offload-arch=gfx950 -O3 -o xcd_handoff_probe xcd_handoff_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kGroups = 8, kWGs = 256, kRowsPerWG = 16, kN = 1024;

struct Ctl {                       // per hop
    unsigned ticket[kGroups];      // work hand-out inside a group (which 16-row slab a workgroup takes)
    unsigned arrived[kGroups];     // producers of the group done
    unsigned err;
    unsigned pad[15];
};

__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 0xf); }   // HW_REG_XCC_ID[3:0]

// MODE 0: producer only (launch pair A).  MODE 1: fused, XCD-local protocol.  MODE 2: fused, agent-scope protocol.
// MODE 3: as 1 with the static map group = blockIdx % 8 (no hand-out ticket; valid only under the observed round-robin placement).
template <int MODE>
__global__ __launch_bounds__(512) void k_produce(const v4i* __restrict__ w, const float* xin, float* vec, float* out, Ctl* ctl, unsigned* census) {
    __shared__ unsigned s_slab;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int grp = (MODE == 0 || MODE == 3) ? (int)(blockIdx.x & 7) : (xcc_id() & 7);
    if (threadIdx.x == 0) {
        s_slab = (MODE == 0 || MODE == 3) ? (blockIdx.x >> 3) : atomicAdd(&ctl->ticket[grp], 1u);
        if (MODE == 3 && (xcc_id() & 7) != grp) atomicAdd(&ctl->err, 1u);      // placement differs from block % 8
        if (census) atomicAdd(&census[grp], 1u);
    }
    __syncthreads();
    const unsigned slab = s_slab;                      // 0..31 when the placement is even
    if (slab >= 32u) { if (threadIdx.x == 0) atomicAdd(&ctl->err, 1u); return; }      // uneven placement: the probe gives up on this workgroup
    // 16 rows of 1 KiB: wave w takes rows 2w, 2w+1
    const int row0 = (grp * 32 + (int)slab) * kRowsPerWG + 2 * wave;
    const v4i* wr = w + (size_t)row0 * (kN / 16);
    const v4i w0 = __builtin_nontemporal_load(wr + lane), w1 = __builtin_nontemporal_load(wr + 64 + lane);
    const v4i x0 = ((const v4i*)xin)[lane];
    int d0 = 0, d1 = 0;
    d0 = __builtin_amdgcn_sdot4(w0.x, x0.x, d0, false); d0 = __builtin_amdgcn_sdot4(w0.y, x0.y, d0, false);
    d0 = __builtin_amdgcn_sdot4(w0.z, x0.z, d0, false); d0 = __builtin_amdgcn_sdot4(w0.w, x0.w, d0, false);
    d1 = __builtin_amdgcn_sdot4(w1.x, x0.x, d1, false); d1 = __builtin_amdgcn_sdot4(w1.y, x0.y, d1, false);
    d1 = __builtin_amdgcn_sdot4(w1.z, x0.z, d1, false); d1 = __builtin_amdgcn_sdot4(w1.w, x0.w, d1, false);
    for (int m = 1; m < 64; m <<= 1) { d0 += __shfl_xor(d0, m); d1 += __shfl_xor(d1, m); }
    float* gv = vec + grp * 512 + (int)slab * kRowsPerWG + 2 * wave;
    if (lane == 0) { gv[0] = (float)(d0 & 0xff) * 0.0078125f; gv[1] = (float)(d1 & 0xff) * 0.0078125f; }
    if (MODE == 0) return;
    // ---- arrive
    __shared__ unsigned s_order;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 2) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_order = __hip_atomic_fetch_add(&ctl->arrived[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            s_order = __hip_atomic_fetch_add(&ctl->arrived[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     // performed in this XCD's L2
        }
    }
    __syncthreads();
    const unsigned order = s_order;
    if (order >= 2u) return;                           // the first two finishers of a group become its consumers (one head each)
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        for (;;) {
            unsigned v;
            if (MODE == 2) v = __hip_atomic_load(&ctl->arrived[grp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(&ctl->arrived[grp]) : "memory");
            if (v >= 32u) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 20000u) { atomicAdd(&ctl->err, 1000u); break; }
        }
        if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    // consumer: read the group's 512 floats, write 128
    const float* gsrc = vec + grp * 512;
    float acc = 0.f;
    if (threadIdx.x < 128) {
        for (int k = 0; k < 4; ++k) {
            float v;
            const float* p = gsrc + threadIdx.x + 128 * k;
            if (MODE == 2) v = *p;
            else asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
            acc += v;
        }
        out[(grp * 2 + (int)order) * 128 + threadIdx.x] = acc;
    }
}

__global__ __launch_bounds__(512) void k_consume(const float* vec, float* out) {
    const int grp = blockIdx.x >> 1, order = blockIdx.x & 1;
    if (threadIdx.x < 128) {
        float acc = 0.f;
        for (int k = 0; k < 4; ++k) acc += vec[grp * 512 + threadIdx.x + 128 * k];
        out[(grp * 2 + order) * 128 + threadIdx.x] = acc;
    }
}

int main() {
    const int NH = 140;
    v4i* w; float *xin, *vec, *out, *outA; Ctl* ctl; unsigned* census;
    const size_t wbytes = (size_t)kWGs * kRowsPerWG * kN;
    CK(hipMalloc(&w, wbytes * 8));                     // 8 rotating weight sets (HBM, not cache)
    CK(hipMemset(w, 1, wbytes * 8));
    CK(hipMalloc(&xin, 4096)); CK(hipMemset(xin, 1, 4096));
    CK(hipMalloc(&vec, 4 * 4096 * (NH + 1))); CK(hipMemset(vec, 0, 4 * 4096 * (NH + 1)));
    CK(hipMalloc(&out, 4 * 2048 * (NH + 1))); CK(hipMalloc(&outA, 4 * 2048 * (NH + 1)));
    CK(hipMalloc(&ctl, sizeof(Ctl) * NH)); CK(hipMalloc(&census, 64)); CK(hipMemset(census, 0, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](int mode, float* o) {
        CK(hipMemsetAsync(ctl, 0, sizeof(Ctl) * NH, st));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int h = 0; h < NH; ++h) {
            const v4i* wh = w + (size_t)(h & 7) * (wbytes / 16);
            if (mode == 0) {
                hipLaunchKernelGGL(k_produce<0>, dim3(kWGs), dim3(512), 0, st, wh, xin, vec + 4096 * h, o + 2048 * h, ctl + h, (unsigned*)nullptr);
                hipLaunchKernelGGL(k_consume, dim3(16), dim3(512), 0, st, vec + 4096 * h, o + 2048 * h);
            } else if (mode == 1) {
                hipLaunchKernelGGL(k_produce<1>, dim3(kWGs), dim3(512), 0, st, wh, xin, vec + 4096 * h, o + 2048 * h, ctl + h, h == 0 ? census : (unsigned*)nullptr);
            } else if (mode == 2) {
                hipLaunchKernelGGL(k_produce<2>, dim3(kWGs), dim3(512), 0, st, wh, xin, vec + 4096 * h, o + 2048 * h, ctl + h, (unsigned*)nullptr);
            } else {
                hipLaunchKernelGGL(k_produce<3>, dim3(kWGs), dim3(512), 0, st, wh, xin, vec + 4096 * h, o + 2048 * h, ctl + h, (unsigned*)nullptr);
            }
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemsetAsync(ctl, 0, sizeof(Ctl) * NH, st));
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        std::vector<Ctl> hc(NH);
        CK(hipMemcpy(hc.data(), ctl, sizeof(Ctl) * NH, hipMemcpyDeviceToHost));
        unsigned errs = 0;
        for (auto& c : hc) errs += c.err;
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        printf("  mode %d: %.2f us per hop (best of 6 replays of %d hops), error words %u\n", mode, best * 1e3f / NH, NH, errs);
        return errs;
    };
    printf("xcd_handoff_probe: QKV -> attention hand-off of the 0.6B shape (256 workgroups x 512 threads, 8 groups x 32 workgroups)\n");
    run(0, outA);
    run(1, out);
    std::vector<float> ha(2048 * NH), hb(2048 * NH);
    CK(hipMemcpy(ha.data(), outA, 4 * 2048 * NH, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), out, 4 * 2048 * NH, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < ha.size(); ++i) bad += ha[i] != hb[i];
    printf("  mode 1 outputs differing from mode 0: %zu of %zu (stale or torn reads)\n", bad, ha.size());
    run(2, out);
    CK(hipMemcpy(hb.data(), out, 4 * 2048 * NH, hipMemcpyDeviceToHost));
    bad = 0;
    for (size_t i = 0; i < ha.size(); ++i) bad += ha[i] != hb[i];
    printf("  mode 2 outputs differing from mode 0: %zu of %zu\n", bad, ha.size());
    run(3, out);
    CK(hipMemcpy(hb.data(), out, 4 * 2048 * NH, hipMemcpyDeviceToHost));
    bad = 0;
    for (size_t i = 0; i < ha.size(); ++i) bad += ha[i] != hb[i];
    printf("  mode 3 outputs differing from mode 0: %zu of %zu\n", bad, ha.size());
    unsigned hcens[8];
    CK(hipMemcpy(hcens, census, 32, hipMemcpyDeviceToHost));
    printf("  workgroups per XCC id (hop 0 of mode 1):");
    for (int i = 0; i < 8; ++i) printf(" %u", hcens[i]);
    printf("\n");
    return 0;
}
