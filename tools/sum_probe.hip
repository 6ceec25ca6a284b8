// Developer probe: shader-cycle cost of the cross-lane primitives the exact sums are built from, and of the whole
// exact sequential sum on realistic data.  One 256-thread workgroup (one wave per SIMD), like the engine's kernels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/sum_probe tools/sum_probe.hip && tools/sum_probe
#include "../qwen3-rs_amd/csrc/q3_kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
using namespace q3;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

// mode: which primitive; every primitive is applied 64 times in a dependent chain
__global__ __launch_bounds__(256) void k_prim(float* out, unsigned long long* cyc, const float* in, int mode) {
    float v = in[threadIdx.x], t = in[256 + threadIdx.x];
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = now();
    if (mode == 0) { for (int i = 0; i < 64; ++i) v = v + t; }
    else if (mode == 1) { for (int i = 0; i < 64; ++i) v = wave_scan_incl(v) * 0.25f; }
    else if (mode == 2) { for (int i = 0; i < 64; ++i) v = wave_prev_lane(v) + t; }
    else if (mode == 3) { for (int i = 0; i < 64; ++i) v = dpp_f<0x114>(v) + t; }
    else if (mode == 4) { for (int i = 0; i < 64; ++i) v = v + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), i)); }
    else if (mode == 5) { for (int i = 0; i < 64; ++i) v = dpp_f<0x142>(v) + t; }
    else if (mode == 6) { for (int i = 0; i < 64; ++i) v = __shfl_up(v, 1) + t; }
    else if (mode == 7) { for (int i = 0; i < 64; ++i) { const unsigned long long b = __ballot(v > t); v = v + (float)__builtin_ctzll(b | 0x8000000000000000ull); } }
    else if (mode == 8) { for (int i = 0; i < 64; ++i) v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)) + t; }
    const unsigned long long t1 = now();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// the plain iteration (round r fixes block r), for comparison with spec_sum_lanes
template <class Fold>
__device__ __forceinline__ float iter_sum_lanes(float tot, int nblk, Fold fold, int* rounds) {
    const int j = threadIdx.x & 63;
    const bool live = j < nblk;
    if (!live) tot = 0.0f;
    float g = wave_prev_lane(wave_scan_incl(tot));
    if (j == 0) g = -0.0f;
    float out = fold(g);
    int r = 0;
    for (int round = 0; round < 65; ++round) {
        float e = wave_prev_lane(out) - g;
        if (j == 0 || !live) e = 0.0f;
        e = wave_scan_incl(e);
        float sc = g + e;
        if (j == 0) sc = -0.0f;
        const float out2 = fold(sc);
        const float prev = wave_prev_lane(out2);
        const bool ok = (j == 0) || !live || (__float_as_uint(prev) == __float_as_uint(sc));
        g = sc; out = out2; ++r;
        if (__all(ok)) break;
    }
    *rounds = r;
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(out), nblk - 1));
}

// variant: 0 = engine's seq_sum_terms (candidate scheme), 1 = plain iteration with blocks of B4*4 terms in registers
template <int B4>
__global__ __launch_bounds__(256) void k_sum(float* out, unsigned long long* cyc, int* rounds, const float* x, int n, int variant) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sq = (float*)smem;
    const int tid = threadIdx.x;
    if (variant == 0) { for (int i = tid; i < n; i += 256) sq[term_index(i, n)] = x[i] * x[i]; }
    else { for (int i = tid; i < n; i += 256) sq[i] = x[i] * x[i]; }
    __syncthreads();
    const unsigned long long t0 = now();
    float s; int r = 0;
    if (variant == 0) s = seq_sum_terms(sq, n);
    else {
        const int lane = tid & 63, nblk = n / (4 * B4);
        v4f rr[B4];
        const v4f* blk = (const v4f*)(sq + (lane < nblk ? lane : 0) * 4 * B4);
#pragma unroll
        for (int k = 0; k < B4; ++k) rr[k] = blk[k];
        float p = 0.f;
#pragma unroll
        for (int k = 0; k < B4; ++k) p += (rr[k].x + rr[k].y) + (rr[k].z + rr[k].w);
        s = iter_sum_lanes(p, nblk, [&](float a) {
#pragma unroll
            for (int k = 0; k < B4; ++k) a = chain4(a, rr[k]);
            return a; }, &r);
    }
    const unsigned long long t1 = now();
    out[tid] = s;
    if (tid == 0) { cyc[0] = t1 - t0; rounds[0] = r; }
}

// Round 6: the exact sum on FOUR waves.  Wave w owns blocks 64 w .. 64 w + 63 of 16 terms (lane = block, terms in
// registers).  Every wave runs the guess -> correct -> verify loop on its own quarter from a GUESSED input -- no barrier inside the
// loop -- ; wave 0's input (-0.0) is exact, so its result is.  Wave w > 0 then takes wave w - 1's exact output from LDS (flag), and
// if it differs from its guess translates its inputs by the difference and re-runs its loop.  Exact by the same argument as
// spec_sum_lanes: every link inside a wave is verified bitwise, and every wave's input is the verified output of the one before.
__device__ __forceinline__ float wave4_exact_sum16(const v4f (&r)[4], int nblk, float* lds, int* rounds_out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* lds_tot = lds;                 // [4] approximate wave totals
    volatile float* lds_val = lds + 8;    // [4] exact output of wave w
    volatile int* lds_flag = (volatile int*)(lds + 16);   // [4]
    const int b = 64 * wave + lane;
    const bool live = b < nblk;
    const int nlive = min(max(nblk - 64 * wave, 0), 64);         // live lanes of this wave (wave-uniform)
    float tot = 0.0f;
    if (live) tot = ((r[0].x + r[0].y) + (r[0].z + r[0].w)) + ((r[1].x + r[1].y) + (r[1].z + r[1].w)) +
                    (((r[2].x + r[2].y) + (r[2].z + r[2].w)) + ((r[3].x + r[3].y) + (r[3].z + r[3].w)));
    const float inc = wave_scan_incl(tot);
    if (lane == 63) lds_tot[wave] = inc;
    if (tid < 4) lds_flag[tid] = 0;
    __syncthreads();
    float off = 0.0f;
    for (int w = 0; w < wave; ++w) off += lds_tot[w];
    float sc = wave_prev_lane(inc) + off;                        // approximate running sum in front of block b
    if (b == 0) sc = -0.0f;
    int nr = 0;
    auto fold = [&](float s0) { s0 = chain4(s0, r[0]); s0 = chain4(s0, r[1]); s0 = chain4(s0, r[2]); s0 = chain4(s0, r[3]); return s0; };
    float out = 0.0f;
    auto converge = [&]() {                                      // lane 0's input is taken as given
        for (int round = 0; round < 66; ++round) {
            out = fold(sc);
            ++nr;
            const float prev = wave_prev_lane(out);
            const bool ok = lane == 0 || !live || (__float_as_uint(prev) == __float_as_uint(sc));
            if (__all(ok)) break;
            float e = prev - sc;
            if (lane == 0 || !live) e = 0.0f;
            sc = sc + wave_scan_incl(e);
        }
    };
    if (nlive > 0) converge();
    float in0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc), 0));
    if (wave > 0) {
        while (lds_flag[wave - 1] == 0) __builtin_amdgcn_s_sleep(1);
        const float exact_in = lds_val[wave - 1];
        if (nlive > 0 && __float_as_uint(exact_in) != __float_as_uint(in0)) {
            const float delta = exact_in - in0;
            sc = lane == 0 ? exact_in : sc + delta;
            converge();
        }
        in0 = exact_in;
    }
    const float res = nlive > 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(out), nlive > 0 ? nlive - 1 : 0)) : in0;
    if (lane == 0) { lds_val[wave] = res; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); lds_flag[wave] = 1; }
    if (rounds_out && wave == 3 && lane == 0) *rounds_out = nr;
    while (lds_flag[3] == 0) __builtin_amdgcn_s_sleep(1);
    return lds_val[3];
}
__global__ __launch_bounds__(256) void k_sum4(float* out, unsigned long long* cyc, int* rounds, const float* x, int n, int variant) {
    __shared__ float lds[32];
    const int tid = threadIdx.x;
    const int nblk = n / 16;
    v4f r[4];
    const v4f* bp = (const v4f*)x + 4 * min(tid, nblk - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) { v4f q = bp[k]; q.x = q.x * q.x; q.y = q.y * q.y; q.z = q.z * q.z; q.w = q.w * q.w; r[k] = q; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t0 = now();
    const float s = wave4_exact_sum16(r, nblk, lds, rounds);
    const unsigned long long t1 = now();
    out[tid] = s;
    if (tid == 0) cyc[0] = t1 - t0;
}

// The FLAT four-wave form: one guess -> correct -> verify loop over all 256 blocks, every round exchanging across the waves through
// LDS (the last lane's output for the link check, then the waves' error totals and verdicts for the corrected inputs): two
// workgroup barriers per round.
__device__ __forceinline__ float wave4_flat_exact_sum16(const v4f (&r)[4], int nblk, float* lds, int* rounds_out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* lds_tot = lds;            // [4]
    float* lds_out = lds + 4;        // [4] output of each wave's last lane
    float* lds_e = lds + 8;          // [4] each wave's error total
    int* lds_ok = (int*)(lds + 12);  // [4]
    const int b = tid;
    const bool live = b < nblk;
    float tot = 0.0f;
    if (live) tot = ((r[0].x + r[0].y) + (r[0].z + r[0].w)) + ((r[1].x + r[1].y) + (r[1].z + r[1].w)) +
                    (((r[2].x + r[2].y) + (r[2].z + r[2].w)) + ((r[3].x + r[3].y) + (r[3].z + r[3].w)));
    const float inc = wave_scan_incl(tot);
    if (lane == 63) lds_tot[wave] = inc;
    __syncthreads();
    float off = 0.0f;
    for (int w = 0; w < wave; ++w) off += lds_tot[w];
    float sc = wave_prev_lane(inc) + off;
    if (b == 0) sc = -0.0f;
    float out = 0.0f;
    int nr = 0;
    for (int round = 0; round < 260; ++round) {
        out = chain4(chain4(chain4(chain4(sc, r[0]), r[1]), r[2]), r[3]);
        ++nr;
        if (lane == 63) lds_out[wave] = out;
        __syncthreads();
        float prev = wave_prev_lane(out);
        if (lane == 0 && wave > 0) prev = lds_out[wave - 1];
        const bool ok = b == 0 || !live || (__float_as_uint(prev) == __float_as_uint(sc));
        float e = prev - sc;
        if (b == 0 || !live) e = 0.0f;
        const float es = wave_scan_incl(e);
        const int wok = __all(ok) ? 1 : 0;
        if (lane == 63) { lds_e[wave] = es; lds_ok[wave] = wok; }
        __syncthreads();
        if (lds_ok[0] & lds_ok[1] & lds_ok[2] & lds_ok[3]) break;
        float eoff = 0.0f;
        for (int w = 0; w < wave; ++w) eoff += lds_e[w];
        sc = sc + (es + eoff);
        if (b == 0) sc = -0.0f;
    }
    if (tid == nblk - 1) lds_tot[0] = out;
    if (rounds_out && tid == 0) *rounds_out = nr;
    __syncthreads();
    return lds_tot[0];
}
__global__ __launch_bounds__(256) void k_sum4f(float* out, unsigned long long* cyc, int* rounds, const float* x, int n, int variant) {
    __shared__ float lds[32];
    const int tid = threadIdx.x;
    const int nblk = n / 16;
    v4f r[4];
    const v4f* bp = (const v4f*)x + 4 * min(tid, nblk - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) { v4f q = bp[k]; q.x = q.x * q.x; q.y = q.y * q.y; q.z = q.z * q.z; q.w = q.w * q.w; r[k] = q; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t0 = now();
    const float s = wave4_flat_exact_sum16(r, nblk, lds, rounds);
    const unsigned long long t1 = now();
    out[tid] = s;
    if (tid == 0) cyc[0] = t1 - t0;
}

int main() {
    float *din, *dout; unsigned long long* dcyc; int* drounds;
    const int n = 1024;
    CK(hipMalloc(&din, 4 * 16384)); CK(hipMalloc(&dout, 4 * 256)); CK(hipMalloc(&dcyc, 8)); CK(hipMalloc(&drounds, 4));
    std::vector<float> h(16384);
    srand(1);
    auto gauss = []() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return (float)(sqrt(-2 * log(u)) * cos(6.283185307 * v)); };
    const char* names[] = {"v_add chain", "wave_scan_incl (6 dpp adds)+mul", "wave_prev_lane (wave_shr:1)+add", "row_shr:4 dpp + add", "v_readlane(const)+add", "row_bcast:15 dpp + add", "__shfl_up(1)+add (ds_bpermute)", "ballot+ctz+cvt+add", "readlane(v,63)+add (dependent)"};
    for (int i = 0; i < 16384; ++i) h[i] = gauss();
    CK(hipMemcpy(din, h.data(), 4 * 16384, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 9; ++mode) {
        unsigned long long c = 0;
        for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k_prim, 1, 256, 0, 0, dout, dcyc, din, mode); CK(hipDeviceSynchronize()); }
        CK(hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost));
        printf("%-36s %7.1f cycles per step\n", names[mode], c / 64.0);
    }
    for (int nn : {1024, 2560, 4096}) {
        for (int trial = 0; trial < 12; ++trial) {
            const int tk = trial & 3;
            const float scale = tk == 0 ? 1.0f : (tk == 1 ? 0.05f : (tk == 2 ? 7.0f : 1.0f));
            for (int i = 0; i < nn; ++i) h[i] = gauss() * scale * (tk == 3 ? expf(gauss()) : 1.0f);
            CK(hipMemcpy(din, h.data(), 4 * nn, hipMemcpyHostToDevice));
            float ref = -0.0f; for (int i = 0; i < nn; ++i) { float q = h[i] * h[i]; ref = ref + q; }
            auto run = [&](const char* what, auto kern, int variant) {
                unsigned long long c = ~0ull; int r = 0; float s = 0;
                for (int rep = 0; rep < 12; ++rep) {                 // minimum of 12: the clocks of an idle part wander
                    unsigned long long c1 = 0;
                    hipLaunchKernelGGL(kern, 1, 256, 4 * (nn + 1024), 0, dout, dcyc, drounds, din, nn, variant); CK(hipDeviceSynchronize());
                    CK(hipMemcpy(&c1, dcyc, 8, hipMemcpyDeviceToHost));
                    c = c1 < c ? c1 : c;
                } CK(hipMemcpy(&r, drounds, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&s, dout, 4, hipMemcpyDeviceToHost));
                printf("n=%5d trial %d  %-34s %6llu cycles  rounds %2d  %s\n", nn, trial, what, c, r, s == ref ? "exact" : "WRONG");
            };
            run("engine seq_sum_terms", k_sum<4>, 0);
            if (nn == 1024) { run("plain iteration, 64 x 16 terms", k_sum<4>, 1); run("plain iteration, 32 x 32 terms", k_sum<8>, 1); run("plain iteration, 16 x 64 terms", k_sum<16>, 1); }
            if (nn == 4096) { run("plain iteration, 64 x 64 terms", k_sum<16>, 1); }
            if (nn != 1024) run("4 waves flat, 2 barriers per round", k_sum4f, 0);
            if (nn != 1024) run("4 waves x 64 lanes x 16 terms", k_sum4, 0);
            else run("4 waves, 64 blocks of 16 (1 wave live)", k_sum4, 0);
        }
    }
    return 0;
}
