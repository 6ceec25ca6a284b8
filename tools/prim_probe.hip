// Developer probe (round 5): cost of the primitives the strict-order chains are built from, measured the amortised way of
// tools/mfma_chain_probe.hip -- 1,024 dependent steps between two s_memtime reads, minimum of 10 launches -- so that the timer's own
// latency (~7 ticks per step in the 64-step loops of tools/sum_probe.hip, profiles/r02_primitive_costs.txt) disappears.
// One workgroup of 64 threads (one wave on one SIMD) and of 256 threads (one wave per SIMD, like the engine's latency-bound kernels).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/prim_probe tools/prim_probe.hip && tools/prim_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int N = 1024;

template <int CTRL, int RM = 0xf, bool BC = true>
__device__ __forceinline__ float dppf(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, RM, 0xf, BC));
}
__device__ __forceinline__ float scan6(float v) {
    v += dppf<0x111>(v); v += dppf<0x112>(v); v += dppf<0x114>(v); v += dppf<0x118>(v);
    v += dppf<0x142, 0xa, false>(v); v += dppf<0x143, 0xc, false>(v);
    return v;
}

template <int mode>
__global__ void k_prim(float* out, unsigned long long* cyc, const float* in) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    __shared__ int idx[1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = in[i & 2047] * 1e-3f;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) idx[i] = (i + 17) & 1023;
    float xs[8];
    for (int i = 0; i < 8; ++i) xs[i] = in[(threadIdx.x + 64 * i) & 1023];
    float a = in[lane], b = in[lane + 64];
    double da = in[lane], db = in[lane + 64] * 1e-3;
    int p = threadIdx.x & 1023;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define LOOP8(...) for (int i = 0; i < N; i += 8) { _Pragma("unroll") for (int u = 0; u < 8; ++u) { __VA_ARGS__; } }
    switch (mode) {
    case 0: LOOP8(a = a + xs[u]); break;                                            // dependent v_add_f32
    case 1: LOOP8(a = dppf<0x111>(a) + xs[u]); break;                               // row_shr:1 hop + add
    case 2: LOOP8(a = dppf<0x114>(a) + xs[u]); break;                               // row_shr:4 hop + add
    case 3: LOOP8(a = dppf<0x138>(a) + xs[u]); break;                               // wave_shr:1 hop + add
    case 4: LOOP8(a = dppf<0x142>(a) + xs[u]); break;                               // row_bcast:15 hop + add
    case 5: LOOP8(a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63)) + xs[u]); break;   // v_readlane(acc) -> SGPR -> add
    case 6: LOOP8(a = a + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs[u]), (i + u) & 63))); break;   // operand by readlane
    case 7: LOOP8(a = scan6(a) * 0.25f); break;                                     // 6-step inclusive wave scan (+ 1 mul)
    case 8: {                                                                       // LDS-fed chain: ds_read_b128 -> 4 adds (hipcc's schedule)
        const v4f* q = (const v4f*)lds;
        for (int i = 0; i < N / 4; ++i) { const v4f v = q[i]; a = a + v.x; a = a + v.y; a = a + v.z; a = a + v.w; }
    } break;
    case 9: {                                                                       // LDS-fed chain, bursts of 8 b128, one lgkmcnt per burst
        const v4f* q = (const v4f*)lds;
        v4f r0[8], r1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r0[k] = q[k];
        for (int i = 0; i < N / 4; i += 16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) r1[k] = q[(i + 8 + k) & 1023];
            __builtin_amdgcn_s_waitcnt(0xC87F); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) { a = a + r0[k].x; a = a + r0[k].y; a = a + r0[k].z; a = a + r0[k].w; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) r0[k] = q[(i + 16 + k) & 1023];
            __builtin_amdgcn_s_waitcnt(0xC87F); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) { a = a + r1[k].x; a = a + r1[k].y; a = a + r1[k].z; a = a + r1[k].w; }
            __builtin_amdgcn_sched_barrier(0);
        }
    } break;
    case 10: LOOP8(a = __shfl_up(a, 1) + xs[u]); break;                             // ds_bpermute hop + add
    case 11: { v2f c = {a, b}; LOOP8(c = c + (v2f){xs[u], xs[7 - u]}); a = c.x + c.y; } break;   // v_pk_add_f32 chain (two lock-step chains)
    case 12: LOOP8(a = a * xs[u]; a = a + xs[7 - u]); break;                        // v_mul + v_add, both dependent
    case 13: LOOP8(da = da + db) ; a = (float)da; break;                           // v_add_f64 chain
    case 14: LOOP8(da = da * db + db); a = (float)da; break;                       // f64 mul + add (contract off)
    case 15: LOOP8(a = xs[u] / a); break;                                           // IEEE f32 division, dependent
    case 16: LOOP8(a = sqrtf(a + xs[u])); break;                                    // correctly rounded sqrt + add
    case 17: for (int i = 0; i < N; ++i) p = idx[p]; a = (float)p; break;          // dependent LDS read (round trip)
    case 18: LOOP8(a = roundf(a * xs[u])); break;                                   // roundf + mul
    case 19: LOOP8(a = fmaxf(a, dppf<0xB1>(a)) + xs[u]); break;                     // quad_perm max + add
    case 20: {                                                                      // two chains through permlane32 swap halves
        LOOP8(int x0 = __float_as_int(a), x1 = __float_as_int(b);
              auto r = __builtin_amdgcn_permlane32_swap(x0, x1, false, false);
              a = __int_as_float(r[0]) + xs[u]; b = __int_as_float(r[1]));
    } break;
    case 21: LOOP8(const unsigned long long m = __ballot(a > xs[u]); a = a + (float)__builtin_popcountll(m)); break;   // ballot -> SALU -> VALU
    case 22: LOOP8(a = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a))) + xs[u]); break;   // readfirstlane + add
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a + b;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int M> static void launch_t(int mode, int threads, float* o, unsigned long long* c, const float* in) {
    if (mode == M) hipLaunchKernelGGL(k_prim<M>, 1, threads, 0, 0, o, c, in);
    else if constexpr (M > 0) launch_t<M - 1>(mode, threads, o, c, in);
}
static void launch(int mode, int threads, float* o, unsigned long long* c, const float* in) { launch_t<22>(mode, threads, o, c, in); }

// s_memtime ticks against the 100 MHz s_memrealtime counter over a long dependent chain
__global__ void k_clock(float* out, unsigned long long* cyc, const float* in) {
    float a = in[threadIdx.x];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < (1 << 18); ++i) a = a + 1.0f;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

int main() {
    float *din, *dout; unsigned long long* dcyc;
    CK(hipMalloc(&din, 4 * 4096)); CK(hipMalloc(&dout, 4 * 1024)); CK(hipMalloc(&dcyc, 16));
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 1.0f + (i % 977) * 1e-3f;
    CK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
    const char* names[] = {"v_add_f32 chain", "row_shr:1 dpp + add", "row_shr:4 dpp + add", "wave_shr:1 dpp + add", "row_bcast:15 dpp + add",
        "v_readlane(acc,63) + add (VALU->SGPR->VALU)", "add with operand by v_readlane (independent)", "6-step wave scan + mul",
        "LDS-fed add chain, hipcc schedule (per add)", "LDS-fed add chain, bursts of 8 b128 (per add)", "ds_bpermute hop + add",
        "v_pk_add_f32 chain (per packed add)", "v_mul + v_add dependent pair", "v_add_f64 chain", "f64 mul + add pair", "IEEE f32 division (dependent)",
        "sqrtf + add", "dependent LDS read round trip", "roundf + mul", "quad_perm max + add", "permlane32_swap + add", "ballot + popcount + cvt + add",
        "v_readfirstlane + add"};
    {
        hipLaunchKernelGGL(k_clock, 1, 64, 0, 0, dout, dcyc, din); CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_clock, 1, 64, 0, 0, dout, dcyc, din); CK(hipDeviceSynchronize());
        unsigned long long c[2]; CK(hipMemcpy(c, dcyc, 16, hipMemcpyDeviceToHost));
        printf("clock: %llu s_memtime ticks in %llu s_memrealtime ticks (100 MHz): %.1f MHz; %.2f ticks per dependent add\n", c[0], c[1], 100.0 * c[0] / c[1], (double)c[0] / (1 << 18));
    }
    for (int threads : {64, 256}) {
        for (int mode = 0; mode < 23; ++mode) {
            unsigned long long best = ~0ull;
            for (int rep = 0; rep < 10; ++rep) {
                launch(mode, threads, dout, dcyc, din); CK(hipDeviceSynchronize());
                unsigned long long c; CK(hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost)); if (c < best) best = c;
            }
            printf("%4d threads  %-50s %7.2f ticks per step\n", threads, names[mode], (double)best / N);
        }
    }
    return 0;
}
