// Developer probe (round 5): is straight-line code in a freshly launched kernel bound by instruction fetch?
// One wave runs N dependent v_add_f32 (a) as straight-line code (.rept: N x 4 or 8 bytes of instructions, executed once) and
// (b) as a 16-instruction loop; s_memtime around both.  Launched repeatedly: back to back (same kernel) and alternating with a
// different large kernel.  Also: 4 waves (one per SIMD) running DIFFERENT straight-line regions at once, like the role-split kernels.
//   hipcc --offload-arch=gfx950 -O3 -o tools/icache_probe tools/icache_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
#define STR2(x) #x
#define STR(x) STR2(x)
#define NADD 2048
__global__ void k_line32(float* out, unsigned long long* cyc, const float* in) {          // 4-byte encodings: 8 KB
    float a = in[threadIdx.x], b = in[64 + threadIdx.x];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile(".rept " STR(NADD) "\n v_add_f32_e32 %0, %0, %1\n .endr" : "+v"(a) : "v"(b));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_line64(float* out, unsigned long long* cyc, const float* in) {          // 8-byte encodings: 16 KB
    float a = in[threadIdx.x], b = in[64 + threadIdx.x];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile(".rept " STR(NADD) "\n v_add_f32_e64 %0, %0, %1\n .endr" : "+v"(a) : "v"(b));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_loop(float* out, unsigned long long* cyc, const float* in) {
    float a = in[threadIdx.x], b = in[64 + threadIdx.x];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < NADD / 16; ++i) asm volatile(".rept 16\n v_add_f32_e64 %0, %0, %1\n .endr" : "+v"(a) : "v"(b));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// four waves, each in its own straight-line region of 512 adds (8-byte encodings)
__global__ void k_roles(float* out, unsigned long long* cyc, const float* in) {
    float a = in[threadIdx.x], b = in[64 + (threadIdx.x & 63)];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave == 0) asm volatile(".rept 512\n v_add_f32_e64 %0, %0, %1\n .endr" : "+v"(a) : "v"(b));
    else if (wave == 1) asm volatile(".rept 512\n v_sub_f32_e64 %0, %0, %1\n .endr" : "+v"(a) : "v"(b));
    else if (wave == 2) asm volatile(".rept 512\n v_mul_f32_e64 %0, %0, %1\n .endr" : "+v"(a) : "v"(b));
    else asm volatile(".rept 512\n v_max_f32_e64 %0, %0, %1\n .endr" : "+v"(a) : "v"(b));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a; if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
int main() {
    float *din, *dout; unsigned long long* dcyc;
    CK(hipMalloc(&din, 4096)); CK(hipMalloc(&dout, 4096)); CK(hipMalloc(&dcyc, 8 * 64));
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 1.0f + i * 1e-3f;
    CK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
    auto run = [&](const char* what, auto kern, int threads, int nres) {
        printf("%-34s", what);
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(kern, 1, threads, 0, 0, dout, dcyc, din); CK(hipDeviceSynchronize());
            unsigned long long c[4]; CK(hipMemcpy(c, dcyc, 8 * nres, hipMemcpyDeviceToHost));
            for (int k = 0; k < nres; ++k) printf(" %6llu", c[k]);
            printf(" |");
        }
        printf("\n");
    };
    printf("# cycles for the timed region, five launches in a row (N = %d dependent adds)\n", NADD);
    run("straight line, 4-byte encodings", k_line32, 64, 1);
    run("straight line, 8-byte encodings", k_line64, 64, 1);
    run("16-instruction loop", k_loop, 64, 1);
    run("straight line, 4-byte, again", k_line32, 64, 1);
    run("4 waves x 512 in 4 regions", k_roles, 256, 4);
    // alternating kernels in one stream without host syncs in between (the engine's pattern)
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 20; ++i) { hipLaunchKernelGGL(k_line64, 1, 64, 0, 0, dout, dcyc, din); hipLaunchKernelGGL(k_line32, 1, 64, 0, 0, dout, dcyc + 1, din); hipLaunchKernelGGL(k_roles, 1, 256, 0, 0, dout, dcyc + 4, din); }
        CK(hipDeviceSynchronize());
        unsigned long long c[8]; CK(hipMemcpy(c, dcyc, 64, hipMemcpyDeviceToHost));
        printf("alternating, last of 20: line64 %llu  line32 %llu  roles %llu %llu %llu %llu\n", c[0], c[1], c[4], c[5], c[6], c[7]);
    }
    return 0;
}
