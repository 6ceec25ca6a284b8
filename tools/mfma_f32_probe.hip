// Developer probe: how do the f32 MFMAs of gfx950 round a k = 1 step, D = A * B + C?
// The reference sums every dot product as separate f32 multiplies and adds (no FMA: RN(RN(a*b) + c)).  If a k = 1 MFMA did
// the same, the ordered dots of attention could run on the matrix cores as rank-1 updates.  Every lane feeds the same
// (a, b, c), so every element of D is the same scalar expression and the operand layout does not matter.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/mfma_f32_probe tools/mfma_f32_probe.hip && tools/mfma_f32_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// kind 0: v_mfma_f32_4x4x1_16b_f32, 1: v_mfma_f32_16x16x1_4b_f32, 2: v_mfma_f32_32x32x1_2b_f32, 3: v_mfma_f32_16x16x4_f32 (k = 4: b = a's
// partner only in the first k slice -- lanes 0..15 -- the other slices multiply by zero)
__global__ __launch_bounds__(64) void k_probe(const float* a, const float* b, const float* c, float* d, int n, int kind) {
    const int lane = threadIdx.x;
    for (int i = 0; i < n; ++i) {
        const float av = a[i], bv = b[i], cv = c[i];
        float r;
        if (kind == 0) {
            v4f acc = {cv, cv, cv, cv};
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc, 0, 0, 0);
            r = acc.x;
        } else if (kind == 1) {
            v16f acc;
            for (int k = 0; k < 16; ++k) acc[k] = cv;
            acc = __builtin_amdgcn_mfma_f32_16x16x1f32(av, bv, acc, 0, 0, 0);
            r = acc[0];
        } else if (kind == 2) {
            typedef float v32f __attribute__((ext_vector_type(32)));
            v32f acc;
            for (int k = 0; k < 32; ++k) acc[k] = cv;
            acc = __builtin_amdgcn_mfma_f32_32x32x1f32(av, bv, acc, 0, 0, 0);
            r = acc[0];
        } else {
            v4f acc = {cv, cv, cv, cv};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(lane < 16 ? av : 0.0f, lane < 16 ? bv : 0.0f, acc, 0, 0, 0);
            r = acc.x;
        }
        if (lane == 0) d[i] = r;
    }
}

int main() {
    const int n = 200000;
    std::vector<float> a(n), b(n), c(n), d(n);
    srand(7);
    auto rnd = []() { return (float)((rand() + 0.5) / (RAND_MAX + 1.0)); };
    for (int i = 0; i < n; ++i) {
        const int kind = i % 5;
        a[i] = (rnd() * 2 - 1) * (kind == 3 ? 1e-19f : 1.0f);
        b[i] = (rnd() * 2 - 1) * (kind == 3 ? 1e-20f : 1.0f);            // kind 3: the product is a denormal
        c[i] = kind == 0 ? (rnd() * 2 - 1) : (kind == 1 ? (rnd() * 2 - 1) * 64.0f : (kind == 2 ? -a[i] * b[i] : (kind == 3 ? 0.0f : (rnd() * 2 - 1) * 1e-39f)));
    }
    a[0] = 1.0f + ldexpf(1.0f, -12); b[0] = a[0]; c[0] = -1.0f;          // fused: 2^-11 + 2^-24; separate roundings: 2^-11
    float *da, *db, *dc, *dd;
    CK(hipMalloc(&da, 4 * n)); CK(hipMalloc(&db, 4 * n)); CK(hipMalloc(&dc, 4 * n)); CK(hipMalloc(&dd, 4 * n));
    CK(hipMemcpy(da, a.data(), 4 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 4 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, c.data(), 4 * n, hipMemcpyHostToDevice));
    const char* names[] = {"v_mfma_f32_4x4x1_16b_f32", "v_mfma_f32_16x16x1_4b_f32", "v_mfma_f32_32x32x1_2b_f32", "v_mfma_f32_16x16x4_f32 (one live k slice)"};
    for (int kind = 0; kind < 4; ++kind) {
        hipLaunchKernelGGL(k_probe, 1, 64, 0, 0, da, db, dc, dd, n, kind);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(d.data(), dd, 4 * n, hipMemcpyDeviceToHost));
        long sep = 0, fus = 0, neither = 0, sep_by_class[5] = {0, 0, 0, 0, 0}, fus_by_class[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < n; ++i) {
            volatile float p = a[i] * b[i];
            const float r_sep = p + c[i], r_fus = fmaf(a[i], b[i], c[i]);
            unsigned ud, us, uf;
            memcpy(&ud, &d[i], 4); memcpy(&us, &r_sep, 4); memcpy(&uf, &r_fus, 4);
            if (ud == us) { ++sep; ++sep_by_class[i % 5]; }
            if (ud == uf) { ++fus; ++fus_by_class[i % 5]; }
            if (ud != us && ud != uf) ++neither;
        }
        printf("%-44s d[0]-2^-11 = %g   == RN(RN(a*b)+c): %ld   == fma(a,b,c): %ld   neither: %ld   of %d\n", names[kind], d[0] - ldexpf(1.0f, -11), sep, fus, neither, n);
        printf("    by class (c ~ product | c large | c = -RN(ab) | denormal product, c = 0 | denormal c):  separate %ld %ld %ld %ld %ld   fused %ld %ld %ld %ld %ld\n",
               sep_by_class[0], sep_by_class[1], sep_by_class[2], sep_by_class[3], sep_by_class[4], fus_by_class[0], fus_by_class[1], fus_by_class[2], fus_by_class[3], fus_by_class[4]);
    }
    return 0;
}
