// q3_kernels.h -- gfx950 (MI355X) device code for the Qwen3 Q8 decode hot path.
//
// One token = one chain of weight-streaming kernels (5 per layer + classifier + bookkeeping), replayed
// from a hipGraph.  Every matmul is an int8 x int8 group-quantized GEMV (1 MAC per weight byte): the
// bound is HBM bandwidth, so the kernels are built around 16-byte-per-lane coalesced non-temporal
// loads of the checkpoint blob (one wavefront-load = 1 KiB of one weight row), v_dot4 integer dots,
// DPP cross-lane reductions and per-wave LDS scratch -- no MFMA (nothing to reuse at batch 1).
//
// Numerics follow the reference (reinterpretcat/qwen3-rs, qwen3-inference/src) operation by operation;
// the file is compiled with -ffp-contract=off so a*b+c is never fused, like rustc's output:
//   * matmul (tensor.rs:23-62): the i32 group dot is exact; each group term ((f32)dot*ws)*xs is formed
//     by one lane and the terms are summed in ascending group order by one lane => bit-identical to the
//     CPU result in every mode.
//   * RMSNorm / attention / softmax sums (layers.rs:109-131,374-419,495-506): default mode reduces with
//     wavefront trees (order differs from the CPU => tolerance); strict mode walks them sequentially in
//     the reference order => bit-identical logits.
//   * expf: glibc's algorithm (double-precision exp2 table + cubic), restated; RoPE cos/sin come from a
//     host-built glibc table.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// Developer instrumentation (ablation switches, in-kernel s_memtime timelines) is compiled in only with -DQ3_DEV
// (`make dev` -> libqwen3_hip_dev.so); the product build carries none of it.
#ifdef Q3_DEV
#define Q3_DEV_ABLATE(args, bit) (((args).debug & (bit)) != 0)
#else
#define Q3_DEV_ABLATE(args, bit) false
#endif

namespace q3 {

// Force a kernel argument into SGPRs at this point: hipcc otherwise issues the scalar loads of a large by-value argument
// struct in several batches, each next to its first use, and every batch costs a full scalar-memory round trip on the
// critical path of these latency-bound kernels.  One batch at the very top, one wait.
#define Q3_PIN_S(x) asm volatile("" ::"s"(x))

constexpr int kWG = 256;       // threads per workgroup
constexpr int kWaves = 4;      // wavefronts (64 lanes) per workgroup
constexpr int kMaxVR = 8;      // weight rows a wave finishes per batch
constexpr float kEps = 1e-6f;  // layers.rs:6

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));   // two lock-step f32 chains: v_pk_mul_f32 / v_pk_add_f32 (IEEE per element)

// ------------------------------------------------------------------------------------------------
// cross-lane helpers.  DPP controls: quad_perm(1,0,3,2)=0xB1, quad_perm(2,3,0,1)=0x4E,
// row_half_mirror=0x141, row_mirror=0x140.  After each step every lane of the (growing) aligned
// group holds the group's reduction, so the sequence is an all-reduce for commutative exact ops.
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(dpp_i<CTRL>(__float_as_int(v)));
}

// sum over aligned groups of `lanes` consecutive lanes (power of two, wave-uniform)
__device__ __forceinline__ int group_sum_i32(int v, int lanes) {
    if (lanes >= 2) v += dpp_i<0xB1>(v);
    if (lanes >= 4) v += dpp_i<0x4E>(v);
    if (lanes >= 8) v += dpp_i<0x141>(v);
    if (lanes >= 16) v += dpp_i<0x140>(v);
    if (lanes >= 32) v += __shfl_xor(v, 16);
    if (lanes >= 64) v += __shfl_xor(v, 32);
    return v;
}
template <int LANES>
__device__ __forceinline__ int group_sum_i32_t(int v) {
    if (LANES >= 2) v += dpp_i<0xB1>(v);
    if (LANES >= 4) v += dpp_i<0x4E>(v);
    if (LANES >= 8) v += dpp_i<0x141>(v);
    if (LANES >= 16) v += dpp_i<0x140>(v);
    return v;
}
__device__ __forceinline__ float group_max_f32(float v, int lanes) {
    if (lanes >= 2) v = fmaxf(v, dpp_f<0xB1>(v));
    if (lanes >= 4) v = fmaxf(v, dpp_f<0x4E>(v));
    if (lanes >= 8) v = fmaxf(v, dpp_f<0x141>(v));
    if (lanes >= 16) v = fmaxf(v, dpp_f<0x140>(v));
    if (lanes == 32) v = fmaxf(v, __shfl_xor(v, 16));
    if (lanes >= 64) {      // every lane of a 16-lane row holds the row result: combine the four rows via v_readlane
        const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
        const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
        const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
        const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
        v = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
    }
    return v;
}
// compile-time lane count (<= 16): straight-line DPP
template <int LANES>
__device__ __forceinline__ float group_max_f32_t(float v) {
    if (LANES >= 2) v = fmaxf(v, dpp_f<0xB1>(v));
    if (LANES >= 4) v = fmaxf(v, dpp_f<0x4E>(v));
    if (LANES >= 8) v = fmaxf(v, dpp_f<0x141>(v));
    if (LANES >= 16) v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}
// tree sum (default mode only: the order differs from the CPU's sequential fold)
__device__ __forceinline__ float group_sum_f32(float v, int lanes) {
    if (lanes >= 2) v += dpp_f<0xB1>(v);
    if (lanes >= 4) v += dpp_f<0x4E>(v);
    if (lanes >= 8) v += dpp_f<0x141>(v);
    if (lanes >= 16) v += dpp_f<0x140>(v);
    if (lanes == 32) v += __shfl_xor(v, 16);
    if (lanes >= 64) {
        const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
        const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
        const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
        const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
        v = (r0 + r1) + (r2 + r3);
    }
    return v;
}

// wave-local LDS hand-off: DS operations of one wave execute in issue order, so a compiler-level
// ordering point is all that is needed between a lane's ds_write and another lane's ds_read.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------
// f32::exp == glibc expf (sysdeps/ieee754/flt-32/e_expf.c, the exp2f_data table with N = 32):
// exp(x) = 2^(k/N) * 2^(r/N), k = round(x*N/ln2) by the 1.5*2^52 shift, cubic in r, all in double.
// ------------------------------------------------------------------------------------------------
static __device__ __constant__ unsigned long long kExp2Tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

// `tab`: the 32-entry exp2 table -- kExp2Tab in constant memory (a dependent global load in the middle of the
// evaluation), or a copy the kernel staged in LDS (latency-critical single-wave phases).
__device__ __forceinline__ float q3_expf_t(float x, const unsigned long long* tab) {
    // Branch-free: the main path is evaluated for every input and the special cases (|x| >= 88: overflow, underflow,
    // infinities, NaN) are patched in with selects afterwards, so several independent exps interleave in one wave
    // instead of serialising behind a divergent range check (an exp is ~12 dependent f64 operations).
    const unsigned ux = __float_as_uint(x);
    const unsigned abstop = (ux >> 20) & 0x7ffu;
    constexpr double kInvLn2N = 0x1.71547652b82fep+0 * 32.0;
    constexpr double kShift = 0x1.8p+52;
    constexpr double kC0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
    constexpr double kC1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
    constexpr double kC2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    const bool special = abstop >= 0x42bu;                 // |x| >= 88.0f (or not finite)
    const double xd = (double)x;                           // out-of-range inputs run through too (bit casts only, table
                                                           // index masked): their result is discarded below
    double z = kInvLn2N * xd;
    double kd = z + kShift;
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= kShift;
    const double r = z - kd;
    const unsigned long long t = tab[ki & 31u] + (ki << 47);
    const double s = __longlong_as_double((long long)t);
    z = kC0 * r + kC1;
    const double r2 = r * r;
    double y = kC2 * r + 1.0;
    y = z * r2 + y;
    y = y * s;
    float res = (float)y;
    // glibc's special cases, in its order of precedence (e_expf.c)
    float sp = 0.0f;                                       // x < -0x1.9fe368p6f: underflow to +0 (also -inf)
    sp = (x > 0x1.62e42ep6f) ? __builtin_inff() : sp;      // overflow
    sp = (special && !(x > 0x1.62e42ep6f) && !(x < -0x1.9fe368p6f)) ? res : sp;   // 88 <= |x| inside the finite range
    sp = (abstop >= 0x7f8u) ? (x + x) : sp;                // +inf / NaN
    sp = (ux == 0xff800000u) ? 0.0f : sp;                  // -inf
    return special ? sp : res;
}
__device__ __forceinline__ float q3_expf(float x) { return q3_expf_t(x, kExp2Tab); }

// f32::total_cmp key (sampler.rs:57-59): unsigned order of the key == IEEE total order
__device__ __forceinline__ unsigned total_order_key(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// Rust `(v).round() as i8`: half away from zero, saturating, NaN -> 0
__device__ __forceinline__ int quant_round_i8(float v) {
    float r = roundf(v);
    r = fminf(fmaxf(r, -128.0f), 127.0f);
    return (r != r) ? 0 : (int)r;
}

// ------------------------------------------------------------------------------------------------
// Device-side run state, advanced by k_next (so a greedy decode loop never returns to the host)
// ------------------------------------------------------------------------------------------------
struct State {
    int token;                      // input token of the current forward
    int pos;                        // position of the current forward
    int step;                       // forwards completed since the host last set the state
    int prompt_len;                 // > 0: tokens 1..prompt_len-1 of the prompt buffer are fed next (chat-mode prefill)
    unsigned long long argmax;      // (total_order_key(logit) << 32) | index, max-reduced
};

// ------------------------------------------------------------------------------------------------
// GEMV arguments
// ------------------------------------------------------------------------------------------------
// PRO_PREQR: the producer kernel already quantized the activation (attention / SwiGLU epilogues); every lane loads its own
// 16 bytes of xq per row chunk straight into registers -- no LDS stage, no barrier (rows of <= 4 KiB: one tile per row)
enum Pro : int { PRO_PREQ = 0, PRO_QUANT = 1, PRO_NORM = 2, PRO_EMBED_NORM = 3, PRO_PREQR = 4 };
enum Epi : int { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_LOGITS = 3, EPI_QKV = 4 };

struct Seg {
    const int8_t* wq;   // [rows][n] int8
    const float* ws;    // [rows][n/G]
    float* out;         // destination vector (indexed by row within the segment)
    int rows;
    int out_pos_stride; // EPI_QKV: out += pos * out_pos_stride (KV cache row); else 0
};

struct GemvArgs {
    // hot first: what the issue phase needs sits in the first 64 bytes of the kernarg segment (one scalar load)
    const float* in;       // PRO_QUANT: f32[n]; PRO_NORM: x f32[n]
    const float* norm_w;   // PRO_NORM*: RMSNorm weight f32[n]
    State* st;             // token / pos
    int n;               // contraction length (bytes per weight row)
    int group;           // quantization group size G
    int total_rows;      // sum of seg rows (EPI_SWIGLU: hidden units * 2 handled via seg[0],seg[1])
    int strict;
    Seg seg[3];
    long long qkv_dw[2], qkv_ds[2], qkv_do[2];  // EPI_QKV: byte deltas seg1-seg0, seg2-seg1 (wq, ws, out)
    int vr;              // rows per wave batch == the kernel's RU template parameter (host bookkeeping)
    int debug;           // developer ablation bits (Q3_ABLATE): 1 skip prologue math, 2 skip tiles, 4 skip ordered sum
    unsigned long long* stamps;  // developer timeline: 8 s_memtime stamps written by (block stamp_block, wave 0)
    int stamp_block;
    // other prologue inputs
    const int8_t* pre_q;   // PRO_PREQ: already-quantized activation
    const float* pre_s;
    const int8_t* emb_q;   // PRO_EMBED_NORM: embedding table
    const float* emb_s;
    float* x_out;          // PRO_EMBED_NORM: residual stream x (written by workgroup 0)
    float* tap_out;        // PRO_NORM: optional copy of the normalised vector (workgroup 0)
    unsigned long long* argmax_slots;  // EPI_LOGITS: one (key<<32|index) per workgroup
    int seq_len;
    // EPI_LOGITS with the bookkeeping of k_next folded in (next_cell != nullptr): every workgroup max-reduces its key into
    // next_cell[0] and takes a ticket from next_cell[1]; the last arriver consumes the cell and advances the state
    unsigned long long* next_cell;
    int32_t* out_tokens;
    int out_cap;
    const int32_t* prompt;
    // specialised NORM launches whose weight stream is shorter than their prologue (QKV of the 4B / 8B shapes): request the
    // weights only after wave 0 holds its block of x -- otherwise x queues behind tens of MB of weight requests of the other
    // workgroups and the exact sum starts ~3.5 us late (r03 stamps, 8B QKV: x after 8,955 cycles)
    int xfirst;
};

// LDS layout of the GEMV kernels (dynamic shared memory, 16-byte aligned carve):
//   [0, n)                    xq   int8   quantized activation
//   [n16, +4*n/G)             xs   f32    activation group scales
//   [.., +4*n)                xf   f32    staged activation (PRO_NORM only)
//   [.., +4*kWaves*vr*NG)     term f32    per-wave group terms
//   [.., +128*4)              red  f32    block reduction scratch + approximate block totals
// Exact speculative sum geometry: the n terms are cut into 64 blocks of n/64 terms, one lane per block (n a multiple of
// 256 in [512, 16384] -> block length a multiple of 4; other n: blocks of 64 terms, up to 64 of them; else a plain chain).
// Short blocks matter: the fold of a block is a dependent chain of 9-cycle adds, 16 of them for dim 1024.
constexpr int kSpecPad = 4;       // floats of padding per block: lane j's b128 reads hit distinct banks ((blen+4)j mod 64)
__host__ __device__ constexpr bool spec_ok(int n) { return n >= 512 && n <= 16384 && ((n % 256) == 0 || ((n % 64) == 0 && n <= 4096)); }
__host__ __device__ constexpr int spec_blen(int n) { return (n % 256) == 0 ? n / 64 : 64; }
__host__ __device__ inline int term_floats(int n) { return n + 64 * kSpecPad; }

struct GemvSmem {
    int8_t* xq;
    float* xs;
    float* xf;
    float* term;
    float* red;
    unsigned long long* etab;   // EPI_SWIGLU: LDS copy of the exp2 table (32 x 8 B)
};
__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~size_t(15); }
// waves: wavefronts per workgroup; fin: the register fold keeps the group terms out of LDS (no per-wave term rows)
__host__ __device__ inline size_t gemv_smem_bytes(int n, int group, int vr, bool stage_f32, int waves = kWaves, bool fin = false) {
    size_t b = align16((size_t)n) + align16(4 * (size_t)(n / group));
    if (stage_f32) b += align16(4 * (size_t)term_floats(n));
    if (!fin) b += align16(4 * (size_t)waves * vr * (n / group));
    b += 128 * 4 + 32 * 8;
    return b;
}
__device__ __forceinline__ GemvSmem gemv_carve(char* base, int n, int group, int vr, bool stage_f32, int waves = kWaves, bool fin = false) {
    GemvSmem s;
    s.xq = (int8_t*)base;
    base += align16((size_t)n);
    s.xs = (float*)base;
    base += align16(4 * (size_t)(n / group));
    s.xf = (float*)base;
    if (stage_f32) base += align16(4 * (size_t)term_floats(n));
    s.term = (float*)base;
    if (!fin) base += align16(4 * (size_t)waves * vr * (n / group));
    s.red = (float*)base;
    s.etab = (unsigned long long*)(base + 128 * 4);
    return s;
}

// block-wide sum of one float per thread (default mode).  Deterministic order: lane tree, then waves.
template <int NW = kWaves>
__device__ __forceinline__ float block_sum_fast(float v, float* red) {
    v = group_sum_f32(v, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float t = red[0];
    for (int w = 1; w < NW; ++w) t += red[w];
    __syncthreads();
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = group_max_f32(v, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float t = red[0];
    for (int w = 1; w < kWaves; ++w) t = fmaxf(t, red[w]);
    __syncthreads();
    return t;
}

// ------------------------------------------------------------------------------------------------
// Exact sequential f32 sums (Rust `iter.sum::<f32>()`, strict left fold) without a 1024-deep dependent
// chain.  `t` holds the terms; the result is bit-identical to  (((-0.0 + t0) + t1) + ...).
//
// seq_chain():  one lane-uniform left fold over a contiguous run, LDS reads software-pipelined.
// seq_sum_terms():  up to 64 lanes fold 64-term blocks concurrently from GUESSED running sums.  Adding a block of
//   small non-negative terms to a large accumulator is (barring ties / binade crossings) a translation,
//   out(s + d) = out(s) + d, so one correction sweep turns approximate guesses into (almost always)
//   exact block inputs; a second fold VERIFIES them bitwise (out_j == in_{j+1} for all j).  If any link
//   fails the loop repeats: block 0's input is exact by construction and round r fixes block r, so it
//   terminates, exact, in <= 64 rounds (1 in practice).  Critical path: two 64-add folds, independent of n.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float chain4(float s, v4f v) {
    s = s + v.x; s = s + v.y; s = s + v.z; s = s + v.w;
    return s;
}
// nq = number of float4 in the run; p 16-byte aligned (LDS).  A chain advances at the wave's issue rate -- ~4.9 cycles per
// INSTRUCTION (tools/mfma_chain_probe.hip), not per dependent add -- so what counts is instructions per add: operands are pulled
// in bursts of 8 float4, the next burst in flight while the current 32 adds run, and ONE explicit lgkmcnt(8) per burst says "all
// but the burst just issued has landed" (left alone hipcc waits once per float4: 1.5 instead of 1.28 instructions per add).
__device__ __forceinline__ float seq_chain(float s, const v4f* p, int nq) {
    int q = 0;
    if (nq >= 8) {
        v4f a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = p[k];
        for (; q + 16 <= nq; q += 16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) b[k] = p[q + 8 + k];
            __builtin_amdgcn_s_waitcnt(0xC87F);                  // lgkmcnt(8)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) s = chain4(s, a[k]);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 24 <= nq) {
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = p[q + 16 + k];
                __builtin_amdgcn_s_waitcnt(0xC87F);
            } else {
                __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): the last burst
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) s = chain4(s, b[k]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q + 8 <= nq) {
#pragma unroll
            for (int k = 0; k < 8; ++k) s = chain4(s, a[k]);
            q += 8;
        }
    }
    for (; q + 4 <= nq; q += 4) {
        const v4f a0 = p[q], a1 = p[q + 1], a2 = p[q + 2], a3 = p[q + 3];
        s = chain4(s, a0); s = chain4(s, a1); s = chain4(s, a2); s = chain4(s, a3);
    }
    for (; q < nq; ++q) s = chain4(s, p[q]);
    return s;
}

// LDS float index of term i in the (possibly padded) term array
__device__ __forceinline__ int term_index(int i, int n) {
    if (!spec_ok(n)) return i;
    const int bl = spec_blen(n);
    return (i / bl) * (bl + kSpecPad) + (i % bl);
}

// Wave-wide inclusive scan and shift in pure DPP (no LDS crossbar): Hillis-Steele inside the 16-lane rows, then
// row_bcast:15 (rows 1,3 += last lane of the row below) and row_bcast:31 (rows 2,3 += lane 31).  Used only for GUESSES
// and corrections whose exactness is verified afterwards, so the association order is free.
__device__ __forceinline__ float wave_scan_incl(float v) {
    v += dpp_f<0x111>(v);
    v += dpp_f<0x112>(v);
    v += dpp_f<0x114>(v);
    v += dpp_f<0x118>(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));
    return v;
}
// lane j gets lane j-1's value (wave_shr:1); lane 0 gets +0.0
__device__ __forceinline__ float wave_prev_lane(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}

// The guess -> correct -> verify loop shared by every exact sum: lane j < nblk owns block j (blocks in sequence order),
// `tot` is any approximation of its block total and fold(s) returns the block's exact left fold started from running
// sum s.  Every lane returns the exact sequential sum over all blocks, started from -0.0.
// Measured on MI355X (tools/sum_probe.hip, 1024 squares of N(0,s) data): a round costs ~400 cycles with 16-term blocks
// (fold 160 + scan 100 + shifts/compare ~140) and 4-8 rounds are needed -- one per power of two the running sum crosses
// inside the speculated range, because the f32 grid coarsens there and a correction of a few fine ulps is no longer a
// translation.  (Tried and dropped: a second round with five candidate inputs per block and a scalar walk over the
// non-transparent links -- exact, deterministic round count, but 1.3-1.6x slower than iterating.)
template <class Fold>
__device__ __forceinline__ float spec_sum_lanes(float tot, int nblk, Fold fold) {
    const int j = threadIdx.x & 63;
    const bool live = j < nblk;
    if (!live) tot = 0.0f;
    // guesses g_j = running sum before block j: exclusive prefix of the approximate totals
    float sc = wave_prev_lane(wave_scan_incl(tot));
    if (j == 0) sc = -0.0f;
    float out = 0.0f;
    for (int round = 0; round < 66; ++round) {
        out = fold(sc);
        // verify every link bitwise: input of block j must equal the output of block j-1
        const float prev = wave_prev_lane(out);
        const bool ok = (j == 0) || !live || (__float_as_uint(prev) == __float_as_uint(sc));
        if (__all(ok)) break;
        // corrected inputs under the translation assumption.  With e_j = out_{j-1} - s_j (the mismatch at link j) the
        // recurrence s'_j = out_{j-1} + (s'_{j-1} - s_{j-1}) unrolls to s'_j = s_j + sum_{i<=j} e_i: another scan.
        // Block 0's input is exact by construction and round r fixes block r, so the loop terminates, exact.
        float e = prev - sc;
        if (j == 0 || !live) e = 0.0f;
        sc = sc + wave_scan_incl(e);
        if (j == 0) sc = -0.0f;
    }
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(out), nblk - 1));   // the last block's output
}

// Exact sequential sum of nblk (<= 64) consecutive blocks of blen terms (blen % 4 == 0); block j starts at
// t + j*stride (16-byte aligned).  Every lane returns the sum.  approx_tot: optional nblk approximate block totals.
// Block lengths of 4..64 terms are compiled as straight-line register code (NQ float4 per lane, pulled from LDS once);
// other lengths fold out of LDS with the software-pipelined chain.
template <int NQ>
__device__ __forceinline__ float seq_sum_blocks_regs(const float* t, int nblk, int stride, const float* approx_tot) {
    const int j = threadIdx.x & 63;
    const bool live = j < nblk;
    const v4f* blk = (const v4f*)(t + (size_t)(live ? j : 0) * stride);
    v4f r[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) r[k] = blk[k];
    float tot = 0.0f;
    if (approx_tot != nullptr) {
        if (live) tot = approx_tot[j];
    } else {
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;      // only a guess: any summation order will do
#pragma unroll
        for (int k = 0; k < NQ; ++k) { p0 += r[k].x; p1 += r[k].y; p2 += r[k].z; p3 += r[k].w; }
        tot = (p0 + p1) + (p2 + p3);
    }
    return spec_sum_lanes(tot, nblk, [&](float s) {
#pragma unroll
        for (int k = 0; k < NQ; ++k) s = chain4(s, r[k]);
        return s;
    });
}
__device__ __forceinline__ float seq_sum_blocks(const float* t, int nblk, int blen, int stride, const float* approx_tot) {
    const int nq = blen >> 2;
    if (nq == 4) return seq_sum_blocks_regs<4>(t, nblk, stride, approx_tot);      // dim 1024
    if (nq == 16) return seq_sum_blocks_regs<16>(t, nblk, stride, approx_tot);    // dim 4096, 64-term blocks
    if (nq == 1) return seq_sum_blocks_regs<1>(t, nblk, stride, approx_tot);      // softmax rows <= 256
    if (nq == 2) return seq_sum_blocks_regs<2>(t, nblk, stride, approx_tot);
    if (nq == 8) return seq_sum_blocks_regs<8>(t, nblk, stride, approx_tot);
    if (nq == 10) return seq_sum_blocks_regs<10>(t, nblk, stride, approx_tot);    // dim 2560 (the 4B shape): 40-term blocks
    const int j = threadIdx.x & 63;
    const bool live = j < nblk;
    const v4f* blk = (const v4f*)(t + (size_t)(live ? j : 0) * stride);
    float tot = 0.0f;
    if (approx_tot != nullptr) {
        if (live) tot = approx_tot[j];
    } else if (live) {
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
        for (int q = 0; q < nq; ++q) {
            const v4f v = blk[q];
            p0 += v.x; p1 += v.y; p2 += v.z; p3 += v.w;
        }
        tot = (p0 + p1) + (p2 + p3);
    }
    return spec_sum_lanes(tot, nblk, [&](float s) { return seq_chain(s, blk, nq); });
}

// every lane returns the exact sequential sum of the n terms stored (term_index layout) at t
__device__ __forceinline__ float seq_sum_terms(const float* t, int n, const float* approx_tot = nullptr) {
    if (!spec_ok(n)) {
        if ((n & 3) == 0) return seq_chain(-0.0f, (const v4f*)t, n >> 2);
        float s = -0.0f;
        for (int i = 0; i < n; ++i) s = s + t[i];
        return s;
    }
    const int bl = spec_blen(n);
    return seq_sum_blocks(t, n / bl, bl, bl + kSpecPad, approx_tot);
}

// quantize 4 consecutive values held by this thread; its quantization group spans `glanes` = G/4
// consecutive threads (tensor.rs:91-119).  Writes the packed int8 dword and (group leader) the scale.
template <int GL_T = 0>
__device__ __forceinline__ void quantize4_to_lds(v4f y, int v_idx, int glanes, bool valid, int8_t* xq, float* xs) {
    float m = fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w)));
    if (!valid) m = 0.0f;
    if (GL_T > 0) { m = group_max_f32_t<GL_T>(m); glanes = GL_T; }
    else m = group_max_f32(m, glanes);
    const float scale = m / 127.0f;
    if (valid) {
        int q0 = 0, q1 = 0, q2 = 0, q3 = 0;
        if (scale != 0.0f) {
            q0 = quant_round_i8(y.x / scale);
            q1 = quant_round_i8(y.y / scale);
            q2 = quant_round_i8(y.z / scale);
            q3 = quant_round_i8(y.w / scale);
        }
        ((int*)xq)[v_idx] = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((q3 & 0xff) << 24);
        if ((v_idx % glanes) == 0) xs[v_idx / glanes] = scale;
    }
}

// ------------------------------------------------------------------------------------------------
// Prologues: build the quantized activation (xq, xs) in LDS.  Every workgroup does this redundantly
// (n <= 12288 floats from L2) so that no extra kernel boundary sits between the producer of the
// activation and the weight stream that consumes it.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stamp(const GemvArgs& a, int idx) {
#ifdef Q3_DEV
    if (a.stamps != nullptr && (int)blockIdx.x == a.stamp_block && threadIdx.x == 0) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        a.stamps[idx] = t;
    }
#else
    (void)a; (void)idx;
#endif
}

// The activation (and RMSNorm weight) loads are ISSUED before the first weight tile and CONSUMED after it
// is in flight: vmcnt retires in order, so the prologue can run at vmcnt(#tile loads) under the weight
// stream.  Up to kProSlots float4 slots per thread are prefetched (n <= 4096); longer vectors load the
// rest inside the loop.
constexpr int kProSlots = 4;          // PRO_NORM: x and norm weight (dim <= 4096 for every listed model)
constexpr int kProSlotsQuant = 12;    // PRO_QUANT: x only, up to n = 12288 (the 8B hidden size) in one round trip
template <int PRO> struct ProSlotCount { static constexpr int value = (PRO == PRO_QUANT) ? kProSlotsQuant : kProSlots; };
template <int PRO>
struct ProRegs {
    v4f x[ProSlotCount<PRO>::value];
    v4f w[kProSlots];
};

template <int PRO>
__device__ __forceinline__ void gemv_prologue_issue(const GemvArgs& a, ProRegs<PRO>& pr) {
    if (PRO == PRO_PREQ) return;
    const int nv = a.n >> 2;
    const int tid = threadIdx.x;
    if (PRO == PRO_QUANT) {
        const int nk = (nv + kWG - 1) / kWG;
#pragma unroll
        for (int k = 0; k < ProSlotCount<PRO>::value; ++k)
            if (k < kProSlots || k < nk)          // wave-uniform: long vectors fetch their extra slots in the same round trip
                pr.x[k] = ((const v4f*)a.in)[min(tid + k * kWG, nv - 1)];
        return;
    }
#pragma unroll
    for (int k = 0; k < kProSlots; ++k) {
        const int v = min(tid + k * kWG, nv - 1);
        if (PRO == PRO_NORM) pr.x[k] = ((const v4f*)a.in)[v];
        if (PRO == PRO_NORM || PRO == PRO_EMBED_NORM) pr.w[k] = ((const v4f*)a.norm_w)[v];
    }
    if (PRO == PRO_EMBED_NORM) {
        // TokenEmbedding::forward over the dequantised table (layers.rs:72-76, tensor.rs:72-80)
        const size_t row = (size_t)a.st->token * (size_t)a.n;
#pragma unroll
        for (int k = 0; k < kProSlots; ++k) {
            const int v = min(tid + k * kWG, nv - 1);
            const size_t e = row + 4 * (size_t)v;
            const int packed = *(const int*)(a.emb_q + e);
            const float sc = a.emb_s[e / (size_t)a.group];
            pr.x[k].x = (float)(int8_t)(packed & 0xff) * sc;
            pr.x[k].y = (float)(int8_t)((packed >> 8) & 0xff) * sc;
            pr.x[k].z = (float)(int8_t)((packed >> 16) & 0xff) * sc;
            pr.x[k].w = (float)(int8_t)((packed >> 24) & 0xff) * sc;
        }
    }
}

template <int PRO>
__device__ __forceinline__ v4f pro_load_x_global(const GemvArgs& a, int v) {
    if (PRO == PRO_EMBED_NORM) {
        const size_t e = (size_t)a.st->token * (size_t)a.n + 4 * (size_t)v;
        const int packed = *(const int*)(a.emb_q + e);
        const float sc = a.emb_s[e / (size_t)a.group];
        v4f xv;
        xv.x = (float)(int8_t)(packed & 0xff) * sc;
        xv.y = (float)(int8_t)((packed >> 8) & 0xff) * sc;
        xv.z = (float)(int8_t)((packed >> 16) & 0xff) * sc;
        xv.w = (float)(int8_t)((packed >> 24) & 0xff) * sc;
        return xv;
    }
    return ((const v4f*)a.in)[v];
}

__device__ __forceinline__ float sumsq4(v4f xv) {
    float s0 = xv.x * xv.x;
    float t = xv.y * xv.y; s0 = s0 + t;
    t = xv.z * xv.z; s0 = s0 + t;
    t = xv.w * xv.w; s0 = s0 + t;
    return s0;
}
__device__ __forceinline__ v4f norm4(v4f w, float f, v4f xv) {
    v4f y;
    y.x = w.x * (f * xv.x);      // layers.rs:117  w * (factor * x)
    y.y = w.y * (f * xv.y);
    y.z = w.z * (f * xv.z);
    y.w = w.w * (f * xv.w);
    return y;
}

template <int PRO, int LPG_T>
__device__ __forceinline__ void gemv_prologue_finish(const GemvArgs& a, const GemvSmem& sm, const ProRegs<PRO>& pr) {
    constexpr int GL = (LPG_T > 0 && LPG_T <= 4) ? 4 * LPG_T : 0;   // threads per quantization group, if known
    const int n = a.n, G = a.group;
    const int nv = n >> 2;          // float4 slots
    const int glanes = G >> 2;      // threads per quantization group
    const int tid = threadIdx.x;
    const int nk = (nv + kWG - 1) / kWG;
    if (PRO == PRO_PREQ) {
        for (int i = tid; i < (n >> 4); i += kWG) ((v4i*)sm.xq)[i] = ((const v4i*)a.pre_q)[i];
        for (int i = tid; i < n / G; i += kWG) sm.xs[i] = a.pre_s[i];
        __syncthreads();
        return;
    }
    if (PRO == PRO_QUANT) {
#pragma unroll
        for (int k = 0; k < ProSlotCount<PRO>::value; ++k) {      // static indices: pr stays in registers
            if (k < nk) {
                const int v = k * kWG + tid;
                quantize4_to_lds<GL>(pr.x[k], v, glanes, v < nv, sm.xq, sm.xs);
            }
        }
        for (int k = ProSlotCount<PRO>::value; k < nk; ++k) {
            const int v = k * kWG + tid;
            const v4f y = ((const v4f*)a.in)[min(v, nv - 1)];
            quantize4_to_lds<GL>(y, v, glanes, v < nv, sm.xq, sm.xs);
        }
        __syncthreads();
        return;
    }
    // PRO_NORM / PRO_EMBED_NORM: x -> RMSNorm (layers.rs:109-119) -> quantize (tensor.rs:91-119)
    float part = 0.0f;
    const int lanes_per_block = spec_blen(n) >> 2;          // float4 slots (consecutive threads) per speculative block
    const bool have_approx = spec_ok(n) && nk <= kProSlots && (nv % kWG) == 0 && lanes_per_block <= 16 &&
                             (lanes_per_block & (lanes_per_block - 1)) == 0;
#pragma unroll
    for (int k = 0; k < kProSlots; ++k) {
        const int v = k * kWG + tid;
        if (k < nk && v < nv) {
            const v4f xv = pr.x[k];
            if (PRO == PRO_EMBED_NORM && blockIdx.x == 0) ((v4f*)a.x_out)[v] = xv;
            v4f sq;
            sq.x = xv.x * xv.x; sq.y = xv.y * xv.y; sq.z = xv.z * xv.z; sq.w = xv.w * xv.w;
            *(v4f*)(sm.xf + term_index(4 * v, n)) = sq;      // squares, layers.rs:113
            const float p4 = sumsq4(xv);
            part = part + p4;
            if (a.strict && have_approx) {
                // block = blen elements = blen/4 consecutive float4 slots = consecutive threads of this slot
                const float bt = group_sum_f32(p4, lanes_per_block);
                if ((v & (lanes_per_block - 1)) == 0) sm.red[64 + v / lanes_per_block] = bt;
            }
        }
    }
    for (int k = kProSlots; k < nk; ++k) {
        const int v = k * kWG + tid;
        if (v < nv) {
            const v4f xv = pro_load_x_global<PRO>(a, v);
            if (PRO == PRO_EMBED_NORM && blockIdx.x == 0) ((v4f*)a.x_out)[v] = xv;
            v4f sq;
            sq.x = xv.x * xv.x; sq.y = xv.y * xv.y; sq.z = xv.z * xv.z; sq.w = xv.w * xv.w;
            *(v4f*)(sm.xf + term_index(4 * v, n)) = sq;
            part = part + sumsq4(xv);
        }
    }
    float ss;
    if (a.strict) {
        stamp(a, 6);
        __syncthreads();
        // one wave walks the exact sum, the others wait: every wave repeating it (the r02 form) costs nothing on a lone latency-bound
        // workgroup but is three quarters of the sum's instructions where thousands of these workgroups queue (dense prefill prologues)
        if (tid < 64) {
            ss = seq_sum_terms(sm.xf, n, have_approx ? sm.red + 64 : nullptr);
            if (tid == 0) sm.red[0] = ss;
        }
        __syncthreads();
        ss = sm.red[0];
        stamp(a, 7);
    } else {
        ss = block_sum_fast(part, sm.red);
    }
    const float f = 1.0f / sqrtf(ss / (float)n + kEps);
#pragma unroll
    for (int k = 0; k < kProSlots; ++k) {
        if (k < nk) {
            const int v = k * kWG + tid;
            const bool valid = v < nv;
            v4f y = {0.f, 0.f, 0.f, 0.f};
            if (valid) {
                y = norm4(pr.w[k], f, pr.x[k]);
                if (a.tap_out != nullptr && blockIdx.x == 0) ((v4f*)a.tap_out)[v] = y;
            }
            quantize4_to_lds<GL>(y, v, glanes, valid, sm.xq, sm.xs);
        }
    }
    for (int k = kProSlots; k < nk; ++k) {
        const int v = k * kWG + tid;
        const bool valid = v < nv;
        v4f y = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            y = norm4(((const v4f*)a.norm_w)[v], f, pro_load_x_global<PRO>(a, v));
            if (a.tap_out != nullptr && blockIdx.x == 0) ((v4f*)a.tap_out)[v] = y;
        }
        quantize4_to_lds<GL>(y, v, glanes, valid, sm.xq, sm.xs);
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Shape-specialised prologue (round 3): contraction length N, workgroup width WGT (256 / 512 / 1024 threads) and elements
// per thread EPT (4 / 2 / 1) are compile-time constants, group size 64.  The generic prologue above walks run-time slot
// counts through a dozen wave-uniform branches and repeats ~150 instructions per float4 in every one of the two 256-thread
// workgroups of a CU, one wave per SIMD pair, i.e. bound by dependent-instruction latency; here the work is straight-line,
// one workgroup per CU spreads it over up to 16 waves (4 per SIMD: latency is hidden by the other waves), and each thread
// handles N / WGT elements.  Same arithmetic per element (layers.rs:109-119, tensor.rs:91-119), so xq / xs are identical.
// Thread v of pass p owns elements [(p*WGT + v)*EPT, +EPT); a quantization group is 64/EPT consecutive threads.
// ------------------------------------------------------------------------------------------------
template <int EPT> __device__ __forceinline__ void load_vec(const float* p, float (&d)[EPT]) {
    if constexpr (EPT == 4) { const v4f t = *(const v4f*)p; d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; }
    else if constexpr (EPT == 2) { const v2f t = *(const v2f*)p; d[0] = t.x; d[1] = t.y; }
    else d[0] = *p;
}
template <int EPT> __device__ __forceinline__ void store_vec(float* p, const float (&d)[EPT]) {
    if constexpr (EPT == 4) { v4f t; t.x = d[0]; t.y = d[1]; t.z = d[2]; t.w = d[3]; *(v4f*)p = t; }
    else if constexpr (EPT == 2) { v2f t; t.x = d[0]; t.y = d[1]; *(v2f*)p = t; }
    else *p = d[0];
}
// max over aligned groups of LANES (16 / 32 / 64) consecutive lanes, compile-time
template <int LANES> __device__ __forceinline__ float group_max_c(float v) {
    if constexpr (LANES <= 16) return group_max_f32_t<LANES>(v);
    else return group_max_f32(v, LANES);
}
#ifndef Q3_BLK_LDS_MIN
#define Q3_BLK_LDS_MIN 4
#endif
template <int PRO, int N, int WGT, int EPT>
struct Pro2 {
    static constexpr int EPP = EPT * WGT;                 // elements per pass of the whole workgroup
    static constexpr int NP = (N + EPP - 1) / EPP;        // passes
    static constexpr bool kFull = (N % EPP) == 0;         // every thread of every pass holds EPT live elements
    static constexpr int GL = 64 / EPT;                   // threads per quantization group
    static constexpr bool kNorm = (PRO == PRO_NORM || PRO == PRO_EMBED_NORM);
    // RMSNorm: the exact sum of squares is the work of ONE wave per workgroup (wave 0).  Lane j of that wave owns block j of
    // the vector (64 blocks of N/64 elements) and pulls it straight from global memory into registers -- no LDS staging of
    // the squares, no barrier in front of the sum -- while the other waves wait at the barrier that publishes the factor.
    // (r03 first cut: every wave of a 16-wave workgroup ran the sum redundantly; four waves per SIMD interleaving the same
    // ~300-instruction loop made it issue-bound and the prologue no faster than with 4-wave workgroups.)
    static constexpr int NQ = kNorm ? N / 256 : 1;        // float4 per lane of wave 0
    static constexpr bool kBlkViaLds = (PRO == PRO_NORM) && NQ >= Q3_BLK_LDS_MIN;   // coalesced loads + LDS transpose (see pro2_issue)
    float x[NP][EPT];
    float w[NP][EPT];
    v4f blk[NQ];
};
#ifdef Q3_DEV
#define PRO_STAMP(a, i) stamp(a, i)
#else
#define PRO_STAMP(a, i) do { } while (0)
#endif

// what: 0 = everything; 1 = only wave 0's block of x (the exact sum's operands); 2 = everything else (GemvArgs::xfirst)
template <int PRO, int N, int WGT, int EPT>
__device__ __forceinline__ void pro2_issue(const GemvArgs& a, Pro2<PRO, N, WGT, EPT>& pr, int what = 0) {
    typedef Pro2<PRO, N, WGT, EPT> P;
    static_assert(N % 64 == 0 && (N % EPT) == 0, "whole quantization groups");
    static_assert(!P::kNorm || (N % 256) == 0, "64 blocks of whole float4 for the exact sum");
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tok = 0;
    if constexpr (PRO == PRO_EMBED_NORM) tok = a.st->token;
    if constexpr (PRO == PRO_NORM) {
        if (wave == 0 && a.strict && what != 2) {         // oldest loads of wave 0: its block of x for the exact sum
            if constexpr (P::kBlkViaLds) {
                // long vectors: COALESCED float4 loads (lane j, slot k <- float4 j + 64k), transposed into blocks through LDS in
                // pro2_finish.  Lane-per-block loads touch 64 cache lines per instruction: 16 of them held the 8B shapes' sum
                // back until ~3,000 cycles after entry (r03 stamps).
                const v4f* bp = (const v4f*)a.in + (tid & 63);
#pragma unroll
                for (int k = 0; k < P::NQ; ++k) pr.blk[k] = bp[64 * k];
            } else {
                const v4f* bp = (const v4f*)(a.in + (size_t)(tid & 63) * (N / 64));
#pragma unroll
                for (int k = 0; k < P::NQ; ++k) pr.blk[k] = bp[k];
            }
        }
    }
    if (what != 1) {
#pragma unroll
        for (int p = 0; p < P::NP; ++p) {
            const int e0 = (p * WGT + tid) * EPT;
            const int ec = P::kFull ? e0 : min(e0, N - EPT);          // threads past the vector re-read its tail (discarded)
            if constexpr (PRO == PRO_NORM || PRO == PRO_QUANT) load_vec<EPT>(a.in + ec, pr.x[p]);
            if constexpr (P::kNorm) load_vec<EPT>(a.norm_w + ec, pr.w[p]);
        }
    }
    if constexpr (PRO == PRO_EMBED_NORM) {
        // TokenEmbedding::forward over the dequantised table (layers.rs:72-76, tensor.rs:72-80)
        const size_t row = (size_t)tok * (size_t)N;
        if (wave == 0 && a.strict && what != 2) {
            const size_t b0 = row + (size_t)(tid & 63) * (N / 64);
#pragma unroll
            for (int k = 0; k < P::NQ; ++k) {
                const size_t e = b0 + 4 * k;
                const int packed = *(const int*)(a.emb_q + e);
                const float sc = a.emb_s[e >> 6];
                v4f t;
                t.x = (float)(int8_t)(packed & 0xff) * sc;
                t.y = (float)(int8_t)((packed >> 8) & 0xff) * sc;
                t.z = (float)(int8_t)((packed >> 16) & 0xff) * sc;
                t.w = (float)(int8_t)((packed >> 24) & 0xff) * sc;
                pr.blk[k] = t;
            }
        }
        if (what != 1)
#pragma unroll
        for (int p = 0; p < P::NP; ++p) {
            const int e0 = (p * WGT + tid) * EPT;
            const size_t e = row + (size_t)(P::kFull ? e0 : min(e0, N - EPT));
            const float sc = a.emb_s[e >> 6];
            int packed;
            if constexpr (EPT == 4) packed = *(const int*)(a.emb_q + e);
            else if constexpr (EPT == 2) packed = *(const short*)(a.emb_q + e);
            else packed = a.emb_q[e];
#pragma unroll
            for (int k = 0; k < EPT; ++k) pr.x[p][k] = (float)(int8_t)((packed >> (8 * k)) & 0xff) * sc;
        }
    }
}

template <int PRO, int N, int WGT, int EPT>
__device__ __forceinline__ void pro2_finish(const GemvArgs& a, const GemvSmem& sm, Pro2<PRO, N, WGT, EPT>& pr) {
    typedef Pro2<PRO, N, WGT, EPT> P;
    constexpr int WAVES = WGT / 64;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (P::kNorm) {
        // x -> RMSNorm (layers.rs:109-119)
        float f;
        if (a.strict) {
            if (wave == 0) {
                __builtin_amdgcn_s_setprio(3);
                if constexpr (P::kBlkViaLds) {
                    // float4 (j + 64k) of x -> LDS (blocks of N/64 floats, padded by 4 against bank conflicts) -> lane j's block
                    constexpr int BLQ = P::NQ + 1;                                           // float4 per padded block
                    const int lane = tid & 63;
#pragma unroll
                    for (int k = 0; k < P::NQ; ++k) {
                        const int f4 = lane + 64 * k;                                        // float4 index in x
                        ((v4f*)sm.xf)[(f4 / P::NQ) * BLQ + (f4 % P::NQ)] = pr.blk[k];
                    }
                    wave_lds_sync();
#pragma unroll
                    for (int k = 0; k < P::NQ; ++k) pr.blk[k] = ((const v4f*)sm.xf)[lane * BLQ + k];
                }
                PRO_STAMP(a, 6);
                float tot = 0.0f;
#pragma unroll
                for (int k = 0; k < P::NQ; ++k) {
                    v4f q = pr.blk[k];
                    q.x = q.x * q.x; q.y = q.y * q.y; q.z = q.z * q.z; q.w = q.w * q.w;      // layers.rs:113
                    pr.blk[k] = q;
                    tot += (q.x + q.y) + (q.z + q.w);                                        // a guess only: any order
                }
                const float ss = spec_sum_lanes(tot, 64, [&](float s0) {
#pragma unroll
                    for (int k = 0; k < P::NQ; ++k) s0 = chain4(s0, pr.blk[k]);
                    return s0;
                });
                PRO_STAMP(a, 7);
                const float fw = 1.0f / sqrtf(ss / (float)N + kEps);
                if ((tid & 63) == 0) sm.red[0] = fw;
                __builtin_amdgcn_s_setprio(0);
            }
            __syncthreads();
            f = sm.red[0];
        } else {
            float part = 0.0f;
#pragma unroll
            for (int p = 0; p < P::NP; ++p) {
                const bool live = P::kFull || (p * WGT + tid) * EPT < N;
                float p4 = 0.0f;
#pragma unroll
                for (int k = 0; k < EPT; ++k) p4 = p4 + pr.x[p][k] * pr.x[p][k];
                if (live) part = part + p4;
            }
            const float ss = block_sum_fast<WAVES>(part, sm.red);
            f = 1.0f / sqrtf(ss / (float)N + kEps);
        }
#pragma unroll
        for (int p = 0; p < P::NP; ++p) {
            const int e0 = (p * WGT + tid) * EPT;
            const bool live = P::kFull || e0 < N;
            if (PRO == PRO_EMBED_NORM && live && blockIdx.x == 0) store_vec<EPT>(a.x_out + e0, pr.x[p]);
#pragma unroll
            for (int k = 0; k < EPT; ++k) pr.x[p][k] = pr.w[p][k] * (f * pr.x[p][k]);       // layers.rs:117  w * (factor * x)
            if (live && a.tap_out != nullptr && blockIdx.x == 0) store_vec<EPT>(a.tap_out + e0, pr.x[p]);
        }
    }
    // quantize (tensor.rs:91-119): group max over the 64/EPT threads of a group, IEEE divisions, round half away
#pragma unroll
    for (int p = 0; p < P::NP; ++p) {
        const int v = p * WGT + tid, e0 = v * EPT;
        const bool live = P::kFull || e0 < N;
        float m = 0.0f;
#pragma unroll
        for (int k = 0; k < EPT; ++k) m = fmaxf(m, fabsf(pr.x[p][k]));
        if (!live) m = 0.0f;
        m = group_max_c<P::GL>(m);
        const float scale = m / 127.0f;
        if (live) {
            int packed = 0;
            if (scale != 0.0f) {
#pragma unroll
                for (int k = 0; k < EPT; ++k) packed |= (quant_round_i8(pr.x[p][k] / scale) & 0xff) << (8 * k);
            }
            if constexpr (EPT == 4) ((int*)sm.xq)[v] = packed;
            else if constexpr (EPT == 2) ((short*)sm.xq)[v] = (short)packed;
            else sm.xq[v] = (int8_t)packed;
            if ((v & (P::GL - 1)) == 0) sm.xs[v / P::GL] = scale;
        }
    }
    PRO_STAMP(a, 8);
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// GEMV body.  A "unit" is one wavefront-load: 64 lanes x 16 B = 1 KiB of one weight row (chunk j of the
// row).  Lane l owns bytes [16(l+64j), +16): LPG = G/16 adjacent lanes share a quantization group.
// A tile = RU rows x JU chunks (RU*JU <= 8 units) is loaded into registers in one go; tiles are
// double-buffered so the next tile's HBM loads are in flight while the current one is reduced, and the
// very first tile is requested BEFORE the activation prologue (weights depend only on kernel arguments).
// ------------------------------------------------------------------------------------------------
template <int LPG_T>
__device__ __forceinline__ int lpg_sum(int v, int lpg) {
    if (LPG_T > 0) return group_sum_i32_t<LPG_T>(v);
    return group_sum_i32(v, lpg);
}

template <int RU, int JU>
struct Tile {
    v4i w[RU][JU];
    float sc[RU][JU];
};

// weight rows of one wave batch: `parts` runs (2 for SwiGLU's w1|w3 pair, else 1) of `hu` consecutive
// rows, the first `cnt` of each run live.  base + row*stride addressing, no pointer arrays.
struct RowSrc {
    const int8_t* w[2];
    const float* s[2];
    float* out;          // destination vector of the batch's segment, already offset to the batch's row 0
    int hu;
    int cnt;
    int row0;            // row index (within segment) of the batch's first row
    int ops;             // EPI_QKV: floats per position of the destination (KV cache row stride), else 0
    float resid;         // EPI_RESID: x[row] of this lane's row, requested together with the batch's first tile
};

// ascending-group sum of one row's terms (Iterator::sum from -0.0 == start at term 0).  Lean on registers
// (4+4 float4): it runs while two weight tiles are live, and VGPRs decide the streaming kernels' occupancy.
__device__ __forceinline__ float ordered_row_sum(const float* t, int ng) {
    if ((ng & 3) == 0) {
        const v4f* p = (const v4f*)t;
        const int nq = ng >> 2;
        float s = -0.0f;
        int q = 0;
        if (nq >= 4) {
            v4f a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
            for (; q + 8 <= nq; q += 8) {
                const v4f b0 = p[q + 4], b1 = p[q + 5], b2 = p[q + 6], b3 = p[q + 7];
                s = chain4(s, a0); s = chain4(s, a1); s = chain4(s, a2); s = chain4(s, a3);
                if (q + 12 <= nq) { a0 = p[q + 8]; a1 = p[q + 9]; a2 = p[q + 10]; a3 = p[q + 11]; }
                s = chain4(s, b0); s = chain4(s, b1); s = chain4(s, b2); s = chain4(s, b3);
            }
            if (q + 4 <= nq) {
                s = chain4(s, a0); s = chain4(s, a1); s = chain4(s, a2); s = chain4(s, a3);
                q += 4;
            }
        }
        for (; q < nq; ++q) s = chain4(s, p[q]);
        return s;
    }
    float acc = t[0];
    for (int g = 1; g < ng; ++g) acc = acc + t[g];
    return acc;
}

// FIN = 1 (G = 64): the group terms never touch LDS.  After the DPP
// all-reduce every lane of a group holds the group's term; the row's sum is folded in ascending group order by a chain
// of 16 DPP adds per 1 KiB chunk (the running sum hops from group to group, lane 4g+3 -> 4g+7), all rows of the tile in
// flight together; the chunk total is read from lane 63 and carries into the next chunk / tile.
// PF = 1 (streaming launches where every wave has at least two tiles): the SECOND tile is requested before the activation
// prologue as well, so 2 x 8 KiB per wave (32 MB chip-wide at 2 workgroups per CU) are in flight while the norm / exact
// sum / quantize run -- the prologue no longer opens a bubble in the HBM stream.
// N_T > 0 (round 3): contraction length, workgroup width WGT and elements per prologue thread EPT are compile-time (group
// 64, register fold): the prologue is pro2_*, every index derived from n folds, and one workgroup of up to 16 waves per CU
// replaces two of four.  N_T == 0: the generic run-time-n kernel (any group size; 256 threads).
template <int PRO, int EPI, int LPG_T, int RU, int JU, int FIN = 0, int PF = 0, int N_T = 0, int WGT = kWG, int EPT = 4>
__global__ __launch_bounds__(WGT) void k_gemv(const GemvArgs a) {
    static_assert(N_T > 0 || (WGT == kWG && EPT == 4), "the generic prologue is written for 256 threads");
    static_assert(N_T == 0 || (LPG_T == 4 && FIN == 1), "specialised shapes: group 64, register fold");
    static_assert(PRO != PRO_PREQR || (N_T > 0 && (N_T + 1023) / 1024 == JU), "register-direct xq: one tile per row");
    constexpr int WAVES = WGT / 64;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    stamp(a, 0);
    // (the grid width is a hidden kernel argument: left alone its scalar load sits next to its first use, a second round trip
    // in front of the first weight request)
    const int gdx = (int)gridDim.x;
    Q3_PIN_S(gdx);
    Q3_PIN_S(a.in); Q3_PIN_S(a.n); Q3_PIN_S(a.group); Q3_PIN_S(a.total_rows); Q3_PIN_S(a.strict);
    Q3_PIN_S(a.seg[0].wq); Q3_PIN_S(a.seg[0].ws); Q3_PIN_S(a.seg[0].out); Q3_PIN_S(a.seg[0].rows);
    if constexpr (PRO == PRO_NORM || PRO == PRO_EMBED_NORM) Q3_PIN_S(a.norm_w);
    if constexpr (PRO == PRO_EMBED_NORM) { Q3_PIN_S(a.emb_q); Q3_PIN_S(a.emb_s); Q3_PIN_S(a.x_out); }
    if constexpr (PRO == PRO_EMBED_NORM || EPI == EPI_QKV) Q3_PIN_S(a.st);
    if constexpr (PRO == PRO_PREQ || PRO == PRO_PREQR) { Q3_PIN_S(a.pre_q); Q3_PIN_S(a.pre_s); }
    if constexpr (EPI == EPI_SWIGLU) { Q3_PIN_S(a.seg[1].wq); Q3_PIN_S(a.seg[1].ws); }
    if constexpr (EPI == EPI_QKV) {
        Q3_PIN_S(a.seg[0].out_pos_stride); Q3_PIN_S(a.seg[1].rows); Q3_PIN_S(a.seg[2].rows);
        Q3_PIN_S(a.seg[1].out_pos_stride); Q3_PIN_S(a.seg[2].out_pos_stride);
        Q3_PIN_S(a.qkv_dw[0]); Q3_PIN_S(a.qkv_dw[1]); Q3_PIN_S(a.qkv_ds[0]); Q3_PIN_S(a.qkv_ds[1]);
        Q3_PIN_S(a.qkv_do[0]); Q3_PIN_S(a.qkv_do[1]);
    }
    if constexpr (EPI == EPI_LOGITS) { Q3_PIN_S(a.argmax_slots); Q3_PIN_S(a.next_cell); }
    // the activation / norm-weight loads go out before anything else is computed (they are the critical path)
    constexpr bool kSpec = N_T > 0;
    constexpr bool kPro2 = kSpec && (PRO == PRO_QUANT || PRO == PRO_NORM || PRO == PRO_EMBED_NORM);
    ProRegs<kPro2 || PRO == PRO_PREQR ? PRO_PREQ : PRO> pr;                          // generic prologue registers (empty for PREQ)
    Pro2<PRO, kPro2 ? N_T : 64, kPro2 ? WGT : 64, kPro2 ? EPT : 4> pr2;              // specialised prologue registers
    unsigned long long etv = 0ull;
    if (EPI == EPI_SWIGLU && threadIdx.x < 32) etv = kExp2Tab[threadIdx.x];   // oldest load: retires first (vmcnt is in order)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    v4i xr[PRO == PRO_PREQR ? JU : 1];        // PRO_PREQR: this lane's 16 bytes of xq of every chunk of a row, and their
    float xsr[PRO == PRO_PREQR ? JU : 1];     // group scales
    if constexpr (PRO == PRO_PREQR) {
#pragma unroll
        for (int j = 0; j < JU; ++j) {
            const int c = min(lane + 64 * j, (N_T >> 4) - 1);
            xr[j] = ((const v4i*)a.pre_q)[c];
            xsr[j] = a.pre_s[c >> 2];
        }
    } else if constexpr (kPro2) {
        if constexpr (PRO == PRO_NORM || PRO == PRO_EMBED_NORM) {
            if (a.xfirst) {                   // wave-uniform (A/B knob, off by default): wave 0's block of x travels alone, everything else behind it
                pro2_issue<PRO, N_T, WGT, EPT>(a, pr2, 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                pro2_issue<PRO, N_T, WGT, EPT>(a, pr2, 2);
            } else pro2_issue<PRO, N_T, WGT, EPT>(a, pr2, 0);
        } else pro2_issue<PRO, N_T, WGT, EPT>(a, pr2, 0);
        if (a.xfirst == 2) {
            // every wave's activation requests enter the CU's address path before any weight request does: a dwordx4 load of a
            // wave occupies that path for 16 cycles, and wave 0's 17 block loads otherwise interleave with the other 15 waves'
            // 16 tile loads each -- its block (the exact sum's input) finished ISSUING ~3,000 cycles after entry (r03 stamps)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        }
    } else gemv_prologue_issue<PRO>(a, pr);
    // EPI_QKV: the position is REQUESTED here (behind the activation, ahead of the weights) and only turned into a scalar
    // after the prologue -- a v_readfirstlane right here made hipcc wait for every load above before the first weight tile
    // was requested (r02 disassembly: s_waitcnt vmcnt(0) in front of the tile loads)
    int pos_v = 0;
    if constexpr (EPI == EPI_QKV) pos_v = a.st->pos;
    __builtin_amdgcn_sched_barrier(0);        // (pin the issue order: vmcnt retires in order)

    constexpr bool kStage = (PRO == PRO_NORM || PRO == PRO_EMBED_NORM);
    constexpr int HU = (EPI == EPI_SWIGLU) ? (RU / 2) : RU;   // rows per run
    static_assert(EPI != EPI_SWIGLU || RU >= 2, "SwiGLU tiles hold a w1 and a w3 row");
    const int n = kSpec ? N_T : a.n, G = kSpec ? 64 : a.group;
    const GemvSmem sm = gemv_carve(smem_raw, n, G, RU, kStage && (!kSpec || n >= 1024), WAVES, FIN != 0);

    const int lpg_shift = (LPG_T > 0) ? __builtin_ctz(LPG_T) : __builtin_ctz(G >> 4);   // G is a power of two >= 16
    const int lpg = 1 << lpg_shift;
    const int ng = n / G;
    const int nchunks = n >> 4;
    const int nj = (n + 1023) >> 10;
    const int njt = (nj + JU - 1) / JU;            // tiles per row batch
    float* term = sm.term + (FIN != 0 ? 0 : wave * RU * ng);
    // batch b of the launch belongs to wave b % nwaves.  Specialised shapes number the waves workgroup-minor, so a launch
    // with fewer row batches than waves (Wo / W2 of the small models under 16-wave workgroups) still puts rows on every CU;
    // the surplus waves only help with the prologue.
    const int gw = kSpec ? wave * gdx + (int)blockIdx.x : (int)blockIdx.x * WAVES + wave;
    const int nwaves = gdx * WAVES;
    const int units = (EPI == EPI_SWIGLU) ? a.seg[0].rows : a.total_rows;   // rows (or hidden units)
    //   // rows (or hidden units)
    const int nb = (units + HU - 1) / HU;
    int pos = 0;                                     // EPI_QKV: set behind the prologue (see pos_v)

    // QKV: segments 1,2 are addressed as byte deltas from segment 0 and blended with 0/1 arithmetic (a
    // select between pointers loaded from the kernarg segment gets folded by LLVM into a VECTOR load of
    // the selected kernarg slot, whose wait would drain the activation loads already in flight).
    const int r0s = a.seg[0].rows, r1s = a.seg[1].rows, r2s = a.seg[2].rows;
    auto batch_rows = [&](int b) {
        RowSrc rs;
        const int row0 = b * HU;
        const int8_t* wq = a.seg[0].wq;
        const float* ws = a.seg[0].ws;
        float* out = a.seg[0].out;
        int rows = r0s, base = 0, ops = a.seg[0].out_pos_stride;
        if (EPI == EPI_QKV) {   // batches never straddle segments (segment rows % RU == 0)
            const long long s1 = row0 >= r0s ? 1 : 0, s2 = row0 >= r0s + r1s ? 1 : 0;
            wq = wq + s1 * a.qkv_dw[0] + s2 * a.qkv_dw[1];
            ws = (const float*)((const char*)ws + s1 * a.qkv_ds[0] + s2 * a.qkv_ds[1]);
            out = (float*)((char*)out + s1 * a.qkv_do[0] + s2 * a.qkv_do[1]);
            rows = r0s + (int)s1 * (r1s - r0s) + (int)s2 * (r2s - r1s);
            ops = ops + (int)s1 * (a.seg[1].out_pos_stride - ops) +
                  (int)s2 * (a.seg[2].out_pos_stride - a.seg[1].out_pos_stride);
            base = (int)s1 * r0s + (int)s2 * r1s;
        }
        const int l0 = row0 - base;
        rs.row0 = l0;
        rs.hu = HU;
        rs.cnt = min(HU, rows - l0);
        rs.w[0] = wq + (size_t)l0 * n;
        rs.s[0] = ws + (size_t)l0 * ng;
        if (EPI == EPI_SWIGLU) {
            rs.w[1] = a.seg[1].wq + (size_t)l0 * n;
            rs.s[1] = a.seg[1].ws + (size_t)l0 * ng;
        } else {
            rs.w[1] = rs.w[0];
            rs.s[1] = rs.s[0];
        }
        rs.out = out + l0;
        rs.ops = ops;
        rs.resid = (EPI == EPI_RESID) ? rs.out[min(lane, rs.cnt - 1)] : 0.0f;
        return rs;
    };
    const bool chunks_fit = (nchunks % (64 * JU)) == 0;      // every lane of every j-tile is inside the row
    auto load_tile = [&](Tile<RU, JU>& T, const RowSrc& rs, int jt) {
        if (chunks_fit && rs.cnt == HU) {
            // exact fit (all listed models' layer shapes): base + constant strides, no clamps
            const int c0 = lane + 64 * JU * jt;
#pragma unroll
            for (int r = 0; r < RU; ++r) {
                const int part = (EPI == EPI_SWIGLU && r >= HU) ? 1 : 0;
                const int lr = r - part * HU;
                const v4i* wrow = (const v4i*)(rs.w[part] + (size_t)lr * n) + c0;
                const float* srow = rs.s[part] + (size_t)lr * ng + (c0 >> lpg_shift);
#pragma unroll
                for (int j = 0; j < JU; ++j) {
                    T.w[r][j] = __builtin_nontemporal_load(wrow + 64 * j);
                    T.sc[r][j] = __builtin_nontemporal_load(srow + ((64 * j) >> lpg_shift));
                }
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < RU; ++r) {
            const int part = (EPI == EPI_SWIGLU && r >= HU) ? 1 : 0;
            const int lr = min(r - part * HU, rs.cnt - 1);     // tail rows re-read the last live row
            const int8_t* wrow = rs.w[part] + (size_t)lr * n;
            const float* srow = rs.s[part] + (size_t)lr * ng;
#pragma unroll
            for (int j = 0; j < JU; ++j) {
                const int c = min(lane + 64 * (jt * JU + j), nchunks - 1);   // tail chunks clamp
                T.w[r][j] = __builtin_nontemporal_load((const v4i*)wrow + c);
                T.sc[r][j] = __builtin_nontemporal_load(srow + (c >> lpg_shift));
            }
        }
    };
    float racc[RU];                                  // FIN: running row sums (wave-uniform)
    auto compute_tile = [&](const Tile<RU, JU>& T, const RowSrc& rs, int jt) {
        if constexpr (FIN != 0) {
            static_assert(FIN == 0 || LPG_T == 4, "the register fold is written for group 64");
            if (jt == 0) {
#pragma unroll
                for (int r = 0; r < RU; ++r) racc[r] = -0.0f;          // Iterator::sum identity (tensor.rs:53-60)
            }
#pragma unroll
            for (int j = 0; j < JU; ++j) {
                // rows that are not a whole number of wave-loads (2560, 9728): lanes past the row end hold a re-read chunk;
                // their term is forced to +0.0, which leaves every running sum unchanged (a sum is -0.0 only before its first
                // term, and the padding comes after the row's last group)
                const int c = lane + 64 * (jt * JU + j);
                const bool cok = chunks_fit || c < nchunks;
                const int cc = chunks_fit ? c : min(c, nchunks - 1);
                v4i xv;
                float xsc;
                if constexpr (PRO == PRO_PREQR) { xv = xr[j]; xsc = xsr[j]; }
                else { xv = ((const v4i*)sm.xq)[cc]; xsc = sm.xs[cc >> 2]; }
                float t[RU], acc[RU];
#pragma unroll
                for (int r = 0; r < RU; ++r) {
                    int d = __builtin_amdgcn_sdot4(T.w[r][j].x, xv.x, 0, false);
                    d = __builtin_amdgcn_sdot4(T.w[r][j].y, xv.y, d, false);
                    d = __builtin_amdgcn_sdot4(T.w[r][j].z, xv.z, d, false);
                    d = __builtin_amdgcn_sdot4(T.w[r][j].w, xv.w, d, false);
                    d = group_sum_i32_t<4>(d);
                    t[r] = (float)d * T.sc[r][j];   // tensor.rs:59  ((dot as f32) * ws) * xs -- identical in the 4 lanes of a group
                    t[r] = t[r] * xsc;
                    t[r] = cok ? t[r] : 0.0f;
                    acc[r] = racc[r] + t[r];
                }
                // ascending-group fold as a chain of DPP adds: group g's running sum lives in lane 4g+3 and moves to lane
                // 4g+7 by row_shr:4 (row_bcast:15 across the 16-lane rows); lane 63 ends with the chunk's sum.  A hop
                // costs 17 cycles; the rows' chains are independent, so they are issued interleaved (step-major).
#pragma unroll
                for (int g = 1; g < 16; ++g) {
#pragma unroll
                    for (int r = 0; r < RU; ++r) {
                        const float prev = (g & 3) ? dpp_f<0x114>(acc[r]) : dpp_f<0x142>(acc[r]);
                        acc[r] = prev + t[r];
                    }
                }
#pragma unroll
                for (int r = 0; r < RU; ++r) racc[r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc[r]), 63));
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < JU; ++j) {
            const int c = lane + 64 * (jt * JU + j);
            const bool cok = c < nchunks;
            const int cc = min(c, nchunks - 1);
            const v4i xv = ((const v4i*)sm.xq)[cc];
            const float xsc = sm.xs[cc >> lpg_shift];
#pragma unroll
            for (int r = 0; r < RU; ++r) {
                int d = __builtin_amdgcn_sdot4(T.w[r][j].x, xv.x, 0, false);
                d = __builtin_amdgcn_sdot4(T.w[r][j].y, xv.y, d, false);
                d = __builtin_amdgcn_sdot4(T.w[r][j].z, xv.z, d, false);
                d = __builtin_amdgcn_sdot4(T.w[r][j].w, xv.w, d, false);
                d = lpg_sum<LPG_T>(d, lpg);
                const int part = (EPI == EPI_SWIGLU && r >= HU) ? 1 : 0;
                if (cok && (c & (lpg - 1)) == 0 && (r - part * HU) < rs.cnt) {
                    float t = (float)d * T.sc[r][j];   // tensor.rs:59  ((dot as f32) * ws) * xs
                    t = t * xsc;
                    term[r * ng + (c >> lpg_shift)] = t;
                }
            }
        }
    };
    unsigned long long best = 0ull;  // EPI_LOGITS running argmax key
    auto finish = [&](const RowSrc& rs) {
        if constexpr (FIN == 0) wave_lds_sync();
        if (lane < rs.cnt) {
            float acc, up = 0.0f;
            if constexpr (FIN != 0) {
                // row r's sum (wave-uniform) goes to lane r.  Each value passes through an opaque scalar: LLVM otherwise turns
                // the select chain into a lane-indexed read of racc[], i.e. a private (scratch) array (r02: 32-48 B/lane)
                auto pick = [&](int base) {
                    float v = 0.0f;
#pragma unroll
                    for (int r = 0; r < HU; ++r) {
                        int sv = __builtin_amdgcn_readfirstlane(__float_as_int(racc[base + r]));
                        asm volatile("" : "+s"(sv));
                        v = (r == 0 || lane == r) ? __int_as_float(sv) : v;
                    }
                    return v;
                };
                acc = pick(0);
                if constexpr (EPI == EPI_SWIGLU) up = pick(HU);
            } else {
                acc = Q3_DEV_ABLATE(a, 4) ? term[lane * ng] : ordered_row_sum(term + lane * ng, ng);
            }
            if (EPI == EPI_STORE) {
                rs.out[lane] = acc;
            } else if (EPI == EPI_QKV) {
                rs.out[(size_t)pos * rs.ops + lane] = acc;    // V rows go straight into the cache row of this position
            } else if (EPI == EPI_RESID) {
                rs.out[lane] = rs.resid + acc;          // ResidualConnection::forward, layers.rs:249-259
            } else if (EPI == EPI_SWIGLU) {
                float u = up;
                if constexpr (FIN == 0) u = ordered_row_sum(term + (lane + HU) * ng, ng);
                const float den = 1.0f + q3_expf_t(-acc, sm.etab);   // layers.rs:472-475
                const float sw = acc * (1.0f / den);
                rs.out[lane] = sw * u;
            } else if (EPI == EPI_LOGITS) {
                rs.out[lane] = acc;
                const unsigned long long key =
                    ((unsigned long long)total_order_key(acc) << 32) | (unsigned)(rs.row0 + lane);
                best = key > best ? key : best;
            }
        }
        if constexpr (FIN == 0) wave_lds_sync();
    };

    // ---- flat tile sequence of this wave: (batch b, tile jt), b = gw, gw+nwaves, ...
    // Loads are never issued under a data-dependent branch next to the compute that waits for them:
    // every compute_tile() sits on a path with a statically known number of younger loads, so hipcc
    // emits counted vmcnt(N) waits and the next tile streams in while the current one is reduced.
    int cb = gw, cjt = 0;
    const bool any = cb < nb;
    Tile<RU, JU> TA, TB;
    RowSrc RA, RB;
    RA = batch_rows(min(cb, nb - 1));         // (idle waves re-read a valid batch; nothing is stored)
    load_tile(TA, RA, 0);                     // ... then the first weight tile ...
    int pb = cb, pjt = 1;                     // PF: coordinates of the tile preloaded into TB
    if constexpr (PF != 0) {
        if (pjt == njt) { pjt = 0; pb = cb + nwaves; }
        // generic kernels: unconditional (static load count for the prologue's vmcnt) -- a wave without a second tile re-reads a
        // valid one.  Specialised shapes skip the request (wave-uniform): their launches have whole sets of waves with a
        // single tile (QKV of the 8B shape: 6,144 rows on 4,096 waves) and the re-reads would be a third of the traffic.
        if (pjt == 0) RB = batch_rows(min(pb, nb - 1)); else RB = RA;
        if (!kSpec || pb < nb) load_tile(TB, RB, pjt);
    }
    __builtin_amdgcn_sched_barrier(0);
    stamp(a, 1);
    if (Q3_DEV_ABLATE(a, 1)) {
        for (int i = threadIdx.x; i < (a.n >> 2); i += kWG) ((int*)sm.xq)[i] = 0x01010101;
        for (int i = threadIdx.x; i < a.n / a.group; i += kWG) sm.xs[i] = 1.0f;
        __syncthreads();
    } else {
    if (EPI == EPI_SWIGLU && threadIdx.x < 32) sm.etab[threadIdx.x] = etv;   // visible after the prologue's barrier
    if constexpr (PRO == PRO_PREQR) { if (EPI == EPI_SWIGLU) __syncthreads(); }
    else if constexpr (kPro2) pro2_finish<PRO, N_T, WGT, EPT>(a, sm, pr2);
    else gemv_prologue_finish<PRO, LPG_T>(a, sm, pr);     // ... and norm + quantize run under the weight loads
    }
    if constexpr (EPI == EPI_QKV) pos = __builtin_amdgcn_readfirstlane(pos_v);   // older than every weight load: a counted wait
    stamp(a, 2);
    if (any && !Q3_DEV_ABLATE(a, 2)) {
        bool enter_mid = false;               // PF: the loop is entered at its midpoint (current tile in TB)
        if constexpr (PF != 0) {
            compute_tile(TA, RA, cjt);        // TB's loads are younger: counted wait
            if (cjt == njt - 1) finish(RA);
            enter_mid = pb < nb;
            cb = pb; cjt = pjt;
        }
        if (PF == 0 || enter_mid) for (;;) {
            int nb_, njt_;
            if (!(PF != 0 && enter_mid)) {
                nb_ = cb; njt_ = cjt + 1;
                if (njt_ == njt) { njt_ = 0; nb_ = cb + nwaves; }
                if (nb_ >= nb) {
                    compute_tile(TA, RA, cjt);
                    stamp(a, 3);
                    if (cjt == njt - 1) finish(RA);
                    stamp(a, 4);
                    break;
                }
                if (njt_ == 0) RB = batch_rows(nb_); else RB = RA;
                load_tile(TB, RB, njt_);
                compute_tile(TA, RA, cjt);
                if (cjt == njt - 1) finish(RA);
                cb = nb_; cjt = njt_;
            }
            enter_mid = false;

            nb_ = cb; njt_ = cjt + 1;
            if (njt_ == njt) { njt_ = 0; nb_ = cb + nwaves; }
            if (nb_ >= nb) {
                compute_tile(TB, RB, cjt);
                if (cjt == njt - 1) finish(RB);
                break;
            }
            if (njt_ == 0) RA = batch_rows(nb_); else RA = RB;
            load_tile(TA, RA, njt_);
            compute_tile(TB, RB, cjt);
            if (cjt == njt - 1) finish(RB);
            cb = nb_; cjt = njt_;
        }
    }
    stamp(a, 5);
    if (EPI == EPI_LOGITS) {
        // Sampler::sample_argmax (sampler.rs:57-59): equal keys -> larger index wins == last maximum.
        // wave max -> workgroup max (LDS) -> one plain store per workgroup; k_next reduces the slots
        // (4096 same-address atomics would serialise at ~12 ns each).
        for (int m = 1; m < 64; m <<= 1) {
            const unsigned lo = __shfl_xor((unsigned)best, m);
            const unsigned hi = __shfl_xor((unsigned)(best >> 32), m);
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            best = o > best ? o : best;
        }
        unsigned long long* wred = (unsigned long long*)sm.red;
        if (lane == 0) wred[wave] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long b = wred[0];
            for (int w = 1; w < WAVES; ++w) b = wred[w] > b ? wred[w] : b;
            if (a.next_cell == nullptr) {
                a.argmax_slots[blockIdx.x] = b;
            } else {
                // k_next folded into the classifier (one launch less per token).  Only device-scope atomics carry data between
                // workgroups, so no fence is needed: the max is performed at the coherence point before its old value returns,
                // the ticket is drawn after that (data dependency on the returned value), and the workgroup that draws the
                // last ticket reads the cell with a device-scope atomic load.
                State* st = a.st;
                const unsigned long long old =
                    __hip_atomic_fetch_max(a.next_cell, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long ticket = __hip_atomic_fetch_add(a.next_cell + 1, old == ~0ull ? 2ull : 1ull,
                                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ticket == (unsigned long long)gdx - 1) {
                    const unsigned long long best_all = __hip_atomic_load(a.next_cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int idx = (int)(unsigned)(best_all & 0xffffffffull);
                    const int step = st->step;
                    if (step < a.out_cap) a.out_tokens[step] = idx;      // the sample is drawn for every forward (generation.rs:120)
                    // chat-mode prefill (generation.rs:116-123): inside the prompt the next input is the next prompt token
                    st->token = (step + 1 < st->prompt_len) ? a.prompt[step + 1] : idx;
                    st->pos = st->pos + 1;
                    st->step = step + 1;
                    st->argmax = best_all;
                    __hip_atomic_store(a.next_cell, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next token
                    __hip_atomic_store(a.next_cell + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

#ifndef Q3_GEMV_ONLY      // (q3_gemv_inst.hip compiles the GEMV kernels only)
// ------------------------------------------------------------------------------------------------
// Attention: QK-RMSNorm + RoPE (layers.rs:346-372) and GQA attention over cache rows 0..=pos
// (layers.rs:374-419).  One workgroup per query head.
// ------------------------------------------------------------------------------------------------
struct AttnArgs {
    float* q;                 // [n_heads*hd] raw projections in; normalised+rotated out (workgroup-local use)
    float* key_cache;         // layer base [seq_len][kv_dim]; row pos is WRITTEN here (normalised + rotated)
    const float* k_raw;       // [kv_dim] raw k projection of the current position (separate buffer: the head
                              // workgroups sharing a kv head all read it while one of them writes the cache row)
    const float* value_cache; // layer base
    const float* q_norm_w;    // [hd]
    const float* k_norm_w;    // [hd]
    const float* rope;        // [seq_len][hd/2][2] (cos,sin) host-built with glibc powf/cosf/sinf
    float* xb;                // [n_heads*hd] out
    float* att_global;        // [n_heads][seq_len] scratch when scores do not fit in LDS (else nullptr)
    const State* st;
    int pos_override;         // >= 0: use this pos instead of st->pos (operator-level entry point)
    int n_heads, n_kv_heads, hd, seq_len;
    int strict;
    int write_q;              // also write the normalised q back (operator-level parity)
    float* q_out;             // split path: normalised q goes here (other chunk workgroups still read the raw q)
    float* att_priv;          // split path: [n_heads][slices][att_stride] private probability rows of k_attn_out
    int att_stride;           // floats per score row (>= seq_len rounded up to 256)
    int debug;                // ablation: 8 = return right after the q/k norm+rope
    unsigned long long* stamps;   // developer timeline (block 0, thread 0)
    // batched decode: blockIdx.y = stream; float strides between streams (all 0 for the single-stream engine)
    long long sb_q, sb_kraw, sb_kv, sb_xb, sb_att;
    int tch;                  // K/V timesteps staged per LDS round (0 = attn_tch(hd))
    // batched decode, per-kv-head kernel: also emit xb quantized (tensor.rs:91-119) in the packed MFMA operand order
    int8_t* pack_q;
    float* pack_s;
    int group;
    int slice_w;              // k_attn_out: output elements per workgroup (attn_slice_w)
    int k_in_cache;           // batched prefill: row pos of the key cache was already written by k_knorm_rope
    // split path with k_attn_scores_kv: [n_heads][cmax_stride] maximum of every 64-timestep block of a score row, written
    // by the scores kernel so that k_attn_out finds the row maximum without a block-wide reduction (nullptr: not available)
    float* att_cmax;
    int cmax_stride;
    // single-stream short-context kernel: also emit xb quantized (tensor.rs:91-119, groups of `xb_group` <= 64 elements, plain
    // order) so that the Wo launch starts from int8 + scales (PRO_PREQR) instead of re-quantizing in every workgroup
    int8_t* xbq;
    float* xbs;
    int xb_group;
    int n_pos;                // k_attn_pf2 (dense prefill): positions in the block
    int att_short_form;       // host bookkeeping: 1 = keep k_attn_short where k_attn_short2 is eligible (A/B)
};

// LDS plan of k_attn (floats): q_s[hd] k_s[hd] raw[2hd] sq[2hd] opart[kWaves*hd] red[64] att[att_lds]
//                              kbuf[TCH][hd+4] vbuf[TCH][hd]
// K and V rows of a chunk of TCH timesteps are staged in LDS (K rows padded by 4 floats so that the
// per-timestep readers hit distinct banks); chunk 0 of both is requested at kernel entry, before q exists.
constexpr int kKPad = 4;
__host__ __device__ inline int attn_tch(int hd) { return hd <= 128 ? 128 : 16384 / hd; }
__host__ __device__ inline size_t attn_smem_bytes(int hd, int att_lds_floats, int tch_override = 0) {
    const int tch = tch_override > 0 ? tch_override : attn_tch(hd);
    return 4 * ((size_t)hd * (6 + kWaves) + 64 + (size_t)((att_lds_floats + 3) & ~3) + (size_t)tch * (2 * hd + kKPad));
}

// per-lane operands of the norm+RoPE of one head: lane l owns rotation pairs (l, l+64, ...) -> at most 2 pairs
// for hd <= 256.  Loaded at kernel entry so their global latency overlaps everything else.
struct RopeRegs { float w_lo[2], w_hi[2], c[2], s[2]; };
__device__ __forceinline__ void rope_regs_load(RopeRegs& r, const float* w, const float* cs, int hd) {
    const int lane = threadIdx.x & 63, half = hd >> 1;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = min(lane + 64 * u, half - 1);
        r.w_lo[u] = w[i];
        r.w_hi[u] = w[i + half];
        r.c[u] = cs[2 * i];
        r.s[u] = cs[2 * i + 1];
    }
}
// one WAVE: RMSNorm over hd raw values (LDS) followed by RoPE -> dst (LDS).  layers.rs:109-119,173-185
__device__ __forceinline__ void wave_norm_rope(float* dst, const float* src, float* sq, const RopeRegs& rr, int hd,
                                               int strict) {
    const int lane = threadIdx.x & 63;
    float ss;
    if (strict) {
        for (int i = lane; i < hd; i += 64) sq[i] = src[i] * src[i];
        wave_lds_sync();
        ss = seq_chain(-0.0f, (const v4f*)sq, hd >> 2);   // hd % 8 == 0
    } else {
        float p = 0.0f;
        for (int i = lane; i < hd; i += 64) p = p + src[i] * src[i];
        ss = group_sum_f32(p, 64);
    }
    const float f = 1.0f / sqrtf(ss / (float)hd + kEps);
    const int half = hd >> 1;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = lane + 64 * u;
        if (i < half) {
            const float xv = rr.w_lo[u] * (f * src[i]);
            const float yv = rr.w_hi[u] * (f * src[i + half]);
            const float a0 = xv * rr.c[u], b0 = yv * rr.s[u];
            const float a1 = xv * rr.s[u], b1 = yv * rr.c[u];
            dst[i] = a0 - b0;            // layers.rs:181-182
            dst[i + half] = a1 + b1;
        }
    }
}

// A chunk of <= TCH timesteps of one kv head is tch*hd <= 16384 floats = 16 float4 per thread: all 16
// loads are issued at once (one round trip) and written to LDS later, so other work overlaps the flight.
constexpr int kStageSlots = 16;
struct StageRegs { v4f v[kStageSlots]; };
// thread `tid` owns float4 column c = tid % (hd/4) of rows r0 + u*rps (u = slot): consecutive slots are a constant
// number of rows apart, so one base address + a stride replaces per-slot index arithmetic.
__device__ __forceinline__ void stage_issue(StageRegs& sr, const float* gbase, size_t kvd, int t0, int cnt, int hd) {
    const int q4s = __builtin_ctz(hd >> 2);            // hd is a power of two: float4 per row = 1 << q4s
    const int rps = kWG >> q4s;                         // rows covered by one slot of the whole workgroup
    const int r0 = (int)threadIdx.x >> q4s, c = (int)threadIdx.x & ((1 << q4s) - 1);
    // threads whose first row lies past the chunk (tiny head_dim: 128 rows per slot) re-read the chunk's last row: their
    // own row may lie past the end of the cache allocation
    const float* p = gbase + (size_t)(t0 + min(r0, max(cnt - 1, 0))) * kvd + 4 * c;
    const size_t stride = (size_t)rps * kvd;
#pragma unroll
    for (int u = 0; u < kStageSlots; ++u) {
        // unconditional (rows past the chunk re-read its first row): a per-slot branch puts every load in its own basic block
        // and hipcc then throttles the burst with conservative vmcnt waits (seen as `s_waitcnt vmcnt(6)` before every load)
        const bool ok = r0 + u * rps < cnt;
        sr.v[u] = *(const v4f*)(ok ? p + u * stride : p);
    }
}
// rows land at lds + r*ld; `skip` (absolute timestep or -1) is left untouched
__device__ __forceinline__ void stage_commit(const StageRegs& sr, float* lds, int ld, int t0, int cnt, int hd, int skip) {
    const int q4s = __builtin_ctz(hd >> 2);
    const int rps = kWG >> q4s;
    const int r0 = (int)threadIdx.x >> q4s, c = (int)threadIdx.x & ((1 << q4s) - 1);
    float* p = lds + r0 * ld + 4 * c;
#pragma unroll
    for (int u = 0; u < kStageSlots; ++u) {
        const int r = r0 + u * rps;
        if (r < cnt && t0 + r != skip) *(v4f*)(p + u * rps * ld) = sr.v[u];
    }
}

#ifdef Q3_DEV
#define ATT_STAMP(i) do { if (a.stamps != nullptr && blockIdx.x == 3 && blockIdx.y == 0 && threadIdx.x == 0) a.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ATT_STAMP(i) do { } while (0)
#endif
__device__ __forceinline__ void attn_body(const AttnArgs& a);
__global__ __launch_bounds__(kWG) void k_attn(const AttnArgs a) { attn_body(a); }
// batched decode: one grid row per stream, each with its own state, scratch rows and KV cache
__global__ __launch_bounds__(kWG) void k_attn_streams(const AttnArgs a0) {
    AttnArgs a = a0;
    const size_t sb = blockIdx.y;
    a.q = a0.q + sb * a0.sb_q;
    a.k_raw = a0.k_raw + sb * a0.sb_kraw;
    a.key_cache = a0.key_cache + sb * a0.sb_kv;
    a.value_cache = a0.value_cache + sb * a0.sb_kv;
    a.xb = a0.xb + sb * a0.sb_xb;
    if (a0.att_global) a.att_global = a0.att_global + sb * a0.sb_att;
    a.st = a0.st + sb;
    attn_body(a);
}
__device__ __forceinline__ void attn_body(const AttnArgs& a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ATT_STAMP(0);
    const int hd = a.hd;
    const int tch = a.tch > 0 ? a.tch : attn_tch(hd);
    const int kld = hd + kKPad;
    float* q_s = (float*)smem_raw;
    float* k_s = q_s + hd;
    float* raw = k_s + hd;            // [2*hd] raw q | raw k
    float* sq = raw + 2 * hd;         // [2*hd]
    float* opart = sq + 2 * hd;       // [kWaves*hd]
    float* red = opart + kWaves * hd; // [64]
    float* att_l = red + 64;
    const int att_lds = a.att_global ? 0 : a.seq_len;
    float* kbuf = att_l + ((att_lds + 3) & ~3);    // [tch][kld]
    float* vbuf = kbuf + tch * kld;   // [tch][hd]

    const int h = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kv_mul = a.n_heads / a.n_kv_heads;
    const int kvh = h / kv_mul;
    const size_t kvd = (size_t)a.n_kv_heads * hd;
    const int pos = __builtin_amdgcn_readfirstlane(a.pos_override >= 0 ? a.pos_override : a.st->pos);   // wave-uniform -> SGPR
    float* att = a.att_global ? a.att_global + (size_t)h * a.seq_len : att_l;
    const float* cs = a.rope + (size_t)pos * hd;  // hd/2 (cos,sin) pairs
    const float* kbase = a.key_cache + (size_t)kvh * hd;
    const float* vbase = a.value_cache + (size_t)kvh * hd;
    const int np = pos + 1;
    const int nch = (np + tch - 1) / tch;

    // ---- raw q head / raw k row first, then chunk 0 of K and V: everything is in flight before q exists
    float rq = 0.f, rk = 0.f;
    if (tid < hd) {
        rq = a.q[(size_t)h * hd + tid];
        rk = a.k_raw[(size_t)kvh * hd + tid];
    }
    const int cnt0 = min(tch, np);
    StageRegs sk, sv;
    RopeRegs rr;
    rope_regs_load(rr, wave == 0 ? a.q_norm_w : a.k_norm_w, cs, hd);
    __builtin_amdgcn_sched_barrier(0);
    stage_issue(sk, kbase, kvd, 0, cnt0, hd);
    stage_issue(sv, vbase, kvd, 0, cnt0, hd);
    __builtin_amdgcn_sched_barrier(0);
    ATT_STAMP(1);
    if (tid < hd) {
        raw[tid] = rq;
        raw[hd + tid] = rk;
    }
    __syncthreads();
    // ---- waves 0/1: QK-RMSNorm + RoPE of q / k (layers.rs:346-372) under the K/V loads
    if (wave == 0) wave_norm_rope(q_s, raw, sq, rr, hd, a.strict);
    else if (wave == 1) wave_norm_rope(k_s, raw + hd, sq + hd, rr, hd, a.strict);
    ATT_STAMP(2);
    stage_commit(sk, kbuf, kld, 0, cnt0, hd, pos);
    stage_commit(sv, vbuf, hd, 0, cnt0, hd, -1);
    __syncthreads();
    ATT_STAMP(3);
    float* krow = a.key_cache + (size_t)pos * kvd + (size_t)kvh * hd;
    if (h % kv_mul == 0)
        for (int i = tid; i < hd; i += kWG) krow[i] = k_s[i];   // K is normalised + rotated in place in the cache
    if (a.write_q)
        for (int i = tid; i < hd; i += kWG) a.q[(size_t)h * hd + i] = q_s[i];
    if (Q3_DEV_ABLATE(a, 8)) { if (tid < hd) a.xb[(size_t)h * hd + tid] = q_s[tid]; return; }

    const float scale = 1.0f / sqrtf((float)hd);  // (head_dim as f32).sqrt().recip()

    // ---- scores: att[t] = (q . K[t]) * scale                                  layers.rs:391-401
    for (int c = 0; c < nch; ++c) {
        const int t0 = c * tch, cnt = min(tch, np - t0);
        if (c > 0) {
            stage_issue(sk, kbase, kvd, t0, cnt, hd);
            __syncthreads();
            stage_commit(sk, kbuf, kld, t0, cnt, hd, pos);
        }
        if (pos >= t0 && pos < t0 + cnt)      // the current position's K comes from this kernel, not the cache
            for (int i = tid; i < hd; i += kWG) kbuf[(pos - t0) * kld + i] = k_s[i];
        __syncthreads();
        if (a.strict) {
            for (int t = tid; t < cnt; t += kWG) {
                const v4f* k4 = (const v4f*)(kbuf + t * kld);
                const v4f* q4 = (const v4f*)q_s;
                float dot = -0.0f;
                const int nq = hd >> 2;
                int i = 0;
                for (; i + 16 <= nq; i += 16) {          // operands first, then the sequential dot (layers.rs:395-400)
                    v4f kk[16], qq[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) { kk[u] = k4[i + u]; qq[u] = q4[i + u]; }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const v4f pr = qq[u] * kk[u];          // products are independent of the chain: packed multiplies
                        dot = dot + pr.x;
                        dot = dot + pr.y;
                        dot = dot + pr.z;
                        dot = dot + pr.w;
                    }
                }
                for (; i < nq; ++i) {
                    const v4f kv = k4[i], qv = q4[i];
                    const v4f pr = qv * kv;          // products are independent of the chain: packed multiplies
                    dot = dot + pr.x;
                    dot = dot + pr.y;
                    dot = dot + pr.z;
                    dot = dot + pr.w;
                }
                att[t0 + t] = dot * scale;
            }
        } else {
            const int lpt = hd >> 2;          // lanes per timestep (float4 each)
            const int tpw = 64 / lpt;         // timesteps per wave step  (hd <= 256)
            const int sub = lane / lpt, li = lane % lpt;
            const v4f qv = *(const v4f*)(q_s + 4 * li);
            for (int tb = wave * tpw; tb < cnt; tb += kWaves * tpw) {
                const int t = tb + sub;
                float p = 0.0f;
                if (t < cnt) {
                    const v4f kv = *(const v4f*)(kbuf + t * kld + 4 * li);
                    p = qv.x * kv.x;
                    p = p + qv.y * kv.y;
                    p = p + qv.z * kv.z;
                    p = p + qv.w * kv.w;
                }
                p = group_sum_f32(p, lpt);
                if (t < cnt && li == 0) att[t0 + t] = p * scale;
            }
        }
    }
    __syncthreads();
    ATT_STAMP(4);

    // ---- softmax                                                              layers.rs:495-506
    float m = -__builtin_inff();
    for (int t = tid; t < np; t += kWG) m = fmaxf(m, att[t]);
    m = block_max(m, red);
    float part = 0.0f;
    for (int t = tid; t < np; t += kWG) {
        const float e = q3_expf(att[t] - m);
        att[t] = e;
        part = part + e;
    }
    float sum;
    if (a.strict) {
        __syncthreads();
        const int nq4 = (((size_t)att & 15) == 0) ? (np >> 2) : 0;
        sum = seq_chain(-0.0f, (const v4f*)att, nq4);
        for (int t = nq4 << 2; t < np; ++t) sum = sum + att[t];
    } else {
        sum = block_sum_fast(part, red);
    }
    const float inv = 1.0f / sum;
    __syncthreads();
    for (int t = tid; t < np; t += kWG) att[t] = att[t] * inv;
    __syncthreads();

    ATT_STAMP(5);
    // ---- xb = sum_t att[t] * V[t]                                              layers.rs:406-417
    float* out = a.xb + (size_t)h * hd;
    float o_s = 0.0f;                       // strict: element tid (fill(0.0) then += in t order)
    v4f o_f = {0.f, 0.f, 0.f, 0.f};         // default: this lane's partial over its timesteps
    const int lpt = hd >> 2, tpw = 64 / lpt;
    const int sub = lane / lpt, li = lane % lpt;
    for (int c = 0; c < nch; ++c) {
        const int t0 = c * tch, cnt = min(tch, np - t0);
        if (c > 0) {
            stage_issue(sv, vbase, kvd, t0, cnt, hd);
            __syncthreads();
            stage_commit(sv, vbuf, hd, t0, cnt, hd, -1);
            __syncthreads();
        }
        if (a.strict) {
            if (tid < hd) {
                const float* v = vbuf + tid;
                const float* w = att + t0;
                int t = 0;
                for (; t + 16 <= cnt; t += 16) {
                    float vv[16], ww[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) { vv[u] = v[(t + u) * hd]; ww[u] = w[t + u]; }
#pragma unroll
                    for (int u = 0; u < 16; ++u) { const float p = ww[u] * vv[u]; o_s = o_s + p; }
                }
                if (t < cnt) {
                    // tail of < 16 timesteps as ONE more batch: absent terms are 0 * 0 = +0.0, and o_s + 0.0 == o_s
                    // (o_s starts from +0.0 and can never be -0.0) -- no per-term LDS round trips
                    float vv[16], ww[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const bool live = t + u < cnt;
                        const int tt = live ? t + u : t;
                        vv[u] = live ? v[tt * hd] : 0.0f;
                        ww[u] = live ? w[tt] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) { const float p = ww[u] * vv[u]; o_s = o_s + p; }
                }
            }
        } else {
            for (int tb = wave * tpw; tb < cnt; tb += kWaves * tpw) {
                const int t = tb + sub;
                if (t < cnt) {
                    const float w = att[t0 + t];
                    const v4f vv = *(const v4f*)(vbuf + t * hd + 4 * li);
                    o_f.x = o_f.x + w * vv.x;
                    o_f.y = o_f.y + w * vv.y;
                    o_f.z = o_f.z + w * vv.z;
                    o_f.w = o_f.w + w * vv.w;
                }
            }
        }
    }
    ATT_STAMP(6);
    if (a.strict) {
        if (tid < hd) out[tid] = o_s;
    } else {
        // combine the tpw sub-groups of the wave (lanes with equal li), then the waves
        for (int msk = lpt; msk < 64; msk <<= 1) {
            o_f.x += __shfl_xor(o_f.x, msk);
            o_f.y += __shfl_xor(o_f.y, msk);
            o_f.z += __shfl_xor(o_f.z, msk);
            o_f.w += __shfl_xor(o_f.w, msk);
        }
        if (sub == 0) *(v4f*)(opart + wave * hd + 4 * li) = o_f;
        __syncthreads();
        for (int i = tid; i < hd; i += kWG) {
            float r = opart[i];
            for (int w = 1; w < kWaves; ++w) r = r + opart[w * hd + i];
            out[i] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Short-context attention (pos < 256, head_dim 64 or 128): one 4-wave workgroup per query head, NO K/V staging in LDS,
// and the waves split by ROLE so that each role's registers hold only what it needs:
//   * score waves (the first 4 - HD/64): wave w owns timesteps 64w + lane (a second pass covers contexts beyond
//     64 x #score waves): lane t keeps its K row in registers (HD/4 dwordx4 loads issued at entry, in flight under the
//     norm) and walks the reference's sequential dot (layers.rs:395-400) against q broadcast from LDS; the current
//     position's key comes from this kernel (LDS).  Waves 0/1 first do the QK-RMSNorm + RoPE of q / k
//     (layers.rs:346-372);
//   * output waves (the last HD/64): lane = output element.  V[t][e] arrives by coalesced 4-byte loads into FIVE register
//     sets of 32 timesteps, all requested at kernel entry -- a context of up to 160 positions is completely in flight
//     before the scores exist (two sets were not enough: folding 32 timesteps takes ~320 cycles, an HBM round trip
//     ~2,000, and the launch period jumped from 5.2 to 7.4 us past position 64).  After the scores barrier every
//     wave reads them back 4 per lane for the max; the score waves take the exp of their own timesteps; the output waves
//     then run the softmax denominator (one chain over LDS for <= 128 timesteps, the speculative scan beyond), the
//     probabilities, and the chain o += p_t * v_t in t order (layers.rs:406-417) with p as LDS float4 bursts.
// Every sum is in the reference's order => bit-identical to k_attn / the CPU path.  Used in both modes (the default mode's
// tolerance is trivially met).  Cost of the pieces as measured in round 2 (tools/sum_probe.hip; ~7 cycles of timer overhead per step included -- a chain really advances at
// ~4.9 cycles per INSTRUCTION, tools/mfma_chain_probe.hip, round 4): a dependent v_add 10 cycles, a DPP
// hop 17, v_readlane + add 23 -- which is why the long chains read their operands from LDS/VGPRs, never cross-lane.
// ------------------------------------------------------------------------------------------------
constexpr int kShortMaxT = 256;
constexpr int kShortVSets = 5;       // register sets of 32 timesteps per output wave
#ifdef Q3_DEV
#define ATTS_STAMP(i, thr) do { if (a.stamps != nullptr && blockIdx.x == 3 && (int)threadIdx.x == (thr)) a.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ATTS_STAMP(i, thr) do { } while (0)
#endif
template <int HD>
__global__ __launch_bounds__(kWG) void k_attn_short(const AttnArgs a) {
    static_assert(HD == 64 || HD == 128, "head dims instantiated");
    constexpr int NQ4 = HD / 4;          // float4 per K row
    constexpr int HALF = HD / 2;         // rotate-half pairing (i, i + HD/2)
    constexpr int NVW = HD / 64;         // output waves (the last NVW of the workgroup)
    constexpr int NSW = 4 - NVW;         // score waves
    constexpr int TPP = 64 * NSW;        // timesteps per score pass
    __shared__ __attribute__((aligned(16))) float q_s[HD];
    __shared__ __attribute__((aligned(16))) float k_s[HD];
    __shared__ __attribute__((aligned(16))) float sq_s[2 * HD];        // squares of raw q | raw k
    __shared__ __attribute__((aligned(16))) float att[kShortMaxT];     // scores
    __shared__ __attribute__((aligned(16))) float att_e[kShortMaxT];   // exp(score - max)
    __shared__ __attribute__((aligned(16))) float att_p[kShortMaxT];   // probabilities
    __shared__ unsigned long long etab[32];                            // exp2 table of q3_expf, staged once
    ATTS_STAMP(0, 0);
    if (Q3_DEV_ABLATE(a, 16) && blockIdx.x != 3) return;      // developer: one workgroup only (launch-period experiments)
    if (Q3_DEV_ABLATE(a, 32)) return;                         // developer: empty kernel with this kernel's resources
    Q3_PIN_S(a.st); Q3_PIN_S(a.pos_override); Q3_PIN_S(a.q); Q3_PIN_S(a.k_raw); Q3_PIN_S(a.key_cache); Q3_PIN_S(a.value_cache);
    Q3_PIN_S(a.q_norm_w); Q3_PIN_S(a.k_norm_w); Q3_PIN_S(a.rope); Q3_PIN_S(a.xb); Q3_PIN_S(a.n_heads); Q3_PIN_S(a.n_kv_heads);
    Q3_PIN_S(a.write_q);

    const int h = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kv_mul = a.n_heads / a.n_kv_heads;
    const int kvh = h / kv_mul;
    const size_t kvd = (size_t)a.n_kv_heads * HD;
    // the position is REQUESTED first and turned into a scalar as late as each role allows: the raw q / k values and the norm
    // weights do not depend on it, and a v_readfirstlane right here put a full memory round trip in front of every other load
    // (r02 stamps: loads issued 2,100 cycles after entry)
    const int pos_v = a.pos_override >= 0 ? a.pos_override : a.st->pos;
    const float* kbase = a.key_cache + (size_t)kvh * HD;
    const float* vbase = a.value_cache + (size_t)kvh * HD;
    const int t4 = 4 * lane;                              // softmax read-back: lane l looks at timesteps 4l .. 4l+3

    if (wave >= NSW) {
        // ================================ output waves ================================
        const int pos = __builtin_amdgcn_readfirstlane(pos_v);   // wave-uniform -> SGPR (the V rows have slack: they are folded last)
        const int np = pos + 1;
        const int e = 64 * (wave - NSW) + lane;           // output element of this lane
        float vv[kShortVSets][32];
        auto v_issue = [&](float (&R)[32], int c) {
            // rows past the context re-read row pos (finite: written by the QKV launch); their probability is +0.0
            const float* vp = vbase + e;
#pragma unroll
            for (int u = 0; u < 32; ++u) R[u] = vp[(size_t)min(32 * c + u, pos) * kvd];
        };
        unsigned long long etv = 0ull;
        if (tid >= kWG - 32) etv = kExp2Tab[tid - (kWG - 32)];
        // A wave can have 63 vector loads outstanding: the first two sets go out now, the rest behind barrier A -- by then
        // the first loads have returned, and the score waves are not held up at the barrier by this wave's issue stalls
        // (requesting all five sets up front pushed barrier A from 4,300 to 10,000 cycles at position 130).
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (32 * c < np) v_issue(vv[c], c);           // wave-uniform
        if (tid >= kWG - 32) etab[tid - (kWG - 32)] = etv;
        __syncthreads();                                  // A: q_s / k_s / etab (this role only publishes etab)
#pragma unroll
        for (int c = 2; c < kShortVSets; ++c)
            if (32 * c < np) v_issue(vv[c], c);           // a context of <= 160 positions is all in flight before the scores exist
        __syncthreads();                                  // B: scores
        __syncthreads();                                  // C: exp(score - max)
        ATTS_STAMP(5, kWG - 64);
        // softmax denominator (layers.rs:495-506), probabilities
        const v4f e4 = ((const v4f*)att_e)[lane];
        float sum;
        if (np <= 128) {
            // one chain over the (zero padded) row: np adds at the wave's issue rate, operands streamed from LDS as float4 -- in whole
            // batches of 8 float4 (the row is +0.0 beyond the context and s + 0.0 == s once the first exp, > 0 or +0.0, is in):
            // the batched path of seq_chain keeps the next reads in flight, its remainder loop pays an LDS round trip per float4
            sum = seq_chain(-0.0f, (const v4f*)att_e, (((np + 3) >> 2) + 7) & ~7);
        } else {
            const float etot = (e4.x + e4.y) + (e4.z + e4.w);
            sum = spec_sum_lanes(etot, (np + 3) >> 2, [&](float s) { return chain4(s, e4); });
        }
        const float inv = 1.0f / sum;
        v4f p4;
        p4.x = e4.x * inv; p4.y = e4.y * inv; p4.z = e4.z * inv; p4.w = e4.w * inv;
        ((v4f*)att_p)[lane] = p4;                         // both output waves write the same values; 0 past the context
        wave_lds_sync();
        // xb = sum_t att[t] * V[t], one chain per output element in t order          layers.rs:406-417
        float o = 0.0f;
        auto fold_chunk = [&](const float (&R)[32], int c) {
            const v4f* pp = (const v4f*)att_p + 8 * c;
            v4f pq[8];
#pragma unroll
            for (int u4 = 0; u4 < 8; ++u4) pq[u4] = pp[u4];   // one burst of LDS reads, not one round trip per step
#pragma unroll
            for (int u4 = 0; u4 < 8; ++u4) {
                if (32 * c + 4 * u4 < np) {               // wave-uniform: the chain stops at the context's last float4
                    // past the context (inside the last float4) p = +0.0 and R holds the finite row pos again: the term
                    // is +-0.0 and o + (+-0.0) == o (o starts from +0.0 and is never -0.0)
                    const v4f pv = pq[u4];
                    typedef float pk2f __attribute__((ext_vector_type(2)));
                    const pk2f t01 = (pk2f){pv.x, pv.y} * (pk2f){R[4 * u4 + 0], R[4 * u4 + 1]};      // (two products per instruction)
                    const pk2f t23 = (pk2f){pv.z, pv.w} * (pk2f){R[4 * u4 + 2], R[4 * u4 + 3]};
                    o = o + t01.x;
                    o = o + t01.y;
                    o = o + t23.x;
                    o = o + t23.y;
                }
            }
        };
#pragma unroll
        for (int c = 0; c < kShortVSets; ++c)
            if (32 * c < np) fold_chunk(vv[c], c);
        // contexts beyond 160 positions (never inside the 128-token benchmark run): the remaining chunks go through set 0 / 1
        for (int c = kShortVSets; 32 * c < np; c += 2) {
            v_issue(vv[0], c);
            if (32 * (c + 1) < np) v_issue(vv[1], c + 1);
            fold_chunk(vv[0], c);
            if (32 * (c + 1) < np) fold_chunk(vv[1], c + 1);
        }
        a.xb[(size_t)h * HD + e] = o;
        if (a.xbq != nullptr) {
            // qwen3.rs:152  quantize(xb): this wave's 64 outputs are whole quantization groups (xb_group divides 64)
            const float m = group_max_f32(fabsf(o), a.xb_group);
            const float scale = m / 127.0f;
            const int qv = (scale != 0.0f) ? quant_round_i8(o / scale) : 0;
            const int idx = h * HD + e;
            a.xbq[idx] = (int8_t)qv;
            if ((idx & (a.xb_group - 1)) == 0) a.xbs[idx / a.xb_group] = scale;
        }
        ATTS_STAMP(6, kWG - 64);
        return;
    }

    // ================================ score waves ================================
    const bool is_q = wave == 0, is_k = wave == 1;
    float r_lo = 0.f, r_hi = 0.f, w_lo = 0.f, w_hi = 0.f, rc = 0.f, rs = 0.f;
    if (wave < 2) {
        const float* rawp = is_q ? a.q + (size_t)h * HD : a.k_raw + (size_t)kvh * HD;
        const int i = min(lane, HALF - 1);
        r_lo = rawp[i];
        r_hi = rawp[i + HALF];
        const float* nw = is_q ? a.q_norm_w : a.k_norm_w;
        w_lo = nw[i];
        w_hi = nw[i + HALF];
    }
    __builtin_amdgcn_sched_barrier(0);
    const int pos = __builtin_amdgcn_readfirstlane(pos_v);       // the oldest load of the wave: a counted wait
    const int np = pos + 1;
    if (wave < 2) {
        const int i = min(lane, HALF - 1);
        const float* cs = a.rope + (size_t)pos * HD;      // HD/2 (cos,sin) pairs of this position
        rc = cs[2 * i];
        rs = cs[2 * i + 1];
    }
    int t = 64 * wave + lane;                             // first pass
    v4f kr[NQ4];
    auto k_issue = [&](int tt) {
        // rows past the context re-read row pos (one cache line for all of them); row pos itself still holds whatever
        // an earlier pass left there -- both are replaced / masked below
        const v4f* kp = (const v4f*)(kbase + (size_t)min(tt, pos) * kvd);
#pragma unroll
        for (int i = 0; i < NQ4; ++i) kr[i] = kp[i];
    };
    if (64 * wave < np) k_issue(t);
    ATTS_STAMP(1, 0);

    // ---- waves 0/1: RMSNorm (layers.rs:109-119) + RoPE (layers.rs:173-185) of q / k
    if (wave < 2) {
        float* sq = sq_s + (is_q ? 0 : HD);
        if (lane < HALF) {
            sq[lane] = r_lo * r_lo;
            sq[lane + HALF] = r_hi * r_hi;
        }
        wave_lds_sync();
        const float ss = seq_chain(-0.0f, (const v4f*)sq, NQ4);      // strict left fold, layers.rs:113
        const float f = 1.0f / sqrtf(ss / (float)HD + kEps);
        if (lane < HALF) {
            const float xv = w_lo * (f * r_lo);
            const float yv = w_hi * (f * r_hi);
            const float a0 = xv * rc, b0 = yv * rs;
            const float a1 = xv * rs, b1 = yv * rc;
            const float lo = a0 - b0, hi = a1 + b1;          // layers.rs:181-182
            float* dst = is_q ? q_s : k_s;
            dst[lane] = lo;
            dst[lane + HALF] = hi;
            if (is_k && (h % kv_mul) == 0) {                  // K is normalised + rotated in place in the cache
                float* krow = a.key_cache + (size_t)pos * kvd + (size_t)kvh * HD;
                krow[lane] = lo;
                krow[lane + HALF] = hi;
            }
            if (is_q && a.write_q) {
                a.q[(size_t)h * HD + lane] = lo;
                a.q[(size_t)h * HD + lane + HALF] = hi;
            }
        }
    }
    ATTS_STAMP(2, 0);
    __syncthreads();                                      // A
    ATTS_STAMP(3, 0);

    // ---- scores: att[t] = (q . K[t]) * scale, the dot walked in index order       layers.rs:391-401
    const float scale = 1.0f / sqrtf((float)HD);
    float sc0 = -__builtin_inff(), sc1 = -__builtin_inff();
    for (int pass = 0; pass < 2; ++pass) {
        const int tb = TPP * pass + 64 * wave;            // wave-uniform first timestep of this wave in this pass
        t = tb + lane;
        if (tb >= np) {                                   // nothing to score: the slots still get their -inf
            if (t < kShortMaxT) att[t] = -__builtin_inff();
            continue;
        }
        if (pass == 1) k_issue(t);                        // contexts beyond one pass: rows requested now
        if (t == pos) {                                   // one lane of one wave: the current position's key is in LDS
#pragma unroll
            for (int i = 0; i < NQ4; ++i) kr[i] = ((const v4f*)k_s)[i];
        }
        // q arrives from LDS (broadcast reads) 8 float4 at a time, the next batch requested before the current one is folded:
        // an LDS round trip behind every float4 (r02 code: 67 waits in this loop) cost ~900 of the pass's ~2,200 cycles,
        // the 128 dependent adds of the reference's dot (layers.rs:395-400) are the other 1,280
        float dot = -0.0f;
        {
            const v4f* q4 = (const v4f*)q_s;
            v4f qa[8], qb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) qa[u] = q4[u];
#pragma unroll
            for (int b = 0; b < NQ4 / 8; ++b) {
                if (b + 1 < NQ4 / 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) qb[u] = q4[8 * (b + 1) + u];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    // two products per v_pk_mul_f32 (each rounded on its own, as layers.rs:397), then the four adds of the chain:
                    // a chain costs its instruction count x ~4.9 cycles (tools/mfma_chain_probe.hip): 7 instead of 9 per float4
                    const v4f qv = qa[u];
                    const v4f kk = kr[8 * b + u];
                    typedef float pk2f __attribute__((ext_vector_type(2)));
                    const pk2f p01 = (pk2f){qv.x, qv.y} * (pk2f){kk.x, kk.y};
                    const pk2f p23 = (pk2f){qv.z, qv.w} * (pk2f){kk.z, kk.w};
                    dot = dot + p01.x;
                    dot = dot + p01.y;
                    dot = dot + p23.x;
                    dot = dot + p23.y;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) qa[u] = qb[u];
            }
        }
        const float sc = t < np ? dot * scale : -__builtin_inff();
        if (t < kShortMaxT) att[t] = sc;                  // all 256 slots are written: -inf beyond the context
        if (pass == 0) sc0 = sc; else sc1 = sc;
    }
    __syncthreads();                                      // B
    ATTS_STAMP(4, 0);

    // ---- softmax numerators (layers.rs:495-506): the max from 4 scores per lane, exp of this wave's own timesteps
    const v4f s4 = ((const v4f*)att)[lane];
    float m = fmaxf(fmaxf(s4.x, s4.y), fmaxf(s4.z, s4.w));
    m = group_max_f32(m, 64);
    (void)t4;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int tb = TPP * pass + 64 * wave;
        const int tt = tb + lane;
        if (tt < kShortMaxT) {
            float ev = 0.0f;                              // +0.0 past the context: leaves every partial sum unchanged
            if (tb < np) {                                // wave-uniform
                const float sc = pass == 0 ? sc0 : sc1;
                ev = q3_expf_t(tt < np ? sc - m : 0.0f, etab);
                ev = tt < np ? ev : 0.0f;
            }
            att_e[tt] = ev;
        }
    }
    __syncthreads();                                      // C
}

#include "q3_attn_short2.h"

// ------------------------------------------------------------------------------------------------
// Long-context attention (pos >= the host's split threshold): the same arithmetic, in the same order, spread
// over more workgroups.  k_attn_scores: grid (heads, T-chunks) -- every chunk's dots are independent.
// k_attn_out: grid (heads, hd/32) -- softmax is recomputed per slice (cheap), the V accumulation is one
// sequential chain per output element and element slices are independent, so both stay in reference order.
// ------------------------------------------------------------------------------------------------
// output elements per k_attn_out workgroup: the widest power of two in [8, 32] that still gives every CU a workgroup
// (n_heads * hd / w >= n_cu): narrow slices stage less V per workgroup, but more than one round over the CUs loses again
__host__ __device__ inline int attn_slice_w(int hd, int n_heads, int n_cu) {
    int w = 32;
    while (w > 8 && (long)n_heads * (hd / w) < (long)n_cu) w >>= 1;
    return hd < w ? hd : w;
}
constexpr int kVChunk = 256;   // timesteps of V staged per LDS round in k_attn_out
constexpr int kVPad = 4;       // reference-order mode keeps the V chunk transposed, [element][kVChunk + kVPad]: each accumulating
                               // thread then reads its element's timesteps as float4 (4x fewer LDS reads than one per term)
constexpr int kPLds = 8192;    // probability rows up to this length live in LDS (32 KiB); longer ones go through HBM/L2

__host__ __device__ inline size_t attn_scores_smem_bytes(int hd) {
    return 4 * ((size_t)hd * 6 + 64 + (size_t)attn_tch(hd) * (hd + kKPad));
}
constexpr int kEscFloats = 64 * (64 + kSpecPad);     // 4352
__host__ __device__ inline size_t attn_out_vtile_floats(int w) {
    const size_t v = 2 * (size_t)(kVChunk + kVPad) * w;
    return v > (size_t)kEscFloats ? v : (size_t)kEscFloats;
}
__host__ __device__ inline bool attn_out_p_in_lds(int seq_len) { return ((seq_len + 255) & ~255) <= kPLds; }
__host__ __device__ inline size_t attn_out_smem_bytes(int hd, int seq_len, int w) {
    const int pl = ((seq_len + 255) & ~255) <= kPLds ? ((seq_len + 255) & ~255) : 0;
    // V / p chunk tiles double buffered; [kAoWaves][w] partials; one exp2 table per wave.  The V tiles double as the padded
    // copy of the exps for the exact sum (k_attn_out `esc`: up to 64 blocks x (64 + kSpecPad) floats), so they hold at least that
    return 4 * (attn_out_vtile_floats(w) + 2 * kVChunk + 64 + (size_t)16 * w + (size_t)pl) + 16 * 32 * 8;
}

__global__ __launch_bounds__(kWG) void k_attn_scores(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ATT_STAMP(0);
    const int hd = a.hd, tch = attn_tch(hd), kld = hd + kKPad;
    float* q_s = (float*)smem_raw;
    float* k_s = q_s + hd;
    float* raw = k_s + hd;
    float* sq = raw + 2 * hd;
    float* red = sq + 2 * hd;
    float* kbuf = red + 64;

    const int h = blockIdx.x, c = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kv_mul = a.n_heads / a.n_kv_heads, kvh = h / kv_mul;
    const size_t kvd = (size_t)a.n_kv_heads * hd;
    const int pos = __builtin_amdgcn_readfirstlane(a.pos_override >= 0 ? a.pos_override : a.st->pos);   // wave-uniform -> SGPR
    const int np = pos + 1;
    const int t0 = c * tch;
    if (t0 >= np) return;                              // chunks beyond the current position: nothing to do
    const int cnt = min(tch, np - t0);
    const bool has_pos = pos < t0 + cnt;               // the chunk that contains the current position
    const float* cs = a.rope + (size_t)pos * hd;
    const float* kbase = a.key_cache + (size_t)kvh * hd;
    float* att = a.att_global + (size_t)h * a.att_stride;

    float rq = 0.f, rk = 0.f;
    if (tid < hd) {
        rq = a.q[(size_t)h * hd + tid];
        if (has_pos) rk = a.k_raw[(size_t)kvh * hd + tid];
    }
    StageRegs sk;
    RopeRegs rr;
    rope_regs_load(rr, wave == 0 ? a.q_norm_w : a.k_norm_w, cs, hd);
    __builtin_amdgcn_sched_barrier(0);
    stage_issue(sk, kbase, kvd, t0, cnt, hd);
    __builtin_amdgcn_sched_barrier(0);
    ATT_STAMP(1);
    if (tid < hd) { raw[tid] = rq; raw[hd + tid] = rk; }
    __syncthreads();
    if (wave == 0) wave_norm_rope(q_s, raw, sq, rr, hd, a.strict);
    else if (wave == 1 && has_pos) wave_norm_rope(k_s, raw + hd, sq + hd, rr, hd, a.strict);
    ATT_STAMP(2);
    stage_commit(sk, kbuf, kld, t0, cnt, hd, pos);
    __syncthreads();
    ATT_STAMP(3);
    if (has_pos) {
        for (int i = tid; i < hd; i += kWG) kbuf[(pos - t0) * kld + i] = k_s[i];
        if (h % kv_mul == 0) {
            float* kdst = a.key_cache + (size_t)pos * kvd + (size_t)kvh * hd;
            for (int i = tid; i < hd; i += kWG) kdst[i] = k_s[i];
        }
    }
    if (a.q_out != nullptr && c == 0)
        for (int i = tid; i < hd; i += kWG) a.q_out[(size_t)h * hd + i] = q_s[i];
    __syncthreads();
    ATT_STAMP(4);
    const float scale = 1.0f / sqrtf((float)hd);
    if (a.strict) {
        for (int t = tid; t < cnt; t += kWG) {
            const v4f* k4 = (const v4f*)(kbuf + t * kld);
            const v4f* q4 = (const v4f*)q_s;
            float dot = -0.0f;
            const int nq = hd >> 2;
            int i = 0;
            for (; i + 16 <= nq; i += 16) {
                v4f kk[16], qq[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { kk[u] = k4[i + u]; qq[u] = q4[i + u]; }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const v4f pr = qq[u] * kk[u];          // products are independent of the chain: packed multiplies
                    dot = dot + pr.x;
                    dot = dot + pr.y;
                    dot = dot + pr.z;
                    dot = dot + pr.w;
                }
            }
            for (; i < nq; ++i) {
                const v4f kv = k4[i], qv = q4[i];
                const v4f pr = qv * kv;          // products are independent of the chain: packed multiplies
                dot = dot + pr.x;
                dot = dot + pr.y;
                dot = dot + pr.z;
                dot = dot + pr.w;
            }
            att[t0 + t] = dot * scale;
        }
    } else {
        const int lpt = hd >> 2, tpw = 64 / lpt;
        const int sub = lane / lpt, li = lane % lpt;
        const v4f qv = *(const v4f*)(q_s + 4 * li);
        for (int tb = wave * tpw; tb < cnt; tb += kWaves * tpw) {
            const int t = tb + sub;
            float p = 0.0f;
            if (t < cnt) {
                const v4f kv = *(const v4f*)(kbuf + t * kld + 4 * li);
                p = qv.x * kv.x;
                p = p + qv.y * kv.y;
                p = p + qv.z * kv.z;
                p = p + qv.w * kv.w;
            }
            p = group_sum_f32(p, lpt);
            if (t < cnt && li == 0) att[t0 + t] = p * scale;
        }
    }
    ATT_STAMP(5);
    ATT_STAMP(6);
}

// k_attn_scores with the staged K chunk shared by the KVM_T query heads of one kv head (head_dim 128, KVM_T 2 or 4):
// grid (kv heads, chunks of 64 * 4/KVM_T timesteps).  k_attn_scores stages the same chunk once per QUERY head, so at a
// position in the thousands every CU pulls KVM_T times the cache through L2 and the staging burst -- not the dot chains
// -- sets the launch time.  Here wave w owns query head w % KVM_T and timesteps (w / KVM_T) * 64 + lane: one sequential
// 128-term dot per lane (attention.rs:96-104 order), q broadcast from LDS, K rows read lane-per-row from the padded tile.
constexpr int kSgHd = 128;
template <int KVM_T> __host__ __device__ constexpr int sg_tch() { return 64 * (kWaves / KVM_T); }
template <int KVM_T> __host__ __device__ constexpr size_t attn_scores_kv_smem_bytes() {
    return 4 * ((size_t)kSgHd * (KVM_T + 1 + (KVM_T + 1) + kWaves) + (size_t)sg_tch<KVM_T>() * (kSgHd + kKPad));
}
template <int KVM_T>
__global__ __launch_bounds__(kWG) void k_attn_scores_kv(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ATT_STAMP(0);
    constexpr int hd = kSgHd, kld = hd + kKPad, TCH = sg_tch<KVM_T>();
    constexpr int NS = TCH * (hd / 4) / kWG;           // float4 staged per thread (8 or 16)
    constexpr int rps = kWG / (hd / 4);                // rows covered by one slot of the whole workgroup
    float* q_s = (float*)smem_raw;                     // [KVM_T][hd]
    float* k_s = q_s + KVM_T * hd;                     // [hd]
    float* raw = k_s + hd;                             // [KVM_T + 1][hd]: raw q per head | raw k
    float* sq = raw + (KVM_T + 1) * hd;                // [kWaves][hd]
    float* kbuf = sq + kWaves * hd;                    // [TCH][kld]

    const int kvh = blockIdx.x, c = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t kvd = (size_t)a.n_kv_heads * hd;
    const int pos = __builtin_amdgcn_readfirstlane(a.pos_override >= 0 ? a.pos_override : a.st->pos);
    const int np = pos + 1;
    const int t0 = c * TCH;
    if (t0 >= np) return;
    const int cnt = min(TCH, np - t0);
    const bool has_pos = pos < t0 + cnt;
    const float* cs = a.rope + (size_t)pos * hd;
    // the wave that normalises the new K row: the first one without a query head, or wave 0 after its own head
    constexpr int kwave = KVM_T < kWaves ? KVM_T : 0;

    float rv[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int idx = tid + u * kWG;                 // (KVM_T + 1) * 128 <= 640 values: raw q of each head, then raw k
        if (idx < KVM_T * hd) rv[u] = a.q[(size_t)kvh * KVM_T * hd + idx];
        else if (idx < (KVM_T + 1) * hd && has_pos) rv[u] = a.k_raw[(size_t)kvh * hd + idx - KVM_T * hd];
    }
    float rv2 = 0.f;
    if (KVM_T == 4 && tid < hd && has_pos) rv2 = a.k_raw[(size_t)kvh * hd + tid];
    RopeRegs rr, rrk;
    rope_regs_load(rr, wave < KVM_T ? a.q_norm_w : a.k_norm_w, cs, hd);
    rrk = rr;
    if (KVM_T == kWaves && wave == kwave && has_pos) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = min(lane + 64 * u, hd / 2 - 1);
            rrk.w_lo[u] = a.k_norm_w[i];
            rrk.w_hi[u] = a.k_norm_w[i + hd / 2];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    v4f sv[NS];
    {
        const int r0 = tid / (hd / 4), c4 = tid % (hd / 4);
        const float* p = a.key_cache + (size_t)kvh * hd + (size_t)(t0 + r0) * kvd + 4 * c4;
        const size_t stride = (size_t)rps * kvd;
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const bool ok = r0 + u * rps < cnt;         // unconditional loads (see stage_issue)
            sv[u] = *(const v4f*)(ok ? p + u * stride : p);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    ATT_STAMP(1);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int idx = tid + u * kWG;
        if (KVM_T == 4) { if (u == 0 || idx < KVM_T * hd) raw[idx] = rv[u]; }
        else if (idx < (KVM_T + 1) * hd) raw[idx] = rv[u];
    }
    if (KVM_T == 4 && tid < hd) raw[KVM_T * hd + tid] = rv2;
    __syncthreads();
    if (wave < KVM_T) wave_norm_rope(q_s + wave * hd, raw + wave * hd, sq + wave * hd, rr, hd, a.strict);
    if (wave == kwave && has_pos) wave_norm_rope(k_s, raw + KVM_T * hd, sq + wave * hd, rrk, hd, a.strict);
    ATT_STAMP(2);
    {
        const int r0 = tid / (hd / 4), c4 = tid % (hd / 4);
        float* p = kbuf + r0 * kld + 4 * c4;
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int r = r0 + u * rps;
            if (r < cnt && t0 + r != pos) *(v4f*)(p + u * rps * kld) = sv[u];
        }
    }
    __syncthreads();
    ATT_STAMP(3);
    if (has_pos) {
        if (tid < hd) {
            const float kv = k_s[tid];
            kbuf[(pos - t0) * kld + tid] = kv;
            a.key_cache[(size_t)pos * kvd + (size_t)kvh * hd + tid] = kv;
        }
    }
    if (a.q_out != nullptr && c == 0)
        for (int i = tid; i < KVM_T * hd; i += kWG) a.q_out[(size_t)kvh * KVM_T * hd + i] = q_s[i];
    if (has_pos) __syncthreads();
    ATT_STAMP(4);
    {
        const int j = wave % KVM_T, t = (wave / KVM_T) * 64 + lane;
        float sc = -__builtin_inff();
        if (t < cnt) {
            const v4f* k4 = (const v4f*)(kbuf + t * kld);
            const v4f* q4 = (const v4f*)(q_s + j * hd);
            float dot = -0.0f;
#pragma unroll
            for (int i = 0; i < hd / 4; i += 16) {
                v4f kk[16], qq[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { kk[u] = k4[i + u]; qq[u] = q4[i + u]; }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const v4f pr = qq[u] * kk[u];          // products are independent of the chain: packed multiplies
                    dot = dot + pr.x;
                    dot = dot + pr.y;
                    dot = dot + pr.z;
                    dot = dot + pr.w;
                }
            }
            sc = dot * (1.0f / sqrtf((float)hd));
            a.att_global[(size_t)(kvh * KVM_T + j) * a.att_stride + t0 + t] = sc;
        }
        if (a.att_cmax != nullptr) {
            // this wave's 64 timesteps are block (t0 / 64 + wave / KVM_T) of the head's row: its maximum (f32::max over the
            // same values in any order) spares k_attn_out a block-wide reduction behind its slowest wave
            const float wm = group_max_f32(sc, 64);
            if (lane == 0) a.att_cmax[(size_t)(kvh * KVM_T + j) * a.cmax_stride + (t0 >> 6) + wave / KVM_T] = wm;
        }
    }
    ATT_STAMP(5);
    ATT_STAMP(6);
}

// W_T: the slice width as a compile-time constant (8 / 16 / 32 cover every listed model; 0 = read a.slice_w): the staging
// pass count and every index derived from it fold, and the per-slot `if (u < npass)` branches disappear -- each of them
// put its load in a basic block of its own, which makes hipcc throttle the burst with conservative vmcnt waits.
// 1024 threads per workgroup: one workgroup per CU either way (256 of them), and the softmax in front of the chain -- the
// score loads and above all the exps, ~70 instructions each on the f64 pipe -- is spread over 16 waves instead of 4
constexpr int kAoThreads = 1024, kAoWaves = kAoThreads / 64;
constexpr int kAoSv = 4096 / kAoThreads;              // scores per thread kept in registers (rows up to 4096 positions)
template <int NW> __device__ __forceinline__ float block_max_n(float v, float* red) {
    v = group_max_f32(v, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) t = fmaxf(t, red[w]);
    __syncthreads();
    return t;
}
template <int NW> __device__ __forceinline__ float block_sum_fast_n(float v, float* red) {
    v = group_sum_f32(v, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += red[w];
    __syncthreads();
    return t;
}
// P_LDS: the probability row lives in LDS (contexts up to kPLds positions) -- a compile-time fact, because a pointer that
// is LDS or global at run time makes every access a FLAT instruction (slower, and it ties the LDS and vector-memory wait
// counters together).
template <int W_T, bool P_LDS>
// (developer timeline, Q3_DEV builds: 1 loads issued, 2 scores in registers, 3 max, 4 exps written, 5 exact sum, 6 end)
__global__ __launch_bounds__(kAoThreads) void k_attn_out(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ATT_STAMP(0);
    const int hd = a.hd;
    const int w = W_T ? W_T : a.slice_w;                 // slice width (power of two >= 8, or hd)
    float* vbuf0 = (float*)smem_raw;                     // 2 x [kVChunk][w]  (reference order: [w][kVChunk + kVPad])
    float* pbuf0 = vbuf0 + attn_out_vtile_floats(w);     // 2 x [kVChunk]  (behind the V tiles / the padded exps, whichever is larger)
    float* red = pbuf0 + 2 * kVChunk;                    // [64]
    float* opart = red + 64;                             // [kAoWaves][w]
    float* p_lds = opart + kAoWaves * w;                   // [npad] when the row fits (see attn_out_smem_bytes)

    const int h = blockIdx.x, sl = blockIdx.y, nsl = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kv_mul = a.n_heads / a.n_kv_heads, kvh = h / kv_mul;
    const size_t kvd = (size_t)a.n_kv_heads * hd;
    const float* src = a.att_global + (size_t)h * a.att_stride;
    const int npad_max = (a.seq_len + 255) & ~255;
    constexpr bool p_in_lds = P_LDS;                     // (host: attn_out_p_in_lds(seq_len))
    float* p;
    if constexpr (P_LDS) p = p_lds; else p = a.att_priv + ((size_t)h * nsl + sl) * a.att_stride;
    float* dummy = opart;                                // LDS word nobody reads before the epilogue: target of masked-off stores
    const float* vbase = a.value_cache + (size_t)kvh * hd + (size_t)sl * w;

    const int w4s = __builtin_ctz(w >> 2);               // float4 per slice row = 1 << w4s
    // Reference-order mode with slices of 8 / 16 elements: the chain lanes live in wave 0, so wave 0 only folds and the
    // other three waves do all the staging (a chunk still fits their 8 register slots); otherwise every thread stages.
    constexpr bool w0_folds = (W_T == 8 || W_T == 16);   // (in both modes, so that the staging shape is a compile-time fact)
    const bool stager = !w0_folds || tid >= 64;          // wave-uniform
    constexpr int nst = w0_folds ? kAoThreads - 64 : kAoThreads;   // staging threads
    const int sid = w0_folds ? max(tid - 64, 0) : tid;
    const int rps = nst >> w4s;                          // rows per staging pass
    const int npass = (kVChunk + rps - 1) / rps;         // 1 / 2 / 2 for slice widths 8 / 16 / 32; <= 8 for a whole head_dim-128 row
    const int r0 = sid >> w4s, c4 = sid & ((1 << w4s) - 1);
    float o_s = 0.0f;
    v4f o_f = {0.f, 0.f, 0.f, 0.f};
    const int lpt = w >> 2, tpw = 64 / lpt;              // default mode: lanes per timestep / timesteps per wave step
    const int sub = lane / lpt, li = lane % lpt;
    // two register sets: the V rows of chunks c+1 AND c+2 are in flight while chunk c is folded (one chunk of lookahead
    // is shorter than an HBM round trip: the fold of 256 timesteps takes ~1 us)
    struct VRegs { v4f v[8]; };
    VRegs vra, vrb;
    // every_wave: the two requests in front of the softmax are made by wave 0 as well (it discards them) -- loads behind
    // a branch make hipcc's wait for the scores conservative, i.e. a wait for the V rows
    auto v_issue = [&](VRegs& R, int c0, bool every_wave = false) {
        if (!every_wave && !stager) return;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u < npass) {                             // (compile-time for the instantiated slice widths)
                // clamped to the cache, not to the context: rows past the current position are allocated memory whose
                // contents are masked at commit time, and the address then does not wait for the position
                const int t = min(c0 + r0 + u * rps, a.seq_len - 1);
                R.v[u] = *(const v4f*)(vbase + (size_t)t * kvd + 4 * c4);
            }
        }
    };
    constexpr int K = kVChunk;

    // exp2 table of q3_expf staged in LDS (a dependent global load per exp otherwise)
    // (one copy per wave: with the block maxima below no barrier separates the staging of the table from its use)
    unsigned long long* etab = (unsigned long long*)(p_lds + (p_in_lds ? npad_max : 0)) + 32 * wave;
    if (lane < 32) etab[lane] = kExp2Tab[lane];

    // ---- softmax (layers.rs:495-506) into this workgroup's private probability row.  The scores are on the critical path
    // and go out first (16 independent loads per thread: one trip covers 4096 positions and the values then stay in
    // registers for the exp pass); the first two V chunks follow and travel under the softmax.
    // The exact sum wants power-of-two blocks of <= 64 terms in registers, one lane each, read conflict-free: besides the
    // contiguous row p[] a copy padded by 4 floats per block goes into the (still unused) first V tile.
    // Nothing above depends on the position: the score row (clamped to its allocated stride; entries past the context are
    // masked below) and the first two V chunks are requested before the position itself has arrived -- one memory round
    // trip less in front of the max.
    // The position is requested FIRST and unconditionally (operator calls pass it by value and have no state: they read a
    // word of the rope table instead): loads retire in order, so a position load behind the V rows -- or one in a basic block
    // of its own, which hipcc closes with vmcnt(0) -- would make every wave wait for its whole V prefetch before the max.
    const int* pos_ptr = a.pos_override >= 0 ? (const int*)a.rope : &a.st->pos;
    const int pos_mem = *pos_ptr;
    __builtin_amdgcn_sched_barrier(0);
    float sv[kAoSv];
#pragma unroll
    for (int u = 0; u < kAoSv; ++u) sv[u] = src[min(u * kAoThreads + tid, a.att_stride - 1)];
    // block maxima of the row from k_attn_scores_kv (one per lane covers 4096 positions); valid memory either way
    const bool have_cmax = a.att_cmax != nullptr;
    const float* cmrow = have_cmax ? a.att_cmax + (size_t)h * a.cmax_stride : src;
    float cmv = cmrow[min(lane, (have_cmax ? a.cmax_stride : a.att_stride) - 1)];
    __builtin_amdgcn_sched_barrier(0);
    v_issue(vra, 0, true);
    v_issue(vrb, K, true);                               // (row indices are clamped to the cache)
    __builtin_amdgcn_sched_barrier(0);
    ATT_STAMP(1);
    const int pos = __builtin_amdgcn_readfirstlane(a.pos_override >= 0 ? a.pos_override : pos_mem);     // wave-uniform -> SGPR
    const int np = pos + 1;
    const int npad = (np + 255) & ~255;                  // whole 64 x (npad/64) blocks for the exact sum
    int bl = 4;
    while (64 * bl < np) bl <<= 1;                       // 4 .. 64 for np <= 4096
    const bool padded = a.strict && bl <= 64;
    const int blsh = __builtin_ctz(bl);
    float* esc = vbuf0;                                  // (np/bl) x (bl + 4) floats <= kEscFloats: the V-tile region reserves that much
    float m = -__builtin_inff();
    const bool one_trip = np <= kAoSv * kAoThreads;
#pragma unroll
    for (int u = 0; u < kAoSv; ++u) sv[u] = (u * kAoThreads + tid < np) ? sv[u] : -__builtin_inff();
    if (have_cmax) {
        // every wave finds the row maximum by itself: no barrier, so a wave that was launched late only delays its own exps
        const int nblk64 = (np + 63) >> 6;
        m = lane < nblk64 ? cmv : m;
        for (int i = 64 + lane; i < nblk64; i += 64) m = fmaxf(m, cmrow[i]);     // rows beyond 4096 positions
        m = group_max_f32(m, 64);
        wave_lds_sync();                                 // this wave's exp2 table
        ATT_STAMP(2);
    } else {
#pragma unroll
        for (int u = 0; u < kAoSv; ++u) m = fmaxf(m, sv[u]);
        for (int t0 = kAoSv * kAoThreads; t0 < np; t0 += kAoSv * kAoThreads) {   // rows beyond 4096 positions
            float s2[kAoSv];
#pragma unroll
            for (int u = 0; u < kAoSv; ++u) s2[u] = src[min(t0 + u * kAoThreads + tid, np - 1)];
#pragma unroll
            for (int u = 0; u < kAoSv; ++u) m = fmaxf(m, s2[u]);
        }
        ATT_STAMP(2);
        m = block_max_n<kAoWaves>(m, red);                               // (its barriers also publish the exp2 tables)
    }
    ATT_STAMP(3);
    float part = 0.0f;
    const int nblk_terms = ((np + bl - 1) >> blsh) << blsh;
    if (one_trip) {
        // exps in groups of four (their f64 chains interleave; a guard around each one would serialise them), whole groups
        // past the padded row skipped
#pragma unroll
        for (int g = 0; g < kAoSv / 4; ++g) {
            if (g * 4 * kAoThreads < npad) {
                float ev[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = (g * 4 + u) * kAoThreads + tid;
                    const float e = q3_expf_t(t < np ? sv[g * 4 + u] - m : 0.0f, etab);
                    ev[u] = t < np ? e : 0.0f;           // +0.0 padding leaves every partial sum unchanged
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = (g * 4 + u) * kAoThreads + tid;
                    part = part + ev[u];
                    if constexpr (P_LDS) {
                        // masked-off stores go to a dummy word instead of sitting behind a branch: LLVM sinks the whole exp
                        // into a guarded block otherwise and the four chains no longer interleave
                        *(t < npad ? p + t : dummy) = ev[u];
                        *((padded && t < nblk_terms) ? esc + (t >> blsh) * (bl + kSpecPad) + (t & (bl - 1)) : dummy) = ev[u];
                    } else if (t < npad) {
                        p[t] = ev[u];
                        if (padded && t < nblk_terms) esc[(t >> blsh) * (bl + kSpecPad) + (t & (bl - 1))] = ev[u];
                    }
                }
            }
        }
    } else {
        for (int t0 = 0; t0 < npad; t0 += 4 * kAoThreads) {
            float s2[4], ev[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) s2[u] = src[min(t0 + u * kAoThreads + tid, np - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + u * kAoThreads + tid;
                const float e = q3_expf_t(t < np ? s2[u] - m : 0.0f, etab);
                ev[u] = t < np ? e : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + u * kAoThreads + tid;
                if (t < npad) {
                    part = part + ev[u];
                    p[t] = ev[u];
                    if (padded && t < nblk_terms) esc[(t >> blsh) * (bl + kSpecPad) + (t & (bl - 1))] = ev[u];
                }
            }
        }
    }
    __syncthreads();
    ATT_STAMP(4);
    float sum;
    if (a.strict) {
        if (wave == 0) {                                 // one wave walks the blocks; the other 15 would only fight it for LDS
            if (padded) sum = seq_sum_blocks(esc, (np + bl - 1) >> blsh, bl, bl + kSpecPad, nullptr);
            else sum = seq_sum_blocks(p, 64, npad >> 6, npad >> 6, nullptr);    // rows beyond 4096 positions: 64 longer blocks out of LDS
            if (tid == 0) red[0] = sum;
        }
        __syncthreads();                                 // (also: the blocks in the first V tile have been read)
        sum = red[0];
    } else sum = block_sum_fast_n<kAoWaves>(part, red);
    ATT_STAMP(5);
    const float inv = 1.0f / sum;
    // reference order: p stays unnormalised, the staging threads form (e * inv) * v themselves (same two roundings as
    // normalising the row first, layers.rs:503-505 then 406-417) -- one pass over the row and one barrier less
    if (!a.strict) {
        for (int t = tid; t < np; t += kAoThreads) p[t] = p[t] * inv;
        __syncthreads();
    }

    // ---- out[e] = sum_t p[t] * V[t][e]  (layers.rs:406-417), V slices staged kVChunk timesteps at a time
    constexpr int VLD = kVChunk + kVPad;
    auto v_commit = [&](const VRegs& R, int c0, int buf) {
        if (!stager) return;
        float* vbuf = vbuf0 + buf * (kVChunk + kVPad) * w;
        float* pbuf = pbuf0 + buf * kVChunk;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + u * rps;
            if (a.strict) {
                // transposed, rows past the context zero-filled: the accumulation below runs in whole float4 steps and
                // p = +0.0 there, so the padded terms add +0.0 (o + 0.0 == o: o is never -0.0, it starts from +0.0)
                // The tile holds the PRODUCTS p[t] * v[t][e]: the chain lanes then issue one LDS read and four adds per four
                // timesteps, which keeps the chain lane's instruction count down (with the multiplies in the chain lane
                // the fold ran at ~21 cycles per timestep).
                if (u < npass && r < kVChunk) {
                    const bool live = c0 + r < np;
                    const float pn = p[min(c0 + r, np - 1)] * inv;
                    const v4f vv = R.v[u];
                    float* dst = vbuf + (4 * c4) * VLD + r;
                    const float x0 = pn * vv.x, x1 = pn * vv.y, x2 = pn * vv.z, x3 = pn * vv.w;
                    dst[0] = live ? x0 : 0.0f;
                    dst[VLD] = live ? x1 : 0.0f;
                    dst[2 * VLD] = live ? x2 : 0.0f;
                    dst[3 * VLD] = live ? x3 : 0.0f;
                }
            } else if (u < npass && r < kVChunk && c0 + r < np) {
                *(v4f*)(vbuf + r * w + 4 * c4) = R.v[u];
            }
        }
        if (!a.strict)
            for (int t = sid; t < kVChunk; t += nst) pbuf[t] = (c0 + t < np) ? p[c0 + t] : 0.0f;
    };
    // fold chunk c0 out of LDS tile `buf` (committed one barrier earlier)
    auto fold = [&](int c0, int buf) {
        const int cnt = min(kVChunk, np - c0);
        const float* vbuf = vbuf0 + buf * (kVChunk + kVPad) * w;
        const float* pbuf = pbuf0 + buf * kVChunk;
        if (a.strict) {
            if (tid < w) {
                // one sequential chain per output element (layers.rs:406-417): the products of 4 timesteps arrive as one
                // float4 and the next 16 timesteps are in flight while the current 16 are added
                const v4f* vr = (const v4f*)(vbuf + tid * VLD);
                const int nq8 = ((cnt + 31) >> 5) << 3;          // float4 steps, whole blocks of 8 (zero padded, <= kVChunk/4)
                auto fold4 = [&](v4f x) {
                    o_s = o_s + x.x;
                    o_s = o_s + x.y;
                    o_s = o_s + x.z;
                    o_s = o_s + x.w;
                };
                v4f av[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) av[u] = vr[u];
                for (int q = 0; q < nq8; q += 8) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) bv[u] = vr[q + 4 + u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) fold4(av[u]);
                    if (q + 8 < nq8) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) av[u] = vr[q + 8 + u];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) fold4(bv[u]);
                }
            }
        } else {
            for (int tb = wave * tpw; tb < cnt; tb += kAoWaves * tpw) {
                const int t = tb + sub;
                if (t < cnt) {
                    const float wt = pbuf[t];
                    const v4f vv = *(const v4f*)(vbuf + t * w + 4 * li);
                    o_f.x = o_f.x + wt * vv.x;
                    o_f.y = o_f.y + wt * vv.y;
                    o_f.z = o_f.z + wt * vv.z;
                    o_f.w = o_f.w + wt * vv.w;
                }
            }
        }
    };
    // Two LDS tiles, one barrier per chunk: while the chain lanes fold chunk c out of one tile, every thread first commits
    // chunk c+1 (already in registers) into the other and requests chunk c+3 into the freed registers -- the staging no longer
    // sits between two barriers in front of every fold.
    v_commit(vra, 0, 0);
    if (2 * K < np) v_issue(vra, 2 * K);
    __syncthreads();
    if (w0_folds && a.strict && wave == 0) {
        // The chain wave (reference order, slices of 8 / 16): one uninterrupted stream of adds over all chunks.  It meets the
        // staging waves' per-chunk barrier 16 timesteps before the end of its chunk -- every read of the current tile has
        // been issued by then, and the staging waves, a whole chunk ahead, are already waiting there -- and uses the last
        // 16 adds to bring in the head of the next tile, so no LDS latency and no barrier sits between two chunks.
        // (Lanes >= w run along on the last row and are dropped at the store: the barrier stays outside divergent code.)
        const int row = min(tid, w - 1);
        auto fold4 = [&](v4f x) {
            o_s = o_s + x.x;
            o_s = o_s + x.y;
            o_s = o_s + x.z;
            o_s = o_s + x.w;
        };
        // Operands run 32 timesteps ahead of the adds (two sets of 8 float4): a dependent v_add_f32 issues every ~4.9 cycles
        // (tools/mfma_chain_probe.hip; the "10 cycles" of rounds 2-3 included the timer's own latency over a 64-step loop), so the
        // 16 adds that used to cover an LDS read were 78 cycles against a ~130-cycle round trip, and the chain ran at 11.3 per step.
        v4f av[8], bv[8];
        {
            const v4f* v0 = (const v4f*)(vbuf0 + row * VLD);
#pragma unroll
            for (int u = 0; u < 8; ++u) av[u] = v0[u];
        }
        for (int c0 = 0, buf = 0; c0 < np; c0 += K, buf ^= 1) {
            const int cnt = min(K, np - c0);
            const int nq16 = ((cnt + 63) >> 6) << 4;         // float4 steps, whole blocks of 16 (zero padded, <= K/4)
            const v4f* vr = (const v4f*)(vbuf0 + buf * (K + kVPad) * w + row * VLD);
            const v4f* vn = (const v4f*)(vbuf0 + (buf ^ 1) * (K + kVPad) * w + row * VLD);
            int q = 0;
            for (; q + 16 < nq16; q += 16) {                 // every 64-timestep block but the chunk's last: one basic block
                // the chain is bound by the wave's issue rate (one instruction per ~4.9 cycles): a burst of 8 reads, ONE wait for the
                // set requested a phase ago, 32 adds -- 1.28 instructions per timestep (a read + a wait behind every fourth add: 1.5)
                // (the explicit lgkmcnt(8) = "everything but the burst just issued has landed": without it hipcc waits once per float4)
#pragma unroll
                for (int u = 0; u < 8; ++u) bv[u] = vr[q + 8 + u];
                __builtin_amdgcn_s_waitcnt(0xC87F);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) fold4(av[u]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) av[u] = vr[q + 16 + u];
                __builtin_amdgcn_s_waitcnt(0xC87F);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) fold4(bv[u]);
                __builtin_amdgcn_sched_barrier(0);
            }
            {
#pragma unroll
                for (int u = 0; u < 8; ++u) { bv[u] = vr[q + 8 + u]; fold4(av[u]); }
                __syncthreads();                             // the staging waves' barrier of this chunk: tile c+1 is complete
                if (c0 + K < np) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) av[u] = vn[u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) fold4(bv[u]);
            }
        }
    } else
    for (int c0 = 0; c0 < np; c0 += 2 * K) {
        if (c0 + K < np) {
            v_commit(vrb, c0 + K, 1);
            if (c0 + 3 * K < np) v_issue(vrb, c0 + 3 * K);
        }
        fold(c0, 0);
        __syncthreads();
        if (c0 + K >= np) break;
        if (c0 + 2 * K < np) {
            v_commit(vra, c0 + 2 * K, 0);
            if (c0 + 4 * K < np) v_issue(vra, c0 + 4 * K);
        }
        fold(c0 + K, 1);
        __syncthreads();
    }
    ATT_STAMP(6);
    float* out = a.xb + (size_t)h * hd + (size_t)sl * w;
    if (a.strict) {
        if (tid < w) out[tid] = o_s;
    } else {
        for (int msk = lpt; msk < 64; msk <<= 1) {
            o_f.x += __shfl_xor(o_f.x, msk);
            o_f.y += __shfl_xor(o_f.y, msk);
            o_f.z += __shfl_xor(o_f.z, msk);
            o_f.w += __shfl_xor(o_f.w, msk);
        }
        __syncthreads();
        if (sub == 0) *(v4f*)(opart + wave * w + 4 * li) = o_f;
        __syncthreads();
        for (int i = tid; i < w; i += kAoThreads) {
            float r = opart[i];
            for (int ww = 1; ww < kAoWaves; ++ww) r = r + opart[ww * w + i];
            out[i] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Bookkeeping: consume the argmax cell, advance (token, pos, step)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void k_next(State* st, const unsigned long long* slots, int nslots, int32_t* out_tokens,
                                              int out_cap, const int32_t* prompt) {
    __shared__ unsigned long long red[kWaves];
    unsigned long long best = 0ull;
    for (int i = threadIdx.x; i < nslots; i += kWG) best = slots[i] > best ? slots[i] : best;
    for (int m = 1; m < 64; m <<= 1) {
        const unsigned lo = __shfl_xor((unsigned)best, m);
        const unsigned hi = __shfl_xor((unsigned)(best >> 32), m);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWaves; ++w) best = red[w] > best ? red[w] : best;
        const int idx = (int)(unsigned)(best & 0xffffffffull);
        const int step = st->step;
        if (step < out_cap) out_tokens[step] = idx;       // the sample is drawn for every forward (generation.rs:120)
        // chat-mode prefill (generation.rs:116-123): inside the prompt the next input is the next prompt token and
        // the sample is discarded; afterwards the sample is fed back (generation.rs:143-147)
        st->token = (step + 1 < st->prompt_len) ? prompt[step + 1] : idx;
        st->pos = st->pos + 1;
        st->step = step + 1;
        st->argmax = best;
    }
}

// ------------------------------------------------------------------------------------------------
// Stand-alone operator kernels (operator-level C ABI; same device functions as above)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void k_op_quantize(int8_t* q, float* s, const float* x, int n, int group) {
    // one workgroup; grid-strides over float4 slots
    const int nv = n >> 2, glanes = group >> 2;
    for (int v0 = 0; v0 < nv; v0 += kWG) {
        const int v = v0 + threadIdx.x;
        const bool valid = v < nv;
        v4f y = {0.f, 0.f, 0.f, 0.f};
        if (valid) y = ((const v4f*)x)[v];
        quantize4_to_lds(y, v, glanes, valid, q, s);
    }
}

__global__ void k_op_dequantize(const int8_t* q, const float* s, float* x, size_t n, int group) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        x[i] = (float)q[i] * s[i / (size_t)group];   // tensor.rs:76-79
}

__global__ __launch_bounds__(kWG) void k_op_rmsnorm(float* out, const float* in, const float* w, int n, int strict) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sq = (float*)smem_raw;               // term_floats(n)
    float* red = sq + term_floats(n);
    float part = 0.0f;
    for (int i = threadIdx.x; i < n; i += kWG) {
        const float v = in[i];
        sq[term_index(i, n)] = v * v;
        part = part + v * v;
    }
    float ss;
    if (strict) {
        __syncthreads();
        ss = seq_sum_terms(sq, n);
    } else {
        ss = block_sum_fast(part, red);
    }
    const float f = 1.0f / sqrtf(ss / (float)n + kEps);
    for (int i = threadIdx.x; i < n; i += kWG) out[i] = w[i] * (f * in[i]);
}

__global__ __launch_bounds__(kWG) void k_op_softmax(float* x, int n, int strict) {
    __shared__ float red[64];
    float m = -__builtin_inff();
    for (int t = threadIdx.x; t < n; t += kWG) m = fmaxf(m, x[t]);
    m = block_max(m, red);
    float part = 0.0f;
    for (int t = threadIdx.x; t < n; t += kWG) {
        const float e = q3_expf(x[t] - m);
        x[t] = e;
        part = part + e;
    }
    float sum;
    if (strict) {
        __syncthreads();
        sum = -0.0f;
        for (int t = 0; t < n; ++t) sum = sum + x[t];
    } else {
        sum = block_sum_fast(part, red);
    }
    const float inv = 1.0f / sum;
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += kWG) x[t] = x[t] * inv;
}

__global__ void k_op_swiglu(float* hb, const float* hb2, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float g = hb[i];
        const float den = 1.0f + q3_expf(-g);
        const float sw = g * (1.0f / den);
        hb[i] = sw * hb2[i];
    }
}

__global__ void k_op_expf(float* x, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        x[i] = q3_expf(x[i]);
}

__global__ __launch_bounds__(kWG) void k_op_argmax(const float* logits, size_t n, unsigned long long* cell) {
    unsigned long long best = 0ull;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long key = ((unsigned long long)total_order_key(logits[i]) << 32) | (unsigned)i;
        best = key > best ? key : best;
    }
    for (int m = 1; m < 64; m <<= 1) {
        const unsigned lo = __shfl_xor((unsigned)best, m);
        const unsigned hi = __shfl_xor((unsigned)(best >> 32), m);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0 && best != 0ull) atomicMax(cell, best);
}

#endif  // Q3_GEMV_ONLY

}  // namespace q3
