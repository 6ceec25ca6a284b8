// q3_gemv.h -- gfx950 (MI355X) device code for the Qwen3 Q8 decode hot path, part 1: shared helpers and the W8A8 GEMV kernel
// (k_gemv).  Part 2 (attention, operators) is q3_kernels.h, which includes this file; q3_gemv_inst.hip compiles this part alone.
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
// glibc's main path alone: THE result for |x| < 88 (q3_expf_special says which inputs those are); a caller that has checked a whole
// batch skips the special-case selects of q3_expf_t (k_attn_out's exp phase is bound by its instruction count)
__device__ __forceinline__ bool q3_expf_special(float x) { return ((__float_as_uint(x) >> 20) & 0x7ffu) >= 0x42bu; }   // |x| >= 88 or not finite
__device__ __forceinline__ float q3_expf_main(float x, const unsigned long long* tab) {
    constexpr double kInvLn2N = 0x1.71547652b82fep+0 * 32.0;
    constexpr double kShift = 0x1.8p+52;
    constexpr double kC0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
    constexpr double kC1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
    constexpr double kC2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    const double xd = (double)x;
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
    return (float)y;
}
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
// one exp per lane on a latency path: the special-case selects (a sixth of the instructions) only when some active lane needs them
__device__ __forceinline__ float q3_expf_wave(float x, const unsigned long long* tab) {
    if (__builtin_expect(__any(q3_expf_special(x)), 0)) return q3_expf_t(x, tab);
    return q3_expf_main(x, tab);
}

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
    // EPI_QKV: transposed copy of this layer's value cache, [kv_dim][seq_len] (nullptr: none).  The long-context output kernel walks
    // 16-element slices of the value rows over thousands of timesteps: in the row-major cache that is a 64-byte piece every 4 KB,
    // here a contiguous run per element (docs/HISTORY.md section 0, round 5).  The value row of the current position is written to both.
    float* v_t;
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
// Developer, kernel-duration mode (debug bit 64, Q3_KSTAMPS=1): lane 0 of EVERY wave stores its entry / exit time -- s_memrealtime,
// the 100 MHz constant clock the whole chip shares -- into its own slot pair of the launch (plain stores: same-address atomics
// from 4,096 waves serialise at ~10 ns each and made the stamped run 7x slower).  Behind the token's last launch
// k_kstamp_reduce takes the minimum begin / maximum end of every launch and re-arms the slots, k_kstamp_fold adds duration and
// gap-to-predecessor to per-launch sums: begin and end of every launch as the hardware ran it under hipGraph replay, where
// rocprofv3 of this image cannot trace.
constexpr int kKstampSlots = 16384;      // wave slots per launch (1,024 workgroups x 16 waves)
#ifdef Q3_DEV
// KSTAMP_BEGIN reads the clock into scalar registers (no wait: the value is first USED at the exit), KSTAMP_END stores both
// times from lane 0 of the wave: the instrumented launch does not stall at entry.
__device__ __forceinline__ void kstamp_store(unsigned long long* st, int debug, unsigned long long t0) {
    if ((debug & 64) != 0 && st != nullptr && (threadIdx.x & 63) == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        const unsigned slot = ((blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (kKstampSlots - 1);
        st[2 * slot] = t0;
        st[2 * slot + 1] = t1;
    }
}
#define KSTAMP_BEGIN(a) const unsigned long long kst0_ = __builtin_amdgcn_s_memrealtime()
#define KSTAMP_END(a) kstamp_store((a).stamps, (a).debug, kst0_)
#else
#define KSTAMP_BEGIN(a) do { } while (0)
#define KSTAMP_END(a) do { } while (0)
#endif

__device__ __forceinline__ void stamp(const GemvArgs& a, int idx) {
#ifdef Q3_DEV
    if (a.stamps != nullptr && (a.debug & 64) == 0 && (int)blockIdx.x == a.stamp_block && threadIdx.x == 0) {
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
    static_assert(N_T == 0 || (LPG_T == 4 && FIN >= 1), "specialised shapes: group 64, register fold");
    static_assert(PRO != PRO_PREQR || (N_T > 0 && (N_T + 1023) / 1024 == JU), "register-direct xq: one tile per row");
    constexpr int WAVES = WGT / 64;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    stamp(a, 0);
    KSTAMP_BEGIN(a);
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
                if constexpr (FIN == 2) {
                    // tolerance mode (Q3_FLAG_FAST): the chunk's 16 group terms as a wavefront TREE -- two mirror steps inside the
                    // 16-lane rows (a term already sits in the 4 lanes of its group), two row broadcasts; the chunk total of lane
                    // 63 joins the running row sum.  4 hops instead of 15; NOT the reference's order (tensor.rs:53-60).
#pragma unroll
                    for (int r = 0; r < RU; ++r) {
                        float v = t[r];
                        v += dpp_f<0x141>(v);
                        v += dpp_f<0x140>(v);
                        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));
                        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));
                        racc[r] = racc[r] + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
                    }
                    continue;
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
                if (rs.ops != 0 && a.v_t != nullptr) a.v_t[(size_t)(rs.row0 + lane) * a.seq_len + pos] = acc;     // ... and into its transposed copy
            } else if (EPI == EPI_RESID) {
                rs.out[lane] = rs.resid + acc;          // ResidualConnection::forward, layers.rs:249-259
            } else if (EPI == EPI_SWIGLU) {
                float u = up;
                if constexpr (FIN == 0) u = ordered_row_sum(term + (lane + HU) * ng, ng);
                const float den = 1.0f + q3_expf_wave(-acc, sm.etab);   // layers.rs:472-475
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
                // k_next folded into the classifier (one launch less per token).  The maximum travels in one atomic cell, the
                // arrival count in another.  Only device-scope read-modify-writes on these two words carry data between the
                // workgroups: the max is performed at the coherence point before its old value returns, the ticket is drawn
                // after that (its operand depends on the returned value), and the workgroup that draws the last ticket reads the
                // cell with a device-scope atomic load -- relaxed orders suffice on this hardware.  The formally ordered form
                // (-DQ3_TICKET_ACQ_REL: acquire + release on the ticket, i.e. an L2 write-back + invalidate around the ticket
                // of each of the ~590 classifier workgroups) was measured in round 5: +1.4 ... +4.8 us per token on the 0.6B
                // device loop in three alternations (profiles/r05_ticket_order.txt), above the 0.5 us the review allowed
                // for it, so relaxed stays the product's form.
#ifdef Q3_TICKET_ACQ_REL
                constexpr int kTicketOrder = __ATOMIC_ACQ_REL;
#else
                constexpr int kTicketOrder = __ATOMIC_RELAXED;
#endif
                State* st = a.st;
                const unsigned long long old =
                    __hip_atomic_fetch_max(a.next_cell, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long ticket = __hip_atomic_fetch_add(a.next_cell + 1, old == ~0ull ? 2ull : 1ull,
                                                                         kTicketOrder, __HIP_MEMORY_SCOPE_AGENT);
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
    KSTAMP_END(a);
}


}  // namespace q3
