// Batched decode (BASELINE configs[3]: B <= 32 independent streams advance one token per step).
//
// The weights are read ONCE per step for all streams: the W8A8 group-quant matmul of tensor.rs:23-62 becomes a
// [rows x K] x [K x B] product on the int8 matrix cores.  One v_mfma_i32_16x16x64_i8 produces the exact i32 dots of
// 16 weight rows x 16 streams over 64 contraction bytes; a quantization group (G = 64 * NJ bytes) is NJ chained
// MFMAs.  Lane l then owns rows 4*(l/16)+i (i = 0..3) of stream l%16 and folds their per-group f32 terms
// ((f32)idot * ws) * xs itself, g ascending, starting from -0.0 -- the same operations in the same order as the
// single-stream GEMV (and the reference), so every stream's logits are bit-identical to its single-stream run.
//
// Memory layout (HBM, 288 GB: a second, MFMA-ordered copy of the weights is cheap).  A row tile = 16 consecutive
// weight rows.  Packed int8: [tile][g][j < NJ][lane 0..63][16 B] where lane (r = l%16, q = l/16) holds bytes
// [g*G + (q*NJ + j)*16, +16) of row r -- a wave's dwordx4 load is 1 KiB contiguous and a tile is one sequential
// 16*K-byte stream.  Packed scales: [tile][g][16 rows] f32.  The quantized activations of the step use the same
// order with streams in place of rows: [stream tile][g][j][lane][16 B] and [stream tile][g][16 streams].
#pragma once
#include "q3_kernels.h"

// developer ablation of the batched matmul, compile-time only (-DQ3_BABLATE=bits): 1 no B loads, 4 no term
// formation / LDS term writes, 8 no A loads, 16 no fold.  Runtime switches would make the loads conditional and change what is being measured.
#ifdef Q3_BABLATE
#define Q3_BABL(bit) ((Q3_BABLATE & (bit)) != 0)
#else
#define Q3_BABL(bit) false
#endif

#ifdef Q3_DEV
#define BG_STAMP(i) do { if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && sidx < 12) a.stamps[sidx * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BG_STAMP(i) do { } while (0)
#endif

namespace q3 {

constexpr int kMaxStreams = 32;

// ------------------------------------------------------------------------------------------------
// One-time repack of a [rows][n] int8 matrix (+ [rows][n/G] scales) into row-tile order.
// Destination tile of source tile t is tile0 + t * tile_stride (q|k|v concatenated; w1/w3 interleaved).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void k_pack_weights(const int8_t* __restrict__ wq, const float* __restrict__ ws,
                                                      int8_t* __restrict__ pq, float* __restrict__ ps, int rows, int n,
                                                      int G, int tile0, int tile_stride) {
    const int ng = n / G, nj = G >> 6;
    const size_t pieces = (size_t)rows * (size_t)(n >> 4);
    for (size_t p = (size_t)blockIdx.x * kWG + threadIdx.x; p < pieces; p += (size_t)gridDim.x * kWG) {
        // destination-major enumeration: p = ((t*ng + g)*nj + j)*64 + lane
        const int lane = (int)(p & 63);
        size_t rest = p >> 6;
        const int j = (int)(rest % nj);
        rest /= nj;
        const int g = (int)(rest % ng);
        const size_t t = rest / ng;
        const int r = lane & 15, q = lane >> 4;
        const size_t src = (t * 16 + r) * (size_t)n + (size_t)g * G + (size_t)(q * nj + j) * 16;
        const size_t dt = (size_t)tile0 + t * (size_t)tile_stride;
        const size_t dst = (((dt * ng + g) * nj + j) * 64 + lane) * 16;
        *(v4i*)(pq + dst) = *(const v4i*)(wq + src);
    }
    const size_t nsc = (size_t)rows * ng;
    for (size_t p = (size_t)blockIdx.x * kWG + threadIdx.x; p < nsc; p += (size_t)gridDim.x * kWG) {
        const int r = (int)(p & 15);
        size_t rest = p >> 4;
        const int g = (int)(rest % ng);
        const size_t t = rest / ng;
        const size_t dt = (size_t)tile0 + t * (size_t)tile_stride;
        ps[(dt * ng + g) * 16 + r] = ws[(t * 16 + r) * (size_t)ng + g];
    }
}

// ------------------------------------------------------------------------------------------------
// Per-stream activation prologue: exactly the GEMV prologue (embedding / RMSNorm / quantize, same code), one
// workgroup per stream, result written in the packed MFMA operand order.
// ------------------------------------------------------------------------------------------------
struct BQuantArgs {
    long long in_stride;     // floats between consecutive streams of GemvArgs::in
    long long x_out_stride;  // PRO_EMBED_NORM: floats between streams of x_out
    int8_t* xq_p;            // packed int8 activations
    float* xs_p;             // packed activation scales
    int n_streams;
};

// N_T > 0 (round 6): the listed models' vector lengths at group 64 as compile-time constants -- the generic prologue's slot counts,
// term_index and group bookkeeping and the pack loop's operand address are run-time integer divisions otherwise (a dense prefill
// block runs one of these workgroups per position: 2,048 per launch, instruction bound).
template <int PRO, int N_T = 0>
__global__ __launch_bounds__(kWG) void k_bquant(const GemvArgs a0, const BQuantArgs b) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr bool kStage = (PRO == PRO_NORM || PRO == PRO_EMBED_NORM);
    const int sidx = blockIdx.y;                 // blockIdx.x == 0: the prologue's "workgroup 0" side outputs are per stream
    GemvArgs a = a0;
    if constexpr (N_T > 0) { a.n = N_T; a.group = 64; }
    a.in = a0.in ? a0.in + (size_t)sidx * b.in_stride : nullptr;
    a.st = a0.st + sidx;
    a.x_out = a0.x_out ? a0.x_out + (size_t)sidx * b.x_out_stride : nullptr;
    a.tap_out = nullptr;
    const GemvSmem sm = gemv_carve(smem_raw, a.n, a.group, 1, kStage);
    ProRegs<PRO> pr;
    gemv_prologue_issue<PRO>(a, pr);
    gemv_prologue_finish<PRO, (N_T > 0 ? 4 : 0)>(a, sm, pr);     // ends with __syncthreads(): sm.xq / sm.xs complete
    const int n = a.n, G = a.group, ng = n / G, nj = G >> 6;
    const int nt = sidx >> 4, s = sidx & 15;
    for (int p = threadIdx.x; p < (n >> 4); p += kWG) {
        const int k0 = p << 4;
        const int g = k0 / G, within = (k0 % G) >> 4;      // 16-byte piece index inside the group: q*nj + j
        const int q = within / nj, j = within % nj;
        const size_t dst = ((((size_t)nt * ng + g) * nj + j) * 64 + (q * 16 + s)) * 16;
        *(v4i*)(b.xq_p + dst) = ((const v4i*)sm.xq)[p];
    }
    for (int g = threadIdx.x; g < ng; g += kWG) b.xs_p[((size_t)nt * ng + g) * 16 + s] = sm.xs[g];
}

// The same prologue spread over `gridDim.x` workgroups per stream (PRO_NORM / PRO_QUANT).  The exact sum of squares is
// the only part that needs the whole vector -- every part recomputes it (one 32-workgroup launch was latency-bound:
// 8-10 us for a few hundred KB) -- while the normalise / divide / round / pack work, which is instruction-issue bound
// (~110 instructions per float4), is cut into slices of whole quantization groups.  Same arithmetic per element as
// k_bquant, so the packed operands are identical.
#ifdef Q3_DEV
#define BQ_STAMP(i) do { if (a.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BQ_STAMP(i) do { } while (0)
#endif
// N_T / PARTS_T > 0 (round 6): vector length and parts per stream at compile time, group 64 -- the listed models' shapes.  The generic
// form computes term_index (two divisions by the block length per float4), the quantizer's group bookkeeping and the packed
// operand address (four more) with run-time integer divisions, ~40 instructions each on this ISA: in-kernel stamps put 1,350
// cycles on the pack loop alone and ~1,000 on the quantize step of ONE float4 (profiles/r06_bquant_stamps.txt).
template <int PRO, int N_T = 0, int PARTS_T = 0>
__global__ __launch_bounds__(kWG) void k_bquant_split(const GemvArgs a, const BQuantArgs b) {
    static_assert(PRO == PRO_NORM || PRO == PRO_QUANT, "embedding rows keep the single-workgroup kernel");
    static_assert((N_T == 0) == (PARTS_T == 0) && (N_T == 0 || (N_T % (PARTS_T * 64)) == 0), "whole groups per part");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int sidx = blockIdx.y, part = blockIdx.x, nparts = PARTS_T ? PARTS_T : (int)gridDim.x;
    const int tid = threadIdx.x;
    BQ_STAMP(0);
    const int n = N_T ? N_T : a.n, G = N_T ? 64 : a.group, nv = n >> 2, glanes = G >> 2;
    const int nvp = nv / nparts;                       // float4 slots of this part (a whole number of groups: host)
    const int v0 = part * nvp;
    const v4f* x4 = (const v4f*)(a.in + (size_t)sidx * b.in_stride);
    char* base = smem_raw;
    float f = 1.0f;
    // this part's slots are requested first, then (PRO_NORM) the whole vector for the sum of squares
    const int nk = (nvp + kWG - 1) / kWG;              // <= 2 for every listed shape
    v4f xv[2], wv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int v = v0 + min(k * kWG + tid, nvp - 1);
        xv[k] = x4[v];
        if (PRO == PRO_NORM) wv[k] = ((const v4f*)a.norm_w)[v];
    }
    if (PRO == PRO_NORM) {
        float* sq = (float*)base;                      // term_floats(n) squares in the exact-sum layout
        float* red = sq + term_floats(n);              // 64 floats + 64 approximate block totals
        base += 4 * (size_t)(term_floats(n) + 128);
        // specialised shapes whose exact-sum block is 4 / 8 / 16 consecutive float4 (= consecutive threads): the approximate block
        // totals the sum wave starts from are formed here by all four waves (a 4-step DPP sum) instead of by the sum wave alone
        // (64 dependent-ish adds, ~300 cycles in front of its first round)
        constexpr int LPB = N_T > 0 ? spec_blen(N_T > 0 ? N_T : 1024) >> 2 : 0;
        constexpr bool kApprox = N_T > 0 && spec_ok(N_T > 0 ? N_T : 1024) && (LPB == 4 || LPB == 8 || LPB == 16) && ((N_T >> 2) % kWG) == 0;
        float part_sum = 0.0f;
        for (int v = tid; v < nv; v += kWG) {
            const v4f t = x4[v];
            v4f q2;
            q2.x = t.x * t.x; q2.y = t.y * t.y; q2.z = t.z * t.z; q2.w = t.w * t.w;   // layers.rs:113
            *(v4f*)(sq + term_index(4 * v, n)) = q2;
            const float p4 = sumsq4(t);
            part_sum = part_sum + p4;
            if constexpr (kApprox) {
                const float bt = group_sum_f32(p4, LPB);
                if ((v & (LPB - 1)) == 0) red[64 + v / LPB] = bt;
            }
        }
        float ss;
        if (a.strict) {
            BQ_STAMP(1);
            __syncthreads();
            BQ_STAMP(2);
            if (tid < 64) {                                // (one wave: see gemv_prologue_finish)
                ss = seq_sum_terms(sq, n, kApprox ? red + 64 : nullptr);
                if (tid == 0) red[0] = ss;
            }
            BQ_STAMP(3);
            __syncthreads();
            ss = red[0];
        } else {
            ss = block_sum_fast(part_sum, red);
        }
        f = 1.0f / sqrtf(ss / (float)n + kEps);
    }
    int8_t* xq = (int8_t*)base;                        // this part's quantized slice
    float* xs = (float*)(base + align16((size_t)nvp * 4));
    for (int k = 0; k < nk; ++k) {
        const int vl = k * kWG + tid;                  // slot inside the part
        const bool valid = vl < nvp;
        v4f y = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            v4f t = xv[0], w = wv[0];
            if (k == 1) { t = xv[1]; w = wv[1]; }
            if (k >= 2) { t = x4[v0 + vl]; if (PRO == PRO_NORM) w = ((const v4f*)a.norm_w)[v0 + vl]; }
            y = (PRO == PRO_NORM) ? norm4(w, f, t) : t;
        }
        if constexpr (N_T > 0) quantize4_to_lds<16>(y, vl, 16, valid, xq, xs);
        else quantize4_to_lds(y, vl, glanes, valid, xq, xs);
    }
    BQ_STAMP(4);
    __syncthreads();
    BQ_STAMP(5);
    // packed MFMA operand order (see k_bquant): 16-byte pieces of this part
    const int ng = n / G, nj = G >> 6;
    const int nt = sidx >> 4, s = sidx & 15;
    for (int pl = tid; pl < (nvp >> 2); pl += kWG) {
        const int p = (v0 >> 2) + pl;
        const int k0 = p << 4;
        const int g = k0 / G, within = (k0 % G) >> 4;
        const int q = within / nj, j = within % nj;
        const size_t dst = ((((size_t)nt * ng + g) * nj + j) * 64 + (q * 16 + s)) * 16;
        *(v4i*)(b.xq_p + dst) = ((const v4i*)xq)[pl];
    }
    const int g0 = (v0 * 4) / G, ngp = (nvp * 4) / G;
    for (int gl = tid; gl < ngp; gl += kWG) b.xs_p[((size_t)nt * ng + g0 + gl) * 16 + s] = xs[gl];
    BQ_STAMP(6);
#ifdef Q3_DEV
    if (a.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.stamps[15] = 0xB0ull + (unsigned)PRO;
#endif
}

// ------------------------------------------------------------------------------------------------
// Batched W8A8 matmul on the matrix cores.
// ------------------------------------------------------------------------------------------------
struct BGemmArgs {
    const int8_t* wq;        // packed row tiles
    const float* ws;         // packed row-tile scales
    const int8_t* xq;        // packed activations of this step
    const float* xs;
    int ng;                  // groups per row
    int ntiles;              // row tiles in the packed matrix (SwiGLU: w1 and w3 tiles interleaved)
    int n_streams;
    float* out0;             // [stream][rows] destination (RESID: residual stream x; LOGITS: logits or nullptr)
    long long out0_stride;
    float* out1;             // QKV: k_raw [stream][kv_dim]
    long long out1_stride;
    float* out2;             // QKV: value cache of this layer, stream stride out2_stride, row st[b].pos
    long long out2_stride;
    int rows0, rows1;        // QKV: q rows, k rows
    int pos_stride;          // QKV: floats per cache row (kv_dim)
    const State* st;         // [stream]
    unsigned long long* stamps;  // developer timeline (Q3_DEV builds, block 0 thread 0): 5 stamps per phase
    unsigned long long* slots;   // LOGITS: [stream][nslots] argmax keys, one per wave of the launch
    int nslots;
    int8_t* pack_q;          // k_dgemm SWIGLU: hq = quantize(hb) in packed operand order (W2's activation), or nullptr
    float* pack_s;
};

// One workgroup (8 waves) per row task (RT row tiles of 16 rows), the contraction walked in phases of PG groups:
//   1. the 8 waves split the phase's groups; each wave issues ALL of its A/B fragment loads at once (a phase of a
//      16-row tile is 64 KiB of weights in flight per CU), runs its MFMAs and writes the f32 group terms
//      ((f32)idot * ws) * xs to LDS  [term tile: PG groups x 32 streams x 16 rows];
//   2. after a barrier each of the 512 threads owns one (stream, row) accumulator and folds that accumulator's PG
//      terms in ascending group order (the strict chain of tensor.rs:53-60), while the next
//      phase's fragments are already in flight.
// The contraction is therefore parallel over K even though every accumulator is summed strictly in order, and a
// 256-tile matrix (4096 rows) still fills all 256 CUs with 8 waves each.
constexpr int kBW = 8;                        // waves per workgroup
constexpr int kBThreads = kBW * 64;
template <int RT> struct BPhase { static constexpr int PG = (RT == 1) ? 32 : 16; };   // 64 KiB term tile: two workgroups per CU
                                                                                      // (one folds while the other runs its MFMAs)
__host__ __device__ inline size_t bgemm_smem_bytes(int RT, int NT) {
    const int PG = RT == 1 ? 32 : 16;
    return 4 * ((size_t)RT * PG * NT * 256 + (size_t)RT * PG * 16 + (size_t)NT * PG * 16);
}

template <int EPI, int RT, int NT, int NJ>
__global__ __launch_bounds__(kBThreads) void k_bgemm(const BGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int PG = BPhase<RT>::PG;        // groups per phase
    constexpr int GW = PG / kBW;              // groups per wave per phase
    float* terms = (float*)smem_raw;                          // [RT][PG][NT*16 streams][16 rows]
    float* wsl = terms + (size_t)RT * PG * NT * 256;          // [RT][PG][16 rows]
    float* xsl = wsl + RT * PG * 16;                          // [NT][PG][16 streams]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, s = lane & 15;
    const int ng = a.ng;
    const int nph = (ng + PG - 1) / PG;
    const int ntasks = a.ntiles / RT;
    const size_t tile_v4 = (size_t)ng * NJ * 64;              // v4i per packed tile (weights or activations)
    const bool fold_thread = tid < NT * 256;                  // (stream, row) accumulator owner
    // The term tile of a (row tile, group, stream tile) is the MFMA result layout itself: lane l = 16 q + s of the producing
    // wave writes its four rows (4q .. 4q+3 of stream s) as ONE contiguous 16-byte piece at float 4 l, and fold thread
    // t reads float t & 255 -- both sides conflict-free.  (Round 2 stored [stream][row]: the 16 lanes of a quarter wave wrote
    // 16-byte pieces 64 bytes apart -- an 8-way bank conflict; r03 PMC: SQ_LDS_BANK_CONFLICT = 55 % of SQ_LDS_IDX_ACTIVE,
    // the LDS pipe busy 58 % of the w1|w3 launch.)
    const int f_stream = ((tid >> 8) << 4) | ((tid >> 2) & 15), f_row = (((tid >> 6) & 3) << 2) | (tid & 3);
    unsigned long long best = 0ull;                           // EPI_LOGITS: running argmax key of f_stream (lanes with f_row == 0)

    struct Frag {
        v4i a[RT][GW][NJ], b[NT][GW][NJ];
        v4f ws, xs;
    };
    auto issue_scales = [&](Frag& F, int task, int p) {
        const int g0 = p * PG;
        if (tid < RT * PG * 4) {
            const int rt = tid / (PG * 4), i = tid - rt * (PG * 4);
            const int gi = min(g0 * 4 + i, ng * 4 - 1);
            F.ws = ((const v4f*)a.ws)[((size_t)(task * RT + rt) * ng) * 4 + gi];
        }
        if (tid < NT * PG * 4) {
            const int nt = tid / (PG * 4), i = tid - nt * (PG * 4);
            const int gi = min(g0 * 4 + i, ng * 4 - 1);
            F.xs = ((const v4f*)a.xs)[((size_t)nt * ng) * 4 + gi];
        }
    };
    auto issue_group = [&](Frag& F, int task, int p, int k) {     // fragments of this wave's k-th group of phase p
        const int g = min(p * PG + wave * GW + k, ng - 1);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                F.a[rt][k][j] = Q3_BABL(8) ? (v4i){lane, g, rt, j}
                                           : __builtin_nontemporal_load((const v4i*)a.wq + (size_t)(task * RT + rt) * tile_v4 + ((size_t)g * NJ + j) * 64 + lane);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                F.b[nt][k][j] = Q3_BABL(1) ? (v4i){lane, g, nt, j} : ((const v4i*)a.xq)[(size_t)nt * tile_v4 + ((size_t)g * NJ + j) * 64 + lane];
    };
    auto issue = [&](Frag& F, int task, int p) {
        issue_scales(F, task, p);                                 // oldest loads: their LDS write comes first
#pragma unroll
        for (int k = 0; k < GW; ++k) issue_group(F, task, p, k);
    };
    auto commit_scales = [&](const Frag& F) {
        if (tid < RT * PG * 4) ((v4f*)wsl)[tid] = F.ws;
        if (tid < NT * PG * 4) ((v4f*)xsl)[tid] = F.xs;
    };

    float acc[RT];
    int sidx = 0;      // developer stamps: phases seen by this workgroup
    (void)sidx;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = -0.0f;                  // Iterator::sum::<f32>() starts from -0.0
    // one phase: `cur` holds this phase's fragments (its scale chunks are already in LDS); `nxt` (the same registers)
    // receives the following phase's as soon as the MFMAs have consumed them
    auto phase = [&](Frag& cur, Frag& nxt, int task, int p) -> bool {
        int ntask = task, np = p + 1;
        if (np == nph) { np = 0; ntask = task + (int)gridDim.x; }
        const bool more = ntask < ntasks;
        const int g0 = p * PG;
        BG_STAMP(0);
        // ---- MFMA: this wave's GW groups of the phase -> f32 terms in LDS.  Each group's fragment registers are
        // re-requested for the next phase as soon as its MFMAs have read them: the loads trickle through the CU's
        // texture path during the math instead of arriving as one burst from all 16 waves after it.
        // (Tried in round 2 and dropped: a second fragment set for the one-tile tasks, the whole next phase requested
        // before the first MFMA -- wo/w2 unchanged at 14.3 -> 14.6 us, QKV 12.5 -> 15.0 us through the lost occupancy.)
        if (more) issue_scales(nxt, ntask, np);
#pragma unroll
        for (int k = 0; k < GW; ++k) {
            const int gg = wave * GW + k;
            v4i cacc[RT][NT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    v4i c = {0, 0, 0, 0};
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(cur.a[rt][k][j], cur.b[nt][k][j], c, 0, 0, 0);
                    cacc[rt][nt] = c;
                }
            if (more) issue_group(nxt, ntask, np, k);
            if (Q3_BABL(4)) {
                if (cacc[0][0].x == 0x7fffffff) terms[tid] = 1.0f;      // keep the MFMAs alive
            } else if (g0 + gg < ng) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const v4f wsv = *(const v4f*)(wsl + (rt * PG + gg) * 16 + 4 * q);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const v4i c = cacc[rt][nt];
                        const float xsc = xsl[(nt * PG + gg) * 16 + s];
                        v4f t;
                        t.x = (float)c.x * wsv.x; t.x = t.x * xsc;      // tensor.rs:59  ((dot as f32) * ws) * xs
                        t.y = (float)c.y * wsv.y; t.y = t.y * xsc;
                        t.z = (float)c.z * wsv.z; t.z = t.z * xsc;
                        t.w = (float)c.w * wsv.w; t.w = t.w * xsc;
                        *(v4f*)(terms + ((size_t)(rt * PG + gg) * NT + nt) * 256 + 4 * lane) = t;
                    }
                }
            }
        }
        BG_STAMP(1);
        BG_STAMP(2);
        __syncthreads();                                           // terms of this phase complete
        BG_STAMP(3);
        // ---- fold: one (stream, row) accumulator per thread, ascending groups
        const int cnt = min(PG, ng - g0);
        if (fold_thread && !Q3_BABL(16)) {
            // (requesting all of the phase's terms before the first add was measured 2 % slower than these 8-term batches)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float* tp = terms + (size_t)rt * PG * NT * 256 + tid;
                float sacc = acc[rt];
                int gg = 0;
                for (; gg + 8 <= cnt; gg += 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = tp[(size_t)(gg + u) * NT * 256];
#pragma unroll
                    for (int u = 0; u < 8; ++u) sacc = sacc + v[u];
                }
                for (; gg < cnt; ++gg) sacc = sacc + tp[(size_t)gg * NT * 256];
                acc[rt] = sacc;
            }
        }
        BG_STAMP(4);
        if (more) commit_scales(nxt);                              // scale chunks are only read by the MFMA stage
        BG_STAMP(5);
        __syncthreads();                                           // terms consumed, next scales visible
        BG_STAMP(6);
        ++sidx;

        if (p == nph - 1) {
            // ---- epilogue: thread (f_stream, f_row) owns out[f_stream][r0 + f_row] of each of the task's RT tiles
            if (fold_thread && f_stream < a.n_streams) {
                const int sb = f_stream;
                if (EPI == EPI_SWIGLU) {
                    // packed tiles alternate w1 | w3 of the same 16 hidden units (RT == 2)   layers.rs:468-475
                    const float g1 = acc[0], u = acc[RT - 1];
                    const float den = 1.0f + q3_expf(-g1);
                    const float sw = g1 * (1.0f / den);
                    a.out0[(size_t)sb * a.out0_stride + (size_t)task * 16 + f_row] = sw * u;
                } else {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const int r0 = (task * RT + rt) * 16;
                        const float o = acc[rt];
                        if (EPI == EPI_QKV) {
                            float* dst;
                            if (r0 < a.rows0) dst = a.out0 + (size_t)sb * a.out0_stride + r0;
                            else if (r0 < a.rows0 + a.rows1) dst = a.out1 + (size_t)sb * a.out1_stride + (r0 - a.rows0);
                            else dst = a.out2 + (size_t)sb * a.out2_stride + (size_t)a.st[sb].pos * a.pos_stride + (r0 - a.rows0 - a.rows1);
                            dst[f_row] = o;
                        } else if (EPI == EPI_RESID) {
                            float* dst = a.out0 + (size_t)sb * a.out0_stride + r0 + f_row;
                            *dst = *dst + o;                                   // layers.rs:249-259
                        } else if (EPI == EPI_LOGITS) {
                            if (a.out0 != nullptr) a.out0[(size_t)sb * a.out0_stride + r0 + f_row] = o;
                            const unsigned long long key = ((unsigned long long)total_order_key(o) << 32) | (unsigned)(r0 + f_row);
                            best = key > best ? key : best;
                        } else {
                            a.out0[(size_t)sb * a.out0_stride + r0 + f_row] = o;
                        }
                    }
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = -0.0f;
        }
        return more;
    };

    int task = blockIdx.x, p = 0;
    if (task < ntasks) {
        Frag F0;
        issue(F0, task, 0);
        commit_scales(F0);
        __syncthreads();
        while (phase(F0, F0, task, p))
            if (++p == nph) { p = 0; task += gridDim.x; }
    }
    if (EPI == EPI_LOGITS) {
        // sampler.rs:57-59 (last maximum): max over the 16 rows of each stream -- 4 adjacent lanes (rows 4q .. 4q+3) in each of
        // the 4 waves q of the stream tile -- one slot per workgroup
        for (int m = 1; m < 4; m <<= 1) {
            const unsigned lo = __shfl_xor((unsigned)best, m);
            const unsigned hi = __shfl_xor((unsigned)(best >> 32), m);
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            best = o > best ? o : best;
        }
        unsigned long long* red = (unsigned long long*)smem_raw;      // [stream][q]; the term tile is idle now
        __syncthreads();
        if (fold_thread && (tid & 3) == 0) red[f_stream * 4 + (f_row >> 2)] = best;
        __syncthreads();
        if (fold_thread && f_row == 0 && f_stream < a.n_streams) {
            unsigned long long b = red[f_stream * 4];
            for (int k = 1; k < 4; ++k) b = red[f_stream * 4 + k] > b ? red[f_stream * 4 + k] : b;
            a.slots[(size_t)f_stream * a.nslots + blockIdx.x] = b;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Dense prefill matmul (round 3): M = 64..256 positions of ONE sequence per weight pass (q3_prefill_batched), group 64.
// At 32 columns the matrix cores idle (MFMA busy ~5 %) and k_bgemm splits K over the waves of a workgroup, exchanging the
// f32 group terms through a 64 KiB LDS tile with two barriers per phase.  With >= 64 positions there are enough
// (row tile, position tile) pairs to give every WAVE an output tile for the whole contraction: RT row tiles x PT position
// tiles, RT*PT MFMAs per quantization group, and lane l (position s = l % 16, rows 4*(l/16)+i) adds its group terms
//     acc[rt][pt][i] += ((f32)idot * ws[row][g]) * xs[pos][g]            g ascending, from -0.0
// itself -- the order of tensor.rs:53-60, no term tile, no barrier, no K split.  Weight and activation fragments are
// requested two groups ahead: the same 1 KiB-per-wave contiguous loads of the packed layout as k_bgemm.
// ------------------------------------------------------------------------------------------------
constexpr int kPgThreads = 256;
typedef float pk2 __attribute__((ext_vector_type(2)));
template <int EPI, int RT, int PT, int DEPTH = 4>
__global__ __launch_bounds__(kPgThreads) void k_pgemm(const BGemmArgs a) {
    static_assert(EPI != EPI_SWIGLU || (RT % 2) == 0, "SwiGLU tasks hold w1 tiles and their w3 tiles");
    // prefetch depth in groups, the SAME for weight and activation fragments: loads retire in order, so a shallower ring for
    // the (L2-resident) activations made every wait for them a wait for the youngest weight request as well (first cut:
    // DA = 4, DB = 2 ran the 53 MB w1|w3 pass at 1.25 TB/s)
    // Each wave has DEPTH - 1 groups = (DEPTH - 1) * RT KiB of weights in flight; under load an HBM round trip is ~2.5 us, so the
    // chip needs >= 12 MB requested to stream at 5 TB/s: prefer many row tiles per wave (weights) over many position tiles.
    constexpr int DA = DEPTH, DB = DEPTH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, s = lane & 15;
    const int ng = a.ng;
    const int nptiles = (a.n_streams + 15) >> 4;
    const int npg = (nptiles + PT - 1) / PT;                     // position groups per row task
    const int ntasks = (a.ntiles / RT) * npg;
    const size_t tile_v4 = (size_t)ng * 64;                      // v4i per packed tile (weights or activations)
    for (int task = blockIdx.x * (kPgThreads / 64) + wave; task < ntasks; task += gridDim.x * (kPgThreads / 64)) {
        const int rtask = task / npg, pg = task - rtask * npg;
        const v4i* wp[RT];
        const float* wsp[RT];
        const v4i* xp[PT];
        const float* xsp[PT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            wp[rt] = (const v4i*)a.wq + (size_t)(rtask * RT + rt) * tile_v4 + lane;
            // one scale dword per lane: lane (q, s) fetches the scale of row 4q + (s & 3) and the four rows of its accumulators
            // come from lanes 16q + 0..3 by DPP row broadcast.  (A v4f per lane moved 1 KiB through the L1 return path for
            // 64 B of scales: with the fragments that path, not the VALU or the matrix core, bounded the kernel.)
            wsp[rt] = a.ws + (size_t)(rtask * RT + rt) * ng * 16 + 4 * q + (s & 3);
        }
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int ptile = min(pg * PT + pt, nptiles - 1);   // position tiles past the block re-read the last one (not stored)
            xp[pt] = (const v4i*)a.xq + (size_t)ptile * tile_v4 + lane;
            xsp[pt] = a.xs + (size_t)ptile * ng * 16 + s;
        }
        v4i fa[DA][RT], fb[DB][PT];
        float fws[DA][RT];
        float fxs[DB][PT];
        auto load_a = [&](int slot, int g) {
            const int gg = min(g, ng - 1);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                // plain (cacheable) loads: the position groups of a row task run as adjacent waves of one workgroup and share the
                // weight lines through L1 / L2 -- with non-temporal loads every position group fetched the tile from HBM again
                // (r03 sweep: w1|w3 time followed the number of position groups, 46 / 46 / 63 us for 2 / 4 / 8 groups)
                fa[slot][rt] = wp[rt][(size_t)gg * 64];
                fws[slot][rt] = wsp[rt][(size_t)gg * 16];
            }
        };
        auto load_b = [&](int slot, int g) {
            const int gg = min(g, ng - 1);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                fb[slot][pt] = xp[pt][(size_t)gg * 64];
                fxs[slot][pt] = xsp[pt][(size_t)gg * 16];
            }
        };
        v4f acc[RT][PT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) acc[rt][pt] = (v4f){-0.0f, -0.0f, -0.0f, -0.0f};     // Iterator::sum::<f32>() identity
#pragma unroll
        for (int d = 0; d < DA - 1; ++d) { load_a(d, d); load_b(d, d); }
        // A group's work is split in two so that the matrix core and the VALU overlap: mfma_group(g + 1) is issued, then the
        // convert / scale / ordered-add chain of group g runs while those MFMAs execute (one wave per SIMD here: a chain that
        // starts right behind its own MFMA waits out the whole MFMA latency -- s_nop 7 in the r03 disassembly -- four times a group).
        v4i cc[RT][PT], cn[RT][PT];
        auto mfma_group = [&](v4i (&c)[RT][PT], int slot) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    c[rt][pt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[slot][rt], fb[slot][pt], (v4i){0, 0, 0, 0}, 0, 0, 0);
        };
        auto math_group = [&](const v4i (&cg)[RT][PT], int slot) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int wsr = __float_as_int(fws[slot][rt]);
                v4f wsv;                                           // rows 4q .. 4q+3: v_mov_b32_dpp row_newbcast:0..3
                wsv.x = __int_as_float(__builtin_amdgcn_mov_dpp(wsr, 0x150, 0xf, 0xf, true));
                wsv.y = __int_as_float(__builtin_amdgcn_mov_dpp(wsr, 0x151, 0xf, 0xf, true));
                wsv.z = __int_as_float(__builtin_amdgcn_mov_dpp(wsr, 0x152, 0xf, 0xf, true));
                wsv.w = __int_as_float(__builtin_amdgcn_mov_dpp(wsr, 0x153, 0xf, 0xf, true));
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    const v4i c = cg[rt][pt];
                    // the position scale as a REAL register pair {xs, xs}, materialised here.  Left to the compiler the packed
                    // multiply broadcast one half of whatever 64-bit pair held xs (op_sel) and so formally read the partner
                    // register too -- usually the scale of a slot still in flight, and hipcc waited for that slot's loads
                    // (r04 disassembly: s_waitcnt vmcnt(6) with 28 loads in the ring, vmcnt(12-20) with 40)
                    pk2 xbc = (pk2){fxs[slot][pt], fxs[slot][pt]};
                    asm("" : "+v"(xbc));
                    // tensor.rs:59  ((dot as f32) * ws) * xs, then the g-ascending add.  The pairs are chosen by hand -- (rows 4q,
                    // 4q+1) and (4q+2, 4q+3) of one (row tile, position tile): their scales are the two halves of the v4f the ring
                    // loaded, xs is broadcast by op_sel -- three packed operations per pair.  Every value passes through an opaque
                    // register so that the SLP vectoriser does not re-pair them across tiles (its pairs forced copies of in-flight
                    // ring slots: s_waitcnt vmcnt(4) with 32 loads in the ring).
                    pk2 t01 = (pk2){(float)c.x, (float)c.y} * (pk2){wsv.x, wsv.y};
                    pk2 t23 = (pk2){(float)c.z, (float)c.w} * (pk2){wsv.z, wsv.w};
                    asm("" : "+v"(t01)); asm("" : "+v"(t23));
                    t01 = t01 * xbc; t23 = t23 * xbc;
                    asm("" : "+v"(t01)); asm("" : "+v"(t23));
                    v4f& ac = acc[rt][pt];
                    pk2 a01 = (pk2){ac.x, ac.y} + t01, a23 = (pk2){ac.z, ac.w} + t23;
                    asm("" : "+v"(a01)); asm("" : "+v"(a23));
                    ac.x = a01.x; ac.y = a01.y; ac.z = a23.x; ac.w = a23.y;
                }
            }
        };
        auto rotate = [&]() {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) cc[rt][pt] = cn[rt][pt];
        };
        // main loop: whole rings of DA groups, NO branch inside (a wave-uniform `if (g < ng)` here split the body into basic
        // blocks and hipcc closed each with s_waitcnt vmcnt(0): every load was waited for in the iteration that issued it)
        mfma_group(cc, 0);
        int g0 = 0;
        for (; g0 + DA <= ng; g0 += DA) {
#pragma unroll
            for (int u = 0; u < DA; ++u) {
                load_a((u + DA - 1) % DA, g0 + u + DA - 1);      // (clamped to the row: the last ring re-reads its last group)
                load_b((u + DB - 1) % DB, g0 + u + DB - 1);
                mfma_group(cn, (u + 1) % DA);                    // group g + 1 (past the row: a re-read group, its result is dropped)
                math_group(cc, u);
                rotate();
                // stage boundary pinned: without it hipcc hoists the four stages' MFMAs to the top of the body behind one
                // s_waitcnt vmcnt(0) and issues all sixteen loads as one burst behind them -- no load is in flight across a stage
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // tail: ng % DA groups, already in the ring
#pragma unroll
        for (int u = 0; u < DA - 1; ++u)
            if (g0 + u < ng) {
                if (u + 1 < DA - 1) mfma_group(cn, u + 1);
                math_group(cc, u);
                rotate();
            }
        // ---- epilogue: lane (s, q) owns out[position pg*PT*16 + pt*16 + s][rows 4q .. 4q+3 of each row tile]
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int sb = (pg * PT + pt) * 16 + s;
            if (pg * PT + pt >= nptiles || sb >= a.n_streams) continue;
            if constexpr (EPI == EPI_SWIGLU) {
                // packed tiles alternate w1 | w3 of the same 16 hidden units            layers.rs:468-475
#pragma unroll
                for (int k = 0; k < RT / 2; ++k) {
                    v4f o;
                    const v4f g1 = acc[2 * k][pt], up = acc[2 * k + 1][pt];
                    { const float den = 1.0f + q3_expf(-g1.x); o.x = (g1.x * (1.0f / den)) * up.x; }
                    { const float den = 1.0f + q3_expf(-g1.y); o.y = (g1.y * (1.0f / den)) * up.y; }
                    { const float den = 1.0f + q3_expf(-g1.z); o.z = (g1.z * (1.0f / den)) * up.z; }
                    { const float den = 1.0f + q3_expf(-g1.w); o.w = (g1.w * (1.0f / den)) * up.w; }
                    *(v4f*)(a.out0 + (size_t)sb * a.out0_stride + (size_t)(rtask * (RT / 2) + k) * 16 + 4 * q) = o;
                }
            } else {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int r0 = (rtask * RT + rt) * 16 + 4 * q;
                    const v4f o = acc[rt][pt];
                    if constexpr (EPI == EPI_QKV) {
                        float* dst;
                        if (r0 < a.rows0) dst = a.out0 + (size_t)sb * a.out0_stride + r0;
                        else if (r0 < a.rows0 + a.rows1) dst = a.out1 + (size_t)sb * a.out1_stride + (r0 - a.rows0);
                        else dst = a.out2 + (size_t)sb * a.out2_stride + (size_t)a.st[sb].pos * a.pos_stride + (r0 - a.rows0 - a.rows1);
                        *(v4f*)dst = o;
                    } else if constexpr (EPI == EPI_RESID) {
                        v4f* dst = (v4f*)(a.out0 + (size_t)sb * a.out0_stride + r0);
                        v4f x = *dst;
                        x.x = x.x + o.x; x.y = x.y + o.y; x.z = x.z + o.z; x.w = x.w + o.w;      // layers.rs:249-259
                        *dst = x;
                    } else {
                        *(v4f*)(a.out0 + (size_t)sb * a.out0_stride + r0) = o;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Dense prefill matmul, round 4: workgroup tile staged in LDS (k_pgemm2).
// k_pgemm gives every WAVE its own RT x PT output tiles and lets it pull its own fragments: 1 KiB through the CU's L1 path
// per MFMA (1.3 GB per W1|W3 pass, r03), the matrix cores 9.8 % busy.  Here a workgroup of 8 waves owns 4 row tiles x 8
// position tiles (64 rows x 128 positions); per quantization group its 4 weight fragments, 8 activation fragments and their
// scales (12.75 KiB) are fetched ONCE by the workgroup (each wave requests one or two KiB, a register ring 4 groups ahead),
// committed to one of two LDS slots, and read back by the waves: wave (wr, wp) computes row tiles {2wr, 2wr+1} x position
// tiles {2wp, 2wp+1} -- 4 MFMAs per group from 4 KiB of ds_read_b128 -- and keeps the four accumulator tiles in its lanes:
// acc += ((f32)idot * ws) * xs, g ascending from -0.0 (tensor.rs:53-60), the same operations in the same order as every other
// matmul of this library.  Global traffic per MFMA drops from 1 KiB to 0.4 KiB (weights: 128 B), one barrier per group.
// ------------------------------------------------------------------------------------------------
// PTW = 8 position tiles per workgroup (wave: 2 x 2 tiles) or 4 (wave: 2 x 1): the smaller tile doubles the number of workgroup
// tiles -- W1|W3 of the 4B shape has 608 tiles of 4 x 8 on 512 resident workgroup slots (two full rounds for 1.19 rounds of work),
// 1,216 of 4 x 4 run as three rounds of half the length; QKV goes from 192 tiles (fewer than CUs) to 384.
constexpr int kP2Waves = 8, kP2Threads = 512, kP2RT = 4, kP2D = 4;
__host__ __device__ inline size_t pgemm2_slot_bytes(int ptw, int rtw) { return (size_t)(rtw + ptw) * 1024 + (size_t)(rtw + ptw) * 64; }     // fragments + scales of one group
__host__ __device__ inline size_t pgemm2_smem_bytes(int ptw, int rtw = 4) { return (ptw == 4 ? 3 : 2) * pgemm2_slot_bytes(ptw, rtw); }

// developer ablation of k_pgemm2, compile-time only (-DQ3_PABLATE=bits; wrong results, time only): 1 no global requests in the
// loop, 2 no convert / scale / add chain, 4 no MFMAs, 8 no barrier in the group loop, 16 no LDS commits
#ifdef Q3_PABLATE
#define Q3_PABL(bit) ((Q3_PABLATE & (bit)) != 0)
#else
#define Q3_PABL(bit) false
#endif
template <int EPI, int PTW, int RTW = 4>
__global__ __launch_bounds__(kP2Threads, 4) void k_pgemm2(const BGemmArgs a) {
    constexpr int D = kP2D;
    constexpr int NF = RTW + PTW;                                // fragments per group: [A0..A(RTW-1)][B0..B(PTW-1)]
    constexpr int NRW = RTW / 2;                                 // row tiles per wave (SwiGLU: the w1 and the w3 tile -> RTW = 4)
    static_assert(RTW == 4 || (RTW == 2 && EPI != EPI_SWIGLU), "2-row-tile workgroups: one row tile per wave");
    constexpr int NPW = PTW / 4;                                 // position tiles per wave
    constexpr size_t kSlot = (size_t)NF * 1024 + (size_t)NF * 64;
    static_assert(PTW == 4 || PTW == 8, "workgroup tile: 4 row tiles x 4 or 8 position tiles");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, s = lane & 15;
    const int wr = wave >> 2, wp = wave & 3;                     // this wave's row-tile pair / position-tile slot inside the workgroup tile
    const int ng = a.ng;
    const int nptiles = (a.n_streams + 15) >> 4;
    const int npb = (nptiles + PTW - 1) / PTW;                   // position blocks
    const int nrb = a.ntiles / RTW;                            // row blocks (host: ntiles % 4 == 0)
    const size_t tile_v4 = (size_t)ng * 64;
    // LDS slot: [NF fragments] v4i x 64, then [4][16] ws, [PTW][16] xs floats
    auto slot_frag = [&](int sl) { return (v4i*)(smem_raw + (size_t)sl * kSlot); };
    auto slot_sc = [&](int sl) { return (float*)(smem_raw + (size_t)sl * kSlot + (size_t)NF * 1024); };
    for (int blk = blockIdx.x; blk < nrb * npb; blk += gridDim.x) {
        const int rb = blk / npb, pb = blk - rb * npb;           // the position blocks of a row block run on neighbouring workgroups
        // ---- loader roles (wave-uniform).  Fragment f of the group goes to wave f % 8 (waves 0..NF-9 carry two); the scales:
        // wave 7 the row scales (4 tiles x 16 lanes), wave 6 / 5 the position scales of tiles 0-3 / 4-7
        auto frag_src = [&](int f) -> const v4i* {
            if (f < RTW) return (const v4i*)a.wq + (size_t)(rb * RTW + f) * tile_v4 + lane;
            return (const v4i*)a.xq + (size_t)min(pb * PTW + (f - RTW), nptiles - 1) * tile_v4 + lane;      // (tiles past the block re-read the last one)
        };
        const bool one = wave < NF;                              // (2 x 4 tiles: six fragments, waves 6 and 7 only carry scales)
        const v4i* src0 = frag_src(one ? wave : 0);
        const bool two = wave + 8 < NF;
        const v4i* src1 = frag_src(two ? wave + 8 : wave);
        const bool ld_ws = wave == 7, ld_xs = wave == 6 || (PTW == 8 && wave == 5);
        const float* ssrc = a.ws;                                // one scale dword per lane and group for the scale loaders
        if (ld_ws) ssrc = a.ws + (size_t)(rb * RTW + min(lane >> 4, RTW - 1)) * ng * 16 + (lane & 15);
        if (ld_xs) ssrc = a.xs + (size_t)min(pb * PTW + (wave == 5 ? 4 : 0) + (lane >> 4), nptiles - 1) * ng * 16 + (lane & 15);
        v4i r0_[D], r1_[D];
        float rs_[D];
        auto issue = [&](int sl, int g) {                        // group g -> register slot sl
            if (Q3_PABL(1) && g >= D + 2) return;
            const int gg = min(g, ng - 1);
            r0_[sl] = src0[(size_t)gg * 64];                     // (unconditional: a wave without a fragment re-reads fragment 0 and drops it --
            if (two) r1_[sl] = src1[(size_t)gg * 64];            //  a branch around the first request made hipcc give up its counted waits: 45 -> 76 us)
            if (ld_ws || ld_xs) rs_[sl] = ssrc[(size_t)gg * 16];
        };
        auto commit = [&](int sl, int ls) {                      // register slot sl -> LDS slot ls
            if (Q3_PABL(16)) return;
            v4i* f = slot_frag(ls);
            if (one) f[wave * 64 + lane] = r0_[sl];
            if (two) f[(wave + 8) * 64 + lane] = r1_[sl];
            float* sc = slot_sc(ls);
            if (ld_ws && lane < RTW * 16) sc[lane] = rs_[sl];                            // ws [RTW][16]
            if (ld_xs) sc[RTW * 16 + (wave == 5 ? 64 : 0) + lane] = rs_[sl];         // xs [PTW][16]
        };
        v4f acc[NRW][NPW];
#pragma unroll
        for (int i = 0; i < NRW; ++i)
#pragma unroll
            for (int j = 0; j < NPW; ++j) acc[i][j] = (v4f){-0.0f, -0.0f, -0.0f, -0.0f};    // Iterator::sum::<f32>() identity
        // (no software pipeline inside a wave: at 4 waves per SIMD the other waves' MFMAs cover this wave's convert / scale / add
        // chain, and a second set of MFMA results would push the kernel past 128 VGPRs, i.e. to one workgroup per CU)
        v4i cc[NRW][NPW];
        v4f wsc[NRW];
        float xsc[NPW];
        auto mfma_group = [&](int ls, v4i (&c)[NRW][NPW], v4f (&w)[NRW], float (&x)[NPW]) {    // MFMAs of the group in LDS slot ls -> c; its scales -> w, x
            const v4i* f = slot_frag(ls);
            const float* sc = slot_sc(ls);
            v4i fa[NRW], fb[NPW];
#pragma unroll
            for (int i = 0; i < NRW; ++i) fa[i] = f[(NRW * wr + i) * 64 + lane];
#pragma unroll
            for (int j = 0; j < NPW; ++j) fb[j] = f[(RTW + NPW * wp + j) * 64 + lane];
#pragma unroll
            for (int i = 0; i < NRW; ++i) w[i] = *(const v4f*)(sc + (NRW * wr + i) * 16 + 4 * q);
#pragma unroll
            for (int j = 0; j < NPW; ++j) x[j] = sc[RTW * 16 + (NPW * wp + j) * 16 + s];
#pragma unroll
            for (int i = 0; i < NRW; ++i)
#pragma unroll
                for (int j = 0; j < NPW; ++j)
                    c[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], (v4i){0, 0, 0, 0}, 0, 0, 0);
        };
        auto math_group = [&](const v4i (&cg)[NRW][NPW], const v4f (&w)[NRW], const float (&x)[NPW]) {
#pragma unroll
            for (int i = 0; i < NRW; ++i)
#pragma unroll
                for (int j = 0; j < NPW; ++j) {
                    const v4i c = cg[i][j];
                    if (Q3_PABL(2)) { acc[i][j].x = __int_as_float(__float_as_int(acc[i][j].x) ^ c.x ^ c.y ^ c.z ^ c.w ^ __float_as_int(w[i].x) ^ __float_as_int(x[j])); continue; }
                    // tensor.rs:59  ((dot as f32) * ws) * xs, then the g-ascending add; pairs (rows 4q, 4q+1), (4q+2, 4q+3)
                    pk2 t01 = (pk2){(float)c.x, (float)c.y} * (pk2){w[i].x, w[i].y};
                    pk2 t23 = (pk2){(float)c.z, (float)c.w} * (pk2){w[i].z, w[i].w};
                    asm("" : "+v"(t01)); asm("" : "+v"(t23));
                    pk2 xb = (pk2){x[j], x[j]};
                    asm("" : "+v"(xb));
                    t01 = t01 * xb; t23 = t23 * xb;
                    asm("" : "+v"(t01)); asm("" : "+v"(t23));
                    v4f& ac = acc[i][j];
                    pk2 a01 = (pk2){ac.x, ac.y} + t01, a23 = (pk2){ac.z, ac.w} + t23;
                    asm("" : "+v"(a01)); asm("" : "+v"(a23));
                    ac.x = a01.x; ac.y = a01.y; ac.z = a23.x; ac.w = a23.y;
                }
        };
        if constexpr (PTW == 8) {
#pragma unroll
            for (int d = 0; d < D; ++d) issue(d, d);
            commit(0, 0);
            issue(0, D);
            __syncthreads();
            // ---- pipeline.  Register slot of group x is x % D, LDS slot x % 2.  Stage g: the MFMAs and the convert / scale / add
            // chain of group g (LDS slot g % 2); meanwhile group g + 1 (requested D - 1 stages ago) goes to the other LDS slot and
            // group g + D is requested into the register slot it leaves; one barrier per group.
            for (int g0 = 0; g0 < ng; g0 += D) {
#pragma unroll
                for (int u = 0; u < D; ++u) {
                    const int g = g0 + u;
                    commit((u + 1) % D, (u + 1) & 1);            // group g + 1 -> the other LDS slot (past the row: a re-read group)
                    issue((u + 1) % D, g + 1 + D);
                    mfma_group(u & 1, cc, wsc, xsc);
                    math_group(cc, wsc, xsc);
                    __syncthreads();                             // group g + 1 visible; slot g % 2 free for group g + 2
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            // ---- 4 x 4 tiles (78 VGPRs: room for a second fragment set): THREE LDS slots, the fragments of group g + 1 are read into
            // registers while the MFMAs and the chain of group g run, so neither the LDS latency nor the barrier wait sits between a
            // group's read and its MFMAs (a lone workgroup per CU -- Wo / W2 -- had nothing else to cover them: 33.7 us).  Stage g:
            // commit group g + 2 -> slot (g + 2) % 3, request group g + 2 + D - 2, read group g + 1 (slot (g + 1) % 3, visible since
            // the barrier of stage g - 1), compute group g from the registers read in stage g - 1, barrier.
            static_assert((D & 1) == 0, "register-set parity is compile-time");
            v4i fa2[2][NRW], fb2[2][NPW];                          // [parity][...]: fragments of the group computed in this / the next stage
            v4f w2[2][NRW];
            float x2[2][NPW];
            auto read_group = [&](int ls, int par) {
                const v4i* f = slot_frag(ls);
                const float* sc = slot_sc(ls);
#pragma unroll
                for (int i = 0; i < NRW; ++i) fa2[par][i] = f[(NRW * wr + i) * 64 + lane];
#pragma unroll
                for (int j = 0; j < NPW; ++j) fb2[par][j] = f[(RTW + NPW * wp + j) * 64 + lane];
#pragma unroll
                for (int i = 0; i < NRW; ++i) w2[par][i] = *(const v4f*)(sc + (NRW * wr + i) * 16 + 4 * q);
#pragma unroll
                for (int j = 0; j < NPW; ++j) x2[par][j] = sc[RTW * 16 + (NPW * wp + j) * 16 + s];
            };
            auto compute = [&](int par) {
#pragma unroll
                for (int i = 0; i < NRW; ++i)
#pragma unroll
                    for (int j = 0; j < NPW; ++j)
                        cc[i][j] = Q3_PABL(4) ? fa2[par][i] ^ fb2[par][j] : __builtin_amdgcn_mfma_i32_16x16x64_i8(fa2[par][i], fb2[par][j], (v4i){0, 0, 0, 0}, 0, 0, 0);
                math_group(cc, w2[par], x2[par]);
            };
#pragma unroll
            for (int d = 0; d < D; ++d) issue(d, d);
            commit(0, 0);                                        // groups 0 and 1 -> LDS slots 0 and 1
            commit(1, 1);
            issue(0, D);
            issue(1, D + 1);
            __syncthreads();
            read_group(0, 0);                                    // group 0 -> register set 0
            int l1 = 1, l2 = 2;                                  // LDS slots of groups g + 1 and g + 2 (run-time: g % 3 does not divide the unroll)
            for (int g0 = 0; g0 < ng; g0 += D) {
#pragma unroll
                for (int u = 0; u < D; ++u) {
                    const int g = g0 + u;
                    commit((u + 2) % D, l2);                     // group g + 2 (requested D - 2 stages ago)
                    issue((u + 2) % D, g + 2 + D);
                    read_group(l1, (u + 1) & 1);                 // group g + 1 (past the row: a re-read group, dropped)
                    compute(u & 1);
                    if (!Q3_PABL(8)) __syncthreads();            // group g + 2 visible; slot g % 3 free for group g + 3
                    l1 = l2;
                    l2 = l2 == 2 ? 0 : l2 + 1;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // ---- epilogue: lane (s, q) owns out[position (pb*PTW + NPW*wp + j)*16 + s][rows 4q .. 4q+3 of row tile rb*4 + 2wr + i]
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int ptj = pb * PTW + NPW * wp + j;
            const int sb = ptj * 16 + s;
            if (ptj >= nptiles || sb >= a.n_streams) continue;
            if constexpr (EPI == EPI_SWIGLU) {
                // packed tiles alternate w1 | w3 of the same 16 hidden units            layers.rs:468-475
                v4f o;
                const v4f g1 = acc[0][j], up = acc[1][j];
                { const float den = 1.0f + q3_expf(-g1.x); o.x = (g1.x * (1.0f / den)) * up.x; }
                { const float den = 1.0f + q3_expf(-g1.y); o.y = (g1.y * (1.0f / den)) * up.y; }
                { const float den = 1.0f + q3_expf(-g1.z); o.z = (g1.z * (1.0f / den)) * up.z; }
                { const float den = 1.0f + q3_expf(-g1.w); o.w = (g1.w * (1.0f / den)) * up.w; }
                *(v4f*)(a.out0 + (size_t)sb * a.out0_stride + (size_t)(rb * (RTW / 2) + wr) * 16 + 4 * q) = o;
            } else {
#pragma unroll
                for (int i = 0; i < NRW; ++i) {
                    const int r0 = (rb * RTW + NRW * wr + i) * 16 + 4 * q;
                    const v4f o = acc[i][j];
                    if constexpr (EPI == EPI_QKV) {
                        float* dst;
                        if (r0 < a.rows0) dst = a.out0 + (size_t)sb * a.out0_stride + r0;
                        else if (r0 < a.rows0 + a.rows1) dst = a.out1 + (size_t)sb * a.out1_stride + (r0 - a.rows0);
                        else dst = a.out2 + (size_t)sb * a.out2_stride + (size_t)a.st[sb].pos * a.pos_stride + (r0 - a.rows0 - a.rows1);
                        *(v4f*)dst = o;
                    } else if constexpr (EPI == EPI_RESID) {
                        v4f* dst = (v4f*)(a.out0 + (size_t)sb * a.out0_stride + r0);
                        v4f x = *dst;
                        x.x = x.x + o.x; x.y = x.y + o.y; x.z = x.z + o.z; x.w = x.w + o.w;      // layers.rs:249-259
                        *dst = x;
                    } else {
                        *(v4f*)(a.out0 + (size_t)sb * a.out0_stride + r0) = o;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_pgemm3: k_pgemm2's workgroup tile with GS quantization groups per pipeline stage.
// The ablation of k_pgemm2 (profiles/r04_prefill_ab.txt) showed a stage that is a serial chain of latencies -- barrier -> ds_read ->
// MFMA -> convert / scale / add chain -> commit of the next group -> barrier, ~700 cycles for ~100 cycles of pipe work -- with
// the arithmetic itself free (no-VALU and no-MFMA builds run at the same speed).  Here one barrier covers GS groups: a stage's
// fragments and scales (GS x 8.5 KiB for 4 x 4 tiles) sit in one of two LDS slots, every wave requests its share of stage s + 2
// into registers while stage s is computed, and commits stage s + 1 (requested two stages ago) after its own compute; inside a
// stage the GS groups are independent until the ordered adds, so their LDS reads, MFMAs and multiply chains overlap.
// Same accumulation order per (row, position): groups ascending, acc += ((f32)idot * ws) * xs from -0.0 (tensor.rs:53-60).
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t pgemm3_slot_bytes(int ptw, int rtw, int gs) { return (size_t)gs * ((size_t)(rtw + ptw) * 1024 + (size_t)(rtw + ptw) * 64); }
__host__ __device__ inline size_t pgemm3_smem_bytes(int ptw, int rtw, int gs) { return 2 * pgemm3_slot_bytes(ptw, rtw, gs); }

template <int EPI, int PTW, int RTW, int GS>
__global__ __launch_bounds__(kP2Threads, 4) void k_pgemm3(const BGemmArgs a) {
    constexpr int NF = RTW + PTW;                                // fragments per group: [A0..A(RTW-1)][B0..B(PTW-1)]
    constexpr int NRW = RTW / 2, NPW = PTW / 4;                  // row / position tiles per wave
    constexpr int NJ = GS * NF;                                  // fragment jobs (1 KiB each) per stage
    constexpr int JW = (NJ + 7) / 8;                             // ... per wave
    constexpr int NSC = GS * NF * 16;                            // scale floats per stage
    constexpr int SW = (NSC + kP2Threads - 1) / kP2Threads;      // ... per thread (1)
    constexpr size_t kSlot = (size_t)GS * ((size_t)NF * 1024 + (size_t)NF * 64);
    static_assert(RTW == 8 || RTW == 4 || (RTW == 2 && EPI != EPI_SWIGLU), "2-row-tile workgroups: one row tile per wave");
    static_assert(SW == 1, "one scale dword per thread and stage");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, s = lane & 15;
    const int wr = wave >> 2, wp = wave & 3;
    const int ng = a.ng;
    const int nst = ng / GS;                                     // stages (host: ng % (2 GS) == 0)
    const int nptiles = (a.n_streams + 15) >> 4;
    const int npb = (nptiles + PTW - 1) / PTW;
    const int nrb = a.ntiles / RTW;
    const size_t tile_v4 = (size_t)ng * 64;
    // LDS slot: [GS][NF] fragments (v4i x 64 each), then [GS][NF][16] scales (row tiles first, then position tiles)
    auto slot_frag = [&](int sl) { return (v4i*)(smem_raw + (size_t)sl * kSlot); };
    auto slot_sc = [&](int sl) { return (float*)(smem_raw + (size_t)sl * kSlot + (size_t)GS * NF * 1024); };
    for (int blk = blockIdx.x; blk < nrb * npb; blk += gridDim.x) {
        const int rb = blk / npb, pb = blk - rb * npb;
        // ---- loader jobs of this wave: job j = wave + 8 i  ->  (group k = j / NF of the stage, fragment f = j % NF)
        const v4i* jsrc[JW];
        int jk[JW];
#pragma unroll
        for (int i = 0; i < JW; ++i) {
            const int j = min(wave + 8 * i, NJ - 1);             // (a wave past the job list re-reads the last job and drops it)
            const int k = j / NF, f = j - k * NF;
            jk[i] = k;
            // (wave-uniform bases: scalar registers; the lane offset is added at the request)
            if (f < RTW) jsrc[i] = (const v4i*)a.wq + (size_t)(rb * RTW + f) * tile_v4;
            else jsrc[i] = (const v4i*)a.xq + (size_t)min(pb * PTW + (f - RTW), nptiles - 1) * tile_v4;
        }
        // scale job of this thread: element e = tid of [GS][NF][16]
        const int e_ = min(tid, NSC - 1);
        const int ek = e_ / (NF * 16), er = e_ - ek * (NF * 16), et = er >> 4, ei = er & 15;
        const float* ssrc = et < RTW ? a.ws + (size_t)(rb * RTW + et) * ng * 16 + ei
                                     : a.xs + (size_t)min(pb * PTW + (et - RTW), nptiles - 1) * ng * 16 + ei;
        v4i rj[2][JW];
        float rsc[2];
        auto issue = [&](int rs, int st) {                       // stage st -> register set rs (past the row: a re-read stage)
            const int g0 = min(st, nst - 1) * GS;
#pragma unroll
            for (int i = 0; i < JW; ++i) rj[rs][i] = (jsrc[i] + (size_t)(g0 + jk[i]) * 64)[lane];
            rsc[rs] = ssrc[(size_t)(g0 + ek) * 16];
        };
        auto commit = [&](int rs, int ls) {
            v4i* f = slot_frag(ls);
#pragma unroll
            for (int i = 0; i < JW; ++i)
                if (wave + 8 * i < NJ) f[(size_t)(wave + 8 * i) * 64 + lane] = rj[rs][i];
            if (tid < NSC) slot_sc(ls)[tid] = rsc[rs];
        };
        v4f acc[NRW][NPW];
#pragma unroll
        for (int i = 0; i < NRW; ++i)
#pragma unroll
            for (int j = 0; j < NPW; ++j) acc[i][j] = (v4f){-0.0f, -0.0f, -0.0f, -0.0f};    // Iterator::sum::<f32>() identity
        // a stage is computed HS groups x IS row tiles at a time (their reads, MFMAs and multiply chains are independent and overlap;
        // everything at once needs ~155 VGPRs), the ordered adds of a sub-step follow its terms.  Per accumulator the order is
        // unchanged: groups ascending.
        constexpr int HS = (GS >= 2 && NPW == 1) ? 2 : 1;
        constexpr int IS = NRW > 2 ? 2 : NRW;
        auto compute = [&](int ls) {
            const v4i* f = slot_frag(ls);
            const float* sc = slot_sc(ls);
#pragma unroll
            for (int k0 = 0; k0 < GS; k0 += HS) {
#pragma unroll
                for (int i0 = 0; i0 < NRW; i0 += IS) {
                    v4i cc[HS][IS][NPW];
                    v4f w[HS][IS];
                    float x[HS][NPW];
#pragma unroll
                    for (int h = 0; h < HS; ++h) {
                        const int k = k0 + h;
                        v4i fa[IS], fb[NPW];
#pragma unroll
                        for (int i = 0; i < IS; ++i) fa[i] = f[(size_t)(k * NF + NRW * wr + i0 + i) * 64 + lane];
#pragma unroll
                        for (int j = 0; j < NPW; ++j) fb[j] = f[(size_t)(k * NF + RTW + NPW * wp + j) * 64 + lane];
#pragma unroll
                        for (int i = 0; i < IS; ++i) w[h][i] = *(const v4f*)(sc + (k * NF + NRW * wr + i0 + i) * 16 + 4 * q);
#pragma unroll
                        for (int j = 0; j < NPW; ++j) x[h][j] = sc[(k * NF + RTW + NPW * wp + j) * 16 + s];
#pragma unroll
                        for (int i = 0; i < IS; ++i)
#pragma unroll
                            for (int j = 0; j < NPW; ++j)
                                cc[h][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], (v4i){0, 0, 0, 0}, 0, 0, 0);
                    }
                    pk2 t01[HS][IS][NPW], t23[HS][IS][NPW];
#pragma unroll
                    for (int h = 0; h < HS; ++h)
#pragma unroll
                        for (int i = 0; i < IS; ++i)
#pragma unroll
                            for (int j = 0; j < NPW; ++j) {
                                const v4i c = cc[h][i][j];
                                // tensor.rs:59  ((dot as f32) * ws) * xs; pairs (rows 4q, 4q+1), (4q+2, 4q+3)
                                pk2 u01 = (pk2){(float)c.x, (float)c.y} * (pk2){w[h][i].x, w[h][i].y};
                                pk2 u23 = (pk2){(float)c.z, (float)c.w} * (pk2){w[h][i].z, w[h][i].w};
                                asm("" : "+v"(u01)); asm("" : "+v"(u23));
                                pk2 xb = (pk2){x[h][j], x[h][j]};
                                asm("" : "+v"(xb));
                                u01 = u01 * xb; u23 = u23 * xb;
                                asm("" : "+v"(u01)); asm("" : "+v"(u23));
                                t01[h][i][j] = u01; t23[h][i][j] = u23;
                            }
#pragma unroll
                    for (int h = 0; h < HS; ++h)
#pragma unroll
                        for (int i = 0; i < IS; ++i)
#pragma unroll
                            for (int j = 0; j < NPW; ++j) {
                                v4f& ac = acc[i0 + i][j];
                                pk2 a01 = (pk2){ac.x, ac.y} + t01[h][i][j], a23 = (pk2){ac.z, ac.w} + t23[h][i][j];
                                asm("" : "+v"(a01)); asm("" : "+v"(a23));
                                ac.x = a01.x; ac.y = a01.y; ac.z = a23.x; ac.w = a23.y;
                            }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        // ---- pipeline: register set of stage x is x % 2, LDS slot x % 2
        issue(0, 0);
        issue(1, 1);
        commit(0, 0);
        issue(0, 2);
        __syncthreads();
        for (int s0 = 0; s0 < nst; s0 += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int st = s0 + u;
                compute(u);                                      // stage st from LDS slot st % 2
                commit(u ^ 1, u ^ 1);                            // stage st + 1 (requested two stages ago) -> the other slot
                issue(u ^ 1, st + 3);
                __syncthreads();                                 // stage st + 1 visible; slot st % 2 free for stage st + 2
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue: lane (s, q) owns out[position (pb*PTW + NPW*wp + j)*16 + s][rows 4q .. 4q+3 of row tile rb*RTW + NRW*wr + i]
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int ptj = pb * PTW + NPW * wp + j;
            const int sb = ptj * 16 + s;
            if (ptj >= nptiles || sb >= a.n_streams) continue;
            if constexpr (EPI == EPI_SWIGLU) {
                // packed tiles alternate w1 | w3 of the same 16 hidden units            layers.rs:468-475
#pragma unroll
                for (int pr = 0; pr < NRW / 2; ++pr) {
                    v4f o;
                    const v4f g1 = acc[2 * pr][j], up = acc[2 * pr + 1][j];
                    { const float den = 1.0f + q3_expf(-g1.x); o.x = (g1.x * (1.0f / den)) * up.x; }
                    { const float den = 1.0f + q3_expf(-g1.y); o.y = (g1.y * (1.0f / den)) * up.y; }
                    { const float den = 1.0f + q3_expf(-g1.z); o.z = (g1.z * (1.0f / den)) * up.z; }
                    { const float den = 1.0f + q3_expf(-g1.w); o.w = (g1.w * (1.0f / den)) * up.w; }
                    *(v4f*)(a.out0 + (size_t)sb * a.out0_stride + (size_t)(rb * (RTW / 2) + wr * (NRW / 2) + pr) * 16 + 4 * q) = o;
                }
            } else {
#pragma unroll
                for (int i = 0; i < NRW; ++i) {
                    const int r0 = (rb * RTW + NRW * wr + i) * 16 + 4 * q;
                    const v4f o = acc[i][j];
                    if constexpr (EPI == EPI_QKV) {
                        float* dst;
                        if (r0 < a.rows0) dst = a.out0 + (size_t)sb * a.out0_stride + r0;
                        else if (r0 < a.rows0 + a.rows1) dst = a.out1 + (size_t)sb * a.out1_stride + (r0 - a.rows0);
                        else dst = a.out2 + (size_t)sb * a.out2_stride + (size_t)a.st[sb].pos * a.pos_stride + (r0 - a.rows0 - a.rows1);
                        *(v4f*)dst = o;
                    } else if constexpr (EPI == EPI_RESID) {
                        v4f* dst = (v4f*)(a.out0 + (size_t)sb * a.out0_stride + r0);
                        v4f x = *dst;
                        x.x = x.x + o.x; x.y = x.y + o.y; x.z = x.z + o.z; x.w = x.w + o.w;      // layers.rs:249-259
                        *dst = x;
                    } else {
                        *(v4f*)(a.out0 + (size_t)sb * a.out0_stride + r0) = o;
                    }
                }
            }
        }
        __syncthreads();                                         // the next block's prologue overwrites slot 0
    }
}

// ------------------------------------------------------------------------------------------------
// Batched decode matmul, round 4: in-lane accumulation at 16-32 columns (k_dgemm).
// k_bgemm splits K over the 8 waves of a workgroup and pays, per phase of 16-32 groups, a 64 KiB LDS term tile, two
// barriers and a 16-32-term dependent fold behind LDS reads: with the weight loads compiled out its launches kept 70 % of
// their time (profiles/r03_batch32_ablation.txt).  Here a WAVE owns one (16-row tile, 16-stream tile) accumulator tile for
// the whole contraction exactly like k_pgemm -- per group one MFMA, four converts and three packed operations, the
// g-ascending add of tensor.rs:53-60 done by the lane that holds the accumulator -- but sized for decode:
//   * one row tile x one stream tile per wave (SwiGLU: the w1 and the w3 tile of 16 hidden units), so a 256-tile matrix
//     gives 512 waves, two per CU on different SIMDs (k_pgemm's 2 x 2 tiles would leave half the CUs idle);
//   * a DEEP register ring (16 groups = 16 KiB of weights per wave in flight): with one wave per SIMD the whole register
//     file is there for it, and 512 waves x 16 KiB = 8 MB is what the HBM stream needs to stay busy;
//   * the group scales of the wave's row tile(s) and stream tile travel as one float4 load per tile per DEPTH groups
//     (chunk k + 1 requested at the head of ring iteration k, committed to a double-buffered wave-private LDS slice at its
//     end) and are read back per group (ds_read_b128 / b32, broadcast): 2 VMEM instructions per group instead of 4, and
//     no DPP broadcast.  (LDS-DMA for the scale tiles was tried first: with a global_load_lds anywhere in the function
//     hipcc waits vmcnt(0) at the head of the ring loop -- the guide's caveat -- and the ring never runs ahead);
//   * SwiGLU epilogue quantizes hb on the spot (tensor.rs:91-119): the four waves of a workgroup hold the 64 hidden units
//     of one quantization group for the same 16 streams, so W2's operand leaves this kernel packed and the separate
//     k_bquant_split<PRO_QUANT> launch (4.8 us per layer) is gone;
//   * classifier epilogue: logits + per-stream argmax slot per wave (sampler.rs:57-59 last-maximum rule).
// Same operations in the same order per (row, stream) as k_bgemm / k_gemv: bit-identical results.
// ------------------------------------------------------------------------------------------------
typedef unsigned dg_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i dg_as_v4i(dg_v4u v) { return (v4i){(int)v.x, (int)v.y, (int)v.z, (int)v.w}; }
// B operand (the step's packed int8 activations) of a workgroup:
//   BMODE 0  every wave pulls its B fragments from L2 inside its ring (waves numbered stream-tile-minor over the grid: the two
//            stream tiles of a row tile are neighbouring waves and share the weight lines through L1)
//   BMODE 1  the workgroup's waves work on ONE stream tile, staged in LDS once (ng KiB); the nst workgroups of a row group get
//            block ids 8 apart -- the same XCD under the observed round-robin placement -- and share the weights through its L2
//   BMODE 2  BOTH stream tiles in LDS (2 ng KiB, one workgroup per CU) and every wave computes both (PT = 2): a weight
//            fragment is requested ONCE per CU and feeds two MFMAs.  (Measured first: the two stream tiles as two waves
//            sharing the weight lines "through L1" -- both requests travel to L2, the CU's fill path carries the weights
//            twice and saturates at ~32 GB/s per CU: W1|W3 25.7-27.5 us, classifier 148 us whatever the depth or balance.)
// LDS per workgroup: [BMODE tiles] + per wave two chunks (double buffer) of DEPTH groups x 16 scales for each of its RT row
// tiles and PT stream tiles + [2][4][16] group maxima of the fused quantizer
__host__ __device__ inline size_t dgemm_smem_bytes(int rt, int waves, int depth, int bmode, int ng) {
    const int pt = bmode == 2 ? 2 : 1;
    return (size_t)bmode * ng * 1024 + 4 * ((size_t)waves * (rt + pt) * depth * 16 + 2 * 4 * 16);
}

// EPI_SWIGLU: RT = 2 (the w1 and the w3 tile of 16 hidden units); the fused quantizer (a.pack_q) needs the four unit tiles of a
// quantization group of the hidden vector in one workgroup: 4 waves.
// A wave walks its row tasks (rtask, rtask + rstride, ...) as ONE stream of groups: the ring, the scale chunks and the B
// prefetch run across task boundaries (the classifier gives every wave ~10-40 tasks; restarting the ring per task left the
// HBM stream idle for a round trip per 64 groups).  Needs ng % DEPTH == 0 (host).
__device__ __attribute__((noinline)) void dg_fill_lds(const v4i* src, char* lds, int kib, int wave, int nwaves, int lane) {
    for (int g = wave; g < kib; g += nwaves)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)g * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(lds + (size_t)g * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// developer ablation of k_dgemm, compile-time only (-DQ3_DABLATE=bits; results are wrong by construction, only time is read):
// 1 no weight (A) loads, 2 no convert / scale / add chain, 4 no MFMAs, 8 no B fragment reads (LDS or L2)
#ifdef Q3_DABLATE
#define Q3_DABL(bit) ((Q3_DABLATE & (bit)) != 0)
#else
#define Q3_DABL(bit) false
#endif
template <int EPI, int DEPTH, int WAVES, int BMODE>
__global__ __launch_bounds__(WAVES * 64) void k_dgemm(const BGemmArgs a) {
    constexpr int RT = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int PT = (BMODE == 2) ? 2 : 1;                     // stream tiles per wave
    // scale chunk = the DEPTH groups of one ring iteration, requested in stage 0 of the iteration BEFORE (so it travels with
    // the fragments of the same groups: a chunk requested only 8 groups ahead of its use made every commit wait out a full
    // memory round trip and drained the ring to 8 groups whatever DEPTH was -- r04 ablation: 12 us of a 28 us W1|W3 launch),
    // committed to the wave's single LDS slice in the last stage, after the last scale read of the current chunk
    constexpr int CH = DEPTH;
    static_assert(DEPTH * 4 <= 128, "a scale chunk is two float4 per lane at most");
    constexpr int SCN = (CH * 4 + 63) / 64;                      // float4 of a tile's scale chunk per lane (1; 2 for the 32-group ring)
    constexpr int CF = CH * 16;                                  // floats per scale chunk (CH groups x 16 rows / streams)
    static_assert(DEPTH % CH == 0, "whole chunks per ring iteration");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, s = lane & 15;
    const int ng = a.ng;
    const int nst = (a.n_streams + 15) >> 4;                    // stream tiles (1 or 2; BMODE 2: 2)
    const size_t tile_v4 = (size_t)ng * 64;                      // v4i per packed tile (weights or activations)
    float* lsc0 = (float*)(smem_raw + (size_t)BMODE * ng * 1024);
    float* lsc = lsc0 + (size_t)wave * (RT + PT) * CF;          // [RT row tiles + PT stream tiles][CF]
    float* red = lsc0 + (size_t)WAVES * (RT + PT) * CF;         // SwiGLU: [2 stream tiles][4 unit tiles][16 streams] maxima
    const int nrt = a.ntiles / RT;                               // row tasks
    int rtask, rstride, pt, slot, ri = wave;                     // ri: row task of the wave inside the workgroup
    if constexpr (BMODE == 0) {
        const int t = blockIdx.x * WAVES + wave;
        rtask = t / nst;
        pt = t - rtask * nst;
        rstride = ((int)gridDim.x * WAVES) / nst;                // (gridDim.x * WAVES is a multiple of nst: host)
        slot = rtask;
    } else if constexpr (BMODE == 1) {
        const int b = blockIdx.x;
        int rg = b / nst;
        pt = b - rg * nst;
        if (nst == 2 && (b | 15) < (int)gridDim.x) { rg = (b >> 4) * 8 + (b & 7); pt = (b >> 3) & 1; }
        rtask = rg * WAVES + wave;
        rstride = ((int)gridDim.x / nst) * WAVES;
        slot = rtask;
    } else {
        pt = 0;                                                  // (both stream tiles)
        rtask = (int)blockIdx.x * WAVES + wave;
        rstride = (int)gridDim.x * WAVES;
        slot = rtask;
    }
    const v4i* lb = (const v4i*)smem_raw;                        // the stream tile(s) in LDS: [PT][ng][64 lanes]
    const v4i* xbase = (const v4i*)a.xq + (size_t)pt * tile_v4;
    const v4f* gxs = (const v4f*)(a.xs + (size_t)pt * ng * 16);  // (PT = 2: the second tile's scales follow at ng * 16 floats)
    if constexpr (BMODE != 0) {
        // activations -> LDS by LDS-DMA: ng KiB per stream tile, wave w copies every WAVES-th KiB, every request in flight at
        // once (through registers it was 16 requests per wave and round trip: three dependent round trips before the first
        // MFMA of the W1|W3 launch).  The copy lives in a function of its own: with a global_load_lds in THIS function hipcc
        // waits vmcnt(0) at the head of the ring loop (see the header comment); a call boundary resets its bookkeeping.
        const int kib = BMODE * ng;                              // BMODE 2: the two tiles are adjacent in a.xq
        dg_fill_lds((const v4i*)a.xq + (BMODE == 2 ? 0 : (size_t)pt * tile_v4), smem_raw, kib, wave, WAVES, lane);
        __syncthreads();
    }
    unsigned long long best[PT];                                 // EPI_LOGITS: running argmax key of stream s (over this lane's rows)
#pragma unroll
    for (int p = 0; p < PT; ++p) best[p] = 0ull;
    if (rtask < nrt) {
        const int ntk = (nrt - rtask + rstride - 1) / rstride;   // row tasks of this wave
        int sl[SCN];                                             // scale chunk: float4 indices of this lane (upper lanes idle)
#pragma unroll
        for (int i = 0; i < SCN; ++i) sl[i] = min(lane + 64 * i, CH * 4 - 1);
        const int nsv = ng * 4;                                  // float4 per scale tile
        // fragments by buffer loads: resource = the packed matrix (SGPRs), scalar offset = tile * ng KiB + group * 1 KiB, vector
        // offset = lane * 16 -- no vector ALU in the address path (flat global loads cost a 64-bit v_lshl_add per request)
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.wq, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, 0x7fffffff, 0x00020000);
        const int tile_b = ng << 10;                             // bytes per packed tile
        int cur = rtask, nxt = ntk > 1 ? rtask + rstride : rtask;        // current / next row task (a wave without a next one re-reads)
        int k = 0;
        v4f sc[RT + PT][SCN];                                    // scale chunk in flight (global -> registers -> LDS)
        auto chunk_load = [&](int task, int c) {                 // chunk c = groups [c*CH, +CH) of row task `task`
            const v4f* gws = (const v4f*)(a.ws + (size_t)(task * RT) * ng * 16);
#pragma unroll
            for (int i = 0; i < SCN; ++i) {
                const int f0 = c * (CH * 4) + sl[i];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) sc[rt][i] = (gws + (size_t)rt * nsv)[f0];
#pragma unroll
                for (int p = 0; p < PT; ++p) sc[RT + p][i] = (gxs + (size_t)p * nsv)[f0];
            }
        };
        auto chunk_commit = [&]() {
#pragma unroll
            for (int i = 0; i < SCN; ++i) {
                if (lane + 64 * i < CH * 4) {
#pragma unroll
                    for (int kk = 0; kk < RT + PT; ++kk) ((v4f*)(lsc + (size_t)kk * CF))[lane + 64 * i] = sc[kk][i];
                }
            }
        };
        v4i fa[DEPTH][RT], fb[BMODE != 0 ? 1 : DEPTH];
        auto load_ab = [&](int slot_, int task, int g) {
            const int goff = g << 10;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                fa[slot_][rt] = Q3_DABL(1) ? (v4i){lane, goff, task, rt}
                                           : dg_as_v4i(__builtin_amdgcn_raw_buffer_load_b128(wr, lane << 4, (task * RT + rt) * tile_b + goff, 0));
            if constexpr (BMODE == 0) fb[slot_] = Q3_DABL(8) ? (v4i){lane, goff, 1, 2} : dg_as_v4i(__builtin_amdgcn_raw_buffer_load_b128(xr, lane << 4, goff, 0));
        };
        chunk_load(cur, 0);                                      // oldest loads
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) load_ab(d, cur, d);
        chunk_commit();                                          // (counted wait: the ring fill stays in flight)
        wave_lds_sync();
        v4f acc[RT][PT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int p = 0; p < PT; ++p) acc[rt][p] = (v4f){-0.0f, -0.0f, -0.0f, -0.0f};   // Iterator::sum::<f32>() identity
        v4i cc[RT][PT], cn[RT][PT];
        v4f wsc[RT], wsn[RT];
        float xsc[PT], xsn[PT];
        v4i bc[PT], bn[PT];                                      // BMODE 1/2: B fragments of the group whose MFMAs are issued next
#pragma unroll
        for (int p = 0; p < PT; ++p) bc[p] = bn[p] = (v4i){0, 0, 0, 0};
        auto bfrag = [&](v4i (&b)[PT], int g) {
            if constexpr (BMODE != 0) {
#pragma unroll
                for (int p = 0; p < PT; ++p) b[p] = Q3_DABL(8) ? (v4i){lane, g, p, 3} : lb[(size_t)p * tile_v4 + (size_t)(g >= ng ? g - ng : g) * 64 + lane];
            }
        };
        auto mfma_group = [&](v4i (&c)[RT][PT], int slot_, const v4i (&bl)[PT]) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int p = 0; p < PT; ++p)
                    c[rt][p] = Q3_DABL(4) ? fa[slot_][rt] + (BMODE != 0 ? bl[p] : fb[BMODE != 0 ? 0 : slot_])
                                          : __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[slot_][rt], BMODE != 0 ? bl[p] : fb[BMODE != 0 ? 0 : slot_], (v4i){0, 0, 0, 0}, 0, 0, 0);
        };
        auto scales = [&](v4f (&w)[RT], float (&x)[PT], int u) {      // scales of group u of the chunk in LDS
            const float* b0 = lsc;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) w[rt] = *(const v4f*)(b0 + rt * CF + u * 16 + 4 * q);
#pragma unroll
            for (int p = 0; p < PT; ++p) x[p] = b0[(RT + p) * CF + u * 16 + s];
        };
        auto math_group = [&](const v4i (&cg)[RT][PT], const v4f (&w)[RT], const float (&x)[PT]) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    const v4i c = cg[rt][p];
                    if (Q3_DABL(2)) { acc[rt][p].x = __int_as_float(__float_as_int(acc[rt][p].x) ^ c.x ^ c.y ^ c.z ^ c.w); continue; }
                    // tensor.rs:59  ((dot as f32) * ws) * xs, then the g-ascending add; pairs (rows 4q, 4q+1), (4q+2, 4q+3)
                    pk2 t01 = (pk2){(float)c.x, (float)c.y} * (pk2){w[rt].x, w[rt].y};
                    pk2 t23 = (pk2){(float)c.z, (float)c.w} * (pk2){w[rt].z, w[rt].w};
                    asm("" : "+v"(t01)); asm("" : "+v"(t23));
                    t01 = t01 * (pk2){x[p], x[p]}; t23 = t23 * (pk2){x[p], x[p]};
                    asm("" : "+v"(t01)); asm("" : "+v"(t23));
                    v4f& ac = acc[rt][p];
                    pk2 a01 = (pk2){ac.x, ac.y} + t01, a23 = (pk2){ac.z, ac.w} + t23;
                    asm("" : "+v"(a01)); asm("" : "+v"(a23));
                    ac.x = a01.x; ac.y = a01.y; ac.z = a23.x; ac.w = a23.y;
                }
        };
        auto rotate = [&]() {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                wsc[rt] = wsn[rt];
#pragma unroll
                for (int p = 0; p < PT; ++p) cc[rt][p] = cn[rt][p];
            }
#pragma unroll
            for (int p = 0; p < PT; ++p) { xsc[p] = xsn[p]; if constexpr (BMODE != 0) bc[p] = bn[p]; }
        };
        // ---- epilogue of row task `task`: lane (s, q) owns out[stream (pt + p)*16 + s][rows 4q .. 4q+3 of the row tile]
        auto epilogue = [&](int task) {
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const int tp = pt + p;                            // stream tile
                const int sb = tp * 16 + s;
                const bool live = sb < a.n_streams;
                if constexpr (EPI == EPI_SWIGLU) {
                    // packed tiles alternate w1 | w3 of the same 16 hidden units            layers.rs:468-475
                    v4f o;
                    const v4f g1 = acc[0][p], up = acc[RT - 1][p];
                    { const float den = 1.0f + q3_expf(-g1.x); o.x = (g1.x * (1.0f / den)) * up.x; }
                    { const float den = 1.0f + q3_expf(-g1.y); o.y = (g1.y * (1.0f / den)) * up.y; }
                    { const float den = 1.0f + q3_expf(-g1.z); o.z = (g1.z * (1.0f / den)) * up.z; }
                    { const float den = 1.0f + q3_expf(-g1.w); o.w = (g1.w * (1.0f / den)) * up.w; }
                    if (a.out0 != nullptr && live) *(v4f*)(a.out0 + (size_t)sb * a.out0_stride + (size_t)task * 16 + 4 * q) = o;
                    if (a.pack_q != nullptr) {
                        // hq = quantize(hb) (tensor.rs:91-119) for W2: group = the 64 hidden units of this workgroup (unit tile =
                        // wave), stream s; the packed operand piece (q' = wave, s) takes this lane's four bytes at 4q.  (host: one
                        // row task per wave, four waves per workgroup)
                        float* rd = red + p * 64;
                        float m = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w)));
                        m = fmaxf(m, __shfl_xor(m, 16));
                        m = fmaxf(m, __shfl_xor(m, 32));
                        if (q == 0) rd[ri * 16 + s] = m;
                        __syncthreads();
                        m = fmaxf(fmaxf(rd[s], rd[16 + s]), fmaxf(rd[32 + s], rd[48 + s]));
                        const float scale = m / 127.0f;
                        int q0 = 0, q1 = 0, q2 = 0, q3 = 0;
                        if (scale != 0.0f) {
                            q0 = quant_round_i8(o.x / scale);
                            q1 = quant_round_i8(o.y / scale);
                            q2 = quant_round_i8(o.z / scale);
                            q3 = quant_round_i8(o.w / scale);
                        }
                        const int gh = task >> 2, ngh = a.ntiles >> 3;               // group index / groups per row of W2's contraction
                        if (live) {
                            int8_t* dst = a.pack_q + ((((size_t)tp * ngh + gh) * 64 + (ri * 16 + s)) * 16 + 4 * q);
                            *(int*)dst = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((q3 & 0xff) << 24);
                            if (ri == 0 && q == 0) a.pack_s[((size_t)tp * ngh + gh) * 16 + s] = scale;
                        }
                    }
                } else {
                    const int r0 = task * 16 + 4 * q;
                    const v4f o = acc[0][p];
                    if constexpr (EPI == EPI_QKV) {
                        if (live) {
                            float* dst;
                            if (r0 < a.rows0) dst = a.out0 + (size_t)sb * a.out0_stride + r0;
                            else if (r0 < a.rows0 + a.rows1) dst = a.out1 + (size_t)sb * a.out1_stride + (r0 - a.rows0);
                            else dst = a.out2 + (size_t)sb * a.out2_stride + (size_t)a.st[sb].pos * a.pos_stride + (r0 - a.rows0 - a.rows1);
                            *(v4f*)dst = o;
                        }
                    } else if constexpr (EPI == EPI_RESID) {
                        if (live) {
                            v4f* dst = (v4f*)(a.out0 + (size_t)sb * a.out0_stride + r0);
                            v4f x = *dst;
                            x.x = x.x + o.x; x.y = x.y + o.y; x.z = x.z + o.z; x.w = x.w + o.w;      // layers.rs:249-259
                            *dst = x;
                        }
                    } else if constexpr (EPI == EPI_LOGITS) {
                        if (live) {
                            if (a.out0 != nullptr) *(v4f*)(a.out0 + (size_t)sb * a.out0_stride + r0) = o;
                            const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const unsigned long long key = ((unsigned long long)total_order_key(ov[i]) << 32) | (unsigned)(r0 + i);
                                best[p] = key > best[p] ? key : best[p];
                            }
                        }
                    } else {
                        if (live) *(v4f*)(a.out0 + (size_t)sb * a.out0_stride + r0) = o;
                    }
                }
            }
        };
        // software pipeline (k_pgemm's): stage u requests the group DEPTH - 1 ahead, issues the MFMAs and the LDS scale reads
        // of the next group (BMODE 1/2: and the B fragment of the one after), then runs the convert / scale / add chain of
        // its own group while those are in flight.  An iteration = DEPTH groups of one row task; it also carries the NEXT
        // iteration's scale chunk: requested in stage 0, committed to the other LDS buffer in the last stage, just before
        // that chunk's first scales are read.  In the last iteration of a row task "ahead" means the next row task.
        bfrag(bc, 0);
        mfma_group(cc, 0, bc);
        bfrag(bc, 1);
        scales(wsc, xsc, 0);
        int g0 = 0;
        for (;;) {
            const bool last = g0 + DEPTH == ng;                  // wave-uniform
            const int ptask = last ? nxt : cur;                  // task / first group of the requests that run past this iteration
            const int pg0 = last ? -DEPTH : g0;
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                if (u == 0) load_ab(DEPTH - 1, cur, g0 + DEPTH - 1);
                else load_ab(u - 1, ptask, pg0 + u + DEPTH - 1);
                if (u == 0) chunk_load(ptask, (pg0 + DEPTH) / CH);    // the next iteration's scales (maybe the next task's chunk 0)
                mfma_group(cn, (u + 1) % DEPTH, bc);
                bfrag(bn, g0 + u + 2);
                if (u == DEPTH - 1) { wave_lds_sync(); chunk_commit(); wave_lds_sync(); scales(wsn, xsn, 0); }
                else scales(wsn, xsn, u + 1);
                math_group(cc, wsc, xsc);
                rotate();
                __builtin_amdgcn_sched_barrier(0);
            }
            g0 += DEPTH;
            if (last) {
                epilogue(cur);
                if (++k == ntk) break;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int p = 0; p < PT; ++p) acc[rt][p] = (v4f){-0.0f, -0.0f, -0.0f, -0.0f};
                g0 = 0;
                cur = nxt;
                nxt = k + 1 < ntk ? cur + rstride : cur;
            }
        }
    }
    if constexpr (EPI == EPI_LOGITS) {
        // sampler.rs:57-59 (last maximum): max over the rows this wave saw for stream s -- lanes s, s+16, s+32, s+48 --
        // one slot per wave of a stream tile (slot = its first row task: < rstride); k_next_batch reduces the slots
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            unsigned long long bp = best[p];
#pragma unroll
            for (int m = 16; m < 64; m <<= 1) {
                const unsigned lo = __shfl_xor((unsigned)bp, m);
                const unsigned hi = __shfl_xor((unsigned)(bp >> 32), m);
                const unsigned long long o = ((unsigned long long)hi << 32) | lo;
                bp = o > bp ? o : bp;
            }
            const int sbl = (pt + p) * 16 + s;
            if (q == 0 && sbl < a.n_streams && slot < a.nslots) a.slots[(size_t)sbl * a.nslots + slot] = bp;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Batched attention: one workgroup per (stream, kv head).  Wave w < kv_mul owns query head kvh*kv_mul + w; the
// extra wave normalises + rotates the key row.  The kv head's K (then V) rows are staged in LDS ONCE for all
// kv_mul query heads (grouped-query sharing), 32 streams x 8 kv heads = 256 workgroups = one per CU.  Every
// sum is walked in the reference order (layers.rs:346-419,495-506), so the result is bit-identical to k_attn in
// reference-order mode; Q3_FLAG_FAST engines use this exact path too (exact is always admissible).
// ------------------------------------------------------------------------------------------------
constexpr int kGqaTch = 64;             // most timesteps staged per LDS round (one score dot per lane and chunk)
constexpr int kGqaSlots = 8;            // float4 staging registers per thread (tch*hd/4 <= slots * threads); 16 made the kernel spill
__host__ __device__ inline int gqa_att_stride(int seq_len) { return (seq_len + 255) & ~255; }
// timesteps per round: the workgroup's (kv_mul + 1) * 64 threads hold a chunk in kGqaSlots float4 registers each
__host__ __device__ inline int gqa_tch(int hd, int kv_mul) {
    int t = kGqaTch;
    while (t > 8 && t * (hd / 4) > kGqaSlots * (kv_mul + 1) * 64) t >>= 1;
    return t;
}
__host__ __device__ inline size_t attn_gqa_smem_bytes(int hd, int kv_mul, int seq_len) {
    const size_t nw = (size_t)kv_mul + 1;
    return 4 * (nw * hd * 3 + (size_t)kv_mul * gqa_att_stride(seq_len) + (size_t)gqa_tch(hd, kv_mul) * (hd + kKPad));
}

#ifdef Q3_DEV
#define GQA_STAMP(i) do { if (a0.stamps != nullptr && blockIdx.x == 3 && blockIdx.y == gridDim.y - 1 && threadIdx.x == 0) a0.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GQA_STAMP(i) do { } while (0)
#endif
// head h's hd outputs of stream sbi: plain f32 store, plus (pack_q != nullptr) the Wo matmul's quantized operand
__device__ __forceinline__ void gqa_store(const AttnArgs& a0, size_t sbi, int h, int hd, int lane, const float (&o)[4]) {
    float* out = a0.xb + sbi * a0.sb_xb + (size_t)h * hd;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < hd) out[lane + 64 * u] = o[u];
    if (a0.pack_q != nullptr) {
        // the Wo matmul's activation prologue, fused: quantize this head's hd outputs (hd % G == 0, so its groups
        // are whole) exactly as tensor.rs:91-119 and store them in the packed operand order of q3_batch.h
        // (G is a power of two >= 64 on this path -- the packed operand order has G / 64 pieces per lane quarter -- so every
        // quotient below is a shift: as run-time divisions they were ~40 instructions each, ten per lane, in every workgroup's tail)
        const int G = a0.group, upg = G >> 6, nj = G >> 6;
        const int lg = __builtin_ctz((unsigned)G), lnj = lg - 6;
        const int ngx = (a0.n_heads * hd) >> lg;
        const int nt = (int)(sbi >> 4), s = (int)(sbi & 15);
        float mu[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) mu[u] = group_max_f32((lane + 64 * u < hd) ? fabsf(o[u]) : 0.0f, 64);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (lane + 64 * u < hd) {
                float m = mu[u];
                if (upg >= 2) m = fmaxf(mu[u & ~1], mu[u | 1]);
                if (upg >= 4) m = fmaxf(fmaxf(mu[0], mu[1]), fmaxf(mu[2], mu[3]));
                const float scale = m / 127.0f;
                const int qv = (scale != 0.0f) ? quant_round_i8(o[u] / scale) : 0;
                const int k0 = h * hd + 64 * u + lane;
                const int g = k0 >> lg, within = (k0 & (G - 1)) >> 4;
                const int qq = within >> lnj, j = within & (nj - 1);
                a0.pack_q[((((size_t)nt * ngx + g) * nj + j) * 64 + (qq * 16 + s)) * 16 + (k0 & 15)] = (int8_t)qv;
                if ((k0 & (G - 1)) == 0) a0.pack_s[((size_t)nt * ngx + g) * 16 + s] = scale;
            }
        }
    }
}

// HD_T / KVM_T: head_dim and kv_mul as compile-time constants for the listed models (128 with 2 or 4 query heads per kv
// head) -- the staging index arithmetic (row = idx / (hd/4), ...) then folds to a pointer increment per slot; 0 = read
// them from the arguments.
template <int HD_T, int KVM_T>
__global__ __launch_bounds__(512) void k_attn_gqa(const AttnArgs a0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    GQA_STAMP(0);
    const int hd = HD_T ? HD_T : a0.hd, kv_mul = KVM_T ? KVM_T : a0.n_heads / a0.n_kv_heads, nw = kv_mul + 1;
    const int kvh = blockIdx.x;
    const size_t sbi = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = nw * 64;
    const bool kwave = wave == kv_mul;
    const bool own_k = a0.k_in_cache == 0;         // decode: this kernel normalises + rotates the position's key row
    const int h = kvh * kv_mul + (kwave ? 0 : wave);
    const size_t kvd = (size_t)a0.n_kv_heads * hd;
    const int ast = gqa_att_stride(a0.seq_len);
    const int kld = hd + kKPad;

    float* q_s = (float*)smem_raw;                 // [kv_mul][hd] normalised + rotated queries, then k_s[hd]
    float* k_s = q_s + kv_mul * hd;
    float* raw = k_s + hd;                         // [nw][hd]
    float* sq = raw + nw * hd;                     // [nw][hd]
    float* att_l = sq + nw * hd;                   // [kv_mul][ast] score / probability rows (the host picks this kernel only when they fit)
    float* buf = att_l + (size_t)kv_mul * ast;     // [tch][hd+4] K rows, later [tch][hd] V rows

    const State* st = a0.st + sbi;
    const int pos = a0.pos_override >= 0 ? a0.pos_override : st->pos;
    const int np = pos + 1;
    const float* qsrc = a0.q + sbi * a0.sb_q + (size_t)h * hd;
    const float* ksrc = a0.k_raw + sbi * a0.sb_kraw + (size_t)kvh * hd;
    float* key_cache = a0.key_cache + sbi * a0.sb_kv;
    const float* kbase = key_cache + (size_t)kvh * hd;
    const float* vbase = a0.value_cache + sbi * a0.sb_kv + (size_t)kvh * hd;
    float* att = att_l + (size_t)(kwave ? 0 : wave) * ast;
    const float* cs = a0.rope + (size_t)pos * hd;

    // ---- raw q heads / raw k row, norm weights and rope pairs; then chunk 0 of K: all in flight together
    float r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = min(lane + 64 * u, hd - 1);
        r[u] = kwave ? (own_k ? ksrc[i] : 0.0f) : qsrc[i];
    }
    RopeRegs rr;
    rope_regs_load(rr, kwave ? a0.k_norm_w : a0.q_norm_w, cs, hd);
    const int q4s = __builtin_ctz(hd >> 2);        // float4 per row = 1 << q4s
    v4f sr[kGqaSlots];
    auto stage_issue_g = [&](const float* gbase, int t0, int cnt) {
        const int total = cnt << q4s;
#pragma unroll
        for (int u = 0; u < kGqaSlots; ++u) {
            // unconditional: slots past the chunk re-read its last float4.  (A per-slot `if` put every load in its own
            // basic block; with the spills that followed, hipcc waited for each load before issuing the next -- 8,000 cycles
            // to request one 64 KB chunk, measured with the dev stamps.)
            const int ii = min(tid + u * nthr, total - 1);
            const int row = ii >> q4s, c = ii & ((1 << q4s) - 1);
            sr[u] = *(const v4f*)(gbase + (size_t)(t0 + row) * kvd + 4 * c);
        }
    };
    auto stage_commit_g = [&](int ld, int t0, int cnt, int skip) {
        const int total = cnt << q4s;
#pragma unroll
        for (int u = 0; u < kGqaSlots; ++u) {
            const int idx = tid + u * nthr;
            if (idx < total) {
                const int row = idx >> q4s, c = idx & ((1 << q4s) - 1);
                if (t0 + row != skip) *(v4f*)(buf + row * ld + 4 * c) = sr[u];
            }
        }
    };
    const int tch = gqa_tch(hd, kv_mul);
    const int nch = (np + tch - 1) / tch;
    stage_issue_g(kbase, 0, min(tch, np));

    float* raw_w = raw + wave * hd;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < hd) raw_w[lane + 64 * u] = r[u];
    wave_lds_sync();
    if (!kwave || own_k) wave_norm_rope(kwave ? k_s : q_s + wave * hd, raw_w, sq + wave * hd, rr, hd, 1);   // layers.rs:346-372
    __syncthreads();
    if (kwave && own_k) {                                   // K is normalised + rotated in place in the cache
        float* krow = key_cache + (size_t)pos * kvd + (size_t)kvh * hd;
        for (int i = lane; i < hd; i += 64) krow[i] = k_s[i];
    }
    const float scale = 1.0f / sqrtf((float)hd);   // (head_dim as f32).sqrt().recip()
    GQA_STAMP(1);

    // ---- scores: att[t] = (q . K[t]) * scale                                   layers.rs:391-401
    for (int c = 0; c < nch; ++c) {
        const int t0 = c * tch, cnt = min(tch, np - t0);
        if (c == 8) GQA_STAMP(8);
        stage_commit_g(kld, t0, cnt, own_k ? pos : -1);
        if (own_k && pos >= t0 && pos < t0 + cnt)  // the current position's K comes from this kernel, not the cache
            for (int i = tid; i < hd; i += nthr) buf[(pos - t0) * kld + i] = k_s[i];
        if (c == 8) GQA_STAMP(9);
        __syncthreads();
        if (c == 8) GQA_STAMP(10);
        if (c + 1 < nch) stage_issue_g(kbase, t0 + tch, min(tch, np - t0 - tch));     // next chunk under the dots
        if (c == 8) GQA_STAMP(11);
        if (!kwave) {
            const v4f* q4 = (const v4f*)(q_s + wave * hd);
            for (int t = lane; t < cnt; t += 64) {
                const v4f* k4 = (const v4f*)(buf + t * kld);
                float dot = -0.0f;
                const int nq = hd >> 2;
                int i = 0;
                for (; i + 8 <= nq; i += 8) {          // 8 float4 of K and q at a time: 64 registers (16 at a time spilled)
                    v4f kk[8], qq[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { kk[u] = k4[i + u]; qq[u] = q4[i + u]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
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
        }
        if (c == 8) GQA_STAMP(12);
        __syncthreads();                           // chunk consumed before the next one lands in buf
        if (c == 8) GQA_STAMP(13);
    }
    GQA_STAMP(2);

    // ---- softmax, one wave per head                                            layers.rs:495-506
    if (!kwave) {
        wave_lds_sync();
        float m = -__builtin_inff();
        for (int t = lane; t < np; t += 64) m = fmaxf(m, att[t]);
        m = group_max_f32(m, 64);
        const int npad = (np + 255) & ~255;
        for (int t = lane; t < npad; t += 64) att[t] = (t < np) ? q3_expf(att[t] - m) : 0.0f;   // +0.0 padding: sums unchanged
        wave_lds_sync();
        float sum;
        if (np < 256) {
            const int nq4 = np >> 2;
            sum = seq_chain(-0.0f, (const v4f*)att, nq4);
            for (int t = nq4 << 2; t < np; ++t) sum = sum + att[t];
        } else {
            sum = seq_sum_blocks(att, 64, npad >> 6, npad >> 6, nullptr);
        }
        const float inv = 1.0f / sum;
        for (int t = lane; t < np; t += 64) att[t] = att[t] * inv;
        wave_lds_sync();
    }

    GQA_STAMP(3);
    // (V chunk 0 is requested only now: holding its 16 float4 staging registers across the softmax -- whose exact sum keeps
    // up to 64 registers of terms -- made the kernel spill, and hipcc then waits on every staging load individually)
    stage_issue_g(vbase, 0, min(tch, np));
    // ---- xb = sum_t att[t] * V[t]  (fill(0.0) then += in t order)              layers.rs:406-417
    float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int c = 0; c < nch; ++c) {
        const int t0 = c * tch, cnt = min(tch, np - t0);
        stage_commit_g(hd, t0, cnt, -1);
        __syncthreads();
        if (c + 1 < nch) stage_issue_g(vbase, t0 + tch, min(tch, np - t0 - tch));
        if (!kwave && hd == 128) {
            // head_dim 128 (every listed model): the lane's two element chains (lane, lane + 64) advance together and
            // share the probability reads, which come as float4 -- 2.25 LDS instructions per timestep instead of 4
            // (measured with tools/lds_probe.hip: these loops cost ~20 cycles per LDS instruction, not per add)
            const v4f* w4 = (const v4f*)(att + t0);
            const float* v0 = buf + lane;
            float o0 = o[0], o1 = o[1];
            int t = 0;
            for (; t + 8 <= cnt; t += 8, v0 += 8 * 128) {
                const v4f pa = w4[t >> 2], pb = w4[(t >> 2) + 1];
                float a0[8], a1[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) { a0[x] = v0[x * 128]; a1[x] = v0[x * 128 + 64]; }
                const float ww[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    const float p0 = ww[x] * a0[x], p1 = ww[x] * a1[x];
                    o0 = o0 + p0;
                    o1 = o1 + p1;
                }
            }
            for (; t < cnt; ++t, v0 += 128) {
                const float wt = att[t0 + t];
                const float p0 = wt * v0[0], p1 = wt * v0[64];
                o0 = o0 + p0;
                o1 = o1 + p1;
            }
            o[0] = o0;
            o[1] = o1;
        } else if (!kwave) {
            const float* w = att + t0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (lane + 64 * u < hd) {          // wave-uniform for hd % 64 == 0; guarded lanes otherwise
                    const float* v = buf + lane + 64 * u;
                    float acc = o[u];
                    int t = 0;
                    for (; t + 16 <= cnt; t += 16) {
                        float vv[16], ww[16];
#pragma unroll
                        for (int x = 0; x < 16; ++x) { vv[x] = v[(t + x) * hd]; ww[x] = w[t + x]; }
#pragma unroll
                        for (int x = 0; x < 16; ++x) { const float p = ww[x] * vv[x]; acc = acc + p; }
                    }
                    for (; t < cnt; ++t) { const float p = w[t] * v[t * hd]; acc = acc + p; }
                    o[u] = acc;
                }
            }
        }
        __syncthreads();
    }
    GQA_STAMP(4);
    if (!kwave) gqa_store(a0, sbi, h, hd, lane, o);
}

// ------------------------------------------------------------------------------------------------
// k_attn_gqa with the roles split by wave (head_dim 128, KVM = 2 or 4 query heads per kv head; 512 threads): waves
// 0..KVM-1 own one query head each and do nothing but arithmetic -- the score dots, the softmax, the two V chains per lane
// -- while the other 8-KVM waves only move data: the K (then V) chunks travel global -> registers -> one of TWO LDS tiles,
// two chunks ahead of the arithmetic, with ONE barrier per chunk.  In k_attn_gqa every wave staged and then computed: at a
// context of 2,048 positions a 64-timestep chunk cost 4,300 cycles in the score loop (commit 870 + issue 1,020 + dots 2,100
// + two barriers) and 3,400 in the V loop; here the head waves see the dots / the fold and one barrier.  During the
// softmax the first KVM staging waves take every other 64-entry block of their head's exps.  Same operations in the same
// order as k_attn_gqa (and k_attn in reference-order mode): bit-identical results.
// ------------------------------------------------------------------------------------------------
constexpr int kG2Threads = 512, kG2Waves = 8, kG2Tch = 64, kG2Hd = 128;
__host__ __device__ inline size_t attn_gqa2_smem_bytes(int kv_mul, int seq_len) {
    return 4 * ((size_t)(kv_mul + 1) * kG2Hd * 3 + (size_t)kv_mul * gqa_att_stride(seq_len) + 64 + 2 * (size_t)kG2Tch * (kG2Hd + kKPad));
}
template <int KVM>
__global__ __launch_bounds__(kG2Threads) void k_attn_gqa2(const AttnArgs a0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    GQA_STAMP(0);
    constexpr int hd = kG2Hd, kld = hd + kKPad, TCH = kG2Tch, nw = KVM + 1;
    constexpr int NST = (kG2Waves - KVM) * 64;                 // staging threads
    constexpr int NF4 = TCH * (hd / 4);                        // float4 per chunk
    constexpr int NSLOT = (NF4 + NST - 1) / NST;               // 8 (KVM 4) / 6 (KVM 2) staging registers per set
    constexpr int TILE = TCH * kld;
    const int kvh = blockIdx.x;
    const size_t sbi = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool head = wave < KVM, kwave = wave == KVM, stager = wave >= KVM;
    const int sid = tid - KVM * 64;                            // staging thread index (stagers only)
    const bool own_k = a0.k_in_cache == 0;
    const int h = kvh * KVM + (head ? wave : 0);
    const size_t kvd = (size_t)a0.n_kv_heads * hd;
    const int ast = gqa_att_stride(a0.seq_len);

    float* q_s = (float*)smem_raw;                 // [KVM][hd]
    float* k_s = q_s + KVM * hd;                   // [hd]
    float* raw = k_s + hd;                         // [nw][hd]
    float* sq = raw + nw * hd;                     // [nw][hd]
    float* att_l = sq + nw * hd;                   // [KVM][ast]
    float* red = att_l + (size_t)KVM * ast;        // [64]: per-head max
    float* tiles = red + 64;                       // 2 x [TCH][kld]  (V: [TCH][hd])

    const State* st = a0.st + sbi;
    const int pos = __builtin_amdgcn_readfirstlane(a0.pos_override >= 0 ? a0.pos_override : st->pos);
    const int np = pos + 1;
    const float* qsrc = a0.q + sbi * a0.sb_q + (size_t)h * hd;
    const float* ksrc = a0.k_raw + sbi * a0.sb_kraw + (size_t)kvh * hd;
    float* key_cache = a0.key_cache + sbi * a0.sb_kv;
    const float* kbase = key_cache + (size_t)kvh * hd;
    const float* vbase = a0.value_cache + sbi * a0.sb_kv + (size_t)kvh * hd;
    float* att = att_l + (size_t)(head ? wave : 0) * ast;
    const float* cs = a0.rope + (size_t)pos * hd;
    const int skip = own_k ? pos : -1;             // this row of the K cache is produced here, not read

    v4f ra[NSLOT], rb[NSLOT];
    // rows clamped to the context (valid, written rows); float4 slots past the chunk re-read its last one
    auto issue = [&](v4f (&R)[NSLOT], const float* gbase, int t0) {
#pragma unroll
        for (int u = 0; u < NSLOT; ++u) {
            const int idx = min(sid + u * NST, NF4 - 1);
            const int row = min(t0 + (idx >> 5), np - 1), c = idx & 31;
            R[u] = *(const v4f*)(gbase + (size_t)row * kvd + 4 * c);
        }
    };
    auto commit = [&](const v4f (&R)[NSLOT], float* tile, int ld, int t0, int skip_row) {
#pragma unroll
        for (int u = 0; u < NSLOT; ++u) {
            const int idx = sid + u * NST;
            const int row = idx >> 5, c = idx & 31;
            if (idx < NF4 && t0 + row < np && t0 + row != skip_row) *(v4f*)(tile + row * ld + 4 * c) = R[u];
        }
    };
    // the position's own key row (normalised + rotated here) goes into the tile that holds timestep pos
    auto fix_k = [&](float* tile, int t0) {
        if (kwave && own_k && pos >= t0 && pos < t0 + TCH) {
            tile[(pos - t0) * kld + lane] = k_s[lane];
            tile[(pos - t0) * kld + lane + 64] = k_s[lane + 64];
        }
    };

    // ---- raw q heads / raw k row, norm weights and rope pairs; the staging waves request K chunks 0 and 1
    float r[2] = {0.0f, 0.0f};
    RopeRegs rr;
    if (wave <= KVM) {
#pragma unroll
        for (int u = 0; u < 2; ++u) r[u] = kwave ? (own_k ? ksrc[lane + 64 * u] : 0.0f) : qsrc[lane + 64 * u];
        rope_regs_load(rr, kwave ? a0.k_norm_w : a0.q_norm_w, cs, hd);
    }
    if (stager) {
        issue(ra, kbase, 0);
        if (TCH < np) issue(rb, kbase, TCH);
    }
    if (wave <= KVM) {
        float* raw_w = raw + wave * hd;
        raw_w[lane] = r[0];
        raw_w[lane + 64] = r[1];
        wave_lds_sync();
        if (head || own_k) wave_norm_rope(kwave ? k_s : q_s + wave * hd, raw_w, sq + wave * hd, rr, hd, 1);   // layers.rs:346-372
        wave_lds_sync();
        if (kwave && own_k) {                                   // K is normalised + rotated in place in the cache
            float* krow = key_cache + (size_t)pos * kvd + (size_t)kvh * hd;
            krow[lane] = k_s[lane];
            krow[lane + 64] = k_s[lane + 64];
        }
    }
    const float scale = 1.0f / sqrtf((float)hd);   // (head_dim as f32).sqrt().recip()
    if (stager) {
        commit(ra, tiles, kld, 0, skip);
        fix_k(tiles, 0);
        if (2 * TCH < np) issue(ra, kbase, 2 * TCH);
    }
    __syncthreads();
    GQA_STAMP(1);

    // ---- scores: att[t] = (q . K[t]) * scale, one timestep per lane                layers.rs:391-401
    auto dots = [&](const float* tile, int t0) {
        const int t = lane;
        if (t0 + t < np) {
            const v4f* k4 = (const v4f*)(tile + t * kld);
            const v4f* q4 = (const v4f*)(q_s + wave * hd);
            float dot = -0.0f;
#pragma unroll
            for (int i = 0; i < hd / 4; i += 8) {
                v4f kk[8], qq[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { kk[u] = k4[i + u]; qq[u] = q4[i + u]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const v4f pr = qq[u] * kk[u];          // products are independent of the chain: packed multiplies
                    dot = dot + pr.x;
                    dot = dot + pr.y;
                    dot = dot + pr.z;
                    dot = dot + pr.w;
                }
            }
            att[t0 + t] = dot * scale;
        }
    };
    for (int c0 = 0; c0 < np; c0 += 2 * TCH) {
        if (stager && c0 + TCH < np) {
            commit(rb, tiles + TILE, kld, c0 + TCH, skip);
            fix_k(tiles + TILE, c0 + TCH);
            if (c0 + 3 * TCH < np) issue(rb, kbase, c0 + 3 * TCH);
        }
        if (c0 == 8 * TCH) GQA_STAMP(8);
        if (head) dots(tiles, c0);
        if (c0 == 8 * TCH) GQA_STAMP(9);
        __syncthreads();
        if (c0 == 8 * TCH) GQA_STAMP(10);
        if (c0 + TCH >= np) break;
        if (stager && c0 + 2 * TCH < np) {
            commit(ra, tiles, kld, c0 + 2 * TCH, skip);
            fix_k(tiles, c0 + 2 * TCH);
            if (c0 + 4 * TCH < np) issue(ra, kbase, c0 + 4 * TCH);
        }
        if (head) dots(tiles + TILE, c0 + TCH);
        __syncthreads();
    }
    GQA_STAMP(2);

    // ---- softmax, one wave per head (+ one helper wave for the exps)            layers.rs:495-506
    if (stager) {                                   // V chunks 0 and 1 travel under the softmax
        issue(ra, vbase, 0);
        if (TCH < np) issue(rb, vbase, TCH);
    }
    const int npad = (np + 255) & ~255;
    if (head) {
        float m = -__builtin_inff();
        for (int t = lane; t < np; t += 64) m = fmaxf(m, att[t]);
        m = group_max_f32(m, 64);
        if (lane == 0) red[wave] = m;
    }
    __syncthreads();
    {
        // head wave j takes the even 64-entry blocks of its row, staging wave KVM + j the odd ones
        const bool helper = wave >= KVM && wave < 2 * KVM;
        if (head || helper) {
            const int j = head ? wave : wave - KVM;
            float* row = att_l + (size_t)j * ast;
            const float m = red[j];
            for (int t = (head ? 0 : 64) + lane; t < npad; t += 128) row[t] = (t < np) ? q3_expf(row[t] - m) : 0.0f;   // +0.0 padding: sums unchanged
        }
    }
    __syncthreads();
    if (head) {
        float sum;
        if (np < 256) {
            const int nq4 = np >> 2;
            sum = seq_chain(-0.0f, (const v4f*)att, nq4);
            for (int t = nq4 << 2; t < np; ++t) sum = sum + att[t];
        } else {
            sum = seq_sum_blocks(att, 64, npad >> 6, npad >> 6, nullptr);
        }
        const float inv = 1.0f / sum;
        for (int t = lane; t < np; t += 64) att[t] = att[t] * inv;
        wave_lds_sync();
    }
    GQA_STAMP(3);

    // ---- xb = sum_t att[t] * V[t]  (fill(0.0) then += in t order)              layers.rs:406-417
    // the lane's two element chains (lane, lane + 64) advance together and share the probability reads.  (Measured and not
    // kept: the pair as packed f32 ops -- v_pk_mul/add issue at half rate, no gain --; four consecutive elements per lane with
    // two heads per wave and 16-byte LDS reads, also software-pipelined with one read behind each step: 41-48 cycles per
    // timestep against 37 for this form.)
    float o0 = 0.0f, o1 = 0.0f;
    auto fold = [&](const float* tile, int t0) {
        const int cnt = min(TCH, np - t0);
        const v4f* w4 = (const v4f*)(att + t0);
        const float* v0 = tile + lane;
        int t = 0;
        for (; t + 8 <= cnt; t += 8, v0 += 8 * hd) {
            const v4f pa = w4[t >> 2], pb = w4[(t >> 2) + 1];
            float x0[8], x1[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) { x0[x] = v0[x * hd]; x1[x] = v0[x * hd + 64]; }
            const float ww[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const float p0 = ww[x] * x0[x], p1 = ww[x] * x1[x];
                o0 = o0 + p0;
                o1 = o1 + p1;
            }
        }
        for (; t < cnt; ++t, v0 += hd) {
            const float wt = att[t0 + t];
            const float p0 = wt * v0[0], p1 = wt * v0[64];
            o0 = o0 + p0;
            o1 = o1 + p1;
        }
    };
    if (stager) {
        commit(ra, tiles, hd, 0, -1);
        if (2 * TCH < np) issue(ra, vbase, 2 * TCH);
    }
    __syncthreads();
    for (int c0 = 0; c0 < np; c0 += 2 * TCH) {
        if (stager && c0 + TCH < np) {
            commit(rb, tiles + TILE, hd, c0 + TCH, -1);
            if (c0 + 3 * TCH < np) issue(rb, vbase, c0 + 3 * TCH);
        }
        if (c0 == 8 * TCH) GQA_STAMP(11);
        if (head) fold(tiles, c0);
        if (c0 == 8 * TCH) GQA_STAMP(12);
        __syncthreads();
        if (c0 == 8 * TCH) GQA_STAMP(13);
        if (c0 + TCH >= np) break;
        if (stager && c0 + 2 * TCH < np) {
            commit(ra, tiles, hd, c0 + 2 * TCH, -1);
            if (c0 + 4 * TCH < np) issue(ra, vbase, c0 + 4 * TCH);
        }
        if (head) fold(tiles + TILE, c0 + TCH);
        __syncthreads();
    }
    GQA_STAMP(4);
    if (head) {
        const float o[4] = {o0, o1, 0.0f, 0.0f};
        gqa_store(a0, sbi, h, hd, lane, o);
    }
}

// ------------------------------------------------------------------------------------------------
// Dense-prefill attention helpers (k_attn_pf2 below)
// ------------------------------------------------------------------------------------------------
// exact sequential sum (from -0.0) of row[0 .. np): NQ float4 per lane, blocks of 4*NQ consecutive terms, terms past np are +0.0
template <int NQ>
__device__ __forceinline__ float row_exact_sum_regs(const float* row, int np) {
    const int j = threadIdx.x & 63;
    constexpr int BL = 4 * NQ;
    v4f r[NQ];
    float tot = 0.0f;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int t = j * BL + 4 * k;
        v4f v = {0.f, 0.f, 0.f, 0.f};
        if (t < np) v = *(const v4f*)(row + t);                 // (rows are padded to whole float4 of valid memory)
        v.x = (t + 0 < np) ? v.x : 0.0f; v.y = (t + 1 < np) ? v.y : 0.0f;
        v.z = (t + 2 < np) ? v.z : 0.0f; v.w = (t + 3 < np) ? v.w : 0.0f;
        r[k] = v;
        tot += (v.x + v.y) + (v.z + v.w);
    }
    return spec_sum_lanes(tot, (np + BL - 1) / BL, [&](float s0) {
#pragma unroll
        for (int k = 0; k < NQ; ++k) s0 = chain4(s0, r[k]);
        return s0;
    });
}
__device__ __forceinline__ float row_exact_sum(const float* row, int np) {
    if (np <= 256) return row_exact_sum_regs<1>(row, np);
    if (np <= 512) return row_exact_sum_regs<2>(row, np);
    if (np <= 1024) return row_exact_sum_regs<4>(row, np);
    if (np <= 2048) return row_exact_sum_regs<8>(row, np);
    if (np <= 4096) return row_exact_sum_regs<16>(row, np);
    // longer rows: 64 blocks folded out of memory every round (rare: contexts beyond 4,096 positions)
    const int j = threadIdx.x & 63;
    const int bl = (((np + 63) >> 6) + 3) & ~3;
    const int t0 = j * bl;
    float tot = 0.0f;
    for (int t = t0; t < min(t0 + bl, np); ++t) tot += row[t];
    return spec_sum_lanes(tot, (np + bl - 1) / bl, [&](float s0) {
        for (int t = t0; t < min(t0 + bl, np); ++t) s0 = s0 + row[t];
        return s0;
    });
}

// ------------------------------------------------------------------------------------------------
// k_attn_pf2 (round 3; head_dim 128, KVM = 2 or 4 query heads per kv head): dense-prefill attention with the wave-uniform
// operands in SCALAR registers and one wave per (position, head PAIR).  (Its first form, k_attn_pf -- q read from LDS, one wave
// per position, the K row held in 128 registers -- ran 240 us per (layer, block) where this one runs 126; removed.)
// One workgroup per kv head x NP consecutive positions; the K / V chunks (64 timesteps) are staged in LDS once for its
// NP x KVM query rows; wave w owns position w / (KVM/2) and the two query heads of pair w % (KVM/2) -- 16 waves for
// Qwen3-4B/8B (four per SIMD: the latency of one wave's scalar loads, barriers and softmax passes is another wave's
// issue slot; with one wave per position the kernel ran two per SIMD and waited half of the time).  Arithmetic and order per
// (head, position) are those of k_attn_gqa2 / k_attn / the reference (layers.rs:346-419,495-506): bit-identical results.
//   * q is normalised + rotated by k_knorm_rope (one launch for the block's K and q heads) and stored pair-interleaved; the
//     score loop reads it with s_load (constant address space) and forms the two heads' products with one v_pk_mul_f32
//     (K element broadcast by op_sel) and extends the two chains with one v_pk_add_f32: 2 VALU per dim and pair, no LDS
//     reads for q.  The K row is consumed in batches of 16 dims (no 128-register row copy);
//   * the row maxima accumulate in registers during the score loop; the softmax writes p = e * inv back to the score row
//     (the same two roundings, layers.rs:503-505) and pads it with +0.0 to a whole chunk; the V pass fetches p with
//     s_load_dwordx16 and needs 2 VALU per (timestep, head): v_pk_mul_f32 (v[e], v[e+64]) * p, v_pk_add_f32 -- no
//     v_readlane, no tail loop (o + 0.0 * v == o: o is never -0.0);
//   * both loops are software pipelines pinned with sched_barrier: [operands of this stage have landed][request the next
//     stage's][this stage's 32 packed operations]; scalar loads return out of order, so the only usable wait is lgkmcnt(0)
//     and a stage is one request deep.
// Scalar-cache coherence: q rows come from the previous launch, score rows from this wave's own vector stores ->
// s_dcache_inv at entry and again (behind s_waitcnt vmcnt(0)) before the first scalar read of the rows.
// ------------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef const float __attribute__((address_space(4))) kfloat;      // constant address space: uniform loads are s_load
constexpr int kWaitLgkm0 = 0xc07f;                                 // s_waitcnt lgkmcnt(0) only (vmcnt 63, expcnt 7)
constexpr int kPfTch = 128;                                        // timesteps per staged chunk: two per lane in the score loop (half the barriers of 64)
__host__ __device__ inline size_t attn_pf2_smem_bytes() { return 4 * (2 * (size_t)kPfTch * (kG2Hd + kKPad)) + 32 * 8; }
__host__ __device__ constexpr int attn_pf2_threads(int kvm, int np) { return 64 * np * (kvm / 2); }

template <int KVM, int NP>
__global__ __launch_bounds__(64 * NP * (KVM / 2), NP * (KVM / 2) / 4) void k_attn_pf2(const AttnArgs a0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int hd = kG2Hd, kld = hd + kKPad, TCH = kPfTch, TILE = TCH * kld;
    constexpr int NPRW = KVM / 2, NW = NP * NPRW, NTHR = 64 * NW, NSL = TCH * (hd / 4) / NTHR;
    constexpr int DS = 16, TS = 16;                                // dims per score stage, timesteps per V stage
    static_assert(NSL >= 1 && hd % DS == 0 && TCH % TS == 0, "");
    float* tiles = (float*)smem_raw;                               // 2 x [TCH][kld]   (V: [TCH][hd])
    unsigned long long* etab = (unsigned long long*)(tiles + 2 * TILE);
    const int kvh = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pw = wave / NPRW, pr = wave - pw * NPRW;             // position within the workgroup, head pair within the kv group
    const int pi0 = blockIdx.y * NP;
    const int n_pos = a0.n_pos;
    const int pi = min(pi0 + pw, n_pos - 1);
    const bool live = pi0 + pw < n_pos;                            // wave-uniform
    const size_t kvd = (size_t)a0.n_kv_heads * hd;
    const int ast = a0.att_stride;
    const int pos = __builtin_amdgcn_readfirstlane(a0.st[pi].pos);
    const int pos_last = __builtin_amdgcn_readfirstlane(a0.st[min(pi0 + NP - 1, n_pos - 1)].pos);
    const int np = pos + 1, np_max = pos_last + 1;                 // positions of a block are consecutive and ascending
    const float* kbase = a0.key_cache + (size_t)kvh * hd;
    const float* vbase = a0.value_cache + (size_t)kvh * hd;
    GQA_STAMP(0);
    __builtin_amdgcn_s_dcache_inv();
    if (tid < 32) etab[tid] = kExp2Tab[tid];

    v4f sr[NSL];
    auto issue = [&](const float* gbase, int t0) {
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            const int idx = tid + u * NTHR;
            const int row = min(t0 + (idx >> 5), np_max - 1), c = idx & 31;
            sr[u] = *(const v4f*)(gbase + (size_t)row * kvd + 4 * c);
        }
    };
    auto commit = [&](float* tile, int ld) {
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            const int idx = tid + u * NTHR;
            *(v4f*)(tile + (idx >> 5) * ld + 4 * (idx & 31)) = sr[u];
        }
    };
    issue(kbase, 0);
    const float scale = 1.0f / sqrtf((float)hd);                   // (head_dim as f32).sqrt().recip()
    const int head0 = kvh * KVM + 2 * pr;                          // this wave's heads: head0, head0 + 1
    float* rows = a0.att_global + ((size_t)pi * a0.n_heads + (size_t)head0) * ast;       // score rows: rows, rows + ast
    // the pair's query rows, interleaved element by element: [d][2]
    kfloat* qk = (kfloat*)(unsigned long long)(a0.q_out + ((size_t)pi * a0.n_heads + (size_t)head0) * hd);

    // ---- scores: att[h][t] = (q_h . K[t]) * scale, one timestep per lane, the head pair as one packed chain   layers.rs:391-401
    float mx0 = -__builtin_inff(), mx1 = -__builtin_inff();
    for (int c0 = 0, it = 0; c0 < np_max; c0 += TCH, ++it) {
        float* tile = tiles + (it & 1) * TILE;
        if (it == 8) GQA_STAMP(8);
        commit(tile, kld);
        __syncthreads();
        if (it == 8) GQA_STAMP(9);
        if (c0 + TCH < np_max) issue(kbase, c0 + TCH);
        if (live && c0 < np) {
            // two timesteps per lane (rows lane and lane + 64 of the chunk): two packed chains on the same scalar q operands
            const int t = c0 + lane, t1 = t + 64;
            const v4f* k4 = (const v4f*)(tile + lane * kld);
            const v4f* k4b = (const v4f*)(tile + (lane + 64) * kld);
            v2f d = {-0.0f, -0.0f}, e = {-0.0f, -0.0f};
            float qc[2 * DS], qn[2 * DS];
            v4f kc[DS / 4], kn[DS / 4], lc[DS / 4], ln[DS / 4];
            auto ldq = [&](float (&q)[2 * DS], int bb) {
#pragma unroll
                for (int i = 0; i < 2 * DS; ++i) q[i] = qk[2 * DS * bb + i];
            };
            ldq(qc, 0);
#pragma unroll
            for (int j = 0; j < DS / 4; ++j) { kc[j] = k4[j]; lc[j] = k4b[j]; }
#pragma unroll
            for (int b = 0; b < hd / DS; ++b) {
                __builtin_amdgcn_s_waitcnt(kWaitLgkm0);
                __builtin_amdgcn_sched_barrier(0);
                if (b + 1 < hd / DS) {
                    ldq(qn, b + 1);
#pragma unroll
                    for (int j = 0; j < DS / 4; ++j) { kn[j] = k4[(DS / 4) * (b + 1) + j]; ln[j] = k4b[(DS / 4) * (b + 1) + j]; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < DS / 4; ++j) {
                    // the (z, w) halves as opaque register pairs: picked out of the float4, hipcc copied every .w element into an even
                    // register before the multiply instead of broadcasting it by op_sel (64 v_mov per chunk on 512 packed operations)
                    v2f kzw = {kc[j].z, kc[j].w}, lzw = {lc[j].z, lc[j].w};
                    asm("" : "+v"(kzw));
                    asm("" : "+v"(lzw));
                    const float kk[4] = {kc[j].x, kc[j].y, kzw.x, kzw.y};
                    const float ll[4] = {lc[j].x, lc[j].y, lzw.x, lzw.y};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const v2f qq = {qc[8 * j + 2 * i], qc[8 * j + 2 * i + 1]};
                        // both products, then both adds: a packed multiply feeding the very next instruction costs an s_nop (81 per chunk)
                        v2f pd = qq * (v2f){kk[i], kk[i]}, pe = qq * (v2f){ll[i], ll[i]};
                        __builtin_amdgcn_sched_barrier(0);
                        d = d + pd;
                        e = e + pe;
                        __builtin_amdgcn_sched_barrier(0);         // (left alone the scheduler batches four multiplies, then four adds two apart)
                    }
                }
                asm volatile("" : "+v"(d), "+v"(e));               // pure arithmetic is not ordered by sched_barrier: pin the chains to their stage
                if (b + 1 < hd / DS) {
#pragma unroll
                    for (int i = 0; i < 2 * DS; ++i) qc[i] = qn[i];
#pragma unroll
                    for (int j = 0; j < DS / 4; ++j) { kc[j] = kn[j]; lc[j] = ln[j]; }
                }
            }
            if (it == 8) GQA_STAMP(10);
            if (t < np) {
                const float s0 = d.x * scale, s1 = d.y * scale;
                rows[t] = s0;
                rows[(size_t)ast + t] = s1;
                mx0 = fmaxf(mx0, s0);                              // running row maxima stay in registers (no pass over the row)
                mx1 = fmaxf(mx1, s1);
            }
            if (t1 < np) {
                const float s0 = e.x * scale, s1 = e.y * scale;
                rows[t1] = s0;
                rows[(size_t)ast + t1] = s1;
                mx0 = fmaxf(mx0, s0);
                mx1 = fmaxf(mx1, s1);
            }
        }
    }
    GQA_STAMP(1);
    __syncthreads();                                               // every wave is done with the K tiles
    issue(vbase, 0);                                               // V chunk 0 travels under the softmax

    // ---- softmax per (position, head) row (layers.rs:495-506): max, exp in place, exact sequential sum, p = e * inv in place;
    // the row is padded with +0.0 to a whole chunk.  Every step of a pass requests the next step's values before it works on
    // its own.
    if (live) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's score stores have landed (it reads them back)
        const int np64 = (np + TCH - 1) & ~(TCH - 1);
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            float* row = rows + (size_t)h * ast;
            const float m = group_max_f32(h ? mx1 : mx0, 64);
            {   // four exps per lane and step (their f64 chains interleave)
                float xc[4], xn[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xc[u] = row[min(64 * u + lane, np - 1)];
#pragma unroll 1
                for (int t0 = 0; t0 < np; t0 += 256) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) xn[u] = row[min(t0 + 256 + 64 * u + lane, np - 1)];
                    float ev[4], xx[4];
                    bool sp = false;
#pragma unroll
                    for (int u = 0; u < 4; ++u) { xx[u] = xc[u] - m; sp = sp || q3_expf_special(xx[u]); }
                    // (wave-uniform: glibc's main path alone unless some lane is outside |x| < 88 -- a sixth of an exp's instructions)
                    if (__builtin_expect(__any(sp), 0)) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) ev[u] = q3_expf_t(xx[u], etab);
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) ev[u] = q3_expf_main(xx[u], etab);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int t = t0 + 64 * u + lane;
                        if (t < np) row[t] = ev[u];
                        xc[u] = xn[u];
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float sum = row_exact_sum(row, np);
            const float inv = 1.0f / sum;
#pragma unroll 1
            for (int t0 = 0; t0 < np64; t0 += 512) {               // layers.rs:503-505; +0.0 past the context
                float e[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) e[u] = row[min(t0 + 64 * u + lane, np - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = t0 + 64 * u + lane;
                    if (t < np64) row[t] = (t < np) ? e[u] * inv : 0.0f;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the probabilities are in L2 ...
        __builtin_amdgcn_s_dcache_inv();                           // ... and no scalar-cache line of these rows predates them
    }

    GQA_STAMP(2);
    // ---- xb = sum_t att[t] * V[t]  (fill(0.0) then += in t order): elements (lane, lane + 64) of a head as one packed chain
    v2f o0 = {0.0f, 0.0f}, o1 = {0.0f, 0.0f};
    for (int c0 = 0, it = 0; c0 < np_max; c0 += TCH, ++it) {
        float* tile = tiles + (it & 1) * TILE;
        if (it == 8) GQA_STAMP(11);
        commit(tile, hd);
        __syncthreads();
        if (it == 8) GQA_STAMP(12);
        if (c0 + TCH < np_max) issue(vbase, c0 + TCH);
        if (live && c0 < np) {
            const float* v0 = tile + lane;
            const float* prow = rows + c0;
            v16f pc0, pc1, pn0, pn1;
            v2f vc[TS], vn[TS];
            asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(pc0) : "s"(prow) : "memory");
            asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(pc1) : "s"(prow + (size_t)ast) : "memory");
#pragma unroll
            for (int u = 0; u < TS; ++u) vc[u] = (v2f){v0[u * hd], v0[u * hd + 64]};
#pragma unroll
            for (int tb = 0; tb < TCH; tb += TS) {
                __builtin_amdgcn_s_waitcnt(kWaitLgkm0);            // this stage's probabilities (scalar) and V elements (LDS) have landed
                __builtin_amdgcn_sched_barrier(0);
                if (tb + TS < TCH) {
                    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(pn0) : "s"(prow + tb + TS) : "memory");
                    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(pn1) : "s"(prow + (size_t)ast + tb + TS) : "memory");
#pragma unroll
                    for (int u = 0; u < TS; ++u) vn[u] = (v2f){v0[(tb + TS + u) * hd], v0[(tb + TS + u) * hd + 64]};
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < TS; ++u) {
                    const float p0 = pc0[u], p1 = pc1[u];
                    o0 = o0 + vc[u] * (v2f){p0, p0};
                    o1 = o1 + vc[u] * (v2f){p1, p1};
                }
                asm volatile("" : "+v"(o0), "+v"(o1));
                if (tb + TS < TCH) {
                    pc0 = pn0; pc1 = pn1;
#pragma unroll
                    for (int u = 0; u < TS; ++u) vc[u] = vn[u];
                }
            }
            if (it == 8) GQA_STAMP(13);
        }
    }
    GQA_STAMP(3);
    if (live) {
        const float oa[4] = {o0.x, o0.y, 0.0f, 0.0f}, ob[4] = {o1.x, o1.y, 0.0f, 0.0f};
        gqa_store(a0, (size_t)pi, head0, hd, lane, oa);
        gqa_store(a0, (size_t)pi, head0 + 1, hd, lane, ob);
    }
    GQA_STAMP(4);
}

// ------------------------------------------------------------------------------------------------
// Batched prefill (T consecutive positions of ONE sequence per step, all sharing the engine's KV cache): the key
// rows of the whole block must be in the cache before any position's attention runs, so QK-norm + RoPE of K
// (layers.rs:346-372) gets its own small kernel.  grid (kv heads, positions), one wave each.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_knorm_rope(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[3 * 256];
    const int hd = a.hd, lane = threadIdx.x;
    const size_t sbi = blockIdx.y;
    const int pos = a.st[sbi].pos;
    float* raw = lds, *sq = lds + hd, *dst = lds + 2 * hd;
    RopeRegs rr;
    if ((int)blockIdx.x >= a.n_kv_heads) {
        // query head (k_attn_pf2 only): normalised + rotated q goes to q_out, the two heads of a pair interleaved element by
        // element ([position][pair][d][2]) so that the score loop fetches (q_h[d], q_h+1[d]) as one scalar pair
        const int h = (int)blockIdx.x - a.n_kv_heads;
        const float* qsrc = a.q + sbi * a.sb_q + (size_t)h * hd;
        rope_regs_load(rr, a.q_norm_w, a.rope + (size_t)pos * hd, hd);
        for (int i = lane; i < hd; i += 64) raw[i] = qsrc[i];
        wave_lds_sync();
        wave_norm_rope(dst, raw, sq, rr, hd, 1);
        wave_lds_sync();
        float* qrow = a.q_out + (sbi * a.n_heads + (size_t)(h & ~1)) * hd + (h & 1);
        for (int i = lane; i < hd; i += 64) qrow[2 * i] = dst[i];
        return;
    }
    const int kvh = blockIdx.x;
    const size_t kvd = (size_t)a.n_kv_heads * hd;
    const float* ksrc = a.k_raw + sbi * a.sb_kraw + (size_t)kvh * hd;
    rope_regs_load(rr, a.k_norm_w, a.rope + (size_t)pos * hd, hd);
    for (int i = lane; i < hd; i += 64) raw[i] = ksrc[i];
    wave_lds_sync();
    wave_norm_rope(dst, raw, sq, rr, hd, 1);
    wave_lds_sync();
    float* krow = a.key_cache + (size_t)pos * kvd + (size_t)kvh * hd;
    for (int i = lane; i < hd; i += 64) krow[i] = dst[i];
}

// The same for a long block (dense prefill): k_knorm_rope spends one wave on every (head, position) vector and all 64 lanes walk the
// same 128-add chain -- 82k waves per layer at 2,048 positions of the 4B shape, ~60 us.  Here a 256-thread workgroup takes 64
// vectors: staged in LDS (row stride 132 floats: the float4 reads of 16 chain lanes cover all 64 banks), lane l of wave 0 folds the
// squares of vector l in index order (layers.rs:113: the same multiply, the same 128 adds from -0.0), then all threads normalise and
// rotate element pairs with the expressions of wave_norm_rope (layers.rs:117, 181-182).  head_dim 128 only; vectors are numbered
// position-major: v = position * (kv heads + query heads) + head slot, kv heads first.
constexpr int kKnbVec = 64, kKnbLd = 132;
__host__ __device__ inline size_t knorm_blk_smem_bytes() { return 4 * ((size_t)kKnbVec * kKnbLd + kKnbVec); }
__global__ __launch_bounds__(256) void k_knorm_rope_blk(const AttnArgs a, int n_pos, int with_q) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int hd = kG2Hd, half = hd / 2;
    float* raw = (float*)smem_raw;                                 // [64][132]
    float* fv = raw + kKnbVec * kKnbLd;                            // [64] RMSNorm factors
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hs_n = a.n_kv_heads + (with_q ? a.n_heads : 0);
    const long nvec = (long)n_pos * hs_n;
    const long v0 = (long)blockIdx.x * kKnbVec;
    auto vec_src = [&](long v, int& p, int& hs) -> const float* {
        p = (int)(v / hs_n);
        hs = (int)(v - (long)p * hs_n);
        return hs < a.n_kv_heads ? a.k_raw + (size_t)p * a.sb_kraw + (size_t)hs * hd
                                 : a.q + (size_t)p * a.sb_q + (size_t)(hs - a.n_kv_heads) * hd;
    };
    // ---- stage: 8 vectors per pass, 32 threads x float4 per vector
    const int sv = tid >> 5, c4 = tid & 31;
    v4f st[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long v = min(v0 + 8 * k + sv, nvec - 1);
        int p, hs;
        const float* src = vec_src(v, p, hs);
        st[k] = ((const v4f*)src)[c4];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) *(v4f*)(raw + (size_t)(8 * k + sv) * kKnbLd + 4 * c4) = st[k];
    __syncthreads();
    // ---- sum of squares, one chain per vector (wave 0)
    if (wave == 0) {
        const v4f* r4 = (const v4f*)(raw + (size_t)lane * kKnbLd);
        float ss = -0.0f;
        v4f cur[8], nxt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = r4[u];
#pragma unroll
        for (int b = 0; b < hd / 32; ++b) {
            if (b + 1 < hd / 32) {
#pragma unroll
                for (int u = 0; u < 8; ++u) nxt[u] = r4[8 * (b + 1) + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const v4f x = cur[u];
                float q = x.x * x.x; ss = ss + q;                  // layers.rs:113
                q = x.y * x.y; ss = ss + q;
                q = x.z * x.z; ss = ss + q;
                q = x.w * x.w; ss = ss + q;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
        }
        fv[lane] = 1.0f / sqrtf(ss / (float)hd + kEps);
    }
    __syncthreads();
    // ---- normalise + rotate: 32 threads per vector, pairs (i, i + 64) for i = c4 and c4 + 32
#pragma unroll 2
    for (int k = 0; k < 8; ++k) {
        const int vl = 8 * k + sv;
        const long v = v0 + vl;
        if (v >= nvec) continue;
        int p, hs;
        (void)vec_src(v, p, hs);
        const bool is_q = hs >= a.n_kv_heads;
        const float* w = is_q ? a.q_norm_w : a.k_norm_w;
        const int pos = a.st[p].pos;
        const float* cs = a.rope + (size_t)pos * hd;
        const float f = fv[vl];
        const float* src = raw + (size_t)vl * kKnbLd;
        float* dst_lo;
        int dstep;
        if (is_q) {
            const int h = hs - a.n_kv_heads;
            dst_lo = a.q_out + ((size_t)p * a.n_heads + (size_t)(h & ~1)) * hd + (h & 1);     // pair-interleaved: [d][2]
            dstep = 2;
        } else {
            dst_lo = a.key_cache + (size_t)pos * ((size_t)a.n_kv_heads * hd) + (size_t)hs * hd;
            dstep = 1;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = c4 + 32 * u;
            const float xv = w[i] * (f * src[i]);
            const float yv = w[i + half] * (f * src[i + half]);
            const float c = cs[2 * i], sn = cs[2 * i + 1];
            const float a0 = xv * c, b0 = yv * sn;
            const float a1 = xv * sn, b1 = yv * c;
            dst_lo[(size_t)dstep * i] = a0 - b0;                   // layers.rs:181-182
            dst_lo[(size_t)dstep * (i + half)] = a1 + b1;
        }
    }
}

// states of one prefill block: position first_pos + i takes prompt token base + i
__global__ void k_set_prefill_states(State* st, const int32_t* prompt, int base, int first_pos, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        st[i].token = prompt[base + i];
        st[i].pos = first_pos + base + i;
        st[i].step = 0;
        st[i].prompt_len = 0;
        st[i].argmax = 0ull;
    }
}

// per-stream k_next: grid = streams
__global__ __launch_bounds__(kWG) void k_next_batch(State* st, const unsigned long long* slots, int slot_stride, int nslots,
                                                    int32_t* out_tokens, int out_cap) {
    __shared__ unsigned long long red[kWaves];
    const int sb = blockIdx.x;
    st += sb;
    slots += (size_t)sb * slot_stride;
    out_tokens += (size_t)sb * out_cap;
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
        if (step < out_cap) out_tokens[step] = idx;
        st->token = idx;                                   // generation.rs:143-147
        st->pos = st->pos + 1;
        st->step = step + 1;
        st->argmax = best;
    }
}

}  // namespace q3
