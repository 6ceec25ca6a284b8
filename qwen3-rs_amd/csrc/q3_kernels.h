// q3_kernels.h -- gfx950 (MI355X) device code for the Qwen3 Q8 decode hot path, part 2: attention kernels and the operator-level
// kernels.  Part 1 (shared helpers, the W8A8 GEMV kernel) is q3_gemv.h.
#pragma once

#include "q3_gemv.h"

namespace q3 {

#ifdef Q3_DEV
// one workgroup per launch: min of the begin slots, max of the end slots -> cells[2i], cells[2i+1]; slots re-armed
__global__ __launch_bounds__(256) void k_kstamp_reduce(unsigned long long* slots, const int* nslots, unsigned long long* cells) {
    __shared__ unsigned long long smin[256], smax[256];
    unsigned long long* s = slots + 2 * (size_t)kKstampSlots * blockIdx.x;
    const int n = nslots[blockIdx.x];
    unsigned long long mn = ~0ull, mx = 0ull;
    for (int i = threadIdx.x; i < n; i += 256) {
        const unsigned long long b = s[2 * i], e = s[2 * i + 1];
        if (b != 0ull && b < mn) mn = b;
        if (e > mx) mx = e;
        s[2 * i] = 0ull;
        s[2 * i + 1] = 0ull;
    }
    smin[threadIdx.x] = mn; smax[threadIdx.x] = mx;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            if (smin[threadIdx.x + w] < smin[threadIdx.x]) smin[threadIdx.x] = smin[threadIdx.x + w];
            if (smax[threadIdx.x + w] > smax[threadIdx.x]) smax[threadIdx.x] = smax[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { cells[2 * blockIdx.x] = smin[0]; cells[2 * blockIdx.x + 1] = smax[0]; }
}
__global__ __launch_bounds__(256) void k_kstamp_fold(const unsigned long long* cells, int n, unsigned long long* acc) {
    // acc: [0] tokens folded, [1] end of the previous token's last launch, then per launch {duration sum, gap sum} in 10 ns ticks.
    // One thread per launch (n <= 256): its own {begin, end} and its predecessor's end.
    const int i = threadIdx.x;
    const unsigned long long last_prev = acc[1];
    unsigned long long e_last = 0ull;
    if (i < n) {
        const unsigned long long b = cells[2 * i], e = cells[2 * i + 1];
        const unsigned long long pe = i > 0 ? cells[2 * i - 1] : last_prev;
        if (b != ~0ull && e >= b) {
            acc[2 + 2 * i] += e - b;
            if (pe != 0ull && b >= pe) acc[3 + 2 * i] += b - pe;
        }
        if (i == n - 1) e_last = e;
    }
    __syncthreads();
    if (i == 0) acc[0] += 1;
    if (i == n - 1) acc[1] = e_last;
}
#endif

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
    int row_steps;            // k_attn_short2: > 0 = 8-row steps of K / V to request at kernel entry (the graph of a position range < 64)
    const float* value_t;     // k_attn_out: transposed value cache of this layer, [kv_dim][seq_len] (nullptr: stage from value_cache)
    // k_attn_short2 / k_attn_out: n_heads / n_kv_heads and ceil(2^20 / n_kv_heads), set by attn_set_heads -- the workgroup -> head mapping
    // without two integer divisions (~55 scalar instructions in front of the first address that needs the head)
    int kv_mul;
    unsigned kvh_magic;
};
inline void attn_set_heads(AttnArgs& a, int n_heads, int n_kv_heads) {
    a.n_heads = n_heads;
    a.n_kv_heads = n_kv_heads;
    a.kv_mul = n_heads / n_kv_heads;
    a.kvh_magic = (1048576u + (unsigned)n_kv_heads - 1u) / (unsigned)n_kv_heads;     // exact quotient for block index * n_kv_heads < 2^20
}

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
#define ATT_STAMP(i) do { if (a.stamps != nullptr && (a.debug & 64) == 0 && blockIdx.x == 3 && blockIdx.y == 0 && threadIdx.x == 0) a.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
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
#define ATTS_STAMP(i, thr) do { if (a.stamps != nullptr && (a.debug & 64) == 0 && blockIdx.x == 3 && (int)threadIdx.x == (thr)) a.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
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
    // The position is requested first; the raw q / k values do not depend on it and go out BEHIND it, before it is waited for
    // (round 5: a readfirstlane right here put the position's whole round trip in front of every other load of the launch; the raw
    // k row is read by every chunk's workgroup now -- one cache line -- and used by the one that holds the position).
    const int pos_v = a.pos_override >= 0 ? a.pos_override : a.st->pos;
    float rv[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int idx = tid + u * kWG;                 // (KVM_T + 1) * 128 <= 640 values: raw q of each head, then raw k
        if (idx < KVM_T * hd) rv[u] = a.q[(size_t)kvh * KVM_T * hd + idx];
        else if (idx < (KVM_T + 1) * hd) rv[u] = a.k_raw[(size_t)kvh * hd + idx - KVM_T * hd];
    }
    float rv2 = 0.f;
    if (KVM_T == 4 && tid < hd) rv2 = a.k_raw[(size_t)kvh * hd + tid];
    __builtin_amdgcn_sched_barrier(0);
    const int pos = __builtin_amdgcn_readfirstlane(pos_v);       // the oldest load: a counted wait
    const int np = pos + 1;
    const int t0 = c * TCH;
    if (t0 >= np) return;
    const int cnt = min(TCH, np - t0);
    const bool has_pos = pos < t0 + cnt;
    const float* cs = a.rope + (size_t)pos * hd;
    // the wave that normalises the new K row: the first one without a query head, or wave 0 after its own head
    constexpr int kwave = KVM_T < kWaves ? KVM_T : 0;

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
// The V chain's inner step, written out (round 5): request the 8 float4 at LDS byte address `addr` into n[], wait until everything but
// those 8 reads has landed (the set c[], requested a phase ago), add c's 32 floats to acc in order.  hipcc's own version of this loop
// reused the last register of a set in flight as the chain's temporary (kernel at its 128-VGPR limit) and paid for it with an
// `s_waitcnt lgkmcnt(0)` -- a whole LDS round trip -- every 64 timesteps: 9-10 cycles per timestep instead of ~6.
__device__ __forceinline__ void chain_request8(v4f (&n)[8], unsigned addr) {
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                 "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\tds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112\n\t"
                 "s_waitcnt lgkmcnt(8)"
                 : "=&v"(n[0]), "=&v"(n[1]), "=&v"(n[2]), "=&v"(n[3]), "=&v"(n[4]), "=&v"(n[5]), "=&v"(n[6]), "=&v"(n[7])
                 : "v"(addr)
                 : "memory");
}
__device__ __forceinline__ void chain_add16(float& acc, v4f c0, v4f c1, v4f c2, v4f c3) {
    asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %0, %0, %4\n\t"
                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %0, %0, %6\n\tv_add_f32 %0, %0, %7\n\tv_add_f32 %0, %0, %8\n\t"
                 "v_add_f32 %0, %0, %9\n\tv_add_f32 %0, %0, %10\n\tv_add_f32 %0, %0, %11\n\tv_add_f32 %0, %0, %12\n\t"
                 "v_add_f32 %0, %0, %13\n\tv_add_f32 %0, %0, %14\n\tv_add_f32 %0, %0, %15\n\tv_add_f32 %0, %0, %16"
                 : "+v"(acc)
                 : "v"(c0.x), "v"(c0.y), "v"(c0.z), "v"(c0.w), "v"(c1.x), "v"(c1.y), "v"(c1.z), "v"(c1.w),
                   "v"(c2.x), "v"(c2.y), "v"(c2.z), "v"(c2.w), "v"(c3.x), "v"(c3.y), "v"(c3.z), "v"(c3.w));
}

// exact sum over blocks of 4 * nq terms, nq = 1 ... 16: every length as straight-line register code
__device__ __forceinline__ float seq_sum_blocks_nq16(const float* t, int nblk, int nq, int stride) {
    switch (nq) {
#define Q3_NQ_CASE(N) case N: return seq_sum_blocks_regs<N>(t, nblk, stride, nullptr);
        Q3_NQ_CASE(1) Q3_NQ_CASE(2) Q3_NQ_CASE(3) Q3_NQ_CASE(4) Q3_NQ_CASE(5) Q3_NQ_CASE(6) Q3_NQ_CASE(7) Q3_NQ_CASE(8)
        Q3_NQ_CASE(9) Q3_NQ_CASE(10) Q3_NQ_CASE(11) Q3_NQ_CASE(12) Q3_NQ_CASE(13) Q3_NQ_CASE(14) Q3_NQ_CASE(15) Q3_NQ_CASE(16)
#undef Q3_NQ_CASE
    }
    return seq_sum_blocks(t, nblk, 4 * nq, stride, nullptr);
}

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

    // workgroup -> head: every slice of every query head of one kv head on the same XCD (block indices equal modulo n_kv_heads, see
    // k_attn_short2): the kv head's value rows enter ONE L2 instead of up to eight
    const int sl = blockIdx.y, nsl = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kv_mul = a.kv_mul, xq = (int)((blockIdx.x * a.kvh_magic) >> 20), kvh = (int)blockIdx.x - xq * a.n_kv_heads;
    const int h = kvh * kv_mul + xq;
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
    // reference order with a transposed value cache: a staging thread owns 4 consecutive timesteps of one element (float4 q of element e:
    // K / 4 float4 per element and chunk), the chunk of an element is one contiguous run of K floats
    const bool vtr = a.value_t != nullptr && a.strict != 0;              // wave-uniform
    const float* vtbase = (vtr ? a.value_t : a.value_cache) + ((size_t)kvh * hd + (size_t)sl * w) * (size_t)a.seq_len;   // (a valid address either way)
    constexpr int KQ = kVChunk / 4;
    const int npass_t = (w * KQ + nst - 1) / nst;                        // 1 / 2 / 2 for slice widths 8 / 16 / 32
    auto v_issue = [&](VRegs& R, int c0, bool every_wave = false) {
        if (!every_wave && !stager) return;
        // ONE load per slot whatever the layout -- the address is selected, not the code path: loads behind a (wave-uniform) branch
        // made hipcc close the block in front of them with vmcnt(0), i.e. wait for the score row before the first value request
        // ("loads issued" 1,850 -> 3,550 cycles after entry when the transposed form first sat behind its own `if`)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u < (vtr ? npass_t : npass)) {           // (compile-time for the instantiated slice widths: both counts agree)
                // row-major: clamped to the cache, not to the context: rows past the current position are allocated memory whose
                // contents are masked at commit time, and the address then does not wait for the position
                const int t = min(c0 + r0 + u * rps, a.seq_len - 1);
                const float* prm = vbase + (size_t)t * kvd + 4 * c4;
                const int idx = min(sid + u * nst, w * KQ - 1);
                const int e = idx / KQ, tq = idx % KQ;
                const int tt = min(c0 + 4 * tq, a.seq_len - 4);      // (seq_len % 4 == 0: host)
                const float* ptr = vtbase + (size_t)e * a.seq_len + tt;
                // (developer ablation 1024, timing only: the slice's rows read as if they were stored back to back)
                const float* pab = a.value_cache + ((size_t)((blockIdx.x & 7) * gridDim.y + blockIdx.y) * a.seq_len + t) * w + 4 * c4;
                // (developer ablation 8192, timing only: every request of the launch reads the same 16 bytes)
                R.v[u] = *(const v4f*)(Q3_DEV_ABLATE(a, 8192) ? a.value_cache : (Q3_DEV_ABLATE(a, 1024) ? pab : (vtr ? ptr : prm)));
            }
        }
    };
    constexpr int K = kVChunk;

    // exp2 table of q3_expf staged in LDS (a dependent global load per exp otherwise)
    // (one copy per wave: with the block maxima below no barrier separates the staging of the table from its use)
    unsigned long long* etab = (unsigned long long*)(p_lds + (p_in_lds ? npad_max : 0)) + 32 * wave;
    // (requested here, written to LDS only behind the score / value requests below: the LDS store needs the loaded value, and placed
    // here it put the table's whole round trip -- a cold constant-memory line -- in front of every other load of the launch)
    unsigned long long etv = 0ull;
    if (lane < 32) etv = kExp2Tab[lane];

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
    if (lane < 32) etab[lane] = etv;                     // (the oldest load of the wave)
    ATT_STAMP(1);
    const int pos = __builtin_amdgcn_readfirstlane(a.pos_override >= 0 ? a.pos_override : pos_mem);     // wave-uniform -> SGPR
    const int np = pos + 1;
    const int npad = (np + 255) & ~255;                  // whole 64 x (npad/64) blocks for the exact sum
    // exact-sum blocks: 64 lanes x the fewest whole float4 that cover the row (np <= 4096: 4 ... 64 terms; the 36 terms of position 2,300
    // fold in 165 cycles per round where the power-of-two block of 64 took 295), stride padded to an odd number of float4 (b128 reads of
    // 16 consecutive lanes on 16 different bank groups).  t / bl as a multiplication: exact for t < 4,160 and bl <= 64 (t * bl < 2^20)
    const bool one_trip = np <= kAoSv * kAoThreads;
    const int blq = (np + 255) >> 8;
    const int bl = 4 * blq, estride = 4 * (blq + 1 + (blq & 1));
    const unsigned bmag = one_trip ? (1048576u + (unsigned)bl - 1u) / (unsigned)bl : 0u;
    const bool padded = a.strict && one_trip;
    float* esc = vbuf0;                                  // (np/bl) x (bl + 4) floats <= kEscFloats: the V-tile region reserves that much
    float m = -__builtin_inff();
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
    const int nblk = (int)(((unsigned)(np + bl - 1) * bmag) >> 20), nblk_terms = nblk * bl;     // (one_trip rows)
    if (one_trip) {
        // Only the exps this WAVE has live slots for (wave-uniform count): the phase is bound by the SIMD's instruction rate -- four waves
        // per SIMD, ~75 instructions per exp -- and at 2,300 positions the unconditional four per thread computed 4,096 exps for 2,304
        // slots (timeline of the 4B shape at position 2,300: 4,750 cycles from the maximum to the last exp written).  Inside a count the
        // exps stay one basic block (their f64 chains interleave; a guard around each one would serialise them).
        static_assert(kAoSv == 4, "one group of four slots per thread");
        const int nu = min(4, max(0, (npad - wave * 64 + kAoThreads - 1) / kAoThreads));
        auto esc_at = [&](int t) { const int b = (int)(((unsigned)t * bmag) >> 20); return b * estride + (t - b * bl); };
        auto exps = [&](auto NU) {
            constexpr int N = decltype(NU)::value;
            float xv[N], ev[N];
            bool sp = false;
#pragma unroll
            for (int u = 0; u < N; ++u) {
                xv[u] = u * kAoThreads + tid < np ? sv[u] - m : 0.0f;
                sp = sp || q3_expf_special(xv[u]);
            }
            // (wave-uniform: the selects of the special cases -- a sixth of an exp's instructions -- only where some lane needs them)
            if (__builtin_expect(__any(sp), 0)) {
#pragma unroll
                for (int u = 0; u < N; ++u) ev[u] = q3_expf_t(xv[u], etab);
            } else {
#pragma unroll
                for (int u = 0; u < N; ++u) ev[u] = q3_expf_main(xv[u], etab);
            }
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const int t = u * kAoThreads + tid;
                ev[u] = t < np ? ev[u] : 0.0f;           // +0.0 padding leaves every partial sum unchanged
                part = part + ev[u];
                if constexpr (P_LDS) {
                    // masked-off stores go to a dummy word instead of sitting behind a branch: LLVM sinks the whole exp
                    // into a guarded block otherwise and the chains no longer interleave
                    *(t < npad ? p + t : dummy) = ev[u];
                    *((padded && t < nblk_terms) ? esc + esc_at(t) : dummy) = ev[u];
                } else if (t < npad) {
                    p[t] = ev[u];
                    if (padded && t < nblk_terms) esc[esc_at(t)] = ev[u];
                }
            }
        };
        if (nu == 4) exps(IntC<4>{});
        else if (nu == 3) exps(IntC<3>{});
        else if (nu == 2) exps(IntC<2>{});
        else if (nu == 1) exps(IntC<1>{});
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
                    p[t] = ev[u];                        // (rows beyond 4096 positions: the exact sum reads p itself)
                }
            }
        }
    }
    __syncthreads();
    ATT_STAMP(4);
    float sum;
    if (a.strict) {
        if (wave == 0) {                                 // one wave walks the blocks; the other 15 would only fight it for LDS
            if (padded) sum = seq_sum_blocks_nq16(esc, nblk, blq, estride);
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
        if (!stager || Q3_DEV_ABLATE(a, 512)) return;
        float* vbuf = vbuf0 + buf * (kVChunk + kVPad) * w;
        float* pbuf = pbuf0 + buf * kVChunk;
        if (vtr) {
            // products p[t] * v[t][e] of four consecutive timesteps: one b128 read of the probabilities, one b128 write of the tile
            // row (the row-major cache needs four scattered b32 writes per float4)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = sid + u * nst;
                if (u < npass_t && idx < w * KQ) {
                    const int e = idx / KQ, tq = idx % KQ;
                    const int t = c0 + 4 * tq;                           // < npad: chunks start below np and npad is a multiple of 256
                    const v4f pe = *(const v4f*)(p + t);
                    const v4f vv = R.v[u];
                    v4f x;
                    x.x = (pe.x * inv) * vv.x; x.y = (pe.y * inv) * vv.y; x.z = (pe.z * inv) * vv.z; x.w = (pe.w * inv) * vv.w;
                    x.x = t + 0 < np ? x.x : 0.0f; x.y = t + 1 < np ? x.y : 0.0f; x.z = t + 2 < np ? x.z : 0.0f; x.w = t + 3 < np ? x.w : 0.0f;
                    *(v4f*)(vbuf + e * (kVChunk + kVPad) + 4 * tq) = x;
                }
            }
            return;
        }
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
        // staging waves' per-chunk barrier 32 timesteps before the end of its chunk -- every read of the current tile has
        // been issued by then -- and uses the last 32 adds to bring in the head of the next tile.
        // (Lanes >= w run along on the last row and are dropped at the store: the barrier stays outside divergent code.)
        const int row = min(tid, w - 1);
        // (round 5: s_setprio 3 for this wave -- it shares its SIMD with three staging waves -- measured, no change: 905.8 vs 905.4 us
        // of attention per token at position 2,300 on the 4B dims; the staging waves do not take the chain's issue slots)
        // Operands run 32 timesteps ahead of the adds (two sets of 8 float4): a dependent v_add_f32 issues every ~4.9 cycles
        // (tools/mfma_chain_probe.hip; the "10 cycles" of rounds 2-3 included the timer's own latency over a 64-step loop), so the
        // 16 adds that used to cover an LDS read were 78 cycles against a ~130-cycle round trip, and the chain ran at 11.3 per step.
        // EVERY LDS read of the two sets goes through chain_request8: one read of av left to hipcc (the hand-over to the next tile was
        // `av[u] = vn[u]`) put those registers on its scoreboard at the loop header, and it then guarded the written-out adds with
        // lgkmcnt(4) / lgkmcnt(0) -- counters that include the burst just issued, i.e. the LDS round trip every 64 timesteps again
        // (timeline: 2,460 cycles per 256-timestep chunk = 9.6 per timestep against ~6 for the chain by itself).
        v4f av[8], bv[8];
        chain_request8(av, (unsigned)(size_t)(vbuf0 + row * VLD));
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
                const unsigned la = (unsigned)(size_t)(vr + q + 8);          // LDS byte address of the next 8 float4
                if (!Q3_DEV_ABLATE(a, 2048)) chain_request8(bv, la);
                if (!Q3_DEV_ABLATE(a, 4096)) { chain_add16(o_s, av[0], av[1], av[2], av[3]); chain_add16(o_s, av[4], av[5], av[6], av[7]); }
                if (!Q3_DEV_ABLATE(a, 2048)) chain_request8(av, la + 128u);
                if (!Q3_DEV_ABLATE(a, 4096)) { chain_add16(o_s, bv[0], bv[1], bv[2], bv[3]); chain_add16(o_s, bv[4], bv[5], bv[6], bv[7]); }
            }
            {
                // the chunk's last 64 timesteps meet the staging waves' barrier.  (Round 5 tried LDS counters instead -- arrival count of
                // the next tile read behind the last burst, "tile consumed" stored behind it: bit-identical in most runs, 906 -> 902 us
                // of attention per token at position 2,300, but 4 runs in 10 of the 33-chunk operator test returned a wrong 8-element
                // slice, with or without a compiler barrier around the store.  The barrier stays; what the chain really waits for is
                // the staging waves' value rows, see docs/HISTORY.md section 0.)
                chain_request8(bv, (unsigned)(size_t)(vr + q + 8));           // (its wait also covers av, requested by the loop above)
                chain_add16(o_s, av[0], av[1], av[2], av[3]);
                chain_add16(o_s, av[4], av[5], av[6], av[7]);
                // (developer timeline: the chain wave in front of / behind the barriers of chunks 2..5)
                if (c0 >= 2 * K && c0 < 6 * K) ATT_STAMP(7 + 2 * (c0 / K - 2));
                if (!Q3_DEV_ABLATE(a, 256)) __syncthreads();     // the staging waves' barrier of this chunk: tile c+1 is complete
                if (c0 >= 2 * K && c0 < 6 * K) ATT_STAMP(8 + 2 * (c0 / K - 2));
                chain_request8(av, (unsigned)(size_t)vn);        // head of the next tile (past the last chunk: a valid address, unused);
                                                                 // its wait = bv has landed
                chain_add16(o_s, bv[0], bv[1], bv[2], bv[3]);
                chain_add16(o_s, bv[4], bv[5], bv[6], bv[7]);
            }
        }
        // the last request is still in flight and hipcc does not know: its registers must not be handed out before it has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else
    for (int c0 = 0; c0 < np; c0 += 2 * K) {
        if (c0 + K < np) {
            v_commit(vrb, c0 + K, 1);
            if (c0 + 3 * K < np) v_issue(vrb, c0 + 3 * K);
        }
        fold(c0, 0);
        if (!Q3_DEV_ABLATE(a, 256)) __syncthreads();
        if (c0 + K >= np) break;
        if (c0 + 2 * K < np) {
            v_commit(vra, c0 + 2 * K, 0);
            if (c0 + 4 * K < np) v_issue(vra, c0 + 4 * K);
        }
        fold(c0 + K, 1);
        if (!Q3_DEV_ABLATE(a, 256)) __syncthreads();
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
// value rows [pos0, pos0 + n) of every layer, row-major [layer][seq_len][kv_dim] -> transposed [layer][kv_dim][seq_len]
// (behind a batched prefill, whose matmul epilogues write the row-major cache only; grid (ceil(kv_dim / 64), ceil(n / 64), layers))
__global__ __launch_bounds__(kWG) void k_value_transpose(const float* __restrict__ v, float* __restrict__ vt, int seq_len, int kvd, int pos0, int n) {
    __shared__ float tile[64][65];
    const size_t lbase = (size_t)blockIdx.z * seq_len * kvd;
    const int e0 = blockIdx.x * 64, t0 = pos0 + blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += kWG) {
        const int r = i >> 6, c = i & 63;                    // row = timestep, column = element
        tile[r][c] = (t0 + r < pos0 + n && e0 + c < kvd) ? v[lbase + (size_t)(t0 + r) * kvd + e0 + c] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += kWG) {
        const int r = i >> 6, c = i & 63;                    // row = element, column = timestep
        if (t0 + c < pos0 + n && e0 + r < kvd) vt[lbase + (size_t)(e0 + r) * seq_len + t0 + c] = tile[c][r];
    }
}

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


}  // namespace q3
