// Device-side Sampler::sample (qwen3-inference/src/sampler.rs:118-139): temperature, softmax, one xorshift64* coin,
// multinomial or top-p -- so that non-greedy decoding does not ship 607 KB of logits to the host per token.
//
// One workgroup of 1024 threads per token.  Everything order-sensitive in the reference is a running f32 sum walked in
// index order (softmax denominator layers.rs:495-506, cdf sampler.rs:62-71, nucleus cumulative + cdf sampler.rs:90-110).
// wg_exact_prefix() reproduces those sums exactly: lane j owns block j of the (zero padded) sequence, folds it from a
// GUESSED input, the guesses are corrected with a scan of the link mismatches and every link is verified bitwise
// (the scheme of seq_sum_blocks in q3_kernels.h, over 1024 lanes); afterwards every lane knows the exact running sum
// before and after its block, and "first index whose running sum crosses x" is a ballot plus one short in-block walk.
//
// Top-p sorts the candidates by (probability descending, index ascending).  The reference uses sort_unstable_by, which
// leaves the order of EQUAL probabilities unspecified; the CPU restatement used by the tests makes the same choice.
#pragma once
#include "q3_kernels.h"

namespace q3 {

constexpr int kSampThreads = 1024;
#ifndef Q3_SAMP_RADIX_MIN
#define Q3_SAMP_RADIX_MIN 2048
#endif
constexpr int kSampRadixMin = Q3_SAMP_RADIX_MIN;   // candidate lists longer than this are radix sorted (bitonic network below that)

struct SamplerState {
    unsigned long long rng;   // sampler.rs:19 rng_state
    float temperature;
    float topp;
    int rounds[4];            // diagnostics: correction rounds of the last draw's exact prefix passes (softmax sum, cdf / nucleus)
    // hand-over between the kernels of the pipelined draw (phase 1 -> chip-wide passes -> phase 2)
    float inv;                // 1 / (exact softmax denominator)
    float coin;               // this draw's uniform
    int discard;              // a prompt-position draw: nothing to do
    int n0;                   // candidates the chip-wide compaction wrote to keys[]
    unsigned tkey0;           // probability-key threshold of the first attempt
};
constexpr int kSampGrid = 256;                         // workgroups of the chip-wide passes
constexpr int kSampHistBins = 2048;

struct SampleArgs {
    const float* logits;      // [n]
    int n;                    // vocab size
    int blen;                 // terms per lane block: 4 * ceil(n / 4096); 1024 * blen >= n
    float* probs;             // [1024 * blen] scratch: e, then p (zero padded)
    unsigned long long* keys; // [2 * n2] sort scratch, n2 = next power of two >= n: bitonic network in place / radix ping-pong
    long long keys2_off;      // n2: element offset of the second half
    float* sp;                // [1024 * blen] sorted candidate probabilities (zero padded)
    SamplerState* ss;
    State* st;
    int32_t* out_tokens;
    int out_cap;
    // batched decode: workgroup b of the launch samples stream b; element strides between streams (0 for one stream)
    long long sb_logits, sb_scratch, sb_keys;
    // the engine's draws: e = exp(logit / T - max) was already written to probs[] by k_sample_exp (all CUs) -- the pass is
    // ~152k IEEE divisions + f64-pipe exps, ~0.2 ms when a single workgroup walks it
    int pre_exp;
    unsigned long long* stamps;   // developer timeline (dev build, Q3_STAMPS=1)
    // pipelined draw (single-stream engine): 0 = k_sample does everything; 1 = front (coin, exact sum -> SamplerState::inv);
    // 2 = tail (sort, exact walks, result) behind k_sample_norm_hist / k_sample_count / k_sample_scatter
    int phase;
    float* hist;              // [kSampHistBins] probability mass per key bin (window below the maximum only)
    int* counts;              // [kSampGrid] candidates per workgroup range
};

__device__ __forceinline__ float key_to_float(unsigned k) {
    const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;    // inverse of total_order_key
    return __uint_as_float(b);
}

// inclusive scan of one float per thread over the whole workgroup (any association: guesses / corrections only): DPP scan inside the
// waves, ONE barrier, every wave scans the 16 wave totals itself.  wtot: 16 floats of LDS nobody else touches until the next barrier.
__device__ __forceinline__ float wg_scan_incl(float v, float* wtot) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    v = wave_scan_incl(v);
    if (lane == 63) wtot[wave] = v;
    __syncthreads();
    float t = lane < kSampThreads / 64 ? wtot[lane] : 0.0f;
    t = wave_scan_incl(t);
    const float off = wave > 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), wave - 1)) : 0.0f;
    return v + off;
}

// Exact running sums of 1024 x 16 non-negative terms folded in index order from `init`, lane j holding terms [16 j, 16 j + 16) in r[]: it
// returns in_j = sum before its block and out_j = sum after it, both bit-exact.  xch: >= 80 floats of LDS.  Every thread of the 1024-thread workgroup must call it.
// Round 5 re-cut (the exact denominator over the 151,936-entry vocabulary took ~65 us = 10 segments x ~7 rounds x ~2,300 cycles): a round is
// the 16-add fold out of REGISTERS, the neighbour's output by DPP (LDS only across the 16 wave boundaries), the verify folded into the
// correction scan (a mismatch flag travels with it), and two barriers instead of seven; buffers alternate by round parity.
__device__ __forceinline__ void wg_exact_prefix(const v4f (&r)[4], float init, float* xch, float& in_j, float& out_j, int* rounds_out = nullptr) {
    const int j = threadIdx.x, lane = j & 63, wave = __builtin_amdgcn_readfirstlane(j >> 6);
    constexpr int NW = kSampThreads / 64;
    float* wtot = xch;                        // [2][NW]
    float* bnd = xch + 2 * NW;                // [2][NW] last output of every wave
    int* flag = (int*)(xch + 4 * NW);         // [2] "some link does not match"
    auto fold = [&](float s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s = chain4(s, r[q]);
        return s;
    };
    // approximate block total (any order)
    const v4f ps = (r[0] + r[1]) + (r[2] + r[3]);
    const float tot = (ps.x + ps.y) + (ps.z + ps.w);
    if (j == 0) { flag[0] = 0; flag[1] = 0; }
    float g = init + (wg_scan_incl(tot, wtot + NW) - tot);      // guessed input of block j (its barrier also publishes the flags)
    if (j == 0) g = init;
    float out = 0.0f;
    for (int round = 0; round < kSampThreads + 1; ++round) {
        const int p = round & 1;
        out = fold(g);
        if (lane == 63) bnd[p * NW + wave] = out;
        __syncthreads();                                        // A: wave-boundary outputs; last round's readers of flag[p ^ 1] are through
        if (j == 0) flag[p ^ 1] = 0;
        float prev = wave_prev_lane(out);                       // lane - 1's output (lane 0: +0.0)
        if (lane == 0 && wave > 0) prev = bnd[p * NW + wave - 1];
        const bool ok = (j == 0) || (__float_as_uint(prev) == __float_as_uint(g));     // every link bitwise
        float e = (j == 0) ? 0.0f : prev - g;                   // mismatch at link j
        if (__any(!ok) && lane == 0) flag[p] = 1;
        float v = wave_scan_incl(e);
        if (lane == 63) wtot[p * NW + wave] = v;
        __syncthreads();                                        // B: flag[p], wave totals of the corrections
        if (flag[p] == 0) {                                     // (uniform) block 0 is exact by construction; round r fixes block r
            if (rounds_out != nullptr && j == 0) *rounds_out = round + 1;
            break;
        }
        float tw = lane < NW ? wtot[p * NW + lane] : 0.0f;
        tw = wave_scan_incl(tw);
        const float off = wave > 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tw), wave - 1)) : 0.0f;
        g = g + (v + off);                                      // corrected inputs under the translation assumption (spec_sum_lanes)
        if (j == 0) g = init;
    }
    in_j = g;
    out_j = out;
}

// first index i (in sequence order) whose running sum c_i satisfies (GT ? c_i > x : x < c_i) -- the same comparison;
// returns -1 if none.  cum receives c_i (or the total when none).  red: 2 ints + 1 float of LDS.
__device__ __forceinline__ int wg_first_crossing(const float* t, int blen, float in_j, float out_j, float x, int* red, float* cum) {
    const int j = threadIdx.x;
    __syncthreads();                                        // a previous call's results have been consumed
    if (j == 0) { red[0] = 0x7fffffff; red[1] = -1; }
    __syncthreads();
    if (x < out_j) atomicMin(&red[0], j);                   // running sums are non-decreasing: first lane whose block crosses
    __syncthreads();
    const int lane_hit = red[0];
    if (lane_hit == 0x7fffffff) {
        if (j == kSampThreads - 1) *(float*)&red[2] = out_j;
        __syncthreads();
        *cum = *(float*)&red[2];
        return -1;
    }
    if (j == lane_hit) {
        float c = in_j;
        const float* b = t + (size_t)j * blen;
        int i = 0;
        for (; i < blen; ++i) {
            c = c + b[i];
            if (x < c) break;
        }
        red[1] = j * blen + i;
        *(float*)&red[2] = c;
    }
    __syncthreads();
    *cum = *(float*)&red[2];
    return red[1];
}

// The sequences are walked in segments of kSegFloats terms: lane j holds terms [base + 16 j, + 16) of a segment in REGISTERS (requested
// straight from global memory, the next segment's before the current segment's rounds start; round 5: the LDS staging pass and its
// 4-way conflicted read-back cost more than the rounds of a late segment) and the exact running sum at the end of a segment is the
// `init` of the next.  visit(seg_base, in_j, out_j) runs after each segment's exact prefix and returns true to stop early; the
// segment's terms are t[seg_base ..] (entries past `len` count as zero: no running sum moves there).  Returns the exact running sum
// after the last segment walked.
constexpr int kSegBlen = 16;
constexpr int kSegFloats = kSampThreads * kSegBlen;
template <class Visit>
__device__ __forceinline__ float wg_walk_segments(const float* t, int len, float init, float* xch, float* carry_lds,
                                                  int* rounds_out, Visit&& visit) {
    const int j = threadIdx.x;
    float carry = init;
    int rounds_total = 0;
    v4f cur[kSegBlen / 4], nx[kSegBlen / 4];
    auto request = [&](v4f (&dst)[kSegBlen / 4], int base) {
#pragma unroll
        for (int u = 0; u < kSegBlen / 4; ++u) {
            const int i = base + kSegBlen * j + 4 * u;
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (i + 3 < len) v = *(const v4f*)(t + i);
            else {
                if (i < len) v.x = t[i];
                if (i + 1 < len) v.y = t[i + 1];
                if (i + 2 < len) v.z = t[i + 2];
            }
            dst[u] = v;
        }
    };
    request(nx, 0);
    for (int base = 0; base < len; base += kSegFloats) {
#pragma unroll
        for (int u = 0; u < kSegBlen / 4; ++u) cur[u] = nx[u];
        if (base + kSegFloats < len) request(nx, base + kSegFloats);
        float in_j, out_j;
        int r = 0;
        wg_exact_prefix(cur, carry, xch, in_j, out_j, j == 0 ? &r : nullptr);
        rounds_total += r;
        const bool stop = visit(base, in_j, out_j);
        __syncthreads();
        if (j == kSampThreads - 1) *carry_lds = out_j;
        __syncthreads();
        carry = *carry_lds;
        if (stop) break;
    }
    if (rounds_out != nullptr && j == 0) *rounds_out = rounds_total;
    return carry;
}

// Stable LSD radix sort (8-bit digits) of keys k0[0 .. n0) by their HIGH 32 bits, descending, by the whole 1024-thread
// workgroup; k1 is a second buffer of >= n0 entries, the returned pointer is the buffer that holds the result.  The
// candidates arrive in ascending token order and the sort is stable, so equal probabilities stay in ascending token order
// -- the order the bitonic network below produces from the (probability, ~index) key.  Wave w owns a contiguous run of
// the input; per pass: per-wave digit histogram (LDS atomics), offsets (digit-major, wave-minor), then the run is walked
// 64 keys at a time with a ballot match giving every lane its rank among the lanes of the same digit.  A pass whose digit
// is the same for every key (the sign/exponent byte of a probability almost always is) is skipped.
// cnt: 16 * 256 + 256 words of LDS; flag: one int of LDS.  ~0.1 ms per pass for 150k keys (one CU), against 2.7 ms for the
// 153-step bitonic network at that size.
__device__ __forceinline__ unsigned long long* wg_radix_sort_desc(unsigned long long* k0, unsigned long long* k1, int n0,
                                                                   unsigned* cnt, int* flag) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = kSampThreads / 64;
    const int per_wave = ((n0 + NW * 64 - 1) / (NW * 64)) * 64;
    const int w0 = min(wave * per_wave, n0), w1 = min(w0 + per_wave, n0);
    unsigned* tot = cnt + NW * 256;
    unsigned long long* src = k0;
    unsigned long long* dst = k1;
    for (int shift = 32; shift < 64; shift += 8) {
        __syncthreads();
        for (int i = tid; i < NW * 256; i += kSampThreads) cnt[i] = 0u;
        __syncthreads();
        for (int i = w0 + lane; i < w1; i += 64) atomicAdd(&cnt[wave * 256 + (255 - (int)((src[i] >> shift) & 255ull))], 1u);
        __syncthreads();
        if (tid < 256) {
            unsigned run = 0u;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const unsigned c = cnt[w * 256 + tid];
                cnt[w * 256 + tid] = run;                  // keys of this digit in earlier waves
                run += c;
            }
            tot[tid] = run;
        }
        __syncthreads();
        if (tid == 0) {
            unsigned run = 0u;
            int same = 0;
            for (int d = 0; d < 256; ++d) {
                const unsigned c = tot[d];
                tot[d] = run;                              // keys of larger digits (descending order)
                run += c;
                same |= (c == (unsigned)n0) ? 1 : 0;
            }
            *flag = same;
        }
        __syncthreads();
        if (*flag) continue;                               // every key has this digit: the pass would be the identity
        for (int base = w0; base < w1; base += 64) {
            const int i = base + lane;
            const bool live = i < w1;
            const unsigned long long key = live ? src[i] : 0ull;
            const int d = 255 - (int)((key >> shift) & 255ull);
            unsigned long long peers = __builtin_amdgcn_ballot_w64(live);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = ((d >> b) & 1) != 0;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(live && bit);
                peers &= bit ? m : ~m;
            }
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0u));
            if (live) {
                const unsigned at = tot[d] + cnt[wave * 256 + d] + (unsigned)rank;
                dst[at] = key;
            }
            wave_lds_sync();                               // every lane has read the running count before its leader moves it
            if (live && (peers >> lane) == 1ull) cnt[wave * 256 + d] += (unsigned)__builtin_popcountll(peers);   // highest lane of the group
            wave_lds_sync();
        }
        unsigned long long* t = src; src = dst; dst = t;
    }
    __syncthreads();
    return src;
}

#ifdef Q3_DEV
#define SAMP_STAMP(i) do { if (a_in.stamps != nullptr && threadIdx.x == 0) a_in.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SAMP_STAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(kSampThreads) void k_sample(const SampleArgs a_in) {
    __shared__ float xch[kSampThreads + 16];
    __shared__ int red[4];
    __shared__ float fred[16];
    __shared__ int ired[20];
    __shared__ float hmass[2048];
    __shared__ float carry_lds;
    extern __shared__ __attribute__((aligned(16))) float seg[];   // kSegFloats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SampleArgs a = a_in;
    {
        const size_t sb = blockIdx.x;                  // stream
        a.logits = a_in.logits + sb * a_in.sb_logits;
        a.probs = a_in.probs + sb * a_in.sb_scratch;
        a.sp = a_in.sp + sb * a_in.sb_scratch;
        a.keys = a_in.keys + sb * a_in.sb_keys;
        a.ss = a_in.ss + sb;
        a.st = a_in.st + sb;
        a.out_tokens = a_in.out_tokens + sb * (size_t)a_in.out_cap;
    }
    State* st = a.st;
    SamplerState* ss = a.ss;
    const float temperature = ss->temperature, topp = ss->topp;

    // prompt positions of a chat-mode prefill: the reference draws (and discards) a sample per prompt token, which
    // advances the rng by exactly one coin; the input of the next forward stays the prompt token k_next selected
    const int phase = a.phase;
    const bool discard = st->step < st->prompt_len;
    float coin, inv;
    const int n = a.n;
    if (phase != 2) {
    __syncthreads();
    unsigned long long rs = ss->rng;
    rs ^= rs >> 12;
    rs ^= rs << 25;
    rs ^= rs >> 27;                                               // sampler.rs:44-49
    const unsigned r32 = (unsigned)((rs * 0x2545F4914F6CDD1Dull) >> 32);
    coin = (float)(r32 >> 8) / 16777216.0f;                      // sampler.rs:52-54
    __syncthreads();
    if (tid == 0) ss->rng = rs;
    if (discard) return;

    SAMP_STAMP(0);
    if (!a.pre_exp) {
        // ---- logits / temperature, max                                              sampler.rs:124-126, layers.rs:496
        float m = -__builtin_inff();
        for (int i = tid; i < n; i += kSampThreads) m = fmaxf(m, a.logits[i] / temperature);
        m = group_max_f32(m, 64);
        if (lane == 0) fred[wave] = m;
        __syncthreads();
        m = fred[0];
        for (int w = 1; w < kSampThreads / 64; ++w) m = fmaxf(m, fred[w]);
        // ---- e = exp(x - max)                                                        layers.rs:497-501
        for (int i = tid; i < n; i += kSampThreads) a.probs[i] = q3_expf(a.logits[i] / temperature - m);
    }
    // ---- exact sum in index order, p = e * (1/sum)                                  layers.rs:502-505
    __syncthreads();
    const float esum = wg_walk_segments(a.probs, n, -0.0f, xch, &carry_lds, &ss->rounds[0],   // Iterator::sum from -0.0
                                        [](int, float, float) { return false; });
    inv = 1.0f / esum;
    SAMP_STAMP(1);
    if (phase == 1) {
        // the chip-wide passes normalise, bin and compact; this workgroup clears their accumulators and hands over
        for (int i = tid; i < kSampHistBins; i += kSampThreads) a.hist[i] = 0.0f;
        if (tid == 0) { ss->inv = inv; ss->coin = coin; }
        return;
    }
    __syncthreads();
    for (int i = tid; i < n; i += kSampThreads) a.probs[i] = a.probs[i] * inv;
    __syncthreads();
    SAMP_STAMP(2);
    } else {
        if (discard) return;
        inv = ss->inv;
        coin = ss->coin;
        SAMP_STAMP(2);
    }

    int result;
    if (topp <= 0.0f || topp >= 1.0f) {
        // ---- sample_mult: first i with coin < cdf_i, cdf from 0.0                 sampler.rs:62-71
        int hit = -1;
        wg_walk_segments(a.probs, n, 0.0f, xch, &carry_lds, &ss->rounds[1], [&](int base, float in_j, float out_j) {
            float cum;
            const int h = wg_first_crossing(a.probs + base, kSegBlen, in_j, out_j, coin, red, &cum);
            if (h >= 0) hit = base + h;
            return h >= 0;
        });
        result = (hit >= 0 && hit < n) ? hit : n - 1;
    } else {
        // ---- sample_topp                                                          sampler.rs:74-112
        const float cutoff = (1.0f - topp) / (float)((n - 1) > 1 ? (n - 1) : 1);
        // Only the head of the sorted candidate list is ever walked (until the cumulative probability exceeds topp), and
        // "all candidates whose probability key is >= T" is exactly a prefix of that list for any T.  A histogram of the keys
        // (mass per 2^-3-binade bin, any order: heuristic only) picks a T that should cover topp; if the exact walk over that
        // prefix does not cross topp after all, the second attempt sorts every candidate.
        // Round 3: every pass over the vocabulary is COALESCED (thread i takes i, i + 1024, ...; the index-ordered compaction
        // gives each wave a contiguous range and ranks with ballots) -- the thread-contiguous blocks of round 2 touched 64
        // cache lines per load instruction, ~0.2 ms per pass on one CU.  The histogram only covers the 64 bins below the
        // largest probability (p_max = 1 * inv: the exp of the maximum is 1): the nucleus of a peaked distribution lives there,
        // the mass of the (many) tokens below the window is irrelevant, and their LDS atomics -- tens of thousands on a
        // handful of hot bins -- were the most expensive part of the draw.  A wave whose 64 values share a bin adds once.
        unsigned tkey0 = 0u;
        if (phase != 2) {
        for (int i = tid; i < 2048; i += kSampThreads) hmass[i] = 0.0f;
        __syncthreads();
        const unsigned bmax = total_order_key(inv) >> 21;
        const unsigned bfloor = bmax > 64u ? bmax - 64u : 0u;
        for (int i0 = 0; i0 < n; i0 += kSampThreads) {
            const int i = i0 + tid;
            const float p = i < n ? a.probs[i] : 0.0f;
            const unsigned bin = total_order_key(p) >> 21;
            const bool in = i < n && p >= cutoff && bin >= bfloor;
            const unsigned tag = in ? bin : 0xffffffffu;
            const unsigned first = (unsigned)__builtin_amdgcn_readfirstlane((int)tag);
            if (__all(tag == first)) {                         // wave-uniform bin (or nothing to add)
                if (first != 0xffffffffu) {
                    const float sw = group_sum_f32(p, 64);
                    if (lane == 0) atomicAdd(&hmass[first], sw);
                }
            } else if (in) atomicAdd(&hmass[bin], p);
        }
        __syncthreads();
        if (tid == 0) {
            float acc = 0.0f;
            int b = 2047;
            bool found = false;
            for (; b > (int)bfloor; --b) {
                acc += hmass[b];
                if (acc > topp * 1.001f + 1e-6f) { found = true; break; }
            }
            ired[16] = (found && b > 0) ? b - 1 : 0;       // one bin of margin; nucleus below the window: every candidate
        }
        __syncthreads();
        tkey0 = (unsigned)ired[16] << 21;
        }
        SAMP_STAMP(3);
        int n0 = 0, hit = 0;
        const unsigned long long* sorted = a.keys;
        constexpr int NW = kSampThreads / 64;
        const int per_wave = ((n + NW * 64 - 1) / (NW * 64)) * 64;
        const int w0 = min(wave * per_wave, n), w1 = min(w0 + per_wave, n);
        for (int attempt = 0; attempt < 2; ++attempt) {
            const unsigned tkey = attempt == 0 ? tkey0 : 0u;
            __syncthreads();
            if (phase == 2 && attempt == 0) {
                n0 = ss->n0;                                   // k_sample_scatter wrote the first attempt's candidates
            } else {
            // candidates in index order: wave w owns the contiguous index range [w0, w1), 64 consecutive indices per load
            int cnt = 0;
            for (int base = w0; base < w1; base += 64) {
                const int i = base + lane;
                const float p = i < w1 ? a.probs[i] : 0.0f;
                const bool c = i < w1 && p >= cutoff && total_order_key(p) >= tkey;
                cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(c));
            }
            if (lane == 0) ired[wave] = cnt;
            __syncthreads();
            int off = 0;
            for (int w = 0; w < wave; ++w) off += ired[w];
            n0 = 0;
            for (int w = 0; w < NW; ++w) n0 += ired[w];
            int run = off;
            for (int base = w0; base < w1; base += 64) {
                const int i = base + lane;
                const float p = i < w1 ? a.probs[i] : 0.0f;
                const bool c = i < w1 && p >= cutoff && total_order_key(p) >= tkey;
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(c);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                if (c) a.keys[run + rank] = ((unsigned long long)total_order_key(p) << 32) | (unsigned)(0xffffffffu - (unsigned)i);
                run += __builtin_popcountll(mask);
            }
            }
            if (attempt == 0) SAMP_STAMP(4);
#ifdef Q3_DEV
            if (a_in.stamps != nullptr && tid == 0 && attempt == 0) a_in.stamps[9] = (unsigned long long)n0;
#endif
            int n2 = 1;
            while (n2 < n0) n2 <<= 1;
            const unsigned long long* ks = a.keys;            // sorted candidates
            if (n0 > kSampRadixMin) {
                // long candidate lists (flat distributions: up to the whole vocabulary): stable radix sort into keys / keys2
                __syncthreads();
                ks = wg_radix_sort_desc(a.keys, a.keys + a.keys2_off, n0, (unsigned*)seg, &ired[17]);
            } else {
                for (int i = n0 + tid; i < n2; i += kSampThreads) a.keys[i] = 0ull;   // below every real key
                __syncthreads();
                // bitonic sort, descending: probability first, then ascending index (see header)
                for (int k = 2; k <= n2; k <<= 1) {
                    for (int jj = k >> 1; jj > 0; jj >>= 1) {
                        for (int i = tid; i < n2; i += kSampThreads) {
                            const int l = i ^ jj;
                            if (l > i) {
                                const unsigned long long x = a.keys[i], y = a.keys[l];
                                const bool desc = (i & k) == 0;
                                if (desc ? (x < y) : (x > y)) { a.keys[i] = y; a.keys[l] = x; }
                            }
                        }
                        __syncthreads();
                    }
                }
            }
            sorted = ks;
            if (attempt == 0) SAMP_STAMP(5);
            for (int i = tid; i < n0; i += kSampThreads) a.sp[i] = key_to_float((unsigned)(ks[i] >> 32));
            __syncthreads();
            // cumulative probability in sorted order, truncation point, then the cdf walk with r = coin * cumulative
            int last_idx = -1;
            float cumulative = 0.0f;
            const float total = wg_walk_segments(a.sp, n0, 0.0f, xch, &carry_lds, &ss->rounds[2 + attempt],
                                                 [&](int base, float in_j, float out_j) {
                float cum;
                const int h = wg_first_crossing(a.sp + base, kSegBlen, in_j, out_j, topp, red, &cum);   // first cum > topp
                if (h >= 0) { last_idx = base + h; cumulative = cum; }
                return h >= 0;
            });
            if (attempt == 0) SAMP_STAMP(6);
            const bool crossed = last_idx >= 0 && last_idx < n0;
            if (!crossed && attempt == 0) continue;                                             // prefix too short: sort everything
            if (!crossed) { last_idx = n0 - 1; cumulative = total; }
            const float r = coin * cumulative;
            hit = -1;
            wg_walk_segments(a.sp, last_idx + 1, 0.0f, xch, &carry_lds, nullptr, [&](int base, float in_j, float out_j) {
                float cum;
                const int h = wg_first_crossing(a.sp + base, kSegBlen, in_j, out_j, r, red, &cum);       // first r < cdf
                if (h >= 0) hit = base + h;
                return h >= 0;
            });
            if (hit < 0 || hit > last_idx) hit = last_idx;
            break;
        }
        result = (n0 > 0) ? (int)(0xffffffffu - (unsigned)(sorted[hit] & 0xffffffffull)) : 0;
    }
    SAMP_STAMP(7);
#ifdef Q3_DEV
    if (a_in.stamps != nullptr && tid == 0) a_in.stamps[10] = (unsigned long long)(unsigned)ss->rounds[0];
    if (a_in.stamps != nullptr && tid == 0) a_in.stamps[8] = (unsigned long long)(topp > 0.0f && topp < 1.0f ? 1 : 0);
#endif
    if (tid == 0) {
        // k_next already advanced (pos, step) and stored the argmax: the sampled token replaces it
        st->token = result;
        const int s = st->step - 1;
        if (s >= 0 && s < a.out_cap) a.out_tokens[s] = result;
    }
}

// The first two passes of Sampler::sample spread over the chip (grid: workgroups x streams): x = logit / T (sampler.rs:124-126),
// e = exp(x - max) (layers.rs:496-501).  The maximum of the scaled logits is RN(l_max / T): division by T > 0 and its rounding
// are monotone, so it is the scaled value of the largest logit -- which the classifier launch already left in State::argmax
// (total-order key of the best logit).  Element-wise and order-free: bit-identical to the in-kernel passes of k_sample.
__global__ __launch_bounds__(256) void k_sample_exp(const SampleArgs a_in) {
    const size_t sb = blockIdx.y;
    const State* st = a_in.st + sb;
    if (st->step < st->prompt_len) return;                     // a discarded prompt-position draw
    const SamplerState* ss = a_in.ss + sb;
    const float temperature = ss->temperature;
    const float* logits = a_in.logits + sb * a_in.sb_logits;
    float* probs = a_in.probs + sb * a_in.sb_scratch;
    const float m = key_to_float((unsigned)(st->argmax >> 32)) / temperature;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a_in.n; i += gridDim.x * 256) probs[i] = q3_expf(logits[i] / temperature - m);
}

// ---- chip-wide passes of the pipelined draw (grid kSampGrid, one stream).  None of them touches an order-sensitive quantity:
// p = e * inv is element-wise, the mass histogram only chooses a threshold (any threshold gives the same draw: the tail kernel
// verifies that the sorted prefix crosses topp and otherwise repeats with every candidate), and the compaction writes the
// candidates in ascending index order exactly as the single-workgroup pass does.
__device__ __forceinline__ bool samp_skip(const SampleArgs& a) {
    const State* st = a.st;
    const SamplerState* ss = a.ss;
    return st->step < st->prompt_len || !(ss->topp > 0.0f && ss->topp < 1.0f);
}
// thread 0's scan of the window bins (same rule as k_sample), broadcast through LDS
__device__ __forceinline__ unsigned samp_threshold(const SampleArgs& a, float inv, float topp, unsigned* lds_word) {
    if (threadIdx.x == 0) {
        const unsigned bmax = total_order_key(inv) >> 21;
        const unsigned bfloor = bmax > 64u ? bmax - 64u : 0u;
        float acc = 0.0f;
        int b = kSampHistBins - 1;
        bool found = false;
        for (b = (int)min(bmax, (unsigned)kSampHistBins - 1u); b > (int)bfloor; --b) {
            acc += a.hist[b];
            if (acc > topp * 1.001f + 1e-6f) { found = true; break; }
        }
        *lds_word = ((found && b > 0) ? (unsigned)(b - 1) : 0u) << 21;
    }
    __syncthreads();
    return *lds_word;
}
__global__ __launch_bounds__(256) void k_sample_norm_hist(const SampleArgs a) {
    __shared__ float hm[66];
    const State* st = a.st;
    const SamplerState* ss = a.ss;
    if (st->step < st->prompt_len) return;
    const float inv = ss->inv, topp = ss->topp;
    const bool nucleus = topp > 0.0f && topp < 1.0f;
    const int n = a.n, tid = threadIdx.x, lane = tid & 63;
    const float cutoff = (1.0f - topp) / (float)((n - 1) > 1 ? (n - 1) : 1);
    const unsigned bmax = total_order_key(inv) >> 21;
    const unsigned bfloor = bmax > 64u ? bmax - 64u : 0u;
    if (tid < 66) hm[tid] = 0.0f;
    __syncthreads();
    for (int i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256) {
        const int i = i0 + tid;
        float p = 0.0f;
        if (i < n) { p = a.probs[i] * inv; a.probs[i] = p; }                 // layers.rs:503-505
        if (!nucleus) continue;
        const unsigned bin = total_order_key(p) >> 21;
        const bool in = i < n && p >= cutoff && bin >= bfloor && bin <= bmax;
        const unsigned tag = in ? bin : 0xffffffffu;
        const unsigned first = (unsigned)__builtin_amdgcn_readfirstlane((int)tag);
        if (__all(tag == first)) {
            if (first != 0xffffffffu) {
                const float sw = group_sum_f32(p, 64);
                if (lane == 0) atomicAdd(&hm[first - bfloor], sw);
            }
        } else if (in) atomicAdd(&hm[bin - bfloor], p);
    }
    __syncthreads();
    if (nucleus && tid < 66 && hm[tid] != 0.0f && bfloor + tid < (unsigned)kSampHistBins) atomicAdd(&a.hist[bfloor + tid], hm[tid]);
}
// workgroup g owns the contiguous index range [g * per, (g + 1) * per): counts its candidates
__global__ __launch_bounds__(256) void k_sample_count(const SampleArgs a) {
    __shared__ unsigned tk;
    __shared__ int wsum[4];
    if (samp_skip(a)) return;
    SamplerState* ss = a.ss;
    const float topp = ss->topp;
    const int n = a.n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float cutoff = (1.0f - topp) / (float)((n - 1) > 1 ? (n - 1) : 1);
    const unsigned tkey = samp_threshold(a, ss->inv, topp, &tk);
    const int per = ((n + (int)gridDim.x * 256 - 1) / ((int)gridDim.x * 256)) * 256;
    const int g0 = min((int)blockIdx.x * per, n), g1 = min(g0 + per, n);
    int cnt = 0;
    for (int i = g0 + tid; i < g1; i += 256) {
        const float p = a.probs[i];
        cnt += (p >= cutoff && total_order_key(p) >= tkey) ? 1 : 0;
    }
    cnt = (int)group_sum_f32((float)cnt, 64);                                   // <= 64 * per / 256: exact in f32
    if (lane == 0) wsum[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        a.counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (blockIdx.x == 0) ss->tkey0 = tkey;
    }
}
// ... and writes them to keys[] in ascending index order behind the candidates of the lower ranges
__global__ __launch_bounds__(256) void k_sample_scatter(const SampleArgs a) {
    __shared__ int wcnt[4];
    __shared__ int base_s;
    if (samp_skip(a)) return;
    SamplerState* ss = a.ss;
    const float topp = ss->topp;
    const int n = a.n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float cutoff = (1.0f - topp) / (float)((n - 1) > 1 ? (n - 1) : 1);
    const unsigned tkey = ss->tkey0;
    const int per = ((n + (int)gridDim.x * 256 - 1) / ((int)gridDim.x * 256)) * 256;
    const int g0 = min((int)blockIdx.x * per, n), g1 = min(g0 + per, n);
    // candidates of the lower workgroup ranges (and, on workgroup 0, of all ranges: n0)
    {
        int lo = 0, all = 0;
        for (int w = tid; w < (int)gridDim.x; w += 256) { const int c = a.counts[w]; all += c; lo += w < (int)blockIdx.x ? c : 0; }
        lo = (int)group_sum_f32((float)lo, 64);                                  // < 2^24 candidates: exact in f32
        all = (int)group_sum_f32((float)all, 64);
        if (lane == 0) { wcnt[wave] = lo; }
        __syncthreads();
        if (tid == 0) base_s = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
        if (lane == 0) wcnt[wave] = all;
        __syncthreads();
        if (tid == 0 && blockIdx.x == 0) ss->n0 = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
    // wave w owns a contiguous quarter of the range
    const int perw = per / 4;
    const int w0 = min(g0 + wave * perw, g1), w1 = min(w0 + perw, g1);
    int cnt = 0;
    for (int b = w0; b < w1; b += 64) {
        const int i = b + lane;
        const float p = i < w1 ? a.probs[i] : 0.0f;
        const bool c = i < w1 && p >= cutoff && total_order_key(p) >= tkey;
        cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(c));
    }
    if (lane == 0) wcnt[wave] = cnt;
    __syncthreads();
    int run = base_s;
    for (int w = 0; w < wave; ++w) run += wcnt[w];
    for (int b = w0; b < w1; b += 64) {
        const int i = b + lane;
        const float p = i < w1 ? a.probs[i] : 0.0f;
        const bool c = i < w1 && p >= cutoff && total_order_key(p) >= tkey;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(c);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        if (c) a.keys[run + rank] = ((unsigned long long)total_order_key(p) << 32) | (unsigned)(0xffffffffu - (unsigned)i);
        run += __builtin_popcountll(mask);
    }
}

// advance the xorshift64* stream by `count` coins (batched prefill: one discarded sample per prompt position)
__global__ void k_rng_skip(SamplerState* ss, int count) {
    unsigned long long rs = ss->rng;
    for (int i = 0; i < count; ++i) {
        rs ^= rs >> 12;
        rs ^= rs << 25;
        rs ^= rs >> 27;
    }
    ss->rng = rs;
}

}  // namespace q3
