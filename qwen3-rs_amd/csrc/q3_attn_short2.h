// q3_attn_short2.h -- short-context attention, second cut (round 5).  Included by q3_kernels.h inside namespace q3.
//
// Same arithmetic, same order as k_attn_short (layers.rs:346-419, 495-506) -- one workgroup per query head, pos < 256,
// head_dim 128 -- with the data movement re-cut around what the r05 probes measured (profiles/r05_primitive_costs.txt):
//   * k_attn_short let lane t of a score wave pull ITS key row with 32 dwordx4 loads: 64 lanes x 32 requests of 16 bytes,
//     each lane in a different 4 KB-strided row.  The CU's address path serialises them -- "loads issued" moved from 1,870
//     to 3,280 cycles after entry between positions 17 and 70 (70 cycles per position) and everything behind it with it.
//     Here the rows 0..pos-1 of the kv head are requested COALESCED (one dwordx4 instruction = two whole rows) by four
//     staging waves that have nothing else to do, written to LDS with a 4-float row pad, and lane t reads its row back
//     with conflict-free ds_read_b128 while it walks the dot.  The current position's key goes from the k-norm wave into
//     the same LDS tile, so the score loop has no special lane.
//   * 8 waves (2 per SIMD) instead of 4: q-norm, k-norm, 2 staging / far-context score waves, 2 output waves, 2 staging
//     waves.  One pass covers 256 timesteps (k_attn_short: two passes of 128).
//   * the staging threads hold their float4 of a key row AND need only ONE float4 of q for it (their column): they form the
//     products q_i * K[t][i] (each rounded on its own, layers.rs:397) before the tile is written, so the score lanes' chain is
//     nothing but ds_read_b128 + four dependent adds per float4 -- 1.25 instructions per term instead of 2 (a chain costs its
//     instruction count x ~4.6 cycles).
//   * contexts of <= 64 positions: the value rows take the same road (coalesced loads by the staging waves, an LDS tile in the
//     upper half of the dynamic allocation), and the output lanes pull their column into registers while the scores run;
//     beyond, the output waves load their column themselves like k_attn_short (the key tile needs the whole allocation).
//     The value rows are NOT part of barrier A: a CU pulls ~16 bytes per clock through its memory path, i.e. ~32 cycles per
//     512-byte row, and the scores must not wait for the second half of that traffic.  The staging waves commit the value
//     tile after barrier A, raise an LDS flag and END (a finished wave no longer counts at s_barrier); the output waves,
//     idle until the probabilities exist, poll the flag.
//   * grid (heads, 2): past 64 positions the workgroup's time is the key + value bytes through ONE CU's memory path, so the two
//     64-element halves of a head's output go to two workgroups (each loads every key row but only its half of the value
//     rows: -0.15 ... -0.3 us per launch at positions 64 ... 255, profiles/r05_attn_short2.txt); up to 64 positions the second
//     workgroup leaves as soon as it knows the position.
//   * AttnArgs::row_steps > 0 (the host knows the position of every forward it enqueues and keeps one captured graph per position
//     range below 64): the staging waves request that many 8-row steps of key and value rows AT KERNEL ENTRY instead of behind
//     the position's own round trip (rows past the context are valid cache rows: requested, never committed); the launch then
//     has ONE dependent memory round trip in front of the tiles instead of two.
// Dynamic LDS: max_t rows x (HD + 4) floats (attn_short2_smem_bytes): product tile [t][HD + 4]; value tile [t][HD] behind row 128.
//
// Barrier contract.  The wave roles meet a DIFFERENT number of s_barrier each, and staging waves end in the middle of the kernel:
// the code relies on the gfx950 rule that a terminated wave no longer counts at its workgroup's s_barrier (and on LDS spin flags
// -- vflag, bcount -- between waves of one workgroup).  That is outside the portable HIP barrier contract; it was validated on
// gfx950 only (every tier boundary: test_attention_short_contexts_head_dim_128, product AND developer build), hence the #error.
//   barriers each role meets, by context length np:
//   role (waves)                       | np <= 64        | 64 < np <= 128   | np > 128
//   q-norm / k-norm + score (0, 1)     | A', A, C        | A', A, C         | A', A, B, C
//   staging + far score (2, 3)         | A', A, then END | A', A, then END  | A', A, B, C
//   output (4, 5)                      | A', A, C        | A', A, C         | A', A, B, C
//   staging only (6, 7)                | A', A, then END | A', A, then END  | A', A, then END
//   (np <= 64: the staging waves commit the value tile BEHIND A, raise vflag and end; "barrier B" of a single score-wave pair is
//   the LDS counter bcount; the second workgroup of grid (heads, 2) leaves before any barrier when np <= 64.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_attn_short2 relies on gfx950 s_barrier behaviour for terminated waves; validated on gfx950 only"
#endif
constexpr int kS2Threads = 512;
constexpr int kS2MaxT = 256;
__host__ __device__ inline size_t attn_short2_smem_bytes(int hd, int max_t) { return 4 * (size_t)max_t * (size_t)(hd + 4); }

template <int V> struct IntC { static constexpr int value = V; };

#ifdef Q3_DEV
#define ATT2_STAMP(i, thr) do { if (a.stamps != nullptr && (a.debug & 64) == 0 && blockIdx.x == 3 && (int)threadIdx.x == (thr)) a.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ATT2_STAMP(i, thr) do { } while (0)
#endif

template <int HD>
__global__ __launch_bounds__(kS2Threads) void k_attn_short2(const AttnArgs a) {
    static_assert(HD == 128, "head dim instantiated");
    constexpr int NQ4 = HD / 4;          // float4 per K row
    constexpr int HALF = HD / 2;         // rotate-half pairing (i, i + HD/2)
    constexpr int LD = HD + 4;           // floats per product row (pad: 16 consecutive lanes' b128 reads cover all 64 banks)
    constexpr int VT = 64;               // contexts up to VT positions keep the value rows in LDS behind product row VT
    extern __shared__ __attribute__((aligned(16))) float ptile[];       // [<= kS2MaxT][LD]  q_i * K[t][i]
    float* const vtile = ptile + VT * LD;                               // [<= VT][HD]        (np <= VT only)
    __shared__ __attribute__((aligned(16))) float q_s[HD];
    __shared__ __attribute__((aligned(16))) float sq_s[2 * HD];        // squares of raw q | raw k
    __shared__ __attribute__((aligned(16))) float att[kS2MaxT];        // scores
    __shared__ __attribute__((aligned(16))) float att_e[kS2MaxT];      // exp(score - max)
    __shared__ __attribute__((aligned(16))) float att_p[kS2MaxT];      // probabilities
    __shared__ unsigned long long etab[32];                            // exp2 table of q3_expf, staged once
    __shared__ unsigned vflag;                                         // staging waves that have committed their value rows
    __shared__ unsigned bcount;                                        // score waves that have published their scores ("barrier B")
    ATT2_STAMP(0, 0);
    KSTAMP_BEGIN(a);
    Q3_PIN_S(a.st); Q3_PIN_S(a.pos_override); Q3_PIN_S(a.q); Q3_PIN_S(a.k_raw); Q3_PIN_S(a.key_cache); Q3_PIN_S(a.value_cache);
    Q3_PIN_S(a.q_norm_w); Q3_PIN_S(a.k_norm_w); Q3_PIN_S(a.rope); Q3_PIN_S(a.xb); Q3_PIN_S(a.n_heads); Q3_PIN_S(a.n_kv_heads);
    Q3_PIN_S(a.write_q);

    // workgroup -> head: the hardware deals workgroups round-robin over the 8 XCDs (linear id % 8), each with its own L2.  The query
    // heads of one kv head read the same key / value rows, so they take block indices that are equal modulo n_kv_heads (8 for every
    // listed model): the rows cross the fabric into ONE L2 and the other workgroups of the group hit there.
    const int kv_mul = a.kv_mul, xq = (int)((blockIdx.x * a.kvh_magic) >> 20);     // (attn_set_heads: no integer division here)
    const int kvh = (int)blockIdx.x - xq * a.n_kv_heads;
    const int h = kvh * kv_mul + xq;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t kvd = (size_t)a.n_kv_heads * HD;
    // the position is requested first by every role and turned into a scalar as late as the role allows
    const int pos_v = a.pos_override >= 0 ? a.pos_override : a.st->pos;
    const float* kbase = a.key_cache + (size_t)kvh * HD;
    const float* vbase = a.value_cache + (size_t)kvh * HD;

    if (wave == 4 || wave == 5) {
        // ================================ output waves ================================
        const int pos = __builtin_amdgcn_readfirstlane(pos_v);
        const int np = pos + 1;
        const bool vsplit = gridDim.y == 2 && np > 64;    // this workgroup owns output elements 64 y .. 64 y + 63
        if (gridDim.y == 2 && blockIdx.y == 1 && np <= 64) return;      // (every wave of the workgroup leaves)
        if (vsplit && wave == 5) {                        // one output wave is enough: keep the barriers company
            __syncthreads();                              // A'
            __syncthreads();                              // A
            if (np > 128) __syncthreads();                // B
            __syncthreads();                              // C
            return;
        }
        const int e = vsplit ? 64 * (int)blockIdx.y + lane : 64 * (wave - 4) + lane;     // output element of this lane
        const unsigned eoff = 4u * (unsigned)e;
        float vv[kShortVSets][32];
        auto v_issue = [&](float (&R)[32], int c) {
            // rows past the context re-read row pos (finite: written by the QKV launch); their probability is +0.0
            // (row base wave-uniform, lane offset e: scalar-base addressing, no 64-bit address pair per load)
#pragma unroll
            for (int u = 0; u < 32; ++u) R[u] = *(const float*)((const char*)(vbase + (size_t)min(32 * c + u, pos) * kvd) + eoff);
        };
        auto v_pull = [&](float (&R)[32], int c) {        // the same column out of the LDS value tile
            const float* vp = vtile + e;
#pragma unroll
            for (int u = 0; u < 32; ++u) R[u] = vp[min(32 * c + u, pos) * HD];
        };
        const bool vlds = np <= VT;                       // wave-uniform
        if (tid == 256) { vflag = 0u; bcount = 0u; }      // (raised only behind barrier A)
        if (!vlds) {
#pragma unroll
            for (int c = 0; c < 2; ++c) v_issue(vv[c], c);
        }
        __syncthreads();                                  // A': q_s
        if (!vlds) {
#pragma unroll
            for (int c = 2; c < kShortVSets; ++c)
                if (32 * c < np) v_issue(vv[c], c);       // a context of <= 160 positions is all in flight before the scores exist
        }
        __syncthreads();                                  // A: product tile / etab
        if (vlds) {
            // the four staging waves commit their value rows behind barrier A and count themselves in
            while (__hip_atomic_load(&vflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u) __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int c = 0; c < VT / 32; ++c)
                if (32 * c < np) v_pull(vv[c], c);        // under the score chains
        }
        if (np > 128) __syncthreads();                    // B: scores (shorter contexts: a score-wave affair, see below)
        __syncthreads();                                  // C: exp(score - max)
        ATT2_STAMP(5, 256);
        // softmax denominator (layers.rs:495-506), probabilities
        const v4f e4 = ((const v4f*)att_e)[lane];
        float sum;
        if (np <= 128) {
            // one chain over the (zero padded) row, operands streamed from LDS in whole bursts of 8 float4 (s + 0.0 == s
            // once the first exp, > 0 or +0.0, is in)
            sum = seq_chain(-0.0f, (const v4f*)att_e, (((np + 3) >> 2) + 7) & ~7);
        } else {
            const float etot = (e4.x + e4.y) + (e4.z + e4.w);
            sum = spec_sum_lanes(etot, (np + 3) >> 2, [&](float s) { return chain4(s, e4); });
        }
        ATT2_STAMP(13, 256);
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
                    const v2f t01 = (v2f){pv.x, pv.y} * (v2f){R[4 * u4 + 0], R[4 * u4 + 1]};      // (two products per instruction)
                    const v2f t23 = (v2f){pv.z, pv.w} * (v2f){R[4 * u4 + 2], R[4 * u4 + 3]};
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
        // contexts beyond 160 positions: the remaining chunks go through set 0 / 1
        for (int c = kShortVSets; 32 * c < np; c += 2) {
            v_issue(vv[0], c);
            if (32 * (c + 1) < np) v_issue(vv[1], c + 1);
            fold_chunk(vv[0], c);
            if (32 * (c + 1) < np) fold_chunk(vv[1], c + 1);
        }
        ATT2_STAMP(14, 256);
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
        ATT2_STAMP(6, 256);
        KSTAMP_END(a);
        return;
    }

    int pos, np;
    if (wave < 2) {
        // ================================ norm waves: RMSNorm (layers.rs:109-119) + RoPE (layers.rs:173-185) of q / k
        const bool is_q = wave == 0;
        const float* rawp = is_q ? a.q + (size_t)h * HD : a.k_raw + (size_t)kvh * HD;
        const float r_lo = rawp[lane], r_hi = rawp[lane + HALF];
        const float* nw = is_q ? a.q_norm_w : a.k_norm_w;
        const float w_lo = nw[lane], w_hi = nw[lane + HALF];
        __builtin_amdgcn_sched_barrier(0);
        pos = __builtin_amdgcn_readfirstlane(pos_v);      // the oldest load of the wave: a counted wait
        np = pos + 1;
        if (gridDim.y == 2 && blockIdx.y == 1 && np <= 64) return;
        const v2f cs = *(const v2f*)(a.rope + (size_t)pos * HD + 2 * lane);   // (cos, sin) of rotation pair `lane`
        ATT2_STAMP(1, 0);
        float* sq = sq_s + (is_q ? 0 : HD);
        sq[lane] = r_lo * r_lo;
        sq[lane + HALF] = r_hi * r_hi;
        wave_lds_sync();
        const float ss = seq_chain(-0.0f, (const v4f*)sq, NQ4);      // strict left fold, layers.rs:113
        const float f = 1.0f / sqrtf(ss / (float)HD + kEps);
        const float xv = w_lo * (f * r_lo);
        const float yv = w_hi * (f * r_hi);
        const float a0 = xv * cs.x, b0 = yv * cs.y;
        const float a1 = xv * cs.y, b1 = yv * cs.x;
        const float lo = a0 - b0, hi = a1 + b1;          // layers.rs:181-182
        if (is_q) {
            q_s[lane] = lo;
            q_s[lane + HALF] = hi;
            if (a.write_q) {
                a.q[(size_t)h * HD + lane] = lo;
                a.q[(size_t)h * HD + lane + HALF] = hi;
            }
        } else if ((h % kv_mul) == 0 && blockIdx.y == 0) {   // K is normalised + rotated in place in the cache
            float* krow = a.key_cache + (size_t)pos * kvd + (size_t)kvh * HD;
            krow[lane] = lo;
            krow[lane + HALF] = hi;
        }
        ATT2_STAMP(2, 0);
        __syncthreads();                                  // A': q_s
        if (!is_q) {                                      // the current position's key: product row pos of the tile
            ptile[pos * LD + lane] = q_s[lane] * lo;
            ptile[pos * LD + lane + HALF] = q_s[lane + HALF] * hi;
        }
    } else {
        // ================================ staging waves: key rows 0 .. pos-1 (value rows 0 .. pos), coalesced, into LDS
        unsigned long long etv = 0ull;
        if (wave == 7 && lane < 32) etv = kExp2Tab[lane];
        // > 0: 8-row steps to request without waiting for the position (pos < 64) -- the position-range graphs lost their A/B
        // (docs/HISTORY.md section 0) and exist in the developer build only; the product does not fetch the field
#ifdef Q3_DEV
        const int hint = a.row_steps;
#else
        constexpr int hint = 0;
#endif
        if (hint <= 0) {
            pos = __builtin_amdgcn_readfirstlane(pos_v);
            np = pos + 1;
            if (gridDim.y == 2 && blockIdx.y == 1 && np <= 64) return;
        }
        ATT2_STAMP(8, 128);
        const int sidx = 64 * ((wave & 1) + (wave >= 6 ? 2 : 0)) + lane;     // 0 .. 255
        const int r0 = sidx >> 5, col = sidx & 31;        // 8 rows per step of the four staging waves
        // addresses: wave-uniform base of the 8-row step (scalar arithmetic) + this thread's byte offset inside the step (32 bits).
        // Steps past the end of the cache (short test contexts) re-read its last whole step; their rows lie past the context.
        const unsigned loff = 4u * ((unsigned)r0 * (unsigned)kvd + 4u * (unsigned)col);
        const int kmax = max(a.seq_len / 8 - 1, 0);
        float* lp = ptile + r0 * LD + 4 * col;
        float* lv = vtile + r0 * HD + 4 * col;
        auto stage = [&](auto tier, auto with_v) {
            constexpr int T = decltype(tier)::value;      // float4 per staging thread: 8 T rows of a tile
            constexpr bool V = decltype(with_v)::value != 0;
            v4f rk[T], rv[V ? T : 1];
#pragma unroll
            for (int k = 0; k < T; ++k) rk[k] = *(const v4f*)((const char*)(kbase + (size_t)min(k, kmax) * 8 * kvd) + loff);
            __builtin_amdgcn_sched_barrier(0);            // (the key rows are the older requests: barrier A waits for them only)
            if constexpr (V) {
#pragma unroll
                for (int k = 0; k < T; ++k) rv[k] = *(const v4f*)((const char*)(vbase + (size_t)min(k, kmax) * 8 * kvd) + loff);
            }
            ATT2_STAMP(9, 128);
            if (hint > 0) {                               // (wave-uniform) the position arrives under the row requests
                pos = __builtin_amdgcn_readfirstlane(pos_v);
                np = pos + 1;
            }
            __syncthreads();                              // A': q_s
            const v4f qv = ((const v4f*)q_s)[col];
#pragma unroll
            for (int k = 0; k < T; ++k) {
                // products of this float4 of the row with q, two per v_pk_mul_f32, each rounded on its own (layers.rs:397)
                const v2f p01 = (v2f){qv.x, qv.y} * (v2f){rk[k].x, rk[k].y};
                const v2f p23 = (v2f){qv.z, qv.w} * (v2f){rk[k].z, rk[k].w};
                v4f pr; pr.x = p01.x; pr.y = p01.y; pr.z = p23.x; pr.w = p23.y;
                if (r0 + 8 * k < pos) *(v4f*)(lp + 8 * k * LD) = pr;
            }
            if (wave == 7 && lane < 32) etab[lane] = etv;
            ATT2_STAMP(11, 128);
            if constexpr (V) {
                __syncthreads();                          // A: product tile / etab
#pragma unroll
                for (int k = 0; k < T; ++k)
                    if (r0 + 8 * k < np) *(v4f*)(lv + 8 * k * HD) = rv[k];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_fetch_add(&vflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                ATT2_STAMP(10, 128);
            }
        };
        // tiers by context length (wave-uniform): straight-line bursts of 1 .. 32 float4 per thread and tile (a run-time trip
        // count would put every load in its own basic block, and hipcc then throttles the burst with conservative vmcnt waits)
        const int tn = hint > 0 ? hint : (pos + 8) >> 3;  // 8-row steps that cover value rows 0 .. pos
        if (tn <= 1) stage(IntC<1>{}, IntC<1>{});
        else if (tn <= 2) stage(IntC<2>{}, IntC<1>{});
        else if (tn <= 3) stage(IntC<3>{}, IntC<1>{});
        else if (tn <= 4) stage(IntC<4>{}, IntC<1>{});
        else if (tn <= 6) stage(IntC<6>{}, IntC<1>{});
        else if (tn <= 8) stage(IntC<8>{}, IntC<1>{});
        else if (tn <= 12) stage(IntC<12>{}, IntC<0>{});
        else if (tn <= 16) stage(IntC<16>{}, IntC<0>{});
        else if (tn <= 24) stage(IntC<24>{}, IntC<0>{});
        else stage(IntC<32>{}, IntC<0>{});
        if (np <= VT) return;                             // value rows committed: nothing left for a staging wave (a finished
                                                          // wave no longer counts at the barriers below)
        __syncthreads();                                  // A (contexts past VT: no value tile, the barrier is met here)
        if (wave >= 6 || np <= 128) return;               // waves 2, 3 go on to score timesteps 128 .. 255
    }
    if (wave < 2) __syncthreads();                        // A (the staging waves met it above)
    ATT2_STAMP(3, 0);

    // ---- scores: att[t] = (q . K[t]) * scale, the products walked in index order       layers.rs:391-401
    const int t = 64 * wave + lane;                       // (waves 0 .. 3 from here on)
    float sc = -__builtin_inff();
    if (64 * wave < np) {                                 // wave-uniform
        const float scale = 1.0f / sqrtf((float)HD);
        const v4f* pr = (const v4f*)(ptile + min(t, pos) * LD);
        float dot = -0.0f;
        // product rows in bursts of 8 float4, the next burst requested before the current 32 adds run, ONE counted wait per
        // burst (DS operations return in order)
        v4f pa[8], pb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) pa[u] = pr[u];
#pragma unroll
        for (int b = 0; b < NQ4 / 8; ++b) {
            if (b + 1 < NQ4 / 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) pb[u] = pr[8 * (b + 1) + u];
                __builtin_amdgcn_s_waitcnt(0xC87F);      // lgkmcnt(8): everything but the burst just issued has landed
            } else {
                __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) dot = chain4(dot, pa[u]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) pa[u] = pb[u];
        }
        sc = t < np ? dot * scale : -__builtin_inff();
    }
    // contexts of <= 64 positions live in ONE score wave: its lanes hold every score, the maximum is a register reduction and
    // nobody meets at barrier B (np is uniform over the workgroup)
    const bool one_wave = np <= 64;
    float m;
    if (!one_wave) {
        // "barrier B" among the score waves only (the output waves may still be waiting for value rows, and must not hold up the
        // exps): publish, count in, poll.  Contexts of <= 128 positions have two score waves; they also fill the far slots.
        att[t] = sc;                                      // all 256 slots are written: -inf beyond the context
        if (np <= 128) {
            att[t + 128] = -__builtin_inff();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) {
                __hip_atomic_fetch_add(&bcount, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (__hip_atomic_load(&bcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 2u) __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            __syncthreads();                              // B (four score waves + the output waves; the staging-only waves are gone)
        }
        ATT2_STAMP(4, 0);
        const v4f s4 = ((const v4f*)att)[lane];
        m = fmaxf(fmaxf(s4.x, s4.y), fmaxf(s4.z, s4.w));
    } else {
        ATT2_STAMP(4, 0);
        m = sc;
    }
    // ---- softmax numerators (layers.rs:495-506): exp of this wave's own timesteps
    m = group_max_f32(m, 64);
    float ev = 0.0f;                                      // +0.0 past the context: leaves every partial sum unchanged
    if (64 * wave < np) {                                 // wave-uniform
        ev = q3_expf_wave(t < np ? sc - m : 0.0f, etab);
        ev = t < np ? ev : 0.0f;
    }
    att_e[t] = ev;
    if (np <= 128) att_e[t + 128] = 0.0f;                 // (waves 2, 3 are gone)
    __syncthreads();                                      // C
    KSTAMP_END(a);
}
