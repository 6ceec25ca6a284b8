// q3_engine.hip -- host side of libqwen3_hip.so: checkpoint loader, launch plan, hipGraph, C ABI.
//
// Mirrors, for the hot path only, what qwen3-inference does around `Transformer::forward`:
//   TransformerBuilder::build            models/mod.rs:55-73      -> q3_create
//   read_config / validate_config        configuration.rs:77-146  -> parse_header
//   MemoryMapper + load_weights          utils.rs, qwen3.rs:199-277 -> Engine::load (mmap -> one HBM blob)
//   TransformerBlockBuffers::new         qwen3.rs:414-445         -> device scratch + zero-filled f32 KV cache
//   Qwen3Transformer::forward            qwen3.rs:62-79           -> Engine::enqueue_forward (kernel chain)
// There is deliberately no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>
#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "../../include/qwen3_hip.h"
#include "q3_kernels.h"
#include "q3_batch.h"
#include "q3_sampler.h"

namespace {

using namespace q3;

thread_local char g_err[1024] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(Q3_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

constexpr int32_t kMagic = 0x616a6331;   // configuration.rs:8
constexpr int32_t kVersion = 1;          // configuration.rs:10
constexpr size_t kHeaderSize = 256;      // configuration.rs:12
constexpr size_t kConfigSize = 52;       // 13 x i32

int32_t rd_i32(const uint8_t* p) {
    return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}

// configuration.rs:77-146
int parse_header(const uint8_t* d, size_t len, q3_config* c) {
    if (!d || !c) return fail(Q3_ERR_ARG, "null argument");
    if (len < kConfigSize)
        return fail(Q3_ERR_FORMAT, "Insufficient data: need %zu bytes, have %zu remaining", kConfigSize, len);
    if (len < kHeaderSize)
        return fail(Q3_ERR_FORMAT, "Cannot skip %zu bytes: insufficient data", kHeaderSize - kConfigSize);
    const int32_t magic = rd_i32(d), version = rd_i32(d + 4);
    if (magic != kMagic)
        return fail(Q3_ERR_FORMAT, "Invalid model configuration: Invalid checkpoint magic number: expected %#x, got %#x",
                    kMagic, (unsigned)magic);
    if (version != kVersion)
        return fail(Q3_ERR_FORMAT, "Invalid model configuration: Unsupported checkpoint version: expected %d, got %d",
                    kVersion, version);
    c->architecture_id = rd_i32(d + 8);
    c->dim = rd_i32(d + 12);
    c->hidden_dim = rd_i32(d + 16);
    c->n_layers = rd_i32(d + 20);
    c->n_heads = rd_i32(d + 24);
    c->n_kv_heads = rd_i32(d + 28);
    c->vocab_size = rd_i32(d + 32);
    c->seq_len = rd_i32(d + 36);
    c->head_dim = rd_i32(d + 40);
    c->shared_classifier = rd_i32(d + 44) != 0;
    c->group_size = rd_i32(d + 48);
    const char* names[8] = {"architecture_id", "dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size", "seq_len", "head_dim"};
    const int32_t vals[8] = {c->architecture_id, c->dim, c->n_layers, c->n_heads, c->n_kv_heads, c->vocab_size, c->seq_len, c->head_dim};
    for (int i = 0; i < 8; ++i)
        if (vals[i] <= 0)
            return fail(Q3_ERR_FORMAT, "Invalid model configuration: Invalid %s: must be positive, got %d", names[i], vals[i]);
    return Q3_OK;
}

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// shapes the kernels cover (everything the reference README lists, plus small test shapes)
int check_supported(const q3_config& c) {
    const int G = c.group_size;
    if (G < 16 || G > 1024 || !is_pow2(G))
        return fail(Q3_ERR_UNSUPPORTED, "group_size %d unsupported: must be a power of two in [16,1024]", G);
    const int ahd = c.n_heads * c.head_dim;
    if (c.hidden_dim <= 0) return fail(Q3_ERR_FORMAT, "Invalid hidden_dim %d", c.hidden_dim);
    if (c.dim % G || ahd % G || c.hidden_dim % G)
        return fail(Q3_ERR_UNSUPPORTED, "dim/all_heads_dim/hidden_dim must be multiples of group_size %d", G);
    if (c.head_dim % 8 || c.head_dim > 256 || !is_pow2(c.head_dim))
        return fail(Q3_ERR_UNSUPPORTED, "head_dim %d unsupported: power of two in [8,256] required", c.head_dim);
    if (c.n_heads % c.n_kv_heads) return fail(Q3_ERR_UNSUPPORTED, "n_heads must be a multiple of n_kv_heads");
    if (c.dim > 16384 || c.hidden_dim > 65536 || ahd > 16384)
        return fail(Q3_ERR_UNSUPPORTED, "dimension too large for the LDS-staged activation");
    return Q3_OK;
}

struct QT {  // QuantizedTensor view (tensor.rs:5-8) in device memory
    const int8_t* q = nullptr;
    const float* s = nullptr;
};

enum Family { F_QKV = 0, F_ATTN, F_WO, F_W13, F_W2, F_LMHEAD, F_NEXT, F_COUNT };
const char* kFamilyNames[F_COUNT] = {"qkv", "attn", "wo", "w13", "w2", "lm_head", "next"};

typedef void (*GemvFn)(const GemvArgs);

struct Launch {
    Family fam;
    bool is_attn = false, is_next = false;
    bool is_kfold = false; // developer (Q3_KSTAMPS): k_kstamp_fold behind the token's last launch
    int scores_kvm = 0;    // attn_kind 1: > 0 = k_attn_scores_kv<scores_kvm> (K chunk shared by the heads of a kv head)
    int attn_kind = 0;     // 0: single-kernel attention, 1: k_attn_scores, 2: k_attn_out (long-context split),
                           // 3: k_attn_short (pos < 256, head_dim 64/128: K rows in registers, no LDS staging)
    unsigned grid_y = 1;
    GemvFn fn = nullptr;
    GemvArgs ga{};
    AttnArgs aa{};
    unsigned grid = 1;
    unsigned block = kWG;  // threads per workgroup (specialised GEMV shapes: 256 / 512 / 1024)
    size_t smem = 0;
};

}  // namespace

// GEMV kernel instantiations live in their own translation unit (q3_gemv_inst.hip: compiled in parallel with this one)
namespace q3inst {
typedef void (*GemvFn)(const q3::GemvArgs);
struct GemvCfg { int pro, epi, n, wgt, ept, ru, ju, pf; GemvFn fn; };
GemvFn gemv_pick(int pro, int epi, int G, int RU, int JU, int FIN, int PF);       // generic run-time-n kernels
const GemvCfg* find_cfg(int pro, int epi, int n, int G, int which, bool fast = false);              // shape-specialised kernels
}
using q3inst::GemvCfg;
using q3inst::find_cfg;

namespace {
template <int PRO, int EPI>
GemvFn pick(int G, int RU, int JU, int FIN = 0, int PF = 0) { return q3inst::gemv_pick(PRO, EPI, G, RU, JU, FIN, PF); }

int set_max_smem(const void* fn, size_t bytes) {
    if (bytes > 48 * 1024) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return Q3_OK;
}

// Environment variables.  The product library reads only the documented ones (include/qwen3_hip.h, "Environment"):
// env_int().  Everything else -- A/B switches between kernel forms, tile / workgroup overrides, ablation and timeline
// switches -- exists in the developer build only (make dev, -DQ3_DEV): dev_knob() is its default in the product.
int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}
#ifdef Q3_DEV
#define dev_knob(name, dflt) env_int(name, dflt)
#else
#define dev_knob(name, dflt) (dflt)
#endif
// activation requests ahead of the weight requests (GemvArgs::xfirst): 2 = workgroup barrier between them, 1 = wait for wave 0's
// block of x, 0 = program order only.  r03 A/B in reference-order mode (tools/gen_loop.py, Q3_STRICT=1): the barrier costs the QKV
// launch 0.15 us at every shape (0.6B: 1,572 -> 1,527 tok/s), gains W13 0.4 us at dim 1024 (+0.3 %) and nothing at 2560 / 4096;
// in Q3_FLAG_FAST mode (no exact sum in front of the quantizer) it is worth +6 % on the 0.6B shape.
// r05 re-sweep: only the 16-wave forms still carried the barrier (8B QKV / W1|W3): 1,895.7 with it vs 1,885.9 us per token without
// (three alternations) -- off everywhere in reference-order mode now; the developer switch keeps the form for the FAST-mode A/B.
// r06, the same question in Q3_FLAG_FAST mode (64-token device loop, two alternations, tok/s without / with the barrier): 0.6B
// 1,827 / 1,820 vs 1,815 / 1,817; 4B 741 / 742 vs 739 / 739; 8B 535 / 535 vs 537 / 540 -- the r03 "+6 %" is gone (the block loads
// of x are coalesced since then): off in both modes.
int xfirst_dflt(int wgt) { return wgt >= 1024 ? dev_knob("Q3_XFIRST_DEFAULT", 0) : 0; }


}  // namespace

struct BatchCtx;
struct q3_engine;
namespace { void batch_free(q3_engine* e); }

struct q3_engine {
    q3_config cfg{};
    uint32_t flags = 0;
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    // device memory
    uint8_t* d_blob = nullptr;
    size_t blob_bytes = 0;
    const float *rms_att = nullptr, *rms_ffn = nullptr, *rms_final = nullptr, *q_ln = nullptr, *k_ln = nullptr;
    QT tok, wcls;
    std::vector<QT> wq, wk, wv, wo, w1, w2, w3;
    float *d_x = nullptr, *d_q = nullptr, *d_kraw = nullptr, *d_xb = nullptr, *d_hb = nullptr, *d_logits = nullptr;
    float *d_tap = nullptr, *d_key = nullptr, *d_value = nullptr, *d_rope = nullptr, *d_att = nullptr;
    float* d_value_t = nullptr;                // transposed copy of the value cache, [L][kv_dim][seq_len]: what k_attn_out streams (long contexts)
    int8_t* d_xbq = nullptr;                   // attention output quantized by k_attn_short (operand of the PRO_PREQR Wo launch)
    float* d_xbs = nullptr;
    State* d_state = nullptr;
    int32_t* d_out_tokens = nullptr;
    int32_t* d_prompt = nullptr;               // chat-mode prefill: prompt token ids (capacity out_cap)
    int out_cap = 0;
    unsigned long long* d_stamps = nullptr;   // developer timeline (Q3_STAMPS=1) / kernel begin-end cells (Q3_KSTAMPS=1)
    unsigned long long* d_kacc = nullptr;     // Q3_KSTAMPS=1: per-launch duration / gap sums (k_kstamp_fold)
    unsigned long long* d_kslots = nullptr;   //   per-wave begin / end slots of every launch, [launch][kKstampSlots][2]
    unsigned long long* d_kcells = nullptr;   //   per-launch {begin, end} of the current token (k_kstamp_reduce)
    int* d_knslots = nullptr;                 //   live wave slots of every launch
    bool kstamps = false;
    unsigned long long* d_argmax_slots = nullptr;
    unsigned long long* d_next_cell = nullptr;   // {argmax cell, ticket counter} of the classifier launch with k_next folded in
    int n_argmax_slots = 0;
    // pinned host staging
    float* h_logits = nullptr;
    State* h_state = nullptr;
    int32_t* h_tokens = nullptr;
    // launch plans: `plan` (one attention kernel per layer) for short contexts, `plan_long` (scores + out
    // kernels over many workgroups) once pos >= split_pos; the host knows pos of every forward it enqueues
    std::vector<Launch> plan, plan_long;
    hipGraph_t graph = nullptr, graph_long = nullptr;
    hipGraphExec_t graph_exec = nullptr, graph_long_exec = nullptr;
    // q3_forward as ONE graph launch: state upload (pinned h_state) -> the token's kernels -> logits download (pinned h_logits)
    hipGraph_t graph_fwd = nullptr, graph_fwd_long = nullptr;
    hipGraphExec_t graph_fwd_exec = nullptr, graph_fwd_long_exec = nullptr;
    // position ranges below 64 (k_attn_short2 only; developer build, Q3_ATT_RANGES=1): one more captured pair per range, whose
    // attention launches request their key / value rows at kernel entry (AttnArgs::row_steps) -- the host knows the position of
    // every forward it enqueues.  Measured in round 5 (three alternations, 0.6B device loop): 577.8-582.8 us per token against
    // 569.1-576.2 without -- the early row requests stand in the CU's address path in front of the raw q / k values the norm
    // chains wait for.  Off; the product captures no range graphs.
    static constexpr int kNRange = 6;
    static constexpr int kRangeSteps[kNRange] = {1, 2, 3, 4, 6, 8};      // 8-row steps: positions < 8, 16, 24, 32, 48, 64
    hipGraph_t graph_rng[kNRange] = {}, graph_fwd_rng[kNRange] = {};
    hipGraphExec_t graph_rng_exec[kNRange] = {}, graph_fwd_rng_exec[kNRange] = {};
    bool ranges = false;
    int range_of(size_t pos) const {
        if (!ranges) return -1;
        for (int i = 0; i < kNRange; ++i)
            if (pos < (size_t)(8 * kRangeSteps[i])) return i;
        return -1;
    }
    float* d_att_priv = nullptr;
    int cmax_stride = 0;
    int att_stride = 0;
    int split_pos = 256;
    // the transposed value cache exists iff the long-context output kernel will read it: reference-order mode (k_attn_out reads it
    // only when strict), a context that reaches split_pos, rows of whole float4, and the caller did not opt out
    bool wants_value_t() const {
        return !(flags & (Q3_FLAG_FAST | Q3_FLAG_NO_VALUE_T)) && (cfg.seq_len % 4) == 0 && (int64_t)cfg.seq_len > (int64_t)split_pos &&
               dev_knob("Q3_VALUE_T", 1) != 0;
    }
    BatchCtx* batch = nullptr;                 // batched decode state (q3_batch_init), see q3_batch_host.inc
    // device-side Sampler (q3_sampler_set): temperature > 0 makes every token draw go through k_sample
    bool sampling = false;
    SamplerState* d_sampler = nullptr;
    float* d_samp_hist = nullptr;        // pipelined draw: mass histogram, per-workgroup candidate counts
    int* d_samp_counts = nullptr;
    float *d_probs = nullptr, *d_sp = nullptr;
    unsigned long long* d_keys = nullptr;
    size_t keys_n2 = 0;
    SampleArgs sargs{};
    int enqueue_sample();

    int load(const char* path, uint32_t ctx_len);
    int build_plan();
    int capture();
    int enqueue_forward(bool eager, size_t pos, bool draw = true);
    int set_state(size_t token, size_t pos);
    void release();
};

namespace {

template <bool P_LDS>
void launch_attn_out_t(const AttnArgs& aa, unsigned gx, unsigned gy, size_t smem, hipStream_t st) {
    switch (aa.slice_w) {
        case 8: hipLaunchKernelGGL((k_attn_out<8, P_LDS>), dim3(gx, gy), dim3(kAoThreads), smem, st, aa); break;
        case 16: hipLaunchKernelGGL((k_attn_out<16, P_LDS>), dim3(gx, gy), dim3(kAoThreads), smem, st, aa); break;
        case 32: hipLaunchKernelGGL((k_attn_out<32, P_LDS>), dim3(gx, gy), dim3(kAoThreads), smem, st, aa); break;
        default: hipLaunchKernelGGL((k_attn_out<0, P_LDS>), dim3(gx, gy), dim3(kAoThreads), smem, st, aa); break;
    }
}
void launch_attn_out(const AttnArgs& aa, unsigned gx, unsigned gy, size_t smem, hipStream_t st) {
    if (attn_out_p_in_lds(aa.seq_len)) launch_attn_out_t<true>(aa, gx, gy, smem, st);
    else launch_attn_out_t<false>(aa, gx, gy, smem, st);
}
template <bool P_LDS>
int set_attn_out_smem_t(size_t bytes) {
    int rc;
    if ((rc = set_max_smem((const void*)k_attn_out<8, P_LDS>, bytes)) || (rc = set_max_smem((const void*)k_attn_out<16, P_LDS>, bytes)) ||
        (rc = set_max_smem((const void*)k_attn_out<32, P_LDS>, bytes)) || (rc = set_max_smem((const void*)k_attn_out<0, P_LDS>, bytes))) return rc;
    return Q3_OK;
}
int set_attn_out_smem(size_t bytes) {
    int rc;
    if ((rc = set_attn_out_smem_t<true>(bytes)) || (rc = set_attn_out_smem_t<false>(bytes))) return rc;
    return Q3_OK;
}


// k_attn_scores_kv applies to head_dim 128 with 2 or 4 query heads per kv head (every listed Qwen3 size up to 8B)
static int scores_kvm_for(int hd, int n_heads, int n_kv_heads) {
    const int kv_mul = n_heads / n_kv_heads;
    if (hd != kSgHd || (kv_mul != 2 && kv_mul != 4) || !dev_knob("Q3_ATT_SCORES_KV", 1)) return 0;
    return kv_mul;
}
struct ScoresShape { int kvm; unsigned gx, gy; size_t smem; };
static ScoresShape scores_shape(int hd, int n_heads, int n_kv_heads, int S) {
    ScoresShape r;
    r.kvm = scores_kvm_for(hd, n_heads, n_kv_heads);
    const int tch = r.kvm == 4 ? sg_tch<4>() : r.kvm == 2 ? sg_tch<2>() : attn_tch(hd);
    r.gx = (unsigned)(r.kvm ? n_kv_heads : n_heads);
    r.gy = (unsigned)((S + tch - 1) / tch);
    r.smem = r.kvm == 4 ? attn_scores_kv_smem_bytes<4>() : r.kvm == 2 ? attn_scores_kv_smem_bytes<2>() : attn_scores_smem_bytes(hd);
    return r;
}
static int set_attn_scores_smem(const ScoresShape& sh) {
    if (sh.kvm == 4) return set_max_smem((const void*)k_attn_scores_kv<4>, sh.smem);
    if (sh.kvm == 2) return set_max_smem((const void*)k_attn_scores_kv<2>, sh.smem);
    return set_max_smem((const void*)k_attn_scores, sh.smem);
}
static void launch_attn_scores(const AttnArgs& a, int kvm, unsigned gx, unsigned gy, size_t smem, hipStream_t st) {
    if (kvm == 4) hipLaunchKernelGGL(k_attn_scores_kv<4>, dim3(gx, gy), dim3(kWG), smem, st, a);
    else if (kvm == 2) hipLaunchKernelGGL(k_attn_scores_kv<2>, dim3(gx, gy), dim3(kWG), smem, st, a);
    else hipLaunchKernelGGL(k_attn_scores, dim3(gx, gy), dim3(kWG), smem, st, a);
}

// Short-context attention (pos < 256): k_attn_short2 (round 5: coalesced key / value staging, product tile, 8 waves) for
// head_dim 128 on caches of a whole number of 8-row steps; k_attn_short (head_dim 64, odd test contexts) otherwise.
static bool attn_short2_ok(const AttnArgs& a) { return a.hd == 128 && a.seq_len >= 8 && (a.seq_len % 8) == 0 && a.att_short_form != 1; }
static int set_attn_short_smem(const AttnArgs& a) {
    if (!attn_short2_ok(a)) return Q3_OK;
    return set_max_smem((const void*)k_attn_short2<128>, attn_short2_smem_bytes(128, kS2MaxT));
}
static void launch_attn_short(const AttnArgs& a0, unsigned n_heads, hipStream_t st, int row_steps = 0) {
    AttnArgs a = a0;
    if (attn_short2_ok(a)) {
        a.row_steps = row_steps;          // > 0: a position range below 64 -- rows requested at entry, one workgroup per head
        hipLaunchKernelGGL(k_attn_short2<128>, dim3(n_heads, row_steps > 0 ? 1 : 2), dim3(kS2Threads), attn_short2_smem_bytes(128, kS2MaxT), st, a);
    }
    else if (a.hd == 128) hipLaunchKernelGGL(k_attn_short<128>, dim3(n_heads), dim3(kWG), 0, st, a);
    else hipLaunchKernelGGL(k_attn_short<64>, dim3(n_heads), dim3(kWG), 0, st, a);
}

// no_next: the classifier launch without the folded bookkeeping (q3_profile replays launches without advancing the state)
void launch_one(const Launch& L, q3_engine* e, bool no_next = false, int row_steps = 0) {
    if (no_next && L.fam == F_LMHEAD && L.ga.next_cell != nullptr) {
        Launch M = L;
        M.ga.next_cell = nullptr;
        hipLaunchKernelGGL(M.fn, dim3(M.grid), dim3(M.block), M.smem, e->stream, M.ga);
        return;
    }
#ifdef Q3_DEV
    if (L.is_kfold) {
        const int nl = (int)e->plan.size() - 1;
        hipLaunchKernelGGL(k_kstamp_reduce, dim3((unsigned)nl), dim3(256), 0, e->stream, e->d_kslots, e->d_knslots, e->d_kcells);
        hipLaunchKernelGGL(k_kstamp_fold, dim3(1), dim3(256), 0, e->stream, e->d_kcells, nl, e->d_kacc);
        return;
    }
#endif
    if (L.is_attn) {
        if (L.attn_kind == 1) launch_attn_scores(L.aa, L.scores_kvm, L.grid, L.grid_y, L.smem, e->stream);
        else if (L.attn_kind == 2) launch_attn_out(L.aa, L.grid, L.grid_y, L.smem, e->stream);
        else if (L.attn_kind == 3) launch_attn_short(L.aa, L.grid, e->stream, row_steps);
        else hipLaunchKernelGGL(k_attn, dim3(L.grid), dim3(kWG), L.smem, e->stream, L.aa);
    } else if (L.is_next) {
        hipLaunchKernelGGL(k_next, dim3(1), dim3(kWG), 0, e->stream, e->d_state, e->d_argmax_slots, e->n_argmax_slots,
                           e->d_out_tokens, e->out_cap, e->d_prompt);
    } else {
        hipLaunchKernelGGL(L.fn, dim3(L.grid), dim3(L.block), L.smem, e->stream, L.ga);
    }
}

// Tile shape + grid for one GEMV launch.  units = output rows (SwiGLU: hidden units, each 2 weight rows).
// JU follows the row length (1 KiB chunks per row); RU is the largest row count per wave batch that keeps
// every wave of the grid busy and minimises max rows per wave; larger kernels grid-stride over batches.
struct GemvShape { int RU, JU; unsigned grid; int FIN, PF; };
GemvShape plan_gemv(int units, int n, int G, bool swiglu, int row_align, int n_cu, int wg_per_cu, bool allow_fin = true) {
    GemvShape g;
    g.FIN = 0;
    g.PF = 0;
    // launches that stream >= 16 MB are bandwidth- rather than latency-bound: give them a second workgroup per CU
    const size_t launch_bytes = (size_t)units * (swiglu ? 2 : 1) * (size_t)n;
    if (launch_bytes >= (16u << 20) && wg_per_cu < 2) wg_per_cu = 2;
    // chunks (1 KiB wave-loads) per tile row: the largest JU <= 4 that tiles the row exactly (every lane of every load
    // inside the row: the kernel's clamp-free fast path); rows that are not a whole number of wave-loads (2560, 9728)
    // take the JU that wastes the fewest clamped loads
    const int nchunks = n / 16, nj = (n + 1023) / 1024;
    g.JU = 0;
    for (int ju = 4; ju >= 1; --ju)
        if (nchunks % (64 * ju) == 0) { g.JU = ju; break; }
    if (g.JU == 0) {
        int best_waste = 1 << 30;
        for (int ju = 4; ju >= 1; --ju) {
            const int waste = ((nj + ju - 1) / ju) * ju * 64 - nchunks;
            if (waste < best_waste) { best_waste = waste; g.JU = ju; }
        }
    }
    const int force_ju = dev_knob("Q3_GEMV_JU", 0);
    if (force_ju >= 1 && force_ju <= 4) g.JU = force_ju;
    const int ru_max = 8 / g.JU, ru_min = swiglu ? 2 : 1;
    const int waves = n_cu * wg_per_cu * kWaves;
    // rows that are a whole number of tiles fold the group terms in registers (k_gemv FIN = 1)
    if (allow_fin && dev_knob("Q3_GEMV_FIN", 1) && G == 64 && (nchunks % 4) == 0 &&
        launch_bytes < ((size_t)dev_knob("Q3_GEMV_FIN_MAXMB", 1 << 20) << 20)) g.FIN = 1;
    int best_ru = ru_min;
    long best_cost = -1;
    for (int ru = ru_max; ru >= ru_min; ru >>= 1) {
        if (G != 64 && ru != ru_min) continue;
        const int hu = swiglu ? ru / 2 : ru;
        if (row_align > 1 && row_align % hu) continue;   // QKV: batches must not straddle q|k|v segments
        const long nb = (units + hu - 1) / hu;
        const long per_wave = (nb + waves - 1) / waves;
        // latency-bound launches (<= 2 batches per wave): fewest sequential rows per wave wins;
        // streaming launches: the largest tile (most bytes in flight per wave) wins.
        const long cost = per_wave <= 2 ? per_wave * ru * 1000 + (ru_max - ru) : 1000000 + (ru_max - ru);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_ru = ru; }
    }
    g.RU = best_ru;
    const int hu = swiglu ? g.RU / 2 : g.RU;
    const long nb = (units + hu - 1) / hu;
    long grid = (nb + kWaves - 1) / kWaves;
    if (grid > (long)n_cu * wg_per_cu) grid = (long)n_cu * wg_per_cu;
    if (grid < 1) grid = 1;
    g.grid = (unsigned)grid;
    // streaming launches (every wave owns at least two tiles): request the second tile before the prologue too
    const long njt_h = (nj + g.JU - 1) / g.JU;
    if (dev_knob("Q3_GEMV_PF", 1) && G == 64 && nb * njt_h >= 2 * grid * kWaves && (g.FIN || !allow_fin)) g.PF = 1;
    return g;
}

// grid / block / LDS of a specialised launch: one row batch (hu rows or hidden units) per wave and round; at least one
// workgroup per CU as long as there are batches for it (fewer batches than waves: see k_gemv's wave numbering)
void apply_cfg(Launch& Ln, GemvArgs& a, const GemvCfg& c, int units, int n_cu) {
    const int waves = c.wgt / 64;
    const int hu = (c.epi == EPI_SWIGLU) ? c.ru / 2 : c.ru;
    const long nb = ((long)units + hu - 1) / hu;
    const int wg_per_cu = c.wgt >= 1024 ? 1 : (c.wgt >= 512 ? 2 : 4);
    long grid = (nb + waves - 1) / waves;
    const long lower = nb < n_cu ? nb : n_cu;
    if (grid < lower) grid = lower;
    if (grid > (long)n_cu * wg_per_cu) grid = (long)n_cu * wg_per_cu;
    a.vr = c.ru;
    Ln.fn = c.fn;
    Ln.grid = (unsigned)grid;
    Ln.block = (unsigned)c.wgt;
    // f32 staging only for the long vectors' block transpose (wave 0 sums out of registers)
    const bool stage = (c.pro == PRO_NORM || c.pro == PRO_EMBED_NORM) && c.n >= 1024;
    Ln.smem = gemv_smem_bytes(c.n, 64, c.ru, stage, waves, true);
}

}  // namespace

void q3_engine::release() {
    batch_free(this);
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (graph) (void)hipGraphDestroy(graph);
    if (graph_long_exec) (void)hipGraphExecDestroy(graph_long_exec);
    if (graph_long) (void)hipGraphDestroy(graph_long);
    if (graph_fwd_exec) (void)hipGraphExecDestroy(graph_fwd_exec);
    if (graph_fwd) (void)hipGraphDestroy(graph_fwd);
    if (graph_fwd_long_exec) (void)hipGraphExecDestroy(graph_fwd_long_exec);
    for (int i = 0; i < kNRange; ++i) {
        if (graph_rng_exec[i]) (void)hipGraphExecDestroy(graph_rng_exec[i]);
        if (graph_rng[i]) (void)hipGraphDestroy(graph_rng[i]);
        if (graph_fwd_rng_exec[i]) (void)hipGraphExecDestroy(graph_fwd_rng_exec[i]);
        if (graph_fwd_rng[i]) (void)hipGraphDestroy(graph_fwd_rng[i]);
    }
    if (graph_fwd_long) (void)hipGraphDestroy(graph_fwd_long);
    void* dptrs[] = {d_value_t, d_kacc, d_kslots, d_kcells, d_knslots, d_next_cell, d_xbq, d_xbs, d_samp_hist, d_samp_counts, d_sampler, d_probs, d_sp, d_keys, d_prompt, d_att_priv, d_argmax_slots, d_stamps, d_blob, d_x, d_q, d_kraw, d_xb, d_hb, d_logits, d_tap, d_key, d_value, d_rope, d_att, d_state, d_out_tokens};
    for (void* p : dptrs)
        if (p) (void)hipFree(p);
    if (h_logits) (void)hipHostFree(h_logits);
    if (h_state) (void)hipHostFree(h_state);
    if (h_tokens) (void)hipHostFree(h_tokens);
    if (stream) (void)hipStreamDestroy(stream);
}

// utils.rs MemoryMapper + qwen3.rs:199-277 load_weights: walk the file with a cursor, upload it once.
int q3_engine::load(const char* path, uint32_t ctx_len) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(Q3_ERR_IO, "Failed to open checkpoint: %s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) {
        close(fd);
        return fail(Q3_ERR_IO, "Failed to create memory mapping: %s", path);
    }
    const size_t flen = (size_t)st.st_size;
    void* map = mmap(nullptr, flen, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return fail(Q3_ERR_IO, "Failed to create memory mapping: %s", path);
    struct Unmap {
        void* p;
        size_t n;
        ~Unmap() { munmap(p, n); }
    } unmap{map, flen};
    const uint8_t* base = (const uint8_t*)map;

    int rc = parse_header(base, flen, &cfg);
    if (rc) return rc;
    if (ctx_len != 0 && (int64_t)ctx_len < (int64_t)cfg.seq_len) cfg.seq_len = (int32_t)ctx_len;  // models/mod.rs:65-67
    if (cfg.architecture_id != 1) return fail(Q3_ERR_FORMAT, "Unknown architecture_id: %d", cfg.architecture_id);
    if ((rc = check_supported(cfg))) return rc;

    const size_t dim = cfg.dim, L = cfg.n_layers, hd = cfg.head_dim, V = cfg.vocab_size, H = cfg.hidden_dim;
    const size_t G = cfg.group_size, ahd = (size_t)cfg.n_heads * hd, kvd = (size_t)cfg.n_kv_heads * hd;

    // cursor walk (offsets only), mirroring get_f32_slice / get_bytes bounds errors (utils.rs:21-56)
    size_t off = kHeaderSize;
    bool ok = true;
    auto take = [&](size_t bytes, const char* what) -> size_t {
        if (!ok) return 0;
        if (off + bytes > flen) {
            fail(Q3_ERR_FORMAT, "Failed to read %s: Insufficient data: need %zu bytes, have %zu remaining", what, bytes, flen - off);
            ok = false;
            return 0;
        }
        const size_t o = off;
        off += bytes;
        return o;
    };
    struct QOff { size_t q, s; };
    auto take_q = [&](size_t count, size_t size_each, std::vector<QOff>& out) {   // models/mod.rs:83-110
        for (size_t i = 0; i < count; ++i) {
            QOff o;
            o.q = take(size_each, "quantized tensor data");
            o.s = take(4 * (size_each / G), "scale factors");
            out.push_back(o);
        }
    };
    const size_t o_rms_att = take(4 * L * dim, "attention normalization weights");
    const size_t o_rms_ffn = take(4 * L * dim, "FFN normalization weights");
    const size_t o_rms_final = take(4 * dim, "final normalization weights");
    const size_t o_q_ln = take(4 * L * hd, "query layer norm weights");
    const size_t o_k_ln = take(4 * L * hd, "key layer norm weights");
    std::vector<QOff> otok, owq, owk, owv, owo, ow1, ow2, ow3, ocls;
    take_q(1, V * dim, otok);
    take_q(L, dim * ahd, owq);
    take_q(L, dim * kvd, owk);
    take_q(L, dim * kvd, owv);
    take_q(L, ahd * dim, owo);
    take_q(L, dim * H, ow1);
    take_q(L, H * dim, ow2);
    take_q(L, dim * H, ow3);
    if (!cfg.shared_classifier) take_q(1, dim * V, ocls);
    if (!ok) return Q3_ERR_FORMAT;

    // every tensor must start 16-byte aligned for dwordx4 loads (true for all listed models: the header
    // is 256 B and every section size is a multiple of 16)
    auto aligned = [](size_t o) { return (o & 15) == 0; };
    bool all_aligned = aligned(o_rms_att) && aligned(o_rms_ffn) && aligned(o_rms_final) && aligned(o_q_ln) && aligned(o_k_ln);
    for (auto* v : {&otok, &owq, &owk, &owv, &owo, &ow1, &ow2, &ow3, &ocls})
        for (auto& o : *v) all_aligned = all_aligned && aligned(o.q) && aligned(o.s);
    if (!all_aligned) return fail(Q3_ERR_UNSUPPORTED, "checkpoint sections are not 16-byte aligned");

    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

    // the whole file becomes one resident HBM blob, layout unchanged (zero repack)
    blob_bytes = off;
    HIP_TRY(hipMalloc((void**)&d_blob, blob_bytes));
    HIP_TRY(hipMemcpy(d_blob, base, blob_bytes, hipMemcpyHostToDevice));
    auto fp = [&](size_t o) { return (const float*)(d_blob + o); };
    auto qt = [&](const QOff& o) {
        QT t;
        t.q = (const int8_t*)(d_blob + o.q);
        t.s = (const float*)(d_blob + o.s);
        return t;
    };
    rms_att = fp(o_rms_att);
    rms_ffn = fp(o_rms_ffn);
    rms_final = fp(o_rms_final);
    q_ln = fp(o_q_ln);
    k_ln = fp(o_k_ln);
    tok = qt(otok[0]);
    for (size_t l = 0; l < L; ++l) {
        wq.push_back(qt(owq[l]));
        wk.push_back(qt(owk[l]));
        wv.push_back(qt(owv[l]));
        wo.push_back(qt(owo[l]));
        w1.push_back(qt(ow1[l]));
        w2.push_back(qt(ow2[l]));
        w3.push_back(qt(ow3[l]));
    }
    wcls = cfg.shared_classifier ? tok : qt(ocls[0]);   // qwen3.rs:252-253

    // run state (qwen3.rs:414-445); KV cache zero-filled: generate mode attends over never-written rows
    const size_t S = cfg.seq_len;
    const size_t kv_elems = L * S * kvd;
    HIP_TRY(hipMalloc((void**)&d_x, 4 * dim));
    HIP_TRY(hipMalloc((void**)&d_q, 4 * ahd));
    HIP_TRY(hipMalloc((void**)&d_kraw, 4 * kvd));
    HIP_TRY(hipMalloc((void**)&d_xb, 4 * ahd));
    HIP_TRY(hipMalloc((void**)&d_hb, 4 * H));
    HIP_TRY(hipMalloc((void**)&d_logits, 4 * V));
    HIP_TRY(hipMalloc((void**)&d_tap, 4 * dim));
    HIP_TRY(hipMalloc((void**)&d_key, 4 * kv_elems));
    HIP_TRY(hipMalloc((void**)&d_value, 4 * kv_elems));
    HIP_TRY(hipMemset(d_key, 0, 4 * kv_elems));
    HIP_TRY(hipMemset(d_value, 0, 4 * kv_elems));
    // the long-context plan streams a transposed copy of the value rows (k_attn_out, strict mode only; contexts that can reach the
    // split position only -- split_pos is set here, once, and build_plan uses the same value)
    split_pos = dev_knob("Q3_ATT_SPLIT_POS", 256);
    if (wants_value_t()) {
        HIP_TRY(hipMalloc((void**)&d_value_t, 4 * kv_elems));
        HIP_TRY(hipMemset(d_value_t, 0, 4 * kv_elems));
    }
    HIP_TRY(hipMemset(d_x, 0, 4 * dim));
    HIP_TRY(hipMalloc((void**)&d_state, sizeof(State)));
    HIP_TRY(hipMemset(d_state, 0, sizeof(State)));
    out_cap = (int)S;
    HIP_TRY(hipMalloc((void**)&d_out_tokens, 4 * (size_t)out_cap));
    HIP_TRY(hipMemset(d_out_tokens, 0, 4 * (size_t)out_cap));
    HIP_TRY(hipMalloc((void**)&d_prompt, 4 * (size_t)out_cap));
    HIP_TRY(hipMemset(d_prompt, 0, 4 * (size_t)out_cap));
    HIP_TRY(hipHostMalloc((void**)&h_logits, 4 * V, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&h_state, sizeof(State), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&h_tokens, 4 * (size_t)out_cap, hipHostMallocDefault));

    // RoPE table with the host libm, exactly RoPE::compute_freqs (layers.rs:161-171)
    {
        const size_t half = hd / 2;
        std::vector<float> tab(S * hd);
        std::vector<float> freq(half);
        for (size_t i = 0; i < half; ++i) freq[i] = powf(1e6f, -((float)i) / (float)half);
        for (size_t p = 0; p < S; ++p)
            for (size_t i = 0; i < half; ++i) {
                const float angle = (float)p * freq[i];
                tab[p * hd + 2 * i] = cosf(angle);
                tab[p * hd + 2 * i + 1] = sinf(angle);
            }
        HIP_TRY(hipMalloc((void**)&d_rope, 4 * tab.size()));
        HIP_TRY(hipMemcpy(d_rope, tab.data(), 4 * tab.size(), hipMemcpyHostToDevice));
    }
    return Q3_OK;
}

// The per-token kernel chain, in the order of TransformerBlock::forward (qwen3.rs:131-176)
int q3_engine::build_plan() {
    const int dim = cfg.dim, L = cfg.n_layers, hd = cfg.head_dim, V = cfg.vocab_size, H = cfg.hidden_dim;
    const int G = cfg.group_size, ahd = cfg.n_heads * hd, kvd = cfg.n_kv_heads * hd, S = cfg.seq_len;
    const int strict = (flags & Q3_FLAG_FAST) ? 0 : 1;
    const bool fast_fold = !strict && dev_knob("Q3_FAST_FOLD", 1) != 0;   // tolerance mode: tree fold of the GEMV group terms too
    const int small_cap = dev_knob("Q3_WG_PER_CU_SMALL", 2);   // two workgroups per CU: half the rows (and fold chains) per wave
    const int big_cap = dev_knob("Q3_WG_PER_CU_LMHEAD", 2);   // all workgroups resident at once (the NORM prologue keeps ~190 VGPRs live)
    const int att_lds_max = dev_knob("Q3_ATT_LDS_MAX", 4096);

    att_stride = (S + 255) & ~255;
    const int slice_w = attn_slice_w(hd, cfg.n_heads, n_cu);
    const int nsl = hd / slice_w;
    const bool use_att_global = S > att_lds_max;
    cmax_stride = (int)((((size_t)S + 63) / 64 + 63) & ~(size_t)63);
    HIP_TRY(hipMalloc((void**)&d_att, 4 * (size_t)cfg.n_heads * ((size_t)att_stride + cmax_stride)));   // score rows, then their 64-block maxima
    HIP_TRY(hipMalloc((void**)&d_att_priv, 4 * (size_t)cfg.n_heads * nsl * att_stride));
    kstamps = dev_knob("Q3_KSTAMPS", 0) != 0;
    if (dev_knob("Q3_STAMPS", 0) || kstamps) {
        HIP_TRY(hipMalloc((void**)&d_stamps, 8 * 16 * (size_t)(5 * L + 4)));
        HIP_TRY(hipMemset(d_stamps, 0, 8 * 16 * (size_t)(5 * L + 4)));
    }
#ifdef Q3_DEV
    if (kstamps) {
        const size_t nl = (size_t)(5 * L + 4);
        HIP_TRY(hipMalloc((void**)&d_kslots, 16 * (size_t)kKstampSlots * nl));
        HIP_TRY(hipMemset(d_kslots, 0, 16 * (size_t)kKstampSlots * nl));
        HIP_TRY(hipMalloc((void**)&d_kcells, 16 * nl));
        HIP_TRY(hipMalloc((void**)&d_knslots, 4 * nl));
        HIP_TRY(hipMemset(d_knslots, 0, 4 * nl));
        HIP_TRY(hipMalloc((void**)&d_kacc, 8 * (2 + 2 * nl)));
        HIP_TRY(hipMemset(d_kacc, 0, 8 * (2 + 2 * nl)));
    }
#endif

    auto base_args = [&](int n) {
        GemvArgs a{};
        a.n = n;
        a.group = G;
        a.strict = strict;
        a.debug = dev_knob("Q3_ABLATE", 0) | (kstamps ? 64 : 0);
        a.st = d_state;
        a.seq_len = S;
        return a;
    };
    int rc;
    const GemvCfg* wo_preq = nullptr;
    std::vector<Launch> wo_long;       // the long-context plan's Wo launches (k_attn_out emits no quantized operand)
    HIP_TRY(hipMalloc((void**)&d_xbq, (size_t)ahd));
    HIP_TRY(hipMalloc((void**)&d_xbs, 4 * (size_t)(ahd / G)));
    const int alias0 = dev_knob("Q3_DEBUG_ALIAS_LAYER0", 0);   // experiment: every layer streams layer 0's weights
    for (int l = 0; l < L; ++l) {
        const size_t kv_off = (size_t)l * S * kvd;
        const int lw = alias0 ? 0 : l;
        {   // xb = RMSNorm_att(x); xq = quantize(xb); q,k,v = W{q,k,v} xq         qwen3.rs:134-136, layers.rs:334-337
            Launch Ln;
            Ln.fam = F_QKV;
            GemvArgs a = base_args(dim);
            a.seg[0] = Seg{wq[lw].q, wq[lw].s, d_q, ahd, 0};
            a.seg[1] = Seg{wk[lw].q, wk[lw].s, d_kraw, kvd, 0};
            a.seg[2] = Seg{wv[lw].q, wv[lw].s, d_value + kv_off, kvd, kvd};
            a.v_t = d_value_t ? d_value_t + kv_off : nullptr;
            a.total_rows = ahd + 2 * kvd;
            for (int k = 0; k < 2; ++k) {
                a.qkv_dw[k] = (const char*)a.seg[k + 1].wq - (const char*)a.seg[k].wq;
                a.qkv_ds[k] = (const char*)a.seg[k + 1].ws - (const char*)a.seg[k].ws;
                a.qkv_do[k] = (const char*)a.seg[k + 1].out - (const char*)a.seg[k].out;
            }
            a.norm_w = rms_att + (size_t)l * dim;
            a.in = d_x;
            const GemvCfg* cfg = find_cfg(l == 0 ? PRO_EMBED_NORM : PRO_NORM, EPI_QKV, dim, G, dev_knob("Q3_CFG_QKV", 0), fast_fold);
            if (cfg && (hd % cfg->ru) != 0) cfg = nullptr;           // batches must not straddle the q|k|v segments
            if (l == 0) {
                a.emb_q = tok.q;
                a.emb_s = tok.s;
                a.x_out = d_x;
            }
            if (cfg) {
                a.xfirst = dev_knob("Q3_XFIRST", (flags & Q3_FLAG_FAST) ? xfirst_dflt(cfg->wgt) : 0);
                apply_cfg(Ln, a, *cfg, a.total_rows, n_cu);
            } else {
                const GemvShape gs = plan_gemv(a.total_rows, dim, G, false, hd, n_cu, small_cap);
                if (l == 0) Ln.fn = pick<PRO_EMBED_NORM, EPI_QKV>(G, gs.RU, gs.JU, gs.FIN, gs.PF);
                else Ln.fn = pick<PRO_NORM, EPI_QKV>(G, gs.RU, gs.JU, gs.FIN, gs.PF);
                a.vr = gs.RU;
                Ln.grid = gs.grid;
                Ln.smem = gemv_smem_bytes(dim, G, a.vr, true);
            }
            Ln.ga = a;
            if (!Ln.fn) return fail(Q3_ERR_UNSUPPORTED, "no kernel instantiated for this tile shape");
            if ((rc = set_max_smem((const void*)Ln.fn, Ln.smem))) return rc;
            if (d_stamps) { Ln.ga.stamps = kstamps ? d_kslots + 2 * (size_t)kKstampSlots * plan.size() : d_stamps + 16 * plan.size(); Ln.ga.stamp_block = dev_knob("Q3_STAMP_BLOCK", 7); }
            plan.push_back(Ln);
        }
        {   // QK-norm + RoPE + attention                                        layers.rs:346-419
            Launch Ln;
            Ln.fam = F_ATTN;
            Ln.is_attn = true;
            AttnArgs a{};
            a.q = d_q;
            a.key_cache = d_key + kv_off;
            a.k_raw = d_kraw;
            a.value_cache = d_value + kv_off;
            a.value_t = d_value_t ? d_value_t + kv_off : nullptr;
            a.q_norm_w = q_ln + (size_t)l * hd;
            a.k_norm_w = k_ln + (size_t)l * hd;
            a.rope = d_rope;
            a.xb = d_xb;
            a.att_global = use_att_global ? d_att : nullptr;
            a.st = d_state;
            a.pos_override = -1;
            attn_set_heads(a, cfg.n_heads, cfg.n_kv_heads);
            a.hd = hd;
            a.seq_len = S;
            a.strict = strict;
            a.write_q = 0;
            a.debug = dev_knob("Q3_ABLATE", 0) | (kstamps ? 64 : 0);
            a.stamps = d_stamps ? (kstamps ? d_kslots + 2 * (size_t)kKstampSlots * plan.size() : d_stamps + 16 * plan.size()) : nullptr;
            Ln.aa = a;
            Ln.grid = (unsigned)cfg.n_heads;
            Ln.smem = attn_smem_bytes(hd, use_att_global ? 0 : S);
            if ((rc = set_max_smem((const void*)k_attn, Ln.smem))) return rc;
            // the short plan only ever runs at pos < split_pos
            if ((hd == 64 || hd == 128) && split_pos <= kShortMaxT && dev_knob("Q3_ATT_SHORT", 1)) {
                Ln.attn_kind = 3;
                Ln.aa.att_short_form = dev_knob("Q3_ATT_SHORT", 1) == 2 ? 1 : 0;      // 2: the round 2-4 kernel (A/B)
                if ((rc = set_attn_short_smem(Ln.aa))) return rc;
            }
            // k_attn_short can hand Wo its operand quantized (qwen3.rs:152 fused into the attention epilogue)
            wo_preq = Ln.attn_kind == 3 ? find_cfg(PRO_PREQR, EPI_RESID, ahd, G, dev_knob("Q3_CFG_WO", 0), fast_fold) : nullptr;
            if (wo_preq) {
                Ln.aa.xbq = d_xbq;
                Ln.aa.xbs = d_xbs;
                Ln.aa.xb_group = G;
            }
            plan.push_back(Ln);
        }
        {   // xq = quantize(xb); x += Wo xq                                      qwen3.rs:152-156
            Launch Ln;
            Ln.fam = F_WO;
            GemvArgs a = base_args(ahd);
            a.seg[0] = Seg{wo[lw].q, wo[lw].s, d_x, dim, 0};
            a.total_rows = dim;
            a.in = d_xb;
            a.pre_q = d_xbq;
            a.pre_s = d_xbs;
            // quantize-in-prologue form: the long-context plan always, the short plan when attention emits no int8
            Launch Lq = Ln;
            GemvArgs aq = a;
            if (const GemvCfg* cq = find_cfg(PRO_QUANT, EPI_RESID, ahd, G, dev_knob("Q3_CFG_WO_LONG", 0), fast_fold)) {
                apply_cfg(Lq, aq, *cq, dim, n_cu);
            } else {
                const GemvShape gs = plan_gemv(dim, ahd, G, false, 1, n_cu, small_cap);
                Lq.fn = pick<PRO_QUANT, EPI_RESID>(G, gs.RU, gs.JU, gs.FIN, gs.PF);
                aq.vr = gs.RU;
                Lq.grid = gs.grid;
                Lq.smem = gemv_smem_bytes(ahd, G, aq.vr, false);
            }
            Lq.ga = aq;
            if (!Lq.fn) return fail(Q3_ERR_UNSUPPORTED, "no kernel instantiated for this tile shape");
            if ((rc = set_max_smem((const void*)Lq.fn, Lq.smem))) return rc;
            wo_long.push_back(Lq);
            if (wo_preq) {
                apply_cfg(Ln, a, *wo_preq, dim, n_cu);
                Ln.ga = a;
                if ((rc = set_max_smem((const void*)Ln.fn, Ln.smem))) return rc;
            } else {
                Ln = Lq;
            }
            if (d_stamps) { Ln.ga.stamps = kstamps ? d_kslots + 2 * (size_t)kKstampSlots * plan.size() : d_stamps + 16 * plan.size(); Ln.ga.stamp_block = dev_knob("Q3_STAMP_BLOCK", 7); }
            plan.push_back(Ln);
        }
        {   // xb = RMSNorm_ffn(x); xq = quantize(xb); hb = silu(W1 xq) * (W3 xq)   qwen3.rs:159-161, layers.rs:468-475
            Launch Ln;
            Ln.fam = F_W13;
            GemvArgs a = base_args(dim);
            a.seg[0] = Seg{w1[lw].q, w1[lw].s, d_hb, H, 0};
            a.seg[1] = Seg{w3[lw].q, w3[lw].s, nullptr, H, 0};
            a.total_rows = 2 * H;
            a.norm_w = rms_ffn + (size_t)l * dim;
            a.in = d_x;
            if (const GemvCfg* cfg = find_cfg(PRO_NORM, EPI_SWIGLU, dim, G, dev_knob("Q3_CFG_W13", 0), fast_fold)) {
                a.xfirst = dev_knob("Q3_XFIRST_W13", dim < 2048 ? xfirst_dflt(cfg->wgt) : 0);
                apply_cfg(Ln, a, *cfg, H, n_cu);
            } else {
                const GemvShape gs = plan_gemv(H, dim, G, true, 1, n_cu, small_cap);
                Ln.fn = pick<PRO_NORM, EPI_SWIGLU>(G, gs.RU, gs.JU, gs.FIN, gs.PF);
                a.vr = gs.RU;
                Ln.grid = gs.grid;
                Ln.smem = gemv_smem_bytes(dim, G, a.vr, true);
            }
            Ln.ga = a;
            if (!Ln.fn) return fail(Q3_ERR_UNSUPPORTED, "no kernel instantiated for this tile shape");
            if ((rc = set_max_smem((const void*)Ln.fn, Ln.smem))) return rc;
            if (d_stamps) { Ln.ga.stamps = kstamps ? d_kslots + 2 * (size_t)kKstampSlots * plan.size() : d_stamps + 16 * plan.size(); Ln.ga.stamp_block = dev_knob("Q3_STAMP_BLOCK", 7); }
            plan.push_back(Ln);
        }
        {   // hq = quantize(hb); x += W2 hq                                      layers.rs:478-479, qwen3.rs:175
            Launch Ln;
            Ln.fam = F_W2;
            GemvArgs a = base_args(H);
            a.seg[0] = Seg{w2[lw].q, w2[lw].s, d_x, dim, 0};
            a.total_rows = dim;
            a.in = d_hb;
            if (const GemvCfg* cfg = find_cfg(PRO_QUANT, EPI_RESID, H, G, dev_knob("Q3_CFG_W2", 0), fast_fold)) {
                a.xfirst = dev_knob("Q3_XFIRST_W2", 0);
                apply_cfg(Ln, a, *cfg, dim, n_cu);
            } else {
                const GemvShape gs = plan_gemv(dim, H, G, false, 1, n_cu, small_cap);
                Ln.fn = pick<PRO_QUANT, EPI_RESID>(G, gs.RU, gs.JU, gs.FIN, gs.PF);
                a.vr = gs.RU;
                Ln.grid = gs.grid;
                Ln.smem = gemv_smem_bytes(H, G, a.vr, false);
            }
            Ln.ga = a;
            if (!Ln.fn) return fail(Q3_ERR_UNSUPPORTED, "no kernel instantiated for this tile shape");
            if ((rc = set_max_smem((const void*)Ln.fn, Ln.smem))) return rc;
            if (d_stamps) { Ln.ga.stamps = kstamps ? d_kslots + 2 * (size_t)kKstampSlots * plan.size() : d_stamps + 16 * plan.size(); Ln.ga.stamp_block = dev_knob("Q3_STAMP_BLOCK", 7); }
            plan.push_back(Ln);
        }
    }
    {   // x = RMSNorm_final(x); xq = quantize(x); logits = Wcls xq (+ argmax)      qwen3.rs:72-76
        Launch Ln;
        Ln.fam = F_LMHEAD;
        GemvArgs a = base_args(dim);
        a.seg[0] = Seg{wcls.q, wcls.s, d_logits, V, 0};
        a.total_rows = V;
        a.norm_w = rms_final;
        a.in = d_x;
        a.tap_out = d_tap;
        GemvShape gs = plan_gemv(V, dim, G, false, 1, n_cu, big_cap, false);
        const GemvCfg* lcfg = find_cfg(PRO_NORM, EPI_LOGITS, dim, G, dev_knob("Q3_CFG_LMHEAD", 0), fast_fold);
        if (lcfg) {
            a.xfirst = dev_knob("Q3_XFIRST_LM", 0);
            apply_cfg(Ln, a, *lcfg, V, n_cu);
            // streaming launch: cap the grid at the resident set (grid-stride over the row batches)
            const unsigned cap = (unsigned)(n_cu * (lcfg->wgt >= 512 ? 1 : 2));
            if (Ln.grid > cap) Ln.grid = cap;
            gs.grid = Ln.grid;
        }
        n_argmax_slots = (int)gs.grid;
        HIP_TRY(hipMalloc((void**)&d_argmax_slots, 8 * (size_t)n_argmax_slots));
        HIP_TRY(hipMemset(d_argmax_slots, 0, 8 * (size_t)n_argmax_slots));
        a.argmax_slots = d_argmax_slots;
        const bool fuse_next = dev_knob("Q3_FUSE_NEXT", 1) != 0;
        if (fuse_next) {
            HIP_TRY(hipMalloc((void**)&d_next_cell, 16));
            HIP_TRY(hipMemset(d_next_cell, 0, 16));
            a.next_cell = d_next_cell;
            a.out_tokens = d_out_tokens;
            a.out_cap = out_cap;
            a.prompt = d_prompt;
        }
        if (!lcfg) {
            Ln.fn = pick<PRO_NORM, EPI_LOGITS>(G, gs.RU, gs.JU, 0, gs.PF);
            a.vr = gs.RU;
            Ln.grid = gs.grid;
            Ln.smem = gemv_smem_bytes(dim, G, a.vr, true);
        }
        Ln.ga = a;
        if (!Ln.fn) return fail(Q3_ERR_UNSUPPORTED, "no kernel instantiated for this tile shape");
        if ((rc = set_max_smem((const void*)Ln.fn, Ln.smem))) return rc;
        if (d_stamps) { Ln.ga.stamps = kstamps ? d_kslots + 2 * (size_t)kKstampSlots * plan.size() : d_stamps + 16 * plan.size(); Ln.ga.stamp_block = dev_knob("Q3_STAMP_BLOCK", 7); }
        plan.push_back(Ln);
    }
    if (d_next_cell == nullptr) {
        Launch Ln;
        Ln.fam = F_NEXT;
        Ln.is_next = true;
        plan.push_back(Ln);
    }
    if (kstamps) {          // (developer build only: kstamps is false in the product)
        std::vector<int> ns(plan.size(), 0);
        for (size_t i = 0; i < plan.size(); ++i) {
            const bool s2 = plan[i].is_attn && plan[i].attn_kind == 3 && attn_short2_ok(plan[i].aa);
            const long w = (long)plan[i].grid * (s2 ? 2 : 1) * ((plan[i].is_attn ? (s2 ? kS2Threads : kWG) : (long)plan[i].block) / 64);
            ns[i] = (int)(w < kKstampSlots ? w : kKstampSlots);
        }
        HIP_TRY(hipMemcpy(d_knslots, ns.data(), 4 * ns.size(), hipMemcpyHostToDevice));
        Launch Ln;
        Ln.fam = F_NEXT;
        Ln.is_kfold = true;
        plan.push_back(Ln);
    }
    // long-context plan: every attention launch becomes k_attn_scores (heads x T-chunks) + k_attn_out (heads x slices)
    size_t wo_i = 0;
    for (const Launch& L0 : plan) {
        if (L0.fam == F_WO) { plan_long.push_back(wo_long[wo_i++]); continue; }
        if (!L0.is_attn) { plan_long.push_back(L0); continue; }
        Launch A = L0, B = L0;
        A.aa.xbq = B.aa.xbq = nullptr;    // the split kernels write f32 xb only; Wo quantizes in its prologue
        A.attn_kind = 1;
        A.aa.stamps = nullptr;            // developer timeline of the long plan: k_attn_out's (Q3_STAMP_SCORES=1: k_attn_scores')
        if (dev_knob("Q3_STAMP_SCORES", 0)) { A.aa.stamps = L0.aa.stamps; B.aa.stamps = nullptr; }
        A.aa.att_global = d_att;
        A.aa.att_stride = att_stride;
        A.aa.q_out = nullptr;
        const ScoresShape ss = scores_shape(hd, A.aa.n_heads, A.aa.n_kv_heads, S);
        A.scores_kvm = ss.kvm;
        A.aa.att_cmax = B.aa.att_cmax = (ss.kvm && dev_knob("Q3_ATT_CMAX", 1)) ? d_att + (size_t)A.aa.n_heads * att_stride : nullptr;
        A.aa.cmax_stride = B.aa.cmax_stride = cmax_stride;
        A.grid = ss.gx;
        A.grid_y = ss.gy;
        A.smem = ss.smem;
        B.attn_kind = 2;
        B.aa.att_global = d_att;
        B.aa.att_priv = d_att_priv;
        B.aa.att_stride = att_stride;
        B.grid_y = (unsigned)nsl;
        B.aa.slice_w = slice_w;
        B.smem = attn_out_smem_bytes(hd, S, slice_w);
        if ((rc = set_attn_scores_smem(ss))) return rc;
        if ((rc = set_attn_out_smem(B.smem))) return rc;
        plan_long.push_back(A);
        plan_long.push_back(B);
    }
    return Q3_OK;
}

int q3_engine::capture() {
    HIP_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    for (const Launch& L : plan) launch_one(L, this);
    HIP_TRY(hipStreamEndCapture(stream, &graph));
    HIP_TRY(hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0));
    HIP_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    for (const Launch& L : plan_long) launch_one(L, this);
    HIP_TRY(hipStreamEndCapture(stream, &graph_long));
    HIP_TRY(hipGraphInstantiate(&graph_long_exec, graph_long, nullptr, nullptr, 0));
    // position ranges: only where the short plan runs k_attn_short2 (head_dim 128, cache a whole number of 8-row steps)
    for (const Launch& L : plan)
        if (L.is_attn && L.attn_kind == 3 && attn_short2_ok(L.aa)) ranges = dev_knob("Q3_ATT_RANGES", 0) != 0 && dev_knob("Q3_FWD_LOGITS_HOST", 0) == 0;
    if (ranges)
        for (int i = 0; i < kNRange; ++i) {
            HIP_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            for (const Launch& L : plan) launch_one(L, this, false, kRangeSteps[i]);
            HIP_TRY(hipStreamEndCapture(stream, &graph_rng[i]));
            HIP_TRY(hipGraphInstantiate(&graph_rng_exec[i], graph_rng[i], nullptr, nullptr, 0));
        }
    if (dev_knob("Q3_FWD_GRAPH", 1)) {
        const size_t lbytes = 4 * (size_t)cfg.vocab_size;
        if (ranges)
            for (int i = 0; i < kNRange; ++i) {
                HIP_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
                HIP_TRY(hipMemcpyAsync(d_state, h_state, sizeof(State), hipMemcpyHostToDevice, stream));
                for (const Launch& L : plan) launch_one(L, this, false, kRangeSteps[i]);
                HIP_TRY(hipMemcpyAsync(h_logits, d_logits, lbytes, hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamEndCapture(stream, &graph_fwd_rng[i]));
                HIP_TRY(hipGraphInstantiate(&graph_fwd_rng_exec[i], graph_fwd_rng[i], nullptr, nullptr, 0));
            }
        for (int lng = 0; lng < 2; ++lng) {
            HIP_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            HIP_TRY(hipMemcpyAsync(d_state, h_state, sizeof(State), hipMemcpyHostToDevice, stream));
            // (developer build, forward-only experiment: d_logits is NOT filled, so device-side sampling after q3_forward is undefined)
            // Q3_FWD_LOGITS_HOST=1: the classifier stores its logits straight into the pinned host buffer (device-visible,
            // fine-grained) and the download node disappears (SURVEY section 7 "Logits egress"; measured in docs/HISTORY.md section 7)
            const bool host_out = dev_knob("Q3_FWD_LOGITS_HOST", 0) != 0;
            for (const Launch& L : (lng ? plan_long : plan)) {
                if (host_out && L.fam == F_LMHEAD && !L.is_attn && !L.is_next) {
                    Launch M = L;
                    M.ga.seg[0].out = h_logits;
                    launch_one(M, this);
                } else launch_one(L, this);
            }
            if (!host_out) HIP_TRY(hipMemcpyAsync(h_logits, d_logits, lbytes, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamEndCapture(stream, lng ? &graph_fwd_long : &graph_fwd));
            HIP_TRY(hipGraphInstantiate(lng ? &graph_fwd_long_exec : &graph_fwd_exec, lng ? graph_fwd_long : graph_fwd, nullptr, nullptr, 0));
        }
    }
    return Q3_OK;
}

// draw = false: logits only (q3_forward: the caller samples, the device sampler must not consume a coin)
int q3_engine::enqueue_forward(bool eager, size_t pos, bool draw) {
    const bool lng = (int64_t)pos >= (int64_t)split_pos;
    const int rng = lng ? -1 : range_of(pos);
    if (graph_exec && !eager) {
        HIP_TRY(hipGraphLaunch(lng ? graph_long_exec : (rng >= 0 ? graph_rng_exec[rng] : graph_exec), stream));
    } else {
        // (eager launches: the same row-request hint the captured range graphs carry -- only when the engine keeps range graphs
        // (developer build, Q3_ATT_RANGES=1): an eager launch is the SAME launch as the captured one, grid (heads, 2) included, so
        // that eager rocprofv3 traces time the kernel shape the graph replays)
        int steps = 0;
        if (!lng && ranges)
            for (int i = kNRange - 1; i >= 0; --i)
                if (pos < (size_t)(8 * kRangeSteps[i])) steps = kRangeSteps[i];
        for (const Launch& L : (lng ? plan_long : plan)) launch_one(L, this, false, steps);
        HIP_TRY(hipGetLastError());
    }
    return draw ? enqueue_sample() : Q3_OK;
}

// Sampler::sample on the logits of the forward just enqueued (after k_next has advanced the state)
int q3_engine::enqueue_sample() {
    if (!sampling) return Q3_OK;
    if (sargs.pre_exp) hipLaunchKernelGGL(k_sample_exp, dim3((unsigned)n_cu, 1), dim3(256), 0, stream, sargs);
    if (sargs.phase == 1) {
        // pipelined draw: the single-workgroup kernel keeps the order-sensitive parts (exact sum; sort + exact walks), the
        // element-wise passes over the vocabulary in between run on the whole chip
        SampleArgs tail = sargs;
        tail.phase = 2;
        hipLaunchKernelGGL(k_sample, dim3(1), dim3(kSampThreads), 4 * kSegFloats, stream, sargs);
        hipLaunchKernelGGL(k_sample_norm_hist, dim3(kSampGrid), dim3(256), 0, stream, sargs);
        hipLaunchKernelGGL(k_sample_count, dim3(kSampGrid), dim3(256), 0, stream, sargs);
        hipLaunchKernelGGL(k_sample_scatter, dim3(kSampGrid), dim3(256), 0, stream, sargs);
        hipLaunchKernelGGL(k_sample, dim3(1), dim3(kSampThreads), 4 * kSegFloats, stream, tail);
    } else {
        hipLaunchKernelGGL(k_sample, dim3(1), dim3(kSampThreads), 4 * kSegFloats, stream, sargs);
    }
    HIP_TRY(hipGetLastError());
    return Q3_OK;
}

int q3_engine::set_state(size_t token, size_t pos) {
    if (token >= (size_t)cfg.vocab_size || pos >= (size_t)cfg.seq_len)
        return fail(Q3_ERR_ARG, "index out of range: token %zu (vocab_size %d), pos %zu (seq_len %d)", token,
                    cfg.vocab_size, pos, cfg.seq_len);
    h_state->token = (int)token;
    h_state->pos = (int)pos;
    h_state->step = 0;
    h_state->prompt_len = 0;
    h_state->argmax = 0ull;
    HIP_TRY(hipMemcpyAsync(d_state, h_state, sizeof(State), hipMemcpyHostToDevice, stream));
    return Q3_OK;
}

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

const char* q3_last_error(void) { return g_err; }
uint32_t q3_abi_version(void) { return Q3_ABI_VERSION; }
// q3_build_id(): csrc/q3_build_id.cpp (its own object, rebuilt whenever any source changes)

int q3_parse_header(const uint8_t* data, size_t len, q3_config* out) {
    g_err[0] = 0;
    return parse_header(data, len, out);
}

int q3_create(const char* checkpoint_path, uint32_t ctx_len, int device, uint32_t flags, q3_engine** out) {
    g_err[0] = 0;
    if (!checkpoint_path || !out) return fail(Q3_ERR_ARG, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(Q3_ERR_HIP, "no HIP device available (libqwen3_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(Q3_ERR_ARG, "device %d out of range (%d devices)", device, ndev);
    q3_engine* e = new q3_engine();
    e->flags = flags;
    e->device = device;
    int rc = e->load(checkpoint_path, ctx_len);
    if (rc == Q3_OK) rc = e->build_plan();
    if (rc == Q3_OK && !(flags & Q3_FLAG_NO_GRAPH)) rc = e->capture();
    if (rc != Q3_OK) {
        e->release();
        delete e;
        return rc;
    }
    *out = e;
    return Q3_OK;
}

void q3_destroy(q3_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    e->release();
    delete e;
}

int q3_get_config(const q3_engine* e, q3_config* out) {
    if (!e || !out) return fail(Q3_ERR_ARG, "null argument");
    *out = e->cfg;
    return Q3_OK;
}

const float* q3_forward(q3_engine* e, size_t token, size_t pos) {
    g_err[0] = 0;
    if (!e) {
        fail(Q3_ERR_ARG, "null engine");
        return nullptr;
    }
    static const bool dbg = getenv("Q3_DEBUG_TIMING") != nullptr;
    struct timespec t0, t1, t2, t3, t4;
    if (dbg) clock_gettime(CLOCK_MONOTONIC, &t0);
    if (hipSetDevice(e->device) != hipSuccess) { fail(Q3_ERR_HIP, "hipSetDevice failed"); return nullptr; }
    if (e->graph_fwd_exec && !e->sampling) {
        // one graph launch: state upload, kernels, logits download (the three separate enqueues cost ~25 us of host time)
        if (token >= (size_t)e->cfg.vocab_size || pos >= (size_t)e->cfg.seq_len) {
            fail(Q3_ERR_ARG, "index out of range: token %zu (vocab_size %d), pos %zu (seq_len %d)", token, e->cfg.vocab_size, pos, e->cfg.seq_len);
            return nullptr;
        }
        e->h_state->token = (int)token;
        e->h_state->pos = (int)pos;
        e->h_state->step = 0;
        e->h_state->prompt_len = 0;
        e->h_state->argmax = 0ull;
        const int rng = (int64_t)pos >= (int64_t)e->split_pos ? -1 : e->range_of(pos);
        hipError_t ge = hipGraphLaunch((int64_t)pos >= (int64_t)e->split_pos ? e->graph_fwd_long_exec : (rng >= 0 && e->graph_fwd_rng_exec[rng] ? e->graph_fwd_rng_exec[rng] : e->graph_fwd_exec), e->stream);
        if (ge == hipSuccess) ge = hipStreamSynchronize(e->stream);
        if (ge != hipSuccess) {
            fail(Q3_ERR_HIP, "forward failed: %s", hipGetErrorString(ge));
            return nullptr;
        }
        return e->h_logits;
    }
    if (e->set_state(token, pos) != Q3_OK) return nullptr;
    if (dbg) clock_gettime(CLOCK_MONOTONIC, &t1);
    if (e->enqueue_forward(false, pos, false) != Q3_OK) return nullptr;
    if (dbg) clock_gettime(CLOCK_MONOTONIC, &t2);
    hipError_t err = hipMemcpyAsync(e->h_logits, e->d_logits, 4 * (size_t)e->cfg.vocab_size, hipMemcpyDeviceToHost, e->stream);
    if (dbg) clock_gettime(CLOCK_MONOTONIC, &t3);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (err != hipSuccess) {
        fail(Q3_ERR_HIP, "forward failed: %s", hipGetErrorString(err));
        return nullptr;
    }
    if (dbg) {
        clock_gettime(CLOCK_MONOTONIC, &t4);
        auto us = [](const timespec& a, const timespec& b) { return (b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3; };
        static int n = 0;
        if (++n % 16 == 0)
            fprintf(stderr, "[q3] forward: set_state %.1f us, graph launch %.1f us, D2H enqueue %.1f us, sync %.1f us\n",
                    us(t0, t1), us(t1, t2), us(t2, t3), us(t3, t4));
    }
    return e->h_logits;
}

int q3_forward_argmax(q3_engine* e, size_t token, size_t pos, int32_t* next_token) {
    g_err[0] = 0;
    if (!e || !next_token) return fail(Q3_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    int rc = e->set_state(token, pos);
    if (rc) return rc;
    if ((rc = e->enqueue_forward(false, pos))) return rc;
    HIP_TRY(hipMemcpyAsync(e->h_tokens, e->d_out_tokens, 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    *next_token = e->h_tokens[0];
    return Q3_OK;
}

int q3_generate_greedy(q3_engine* e, size_t first_token, size_t first_pos, size_t n_tokens, int32_t* out_tokens) {
    g_err[0] = 0;
    if (!e || (!out_tokens && n_tokens)) return fail(Q3_ERR_ARG, "null argument");
    if (n_tokens == 0) return Q3_OK;
    if (first_pos + n_tokens > (size_t)e->cfg.seq_len)
        return fail(Q3_ERR_ARG, "first_pos %zu + n_tokens %zu exceeds seq_len %d", first_pos, n_tokens, e->cfg.seq_len);
    HIP_TRY(hipSetDevice(e->device));
    int rc = e->set_state(first_token, first_pos);
    if (rc) return rc;
    struct timespec t0, t1, t2;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t k = 0; k < n_tokens; ++k)
        if ((rc = e->enqueue_forward(false, first_pos + k))) return rc;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    HIP_TRY(hipMemcpyAsync(e->h_tokens, e->d_out_tokens, 4 * n_tokens, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    clock_gettime(CLOCK_MONOTONIC, &t2);
    if (e->d_stamps && !e->kstamps) {
        const size_t nl = e->plan.size();
        std::vector<unsigned long long> h(16 * nl);
        HIP_TRY(hipMemcpy(h.data(), e->d_stamps, 8 * 16 * nl, hipMemcpyDeviceToHost));
        double acc[F_COUNT][16] = {}; int cnt[F_COUNT] = {};
        for (size_t i = 0; i < nl; ++i) {
            if (e->plan[i].is_next || h[16 * i] == 0) continue;
            for (int k = 1; k < 16; ++k)
                if (h[16 * i + k] >= h[16 * i]) acc[e->plan[i].fam][k] += (double)(h[16 * i + k] - h[16 * i]);
            cnt[e->plan[i].fam]++;
        }
        for (int f = 0; f < F_COUNT; ++f)
            if (cnt[f]) fprintf(stderr, "[q3 stamps] %-8s n=%d  issue %.0f  prologue %.0f  tile %.0f  finish %.0f  end %.0f  (s_memtime ticks after entry)\n", kFamilyNames[f], cnt[f], acc[f][1] / cnt[f], acc[f][2] / cnt[f], acc[f][3] / cnt[f], acc[f][4] / cnt[f], acc[f][5] / cnt[f]);
        for (int f = 0; f < F_COUNT; ++f) if (cnt[f] && acc[f][7] > 0 && f != F_ATTN) fprintf(stderr, "[q3 stamps] %-8s prologue detail: x/sum start %.0f  sum end %.0f  quantized %.0f\n", kFamilyNames[f], acc[f][6] / cnt[f], acc[f][7] / cnt[f], acc[f][8] / cnt[f]);
        if (e->sampling) {
            unsigned long long hs[16];
            HIP_TRY(hipMemcpy(hs, e->d_stamps + 16 * (size_t)(5 * e->cfg.n_layers + 3), sizeof(hs), hipMemcpyDeviceToHost));
            if (hs[0]) fprintf(stderr, "[q3 stamps] sample (last draw): sum %llu  normalise %llu  histogram %llu  compaction %llu  sort %llu  cumulative walk %llu  cdf walk %llu  total %llu ticks, %llu candidates, %llu rounds of the exact denominator\n",
                               hs[1] - hs[0], hs[2] - hs[1], hs[3] - hs[2], hs[4] - hs[3], hs[5] - hs[4], hs[6] - hs[5], hs[7] - hs[6], hs[7] - hs[0], hs[9], hs[10]);
        }
        if (cnt[F_ATTN] && acc[F_ATTN][7] > 0) {           // k_attn_out: the chain wave in front of each chunk's barrier
            fprintf(stderr, "[q3 stamps] attn chain wave at the chunk barriers:");
            for (int k = 7; k < 16; ++k) fprintf(stderr, " %.0f", acc[F_ATTN][k] / cnt[F_ATTN]);
            fprintf(stderr, "\n");
        }
        if (cnt[F_ATTN]) fprintf(stderr, "[q3 stamps] attn: issued %.0f  norm %.0f  staged %.0f  scores %.0f  softmax %.0f  vsum %.0f\n", acc[F_ATTN][1] / cnt[F_ATTN], acc[F_ATTN][2] / cnt[F_ATTN], acc[F_ATTN][3] / cnt[F_ATTN], acc[F_ATTN][4] / cnt[F_ATTN], acc[F_ATTN][5] / cnt[F_ATTN], acc[F_ATTN][6] / cnt[F_ATTN]);
    }
#ifdef Q3_DEV
    if (e->kstamps && e->d_kacc) {
        // per family: average duration of a launch (first wave in .. last wave out) and average gap to its predecessor's end,
        // both from the in-kernel 100 MHz clock, averaged over every token folded so far; the sum over a token against the
        // host's wall clock for this call
        const size_t nl = e->plan.size() - 1;
        std::vector<unsigned long long> h(2 + 2 * nl);
        HIP_TRY(hipMemcpy(h.data(), e->d_kacc, 8 * h.size(), hipMemcpyDeviceToHost));
        const double ntok = (double)h[0];
        double dur[F_COUNT] = {}, gap[F_COUNT] = {}; int cnt[F_COUNT] = {};
        double tot = 0.0;
        for (size_t i = 0; i < nl; ++i) {
            const int f = e->plan[i].fam;
            dur[f] += (double)h[2 + 2 * i] * 0.01 / ntok;      // 10 ns ticks -> us
            gap[f] += (double)h[3 + 2 * i] * 0.01 / ntok;
            cnt[f]++;
            tot += ((double)h[2 + 2 * i] + (double)h[3 + 2 * i]) * 0.01 / ntok;
        }
        const double wall_us = ((t2.tv_sec - t0.tv_sec) * 1e6 + (t2.tv_nsec - t0.tv_nsec) * 1e-3) / (double)n_tokens;
        fprintf(stderr, "[q3 kstamps] {\"tokens_folded\": %.0f, \"sum_duration_plus_gap_us_per_token\": %.2f, \"wall_us_per_token_this_call\": %.2f, \"families\": {", ntok, tot, wall_us);
        bool first = true;
        for (int f = 0; f < F_COUNT; ++f)
            if (cnt[f] && f != F_NEXT) {
                fprintf(stderr, "%s\"%s\": {\"launches_per_token\": %d, \"avg_duration_us\": %.3f, \"avg_gap_to_predecessor_us\": %.3f}", first ? "" : ", ",
                        kFamilyNames[f], cnt[f], dur[f] / cnt[f], gap[f] / cnt[f]);
                first = false;
            }
        fprintf(stderr, "}}\n");
    }
#endif
    if (getenv("Q3_DEBUG_TIMING"))
        fprintf(stderr, "[q3] generate_greedy n=%zu enqueue %.1f us, drain %.1f us\n", n_tokens,
                (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3,
                (t2.tv_sec - t1.tv_sec) * 1e6 + (t2.tv_nsec - t1.tv_nsec) * 1e-3);
    memcpy(out_tokens, e->h_tokens, 4 * n_tokens);
    return Q3_OK;
}

int q3_prefill(q3_engine* e, const int32_t* tokens, size_t n_tokens, size_t first_pos, int32_t* next_token) {
    g_err[0] = 0;
    if (!e || !tokens || n_tokens == 0) return fail(Q3_ERR_ARG, "null or empty prompt");
    if (first_pos + n_tokens > (size_t)e->cfg.seq_len)
        return fail(Q3_ERR_ARG, "first_pos %zu + n_tokens %zu exceeds seq_len %d", first_pos, n_tokens, e->cfg.seq_len);
    for (size_t i = 0; i < n_tokens; ++i)
        if (tokens[i] < 0 || tokens[i] >= e->cfg.vocab_size)
            return fail(Q3_ERR_ARG, "index out of range: token %d (vocab_size %d)", tokens[i], e->cfg.vocab_size);
    HIP_TRY(hipSetDevice(e->device));
    memcpy(e->h_tokens, tokens, 4 * n_tokens);
    HIP_TRY(hipMemcpyAsync(e->d_prompt, e->h_tokens, 4 * n_tokens, hipMemcpyHostToDevice, e->stream));
    int rc = e->set_state((size_t)tokens[0], first_pos);
    if (rc) return rc;
    // set_state queued {token, pos, step 0, prompt_len 0}; patch prompt_len before anything runs
    e->h_state->prompt_len = (int)n_tokens;
    HIP_TRY(hipMemcpyAsync(e->d_state, e->h_state, sizeof(State), hipMemcpyHostToDevice, e->stream));
    for (size_t k = 0; k < n_tokens; ++k)
        if ((rc = e->enqueue_forward(false, first_pos + k))) return rc;
    HIP_TRY(hipMemcpyAsync(e->h_tokens, e->d_out_tokens + (n_tokens - 1), 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (next_token) *next_token = e->h_tokens[0];
    return Q3_OK;
}

int q3_sampler_set(q3_engine* e, float temperature, float topp, uint64_t rng_seed) {
    g_err[0] = 0;
    if (!e) return fail(Q3_ERR_ARG, "null engine");
    if (!(temperature >= 0.0f)) return fail(Q3_ERR_ARG, "Temperature must be non-negative");            // sampler.rs:32
    if (!(topp >= 0.0f && topp <= 1.0f)) return fail(Q3_ERR_ARG, "Top-p must be between 0.0 and 1.0");  // sampler.rs:33
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const int n = e->cfg.vocab_size;
    const int blen = 4 * ((n + 4095) / 4096);
    if (!e->d_sampler) {
        size_t n2 = 1;
        while (n2 < (size_t)n) n2 <<= 1;
        HIP_TRY(hipMalloc((void**)&e->d_sampler, sizeof(SamplerState)));
        HIP_TRY(hipMalloc((void**)&e->d_probs, 4 * (size_t)kSampThreads * blen));
        HIP_TRY(hipMalloc((void**)&e->d_sp, 4 * (size_t)kSampThreads * blen));
        HIP_TRY(hipMalloc((void**)&e->d_keys, 2 * 8 * n2));
        HIP_TRY(hipMalloc((void**)&e->d_samp_hist, 4 * kSampHistBins));
        HIP_TRY(hipMalloc((void**)&e->d_samp_counts, 4 * kSampGrid));
        e->keys_n2 = n2;
    }
    SamplerState h{};
    h.rng = rng_seed; h.temperature = temperature; h.topp = topp;
    int rc = set_max_smem((const void*)k_sample, 4 * kSegFloats);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(e->d_sampler, &h, sizeof(h), hipMemcpyHostToDevice));
    SampleArgs a{};
    a.logits = e->d_logits;
    a.n = n;
    a.blen = blen;
    a.probs = e->d_probs;
    a.keys = e->d_keys;
    a.keys2_off = (long long)e->keys_n2;
    a.sp = e->d_sp;
    a.ss = e->d_sampler;
    a.st = e->d_state;
    a.out_tokens = e->d_out_tokens;
    a.out_cap = e->out_cap;
    a.pre_exp = dev_knob("Q3_SAMPLER_PRE_EXP", 1);
    a.phase = (a.pre_exp && dev_knob("Q3_SAMPLER_PIPELINE", 1)) ? 1 : 0;
    a.hist = e->d_samp_hist;
    a.counts = e->d_samp_counts;
    a.stamps = e->d_stamps ? e->d_stamps + 16 * (size_t)(5 * e->cfg.n_layers + 3) : nullptr;
    e->sargs = a;
    e->sampling = temperature != 0.0f;                       // sampler.rs:119-120: temperature 0 is the argmax path
    return Q3_OK;
}

int q3_sampler_get_rng(q3_engine* e, uint64_t* rng_state) {
    g_err[0] = 0;
    if (!e || !rng_state || !e->d_sampler) return fail(Q3_ERR_ARG, "q3_sampler_set has not been called");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    SamplerState h;
    HIP_TRY(hipMemcpy(&h, e->d_sampler, sizeof(h), hipMemcpyDeviceToHost));
    *rng_state = h.rng;
    return Q3_OK;
}

int q3_forward_sample(q3_engine* e, size_t token, size_t pos, int32_t* next_token) { return q3_forward_argmax(e, token, pos, next_token); }

int q3_generate_sampled(q3_engine* e, size_t first_token, size_t first_pos, size_t n_tokens, int32_t* out_tokens) {
    return q3_generate_greedy(e, first_token, first_pos, n_tokens, out_tokens);
}

int q3_reset_kv(q3_engine* e) {
    if (!e) return fail(Q3_ERR_ARG, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    const size_t bytes = 4 * (size_t)e->cfg.n_layers * e->cfg.seq_len * e->cfg.n_kv_heads * e->cfg.head_dim;
    HIP_TRY(hipMemsetAsync(e->d_key, 0, bytes, e->stream));
    HIP_TRY(hipMemsetAsync(e->d_value, 0, bytes, e->stream));
    if (e->d_value_t) HIP_TRY(hipMemsetAsync(e->d_value_t, 0, bytes, e->stream));
    // the classifier's {argmax cell, ticket} pair is self-clearing per token; a reset also recovers it after a launch that
    // did not run to completion
    if (e->d_next_cell) HIP_TRY(hipMemsetAsync(e->d_next_cell, 0, 16, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return Q3_OK;
}

int q3_read_state(q3_engine* e, int kind, size_t offset, size_t count, float* out) {
    if (!e || !out) return fail(Q3_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    const size_t kv = (size_t)e->cfg.n_layers * e->cfg.seq_len * e->cfg.n_kv_heads * e->cfg.head_dim;
    const float* src = nullptr;
    size_t lim = 0;
    if (kind == 0) { src = e->d_key; lim = kv; }
    else if (kind == 1) { src = e->d_value; lim = kv; }
    else if (kind == 2) { src = e->d_tap; lim = (size_t)e->cfg.dim; }
    else return fail(Q3_ERR_ARG, "unknown state kind %d", kind);
    if (offset + count > lim) return fail(Q3_ERR_ARG, "state range out of bounds");
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(out, src + offset, 4 * count, hipMemcpyDeviceToHost));
    return Q3_OK;
}

const char* q3_profile_name(int family) { return (family >= 0 && family < F_COUNT) ? kFamilyNames[family] : nullptr; }

int q3_profile(q3_engine* e, size_t token, size_t pos, int reps, float* ms, int32_t* launches, int cap) {
    g_err[0] = 0;
    if (!e || !ms || !launches || cap < F_COUNT || reps <= 0) return fail(Q3_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    for (int i = 0; i < cap; ++i) { ms[i] = 0.f; launches[i] = 0; }
    const std::vector<Launch>& P = ((int64_t)pos >= (int64_t)e->split_pos) ? e->plan_long : e->plan;
    const size_t nl = P.size();
    // events are released on every return path (RAII); the replayed launches overwrite x, the logits and KV row `pos`
    // exactly as a forward(token, pos) would -- callers treat q3_profile as a forward that returns timings
    struct Events {
        std::vector<hipEvent_t> ev;
        ~Events() { for (hipEvent_t x : ev) if (x) (void)hipEventDestroy(x); }
    } evs;
    evs.ev.assign(2 * F_COUNT, nullptr);
    std::vector<hipEvent_t>& ev = evs.ev;
    for (auto& x : ev) HIP_TRY(hipEventCreate(&x));
    for (int r = 0; r < reps; ++r) {
        int rc = e->set_state(token, pos);
        if (rc) return rc;
        // one full forward first so every family runs on live data, then each family's launches back to back
        // between ONE pair of events: the average is the launch period (kernel + boundary), the same quantity
        // rocprofv3's per-dispatch durations sum to on a serialised stream.
        for (size_t i = 0; i < nl; ++i) if (!P[i].is_next) launch_one(P[i], e, true);
        for (int f = 0; f < F_COUNT; ++f) {
            HIP_TRY(hipEventRecord(ev[2 * f], e->stream));
            for (size_t i = 0; i < nl; ++i)
                if (P[i].fam == f) { launch_one(P[i], e, true); launches[f] += 1; }
            HIP_TRY(hipEventRecord(ev[2 * f + 1], e->stream));
        }
        HIP_TRY(hipStreamSynchronize(e->stream));
        for (int f = 0; f < F_COUNT; ++f) {
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, ev[2 * f], ev[2 * f + 1]));
            ms[f] += t;
        }
    }
    return F_COUNT;
}

// The reference's decode loop as its host runs it (generation.rs:31-46,153-162 with temperature 0), in compiled host
// code on top of q3_forward: forward -> `logits.to_vec()` (a 4*vocab byte copy) -> Sampler::sample_argmax on the host
// (last maximum under total_cmp, sampler.rs:57-59) -> feed back.  *seconds follows TokenMetrics (generation.rs:198-233):
// the clock starts before the first forward and stops after the last sample.  This is what a Rust caller of the
// Transformers::Qwen3Hip shim observes; the logits cross PCIe every token.
}  // extern "C"

namespace {
// copy n floats and return the index of the LAST maximum under f32::total_cmp (key = bits ^ ((bits >> 31) & 0x7fffffff) as
// int32: negative floats order reversed).  AVX2 when the host has it (8 keys per step, streaming loads of the pinned
// buffer), scalar otherwise.
#if !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("avx2"))) size_t copy_argmax_last_avx2(const float* src, float* dst, size_t n) {
    // ONE pass: 8 lanes each keep (best key, index of its last occurrence); `key >= best` replaces, so a later equal key wins.
    // dst is 32-byte aligned (caller): non-temporal stores, the copy is not read again here.
    const __m256i m7f = _mm256_set1_epi32(0x7fffffff);
    __m256i best = _mm256_set1_epi32(INT32_MIN), bidx = _mm256_setzero_si256();
    __m256i cur = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
    const __m256i step = _mm256_set1_epi32(8);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256i b = _mm256_loadu_si256((const __m256i*)(src + i));
        _mm256_stream_si256((__m256i*)(dst + i), b);
        const __m256i key = _mm256_xor_si256(b, _mm256_and_si256(_mm256_srai_epi32(b, 31), m7f));
        const __m256i lt = _mm256_cmpgt_epi32(best, key);             // best > key: keep
        best = _mm256_max_epi32(best, key);
        bidx = _mm256_blendv_epi8(cur, bidx, lt);
        cur = _mm256_add_epi32(cur, step);
    }
    _mm_sfence();
    alignas(32) int32_t kk[8], ii[8];
    _mm256_store_si256((__m256i*)kk, best);
    _mm256_store_si256((__m256i*)ii, bidx);
    int32_t best_key = INT32_MIN;
    size_t best_i = 0;
    for (int l = 0; l < 8; ++l)
        if (kk[l] > best_key || (kk[l] == best_key && (size_t)ii[l] > best_i)) { best_key = kk[l]; best_i = (size_t)ii[l]; }
    for (; i < n; ++i) {
        int32_t b;
        memcpy(&b, src + i, 4);
        dst[i] = src[i];
        const int32_t key = b ^ ((b >> 31) & 0x7fffffff);
        if (key >= best_key) { best_key = key; best_i = i; }
    }
    return best_i;
}
#endif
size_t host_copy_argmax_last(const float* src, float* dst, size_t n) {
#if !defined(__HIP_DEVICE_COMPILE__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return copy_argmax_last_avx2(src, dst, n);
#endif
    const int32_t* sb = (const int32_t*)src;
    int32_t best_key = INT32_MIN;
    size_t best_i = 0;
    for (size_t i = 0; i < n; ++i) {
        const int32_t b = sb[i];
        dst[i] = src[i];
        const int32_t key = b ^ ((b >> 31) & 0x7fffffff);
        if (key >= best_key) { best_key = key; best_i = i; }
    }
    return best_i;
}
}  // namespace

extern "C" {

size_t q3_host_sample_argmax(const float* logits, size_t n, float* copy) {
    if (!logits || n == 0) return 0;
    if (copy && ((uintptr_t)copy & 31) == 0) return host_copy_argmax_last(logits, copy, n);
    // no (or an unaligned) destination: the same pass into a scratch block, then a plain copy
    std::vector<float> tmp(n + 16);
    float* al = (float*)(((uintptr_t)tmp.data() + 31) & ~(uintptr_t)31);
    const size_t r = host_copy_argmax_last(logits, al, n);
    if (copy) memcpy(copy, al, 4 * n);
    return r;
}

int q3_host_generate(q3_engine* e, size_t first_token, size_t first_pos, size_t n_tokens, int32_t* out_tokens, double* seconds) {
    g_err[0] = 0;
    if (!e || (!out_tokens && n_tokens)) return fail(Q3_ERR_ARG, "null argument");
    if (first_pos + n_tokens > (size_t)e->cfg.seq_len)
        return fail(Q3_ERR_ARG, "first_pos %zu + n_tokens %zu exceeds seq_len %d", first_pos, n_tokens, e->cfg.seq_len);
    const size_t V = (size_t)e->cfg.vocab_size;
    std::vector<float> copy_store(V + 16);
    float* const copy = (float*)(((uintptr_t)copy_store.data() + 31) & ~(uintptr_t)31);      // 32-byte aligned (non-temporal stores)
    static const bool dbg = getenv("Q3_DEBUG_TIMING") != nullptr;
    double t_fwd = 0.0, t_host = 0.0;
    struct timespec t0, t1, ta, tb, tc;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    size_t token = first_token;
    for (size_t k = 0; k < n_tokens; ++k) {
        if (dbg) clock_gettime(CLOCK_MONOTONIC, &ta);
        const float* lg = q3_forward(e, token, first_pos + k);
        if (!lg) return Q3_ERR_HIP;
        if (dbg) clock_gettime(CLOCK_MONOTONIC, &tb);
        // generation.rs:160 `logits.to_vec()` and Sampler::sample_argmax in ONE pass over the pinned buffer: copy + the maximum
        // total_cmp key (sampler.rs:57-59: Iterator::max_by keeps the LAST maximum under the IEEE total order), then a scan
        // from the END of the copy for its first match
        const size_t best = host_copy_argmax_last(lg, copy, V);
        out_tokens[k] = (int32_t)best;
        token = best;
        if (dbg) {
            clock_gettime(CLOCK_MONOTONIC, &tc);
            t_fwd += (tb.tv_sec - ta.tv_sec) * 1e6 + (tb.tv_nsec - ta.tv_nsec) * 1e-3;
            t_host += (tc.tv_sec - tb.tv_sec) * 1e6 + (tc.tv_nsec - tb.tv_nsec) * 1e-3;
        }
    }
    if (dbg && n_tokens) fprintf(stderr, "[q3] host_generate: q3_forward %.1f us/token, copy + argmax %.1f us/token\n", t_fwd / n_tokens, t_host / n_tokens);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds) *seconds = (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
    return Q3_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// operator-level entry points
// ------------------------------------------------------------------------------------------------
namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
        return Q3_OK;
    }
    int upload(const void* src, size_t bytes) {
        int rc = alloc(bytes);
        if (rc) return rc;
        if (bytes) HIP_TRY(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        return Q3_OK;
    }
    template <class T> T* as() { return (T*)p; }
};
int op_begin(int device) {
    g_err[0] = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(Q3_ERR_HIP, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(Q3_ERR_ARG, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    return Q3_OK;
}
int op_end() {
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return Q3_OK;
}
bool group_ok(size_t G) { return G >= 16 && G <= 1024 && (G & (G - 1)) == 0; }
}  // namespace

extern "C" {

int q3_op_quantize(int8_t* q, float* s, const float* x, size_t size, size_t group_size, int device) {
    int rc = op_begin(device);
    if (rc) return rc;
    if (!group_ok(group_size) || size % group_size) return fail(Q3_ERR_UNSUPPORTED, "unsupported size/group_size");
    DevBuf dx, dq, ds;
    if ((rc = dx.upload(x, 4 * size)) || (rc = dq.alloc(size)) || (rc = ds.alloc(4 * (size / group_size)))) return rc;
    hipLaunchKernelGGL(k_op_quantize, dim3(1), dim3(kWG), 0, 0, dq.as<int8_t>(), ds.as<float>(), dx.as<float>(), (int)size, (int)group_size);
    if ((rc = op_end())) return rc;
    HIP_TRY(hipMemcpy(q, dq.p, size, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(s, ds.p, 4 * (size / group_size), hipMemcpyDeviceToHost));
    return Q3_OK;
}

int q3_op_dequantize(const int8_t* q, const float* s, float* x, size_t size, size_t group_size, int device) {
    int rc = op_begin(device);
    if (rc) return rc;
    if (group_size == 0 || size % group_size) return fail(Q3_ERR_ARG, "size must be a multiple of group_size");
    DevBuf dx, dq, ds;
    if ((rc = dq.upload(q, size)) || (rc = ds.upload(s, 4 * (size / group_size))) || (rc = dx.alloc(4 * size))) return rc;
    hipLaunchKernelGGL(k_op_dequantize, dim3(256), dim3(256), 0, 0, dq.as<int8_t>(), ds.as<float>(), dx.as<float>(), size, (int)group_size);
    if ((rc = op_end())) return rc;
    HIP_TRY(hipMemcpy(x, dx.p, 4 * size, hipMemcpyDeviceToHost));
    return Q3_OK;
}

int q3_op_matmul(float* xout, const int8_t* xq, const float* xs, const int8_t* wq, const float* ws, size_t n, size_t d,
                 size_t group_size, int device) {
    int rc = op_begin(device);
    if (rc) return rc;
    if (!group_ok(group_size) || n % group_size || n % 16 || n == 0 || d == 0 || n > 65536)
        return fail(Q3_ERR_UNSUPPORTED, "unsupported n/group_size");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    DevBuf dxq, dxs, dwq, dws, dout;
    if ((rc = dxq.upload(xq, n)) || (rc = dxs.upload(xs, 4 * (n / group_size))) || (rc = dwq.upload(wq, n * d)) ||
        (rc = dws.upload(ws, 4 * (n * d / group_size))) || (rc = dout.alloc(4 * d)))
        return rc;
    GemvArgs a{};
    a.n = (int)n;
    a.group = (int)group_size;
    a.seg[0] = Seg{dwq.as<int8_t>(), dws.as<float>(), dout.as<float>(), (int)d, 0};
    a.total_rows = (int)d;
    a.pre_q = dxq.as<int8_t>();
    a.pre_s = dxs.as<float>();
    const GemvShape gs = plan_gemv((int)d, (int)n, (int)group_size, false, 1, prop.multiProcessorCount, 4);
    a.vr = gs.RU;
    const unsigned grid = gs.grid;
    const size_t smem = gemv_smem_bytes((int)n, (int)group_size, a.vr, false);
    GemvFn fn = pick<PRO_PREQ, EPI_STORE>((int)group_size, gs.RU, gs.JU, gs.FIN, gs.PF);
    if (!fn) return fail(Q3_ERR_UNSUPPORTED, "no kernel for tile %dx%d", gs.RU, gs.JU);
    if ((rc = set_max_smem((const void*)fn, smem))) return rc;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kWG), smem, 0, a);
    if ((rc = op_end())) return rc;
    HIP_TRY(hipMemcpy(xout, dout.p, 4 * d, hipMemcpyDeviceToHost));
    return Q3_OK;
}

int q3_op_rmsnorm(float* out, const float* in, const float* weight, size_t n, uint32_t flags, int device) {
    int rc = op_begin(device);
    if (rc) return rc;
    if (n == 0 || n > 32768) return fail(Q3_ERR_UNSUPPORTED, "unsupported n");
    DevBuf di, dw, dout;
    if ((rc = di.upload(in, 4 * n)) || (rc = dw.upload(weight, 4 * n)) || (rc = dout.alloc(4 * n))) return rc;
    const size_t smem = 4 * (size_t)term_floats((int)n) + 512;
    if ((rc = set_max_smem((const void*)k_op_rmsnorm, smem))) return rc;
    hipLaunchKernelGGL(k_op_rmsnorm, dim3(1), dim3(kWG), smem, 0, dout.as<float>(), di.as<float>(), dw.as<float>(), (int)n,
                       (flags & Q3_FLAG_FAST) ? 0 : 1);
    if ((rc = op_end())) return rc;
    HIP_TRY(hipMemcpy(out, dout.p, 4 * n, hipMemcpyDeviceToHost));
    return Q3_OK;
}

int q3_op_softmax(float* x, size_t n, uint32_t flags, int device) {
    int rc = op_begin(device);
    if (rc) return rc;
    if (n == 0) return Q3_OK;
    DevBuf dx;
    if ((rc = dx.upload(x, 4 * n))) return rc;
    hipLaunchKernelGGL(k_op_softmax, dim3(1), dim3(kWG), 0, 0, dx.as<float>(), (int)n, (flags & Q3_FLAG_FAST) ? 0 : 1);
    if ((rc = op_end())) return rc;
    HIP_TRY(hipMemcpy(x, dx.p, 4 * n, hipMemcpyDeviceToHost));
    return Q3_OK;
}

int q3_op_swiglu(float* hb, const float* hb2, size_t n, int device) {
    int rc = op_begin(device);
    if (rc) return rc;
    DevBuf a, b;
    if ((rc = a.upload(hb, 4 * n)) || (rc = b.upload(hb2, 4 * n))) return rc;
    hipLaunchKernelGGL(k_op_swiglu, dim3(256), dim3(256), 0, 0, a.as<float>(), b.as<float>(), n);
    if ((rc = op_end())) return rc;
    HIP_TRY(hipMemcpy(hb, a.p, 4 * n, hipMemcpyDeviceToHost));
    return Q3_OK;
}

int q3_op_expf(float* x, size_t n, int device) {
    int rc = op_begin(device);
    if (rc) return rc;
    DevBuf a;
    if ((rc = a.upload(x, 4 * n))) return rc;
    hipLaunchKernelGGL(k_op_expf, dim3(512), dim3(256), 0, 0, a.as<float>(), n);
    if ((rc = op_end())) return rc;
    HIP_TRY(hipMemcpy(x, a.p, 4 * n, hipMemcpyDeviceToHost));
    return Q3_OK;
}

int q3_op_attention(float* xb, float* q, float* key_cache_layer, const float* value_cache_layer, const float* q_norm_w,
                    const float* k_norm_w, size_t pos, size_t seq_len, size_t n_heads, size_t n_kv_heads, size_t head_dim,
                    uint32_t flags, int device) {
    int rc = op_begin(device);
    if (rc) return rc;
    if (pos >= seq_len || n_kv_heads == 0 || n_heads % n_kv_heads || head_dim % 8 || head_dim > 256 ||
        (head_dim & (head_dim - 1)))
        return fail(Q3_ERR_UNSUPPORTED, "unsupported attention shape");
    const size_t ahd = n_heads * head_dim, kvd = n_kv_heads * head_dim;
    DevBuf dq, dk, dv, dqw, dkw, dxb, drope, dkraw, datt;
    if ((rc = dq.upload(q, 4 * ahd)) || (rc = dk.upload(key_cache_layer, 4 * seq_len * kvd)) ||
        (rc = dv.upload(value_cache_layer, 4 * seq_len * kvd)) || (rc = dqw.upload(q_norm_w, 4 * head_dim)) ||
        (rc = dkw.upload(k_norm_w, 4 * head_dim)) || (rc = dxb.alloc(4 * ahd)) ||
        (rc = dkraw.upload(key_cache_layer + pos * kvd, 4 * kvd)))
        return rc;
    const size_t half = head_dim / 2;
    std::vector<float> tab(seq_len * head_dim);
    for (size_t p = 0; p < seq_len; ++p)
        for (size_t i = 0; i < half; ++i) {
            const float freq = powf(1e6f, -((float)i) / (float)half);
            const float angle = (float)p * freq;
            tab[p * head_dim + 2 * i] = cosf(angle);
            tab[p * head_dim + 2 * i + 1] = sinf(angle);
        }
    if ((rc = drope.upload(tab.data(), 4 * tab.size()))) return rc;
    const bool split = pos >= 256;                       // same rule as the engine's long-context plan
    const bool att_global = split || seq_len > 4096;
    const int att_stride = (int)((seq_len + 255) & ~(size_t)255);
    const int slice_w = attn_slice_w((int)head_dim, (int)n_heads, 256);
    const int nsl = (int)head_dim / slice_w;
    DevBuf dpriv, dqout;
    if (att_global && (rc = datt.alloc(4 * n_heads * (size_t)att_stride))) return rc;
    if (split && ((rc = dpriv.alloc(4 * n_heads * nsl * (size_t)att_stride)) || (rc = dqout.alloc(4 * ahd)))) return rc;
    AttnArgs a{};
    a.q = dq.as<float>();
    a.key_cache = dk.as<float>();
    a.k_raw = dkraw.as<float>();
    a.value_cache = dv.as<float>();
    a.q_norm_w = dqw.as<float>();
    a.k_norm_w = dkw.as<float>();
    a.rope = drope.as<float>();
    a.xb = dxb.as<float>();
    a.att_global = att_global ? datt.as<float>() : nullptr;
    a.st = nullptr;
    a.pos_override = (int)pos;
    attn_set_heads(a, (int)n_heads, (int)n_kv_heads);
    a.hd = (int)head_dim;
    a.seq_len = (int)seq_len;
    a.strict = (flags & Q3_FLAG_FAST) ? 0 : 1;
    a.write_q = 1;
    a.stamps = nullptr;
    DevBuf dvt;
    if (split && (seq_len % 4) == 0) {
        // the engine's long-context plan streams a transposed copy of the value rows: the operator builds it the same way
        if ((rc = dvt.alloc(4 * seq_len * kvd))) return rc;
        hipLaunchKernelGGL(k_value_transpose, dim3((unsigned)((kvd + 63) / 64), (unsigned)((seq_len + 63) / 64), 1u), dim3(kWG), 0, 0,
                           (const float*)dv.p, dvt.as<float>(), (int)seq_len, (int)kvd, 0, (int)seq_len);
        a.value_t = dvt.as<float>();
    }
    if (split) {
        a.att_stride = att_stride;
        a.att_priv = dpriv.as<float>();
        a.q_out = dqout.as<float>();
        const size_t sm2 = attn_out_smem_bytes((int)head_dim, (int)seq_len, slice_w);
        a.slice_w = slice_w;
        const ScoresShape ss = scores_shape((int)head_dim, (int)n_heads, (int)n_kv_heads, (int)seq_len);
        if ((rc = set_attn_scores_smem(ss)) || (rc = set_attn_out_smem(sm2))) return rc;
        DevBuf dcmax;
        if (ss.kvm && dev_knob("Q3_ATT_CMAX", 1)) {               // 64-timestep block maxima, as in the engine's long plan
            a.cmax_stride = (int)(((seq_len + 63) / 64 + 63) & ~(size_t)63);
            if ((rc = dcmax.alloc(4 * n_heads * (size_t)a.cmax_stride))) return rc;
            a.att_cmax = dcmax.as<float>();
        }
        launch_attn_scores(a, ss.kvm, ss.gx, ss.gy, ss.smem, 0);
        launch_attn_out(a, (unsigned)n_heads, (unsigned)nsl, sm2, 0);
        if ((rc = op_end())) return rc;
        HIP_TRY(hipMemcpy(dq.p, dqout.p, 4 * ahd, hipMemcpyDeviceToDevice));
    } else if ((head_dim == 64 || head_dim == 128) && dev_knob("Q3_ATT_SHORT", 1)) {
        a.att_short_form = dev_knob("Q3_ATT_SHORT", 1) == 2 ? 1 : 0;
        if ((rc = set_attn_short_smem(a))) return rc;
        launch_attn_short(a, (unsigned)n_heads, 0);
    } else {
        const size_t smem = attn_smem_bytes((int)head_dim, att_global ? 0 : (int)seq_len);
        if ((rc = set_max_smem((const void*)k_attn, smem))) return rc;
        hipLaunchKernelGGL(k_attn, dim3((unsigned)n_heads), dim3(kWG), smem, 0, a);
    }
    if ((rc = op_end())) return rc;
    HIP_TRY(hipMemcpy(xb, dxb.p, 4 * ahd, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(q, dq.p, 4 * ahd, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(key_cache_layer + pos * kvd, (float*)dk.p + pos * kvd, 4 * kvd, hipMemcpyDeviceToHost));
    return Q3_OK;
}

#ifdef Q3_DEV
/* developer micro-benchmark (not part of the drop-in surface; exported by the -DQ3_DEV build only): average device time of the stand-alone
 * W8A8 GEMV (PRO_PREQ/EPI_STORE) over `reps` launches on random weights resident in HBM.  ru/ju/wg_per_cu
 * <= 0 pick the planner's choice.  Distinct weight copies are cycled so nothing stays cache-resident. */
int q3_dev_bench_gemv(size_t n, size_t d, size_t group_size, int wg_per_cu, int ru, int ju, int reps, int device,
                      float* avg_us, int32_t* used /*[3]: RU, JU, grid*/) {
    int rc = op_begin(device);
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    const size_t wbytes = n * d, sbytes = 4 * (n * d / group_size);
    size_t copies = (512ull << 20) / (wbytes + sbytes) + 1;
    if (copies > 16) copies = 16;
    if (copies < 2) copies = 2;
    DevBuf dw, ds, dxq, dxs, dout;
    if ((rc = dw.alloc(wbytes * copies)) || (rc = ds.alloc(sbytes * copies)) || (rc = dxq.alloc(n)) ||
        (rc = dxs.alloc(4 * (n / group_size))) || (rc = dout.alloc(4 * d)))
        return rc;
    HIP_TRY(hipMemset(dw.p, 0x11, wbytes * copies));
    HIP_TRY(hipMemset(ds.p, 0, sbytes * copies));
    HIP_TRY(hipMemset(dxq.p, 1, n));
    HIP_TRY(hipMemset(dxs.p, 0, 4 * (n / group_size)));
    GemvShape gs = plan_gemv((int)d, (int)n, (int)group_size, false, 1, prop.multiProcessorCount, wg_per_cu > 0 ? wg_per_cu : 4);
    if (ru > 0) gs.RU = ru;
    if (ju > 0) gs.JU = ju;
    if (ru > 0 || ju > 0) {
        const long nb = ((long)d + gs.RU - 1) / gs.RU;
        long grid = (nb + kWaves - 1) / kWaves;
        const long cap = (long)prop.multiProcessorCount * (wg_per_cu > 0 ? wg_per_cu : 4);
        gs.grid = (unsigned)(grid > cap ? cap : grid);
    }
    if (ru > 0 || ju > 0) { gs.FIN = 0; gs.PF = 0; }
    GemvFn fn = pick<PRO_PREQ, EPI_STORE>((int)group_size, gs.RU, gs.JU, gs.FIN, gs.PF);
    if (!fn) return fail(Q3_ERR_UNSUPPORTED, "no kernel for tile %dx%d", gs.RU, gs.JU);
    const size_t smem = gemv_smem_bytes((int)n, (int)group_size, gs.RU, false);
    if ((rc = set_max_smem((const void*)fn, smem))) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    auto launch = [&](int i) {
        GemvArgs a{};
        a.n = (int)n;
        a.group = (int)group_size;
        a.vr = gs.RU;
        a.seg[0] = Seg{dw.as<int8_t>() + (size_t)(i % copies) * wbytes, (const float*)((char*)ds.p + (size_t)(i % copies) * sbytes),
                       dout.as<float>(), (int)d, 0};
        a.total_rows = (int)d;
        a.pre_q = dxq.as<int8_t>();
        a.pre_s = dxs.as<float>();
        hipLaunchKernelGGL(fn, dim3(gs.grid), dim3(kWG), smem, 0, a);
    };
    for (int i = 0; i < 3; ++i) launch(i);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch(i);
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (avg_us) *avg_us = ms * 1e3f / reps;
    if (used) { used[0] = gs.RU; used[1] = gs.JU; used[2] = (int)gs.grid; }
    return op_end();
}

#endif  // Q3_DEV

int q3_op_sample(const float* logits, size_t n, float temperature, float topp, uint64_t* rng_state, int32_t* index, int device) {
    int rc = op_begin(device);
    if (rc) return rc;
    if (!logits || !rng_state || !index || n == 0) return fail(Q3_ERR_ARG, "null argument");
    if (!(temperature > 0.0f)) return fail(Q3_ERR_ARG, "temperature must be positive (0 is q3_op_argmax)");
    if (!(topp >= 0.0f && topp <= 1.0f)) return fail(Q3_ERR_ARG, "Top-p must be between 0.0 and 1.0");
    const int blen = 4 * (((int)n + 4095) / 4096);
    size_t n2 = 1;
    while (n2 < n) n2 <<= 1;
    DevBuf dl, dp, ds, dk, dss, dst, dout;
    SamplerState h{*rng_state, temperature, topp, {0, 0, 0, 0}};
    State st{};
    st.step = 1;                                            // "one forward done": the draw is stored in out_tokens[0]
    if ((rc = dl.upload(logits, 4 * n)) || (rc = dp.alloc(4 * (size_t)kSampThreads * blen)) ||
        (rc = ds.alloc(4 * (size_t)kSampThreads * blen)) || (rc = dk.alloc(2 * 8 * n2)) || (rc = dss.upload(&h, sizeof(h))) ||
        (rc = dst.upload(&st, sizeof(st))) || (rc = dout.alloc(16)))
        return rc;
    if ((rc = set_max_smem((const void*)k_sample, 4 * kSegFloats))) return rc;
    SampleArgs a{};
    a.logits = dl.as<float>();
    a.n = (int)n;
    a.blen = blen;
    a.probs = dp.as<float>();
    a.sp = ds.as<float>();
    a.keys = dk.as<unsigned long long>();
    a.keys2_off = (long long)n2;
    a.ss = dss.as<SamplerState>();
    a.st = dst.as<State>();
    a.out_tokens = dout.as<int32_t>();
    a.out_cap = 4;
    hipLaunchKernelGGL(k_sample, dim3(1), dim3(kSampThreads), 4 * kSegFloats, 0, a);
    if ((rc = op_end())) return rc;
    HIP_TRY(hipMemcpy(&h, dss.p, sizeof(h), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(index, dout.p, 4, hipMemcpyDeviceToHost));
    *rng_state = h.rng;
    return Q3_OK;
}

int q3_op_argmax(const float* logits, size_t n, int32_t* index, int device) {
    int rc = op_begin(device);
    if (rc) return rc;
    if (!index) return fail(Q3_ERR_ARG, "null argument");
    if (n == 0) { *index = 0; return Q3_OK; }   // unwrap_or_default
    DevBuf dl, dc;
    unsigned long long zero = 0;
    if ((rc = dl.upload(logits, 4 * n)) || (rc = dc.upload(&zero, 8))) return rc;
    hipLaunchKernelGGL(k_op_argmax, dim3(256), dim3(kWG), 0, 0, dl.as<float>(), n, dc.as<unsigned long long>());
    if ((rc = op_end())) return rc;
    unsigned long long cell = 0;
    HIP_TRY(hipMemcpy(&cell, dc.p, 8, hipMemcpyDeviceToHost));
    *index = (int32_t)(cell & 0xffffffffull);
    return Q3_OK;
}

}  // extern "C"

#include "q3_batch_host.inc"
