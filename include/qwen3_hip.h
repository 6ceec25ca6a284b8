/*
 * qwen3_hip.h -- C ABI of libqwen3_hip.so, the MI355X (gfx950) Qwen3 Q8 decode engine.
 *
 * Drop-in boundary: the reference has no FFI; its seam is the Rust trait
 *     pub trait Transformer { fn forward(&mut self, token: usize, pos: usize) -> &[f32];
 *                             fn get_config(&self) -> &ModelConfig; }
 * (qwen3-inference/src/models/mod.rs:13-18), built by TransformerBuilder::build (models/mod.rs:55-73).
 * A `Transformers::Qwen3Hip` variant inside qwen3-inference binds exactly the entry points of section 1
 * (INTEGRATION.md shows the ~80-line Rust shim).  Plain pointers and sizes only; no torch / HIP types.
 *
 * Threading: one caller at a time per engine (mirrors `&mut self`); engines are independent, one per
 * GPU, no communication (replicas only -- the path does not shard).
 * Errors: 0 on success, negative q3_status otherwise; q3_last_error() gives the calling thread's
 * message.  There is NO CPU fallback: without a usable HIP device q3_create fails.
 *
 * Environment: libqwen3_hip.so reads exactly these variables (tests/test_host_and_abi.py checks the binary):
 *     Q3_PREFILL_M=<16..4096>   positions per weight pass of q3_prefill_batched (default 2048; <= 32 selects the batch-32
 *                               kernels).  The dense attention scratch grows with it: 4 * M * n_heads * context bytes.
 *     Q3_DEBUG_TIMING=1         host-side timing of q3_forward / q3_host_generate phases on stderr.
 *
 * Build options (qwen3-rs_amd/Makefile, `make abl EXTRA=-D... ABL=name` -> libq3_<name>.so; never part of libqwen3_hip.so):
 *     -DQ3_TICKET_ACQ_REL       the classifier's last-arriver ticket (csrc/q3_gemv.h, EPI_LOGITS) with acquire + release ordering
 *                               on its two device-scope read-modify-writes instead of relaxed.  The shipped library keeps
 *                               them relaxed: every workgroup's fetch_max on the argmax cell and its fetch_add on the ticket
 *                               are device-scope RMWs executed at the L2 / memory coherence point, and the ticket's operand
 *                               DATA-DEPENDS on the maximum's return value, so the add cannot issue before the max has
 *                               returned; the last arriver then reads the cell with another RMW.  That is an argument about
 *                               gfx950 (RMWs are performed at one point of coherence, in issue order per address
 *                               dependence), not about the HIP memory model, which would want release on the max and acquire
 *                               on the ticket.  Measured cost of the ordered form: +1.4 ... +4.8 us per token
 *                               (profiles/r05_ticket_order.txt: a release is a buffer_wbl2 in each of ~590 workgroups).
 *                               Build it when porting to another target or toolchain.
 * Every other Q3_* switch of earlier rounds (kernel-form A/B, tile and workgroup overrides, ablation bits, in-kernel
 * timelines) exists only in the developer build, libqwen3_hip_dev.so (`make -C qwen3-rs_amd dev`, -DQ3_DEV), together with
 * the kernel forms that lost their A/B; results are identical in both builds.
 */
#ifndef QWEN3_HIP_H
#define QWEN3_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Q3_ABI_VERSION 1

typedef enum q3_status {
    Q3_OK = 0,
    Q3_ERR_IO = -1,        /* open / mmap failed                        (models/mod.rs:56-59) */
    Q3_ERR_FORMAT = -2,    /* bad magic / version / dims / truncated    (configuration.rs:116-146, utils.rs:21-56) */
    Q3_ERR_ARG = -3,       /* null pointer, token/pos out of range (the reference panics: layers.rs:73-75,335) */
    Q3_ERR_HIP = -4,       /* HIP runtime error / no device */
    Q3_ERR_UNSUPPORTED = -5 /* shape the kernels do not cover (e.g. group_size not a multiple of 16) */
} q3_status;

/* ModelConfig, qwen3-inference/src/configuration.rs:18-30 (seq_len already clamped by ctx_len) */
typedef struct q3_config {
    int32_t architecture_id;
    int32_t dim;
    int32_t hidden_dim;
    int32_t n_layers;
    int32_t n_heads;
    int32_t n_kv_heads;
    int32_t head_dim;
    int32_t seq_len;
    int32_t vocab_size;
    int32_t group_size;
    int32_t shared_classifier; /* bool */
} q3_config;

typedef struct q3_engine q3_engine;

/* q3_create flags.
 * Default (flags = 0): every f32 sum is reduced in the reference's sequential order, so logits are
 * BIT-IDENTICAL to the CPU path and greedy token sequences are identical by construction. */
#define Q3_FLAG_FAST 1u /* opt-in: wavefront-tree reductions for the RMSNorm / attention sums.  Not
                           bit-exact: a 28-layer W8A8 stack amplifies a 1e-7 reordering difference to its
                           own int8 quantization-noise floor (logit deltas of ~0.1), so
                           greedy tokens can differ from the CPU path.  The int8 group-quant matmul itself
                           is bit-exact in both modes. */
#define Q3_FLAG_NO_GRAPH 2u /* launch kernels eagerly instead of replaying a captured hipGraph */
#define Q3_FLAG_NO_VALUE_T 4u /* do not keep the TRANSPOSED copy of the value cache.  Reference-order engines whose context
                           can reach the split attention path (seq_len > 256, a multiple of 4) keep the value cache twice:
                           row-major [L][seq_len][kv_dim] and transposed [L][kv_dim][seq_len], n_layers * kv_dim * seq_len * 4
                           bytes each (Qwen3-8B: 6.0 GB at 40,960 positions, 19.3 GB at 131,072; 0.6B: 4.7 GB at 40,960).
                           The copy lets the long-context output kernel stream 1 KB runs instead of 64-byte pieces
                           (config-3 decode 468 -> 503 tok/s when it went in).  With this flag the engine allocates the
                           row-major cache only and the long-context kernel reads it (same results bit for bit).
                           Q3_FLAG_FAST engines never allocate the copy (their output kernel does not read it). */

/* ------------------------------------------------------------------------------------------------
 * 1. The reference surface
 * ---------------------------------------------------------------------------------------------- */

/* TransformerBuilder::new(path).with_ctx_length(ctx).build()          models/mod.rs:45-73
 * ctx_len 0 = keep the checkpoint's seq_len.  device = HIP device ordinal. */
int q3_create(const char* checkpoint_path, uint32_t ctx_len, int device, uint32_t flags, q3_engine** out);

/* Transformer::get_config                                              models/mod.rs:17 */
int q3_get_config(const q3_engine* e, q3_config* out);

/* Transformer::forward(token, pos) -> &[f32; vocab_size]               models/qwen3.rs:62-79
 * Returns a host pointer to vocab_size logits, valid until the next call on this engine (same lifetime
 * as the reference's borrow from &mut self), or NULL on error (the Rust shim panics, as the reference
 * does on out-of-range indices). */
const float* q3_forward(q3_engine* e, size_t token, size_t pos);

/* Drop for the transformer */
void q3_destroy(q3_engine* e);

const char* q3_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * 2. Extensions outside the reference surface (same arithmetic, less egress)
 * ---------------------------------------------------------------------------------------------- */

/* forward + Sampler::sample_argmax (sampler.rs:57-59: last maximum under f32::total_cmp) on the device;
 * only the 4-byte token id crosses PCIe. */
int q3_forward_argmax(q3_engine* e, size_t token, size_t pos, int32_t* next_token);

/* The greedy inner loop of `generate` (generation.rs:31-46,153-162 with temperature 0) kept on the
 * device: step k runs forward(tok_k, first_pos + k) and tok_{k+1} = argmax.  tok_0 = first_token.
 * Writes tok_1..tok_n into out_tokens (n = n_tokens).  No termination check: callers cut at BOS/EOS
 * (generation.rs:35) afterwards.  Requires first_pos + n_tokens <= seq_len. */
int q3_generate_greedy(q3_engine* e, size_t first_token, size_t first_pos, size_t n_tokens, int32_t* out_tokens);

/* The reference's own decode loop on top of q3_forward, in compiled host code: forward -> `logits.to_vec()` ->
 * Sampler::sample_argmax on the HOST (generation.rs:31-46,153-162; sampler.rs:57-59) -> feed back.  The logits cross
 * PCIe every token, exactly what a Rust caller of the Transformers::Qwen3Hip shim observes.  *seconds (may be NULL) is
 * the TokenMetrics interval (generation.rs:198-233): from before the first forward to after the last sample.
 * Tokens are identical to q3_generate_greedy.  bench.py reports this as `forward_surface`. */
int q3_host_generate(q3_engine* e, size_t first_token, size_t first_pos, size_t n_tokens, int32_t* out_tokens, double* seconds);

/* The host half of generate_next_token as q3_host_generate runs it (no device involved): copy n logits to `copy`
 * (generation.rs:160 `logits.to_vec()`; may be NULL) and return Sampler::sample_argmax of them -- the index of the LAST maximum
 * under f32::total_cmp (sampler.rs:57-59, Iterator::max_by).  One vectorised pass (AVX2 when the host has it). */
size_t q3_host_sample_argmax(const float* logits, size_t n, float* copy);

/* The prompt loop of `chat` (handle_user_turn, generation.rs:116-123) kept on the device: every prompt token is
 * forwarded in order at first_pos, first_pos+1, ... (sequential prefill: identical K/V rows and logits to n calls
 * of q3_forward), the per-token sample is discarded, and the argmax after the LAST prompt token -- the first
 * generated token -- is returned.  Continue with q3_generate_greedy(e, *next_token, first_pos + n_tokens, ...). */
int q3_prefill(q3_engine* e, const int32_t* tokens, size_t n_tokens, size_t first_pos, int32_t* next_token);

/* Sampler::new(vocab_size, temperature, topp, rng_seed) + Sampler::sample on the device (sampler.rs:29-42,118-139):
 * temperature scaling, softmax, one xorshift64* coin per draw, multinomial (topp <= 0 or >= 1) or top-p.  After this
 * call with temperature > 0, every token the engine draws itself -- q3_forward_argmax / q3_forward_sample,
 * q3_generate_greedy / q3_generate_sampled, and the last position of q3_prefill / q3_prefill_batched -- comes from
 * Sampler::sample instead of the argmax, and a chat-mode prefill advances the rng by one coin per prompt position like
 * the reference's discarded samples (generation.rs:116-123).  temperature 0 restores the argmax path (no coin drawn).
 * All running sums are walked in the reference's order (bit-identical probabilities and tokens); candidates of EQUAL
 * probability in top-p are ordered by ascending token id (the reference's sort_unstable_by leaves that unspecified).
 * q3_forward() is unaffected: it returns logits and leaves sampling to the caller.  Batched decode has its own
 * per-stream samplers (q3_batch_sampler_set). */
int q3_sampler_set(q3_engine* e, float temperature, float topp, uint64_t rng_seed);
int q3_sampler_get_rng(q3_engine* e, uint64_t* rng_state);
int q3_forward_sample(q3_engine* e, size_t token, size_t pos, int32_t* next_token);
int q3_generate_sampled(q3_engine* e, size_t first_token, size_t first_pos, size_t n_tokens, int32_t* out_tokens);

/* Fresh-engine state: zero the KV cache (models/qwen3.rs:439-440) */
int q3_reset_kv(q3_engine* e);

/* Copy device state back for parity tests: kind 0 = key cache, 1 = value cache ([L][seq_len][kv_dim]),
 * 2 = x after the final RMSNorm ([dim]).  `count` floats starting at `offset`. */
int q3_read_state(q3_engine* e, int kind, size_t offset, size_t count, float* out);

/* Per-kernel timing of one forward(token,pos), launched eagerly with HIP events around every kernel
 * on the engine's stream.  Kernel family i is named by q3_profile_name(i); ms[i] / launches[i] are
 * accumulated over `reps` forwards.  Returns the number of families (<= cap). */
int q3_profile(q3_engine* e, size_t token, size_t pos, int reps, float* ms, int32_t* launches, int cap);
const char* q3_profile_name(int family);

/* Header parse + validation only (configuration.rs:77-146); does not touch the GPU. */
int q3_parse_header(const uint8_t* data, size_t len, q3_config* out);

uint32_t q3_abi_version(void);
/* First 16 hex digits of the SHA-256 over the library's sources (csrc/ + this header, in sorted file-name order), baked in by
 * the Makefile: lets a test harness detect a prebuilt libqwen3_hip.so that does not match the checked-out sources. */
const char* q3_build_id(void);

/* ------------------------------------------------------------------------------------------------
 * 2b. Batched decode: up to 32 independent streams (each its own KV cache, token and position) advance
 * one token per step while the weights are streamed ONCE per step through the int8 matrix cores.  This
 * serves N concurrent `generate` loops (generation.rs:9-48), which the reference can only run as N
 * processes.  Every stream's logits are bit-identical to the same (token, pos) sequence through
 * q3_forward on a fresh engine.  Stream i of a call always uses KV slot i.
 * ------------------------------------------------------------------------------------------------ */

/* Allocate the batched state: an MFMA-ordered copy of the weights, max_streams (1..32) zero-filled KV caches
 * of min(ctx_len, engine seq_len) rows (0 = engine seq_len) and scratch.  Needs group_size 64/128/256 and
 * every matrix height a multiple of 16 (Q3_ERR_UNSUPPORTED otherwise).  Calling it again re-allocates
 * (and clears) the state. */
int q3_batch_init(q3_engine* e, int max_streams, uint32_t ctx_len);

/* One step: stream i runs forward(tokens[i], pos[i]).  logits_out ([n_streams][vocab_size], host) and
 * argmax_out ([n_streams], sample_argmax of each row, sampler.rs:57-59) may each be NULL. */
int q3_forward_batch(q3_engine* e, const int32_t* tokens, const int32_t* pos, int n_streams, float* logits_out,
                     int32_t* argmax_out);

/* n_steps greedy steps for every stream with no host round trip; out_tokens is [n_streams][n_steps]: row i
 * equals q3_generate_greedy(first_tokens[i], first_pos[i], n_steps) on a fresh engine. */
int q3_generate_greedy_batch(q3_engine* e, const int32_t* first_tokens, const int32_t* first_pos, int n_streams,
                             size_t n_steps, int32_t* out_tokens);

/* Per-stream Sampler::sample for the batched decode (one sampler per stream: same temperature / top-p, stream i seeded
 * with rng_seeds[i], [max_streams] values).  With temperature > 0 the tokens of q3_forward_batch (argmax_out) and
 * q3_generate_greedy_batch are drawn exactly as q3_sampler_set + q3_forward_sample would draw them for that stream alone;
 * temperature 0 restores the argmax. */
int q3_batch_sampler_set(q3_engine* e, float temperature, float topp, const uint64_t* rng_seeds);

/* Zero every stream's KV cache. */
int q3_batch_reset_kv(q3_engine* e);

/* q3_prefill with the prompt walked in blocks of up to 2,048 consecutive positions (env Q3_PREFILL_M; blocks of <= 32 use
 * the batch-32 kernels), each block ONE pass over the weights on the matrix cores; the positions read and write the engine's own KV cache (the one q3_forward uses).
 * Sequential-equivalent: cache rows and the returned first generated token are bit-identical to q3_prefill, i.e. to
 * the prompt loop of `chat` (generation.rs:116-123).  Same shape limits as q3_batch_init (Q3_ERR_UNSUPPORTED otherwise);
 * does not need q3_batch_init (allocates the MFMA-ordered weight copy on first use, no per-stream caches). */
int q3_prefill_batched(q3_engine* e, const int32_t* tokens, size_t n_tokens, size_t first_pos, int32_t* next_token);

/* Parity tap: kind 0 key cache / 1 value cache ([n_layers][ctx][kv_dim]) / 2 residual stream x of one stream. */
int q3_batch_read_state(q3_engine* e, int stream, int kind, size_t offset, size_t count, float* out);

/* ------------------------------------------------------------------------------------------------
 * 3. Operator-level entry points: the reference's public free functions (tensor.rs, layers.rs) run on
 *    the device over caller (host) buffers.  Used by the parity tests; same kernels/device functions
 *    as the fused forward.  `device` as in q3_create; flags: 0 (reference order) or Q3_FLAG_FAST.
 * ---------------------------------------------------------------------------------------------- */

/* tensor::quantize(qx, x, size, group_size)                             tensor.rs:91-119 */
int q3_op_quantize(int8_t* q, float* s, const float* x, size_t size, size_t group_size, int device);
/* tensor::dequantize                                                    tensor.rs:72-80 */
int q3_op_dequantize(const int8_t* q, const float* s, float* x, size_t size, size_t group_size, int device);
/* tensor::matmul(xout, x, w, n, d, group_size)                          tensor.rs:23-62 */
int q3_op_matmul(float* xout, const int8_t* xq, const float* xs, const int8_t* wq, const float* ws, size_t n,
                 size_t d, size_t group_size, int device);
/* RMSNorm::forward                                                      layers.rs:109-119 */
int q3_op_rmsnorm(float* out, const float* in, const float* weight, size_t n, uint32_t flags, int device);
/* layers::softmax                                                       layers.rs:495-506 */
int q3_op_softmax(float* x, size_t n, uint32_t flags, int device);
/* FeedForward SwiGLU: hb = hb*sigmoid(hb)*hb2                           layers.rs:472-475 */
int q3_op_swiglu(float* hb, const float* hb2, size_t n, int device);
/* f32::exp as the device computes it (glibc expf algorithm), elementwise */
int q3_op_expf(float* x, size_t n, int device);
/* MultiHeadAttention: QK-RMSNorm + RoPE + GQA attention of ONE layer    layers.rs:346-419
 * key/value: the layer's cache [seq_len][kv_dim] (row `pos` holds the raw k/v projections on entry;
 * on return the K row is normalised+rotated in place).  q: [n_heads*head_dim] in/out.  xb: output. */
int q3_op_attention(float* xb, float* q, float* key_cache_layer, const float* value_cache_layer,
                    const float* q_norm_w, const float* k_norm_w, size_t pos, size_t seq_len, size_t n_heads,
                    size_t n_kv_heads, size_t head_dim, uint32_t flags, int device);
/* Sampler::sample with temperature > 0 on caller logits: one draw, *rng_state advanced by one coin   sampler.rs:118-139 */
int q3_op_sample(const float* logits, size_t n, float temperature, float topp, uint64_t* rng_state, int32_t* index, int device);
/* Sampler::sample_argmax                                                sampler.rs:57-59 */
int q3_op_argmax(const float* logits, size_t n, int32_t* index, int device);

#ifdef __cplusplus
}
#endif
#endif /* QWEN3_HIP_H */
