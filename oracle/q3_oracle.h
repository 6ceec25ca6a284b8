/*
 * q3_oracle.h -- CPU ORACLE for the Qwen3 Q8 decode hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the arithmetic of reinterpretcat/qwen3-rs's CPU forward
 * pass (qwen3-inference/src/{tensor,layers}.rs, models/qwen3.rs) and of the exporter's
 * checkpoint quantizer (qwen3-export/src/model_exporter.rs).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product library
 * (libqwen3_hip.so) never links, loads or calls anything in this directory.
 *
 * PARITY PINNING STATUS
 *   - checkpoint format + exporter quantizer: PINNED against the reference's own known-answer
 *     tests (qwen3-export/tests/unit/model_exporter_test.rs:27-45,48-87,104-161,397-427),
 *     restated in tests/test_oracle_golden.py.
 *   - forward(token,pos): PARITY UNPINNED by the reference.  qwen3-inference has zero tests and
 *     no golden logits, and the reference is Rust with no cargo/rustc in this image, so it can
 *     neither be built into oracle/_ref nor imported.  The restatement is cross-checked against
 *     an independent numpy-float32 restatement (oracle/np_oracle.py) and committed fixtures
 *     (tests/golden/), which guards against transcription slips but not a shared misreading.
 *   - Sampler (sampler.rs): PARITY UNPINNED by the reference as well (no tests there); cross-checked
 *     against oracle/np_oracle.py (NpSampler) and, for the xorshift64* stream, against an
 *     arbitrary-precision evaluation of the recurrence.  Top-p ties: see q3o_sample_topp.
 *
 * Arithmetic rules honoured (each cited at the function):  strict left-to-right f32 sums
 * (Rust Iterator::sum), no FMA contraction (build with -ffp-contract=off), no fast-math,
 * glibc powf/cosf/sinf/expf/sqrtf (what Rust std lowers to on linux-gnu), roundf == f32::round
 * (half away from zero), saturating float->i8 casts (Rust `as i8`).
 */
#ifndef Q3_ORACLE_H
#define Q3_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- checkpoint header: configuration.rs:18-50, model_exporter.rs:164-191 ---- */
typedef struct q3o_config {
    int32_t architecture_id;
    int32_t dim;
    int32_t hidden_dim;
    int32_t n_layers;
    int32_t n_heads;
    int32_t n_kv_heads;
    int32_t head_dim;
    int32_t seq_len;          /* already clamped by ctx_len (models/mod.rs:65-67) */
    int32_t vocab_size;
    int32_t group_size;
    int32_t shared_classifier;
} q3o_config;

typedef struct q3o_model q3o_model;

/* last error text of the calling thread ("" if none) */
const char* q3o_last_error(void);

/* ---- free functions of tensor.rs / layers.rs (operate on caller buffers) ---- */

/* tensor.rs:91-119  runtime activation quantizer (half-away rounding, scale 0 for zero group) */
void q3o_quantize(int8_t* q, float* s, const float* x, size_t size, size_t group_size);
/* tensor.rs:72-80 */
void q3o_dequantize(const int8_t* q, const float* s, float* x, size_t size, size_t group_size);
/* tensor.rs:23-62   xout[i] = sum_g ((f32)idot * ws) * xs ; rows i in [0,d) */
void q3o_matmul(float* xout, const int8_t* xq, const float* xs, const int8_t* wq, const float* ws,
                size_t n, size_t d, size_t group_size);
/* layers.rs:109-131 (out may alias in: forward_inplace has identical arithmetic) */
void q3o_rmsnorm(float* out, const float* in, const float* weight, size_t n);
/* layers.rs:161-171  freqs_cs = [cos0,sin0,cos1,sin1,...] (head_dim/2 pairs) */
void q3o_rope_freqs(float* freqs_cs, size_t head_dim, size_t pos);
/* layers.rs:173-185 */
void q3o_rope_apply(float* slice, size_t head_dim, const float* freqs_cs);
/* layers.rs:495-506 */
void q3o_softmax(float* x, size_t n);
/* layers.rs:472-475  hb[i] = (hb[i] * (1/(1+exp(-hb[i])))) * hb2[i] */
void q3o_swiglu(float* hb, const float* hb2, size_t n);
/* layers.rs:346-419: QK-norm + RoPE on q (all heads) and on K row `pos` in the cache, then GQA
 * attention over cache rows 0..=pos of one layer.  key/value point at the LAYER's cache base
 * ([seq_len][kv_dim]); q is [n_heads*head_dim] in/out (normalised+rotated), xb is the output. */
void q3o_attention(float* xb, float* q, float* key_cache_layer, const float* value_cache_layer,
                   const float* q_norm_w, const float* k_norm_w, size_t pos, size_t n_heads,
                   size_t n_kv_heads, size_t head_dim);
/* sampler.rs:57-59  last maximum under IEEE total order wins */
size_t q3o_sample_argmax(const float* logits, size_t n);

/* sampler.rs:44-54  xorshift64* stream; random_f32 = (u32 >> 8) / 2^24 in [0,1) */
uint32_t q3o_random_u32(uint64_t* rng_state);
float q3o_random_f32(uint64_t* rng_state);
/* sampler.rs:62-71  first index whose running sum (from 0.0, index order) exceeds coin; else n-1 */
size_t q3o_sample_mult(const float* probs, size_t n, float coin);
/* sampler.rs:74-112  nucleus sampling.  The reference sorts the candidates with sort_unstable_by(total_cmp, descending):
 * the order of EQUAL probabilities is unspecified there; this restatement (and the device) breaks ties by
 * ascending token index. */
size_t q3o_sample_topp(const float* probs, size_t n, float topp, float coin);
/* sampler.rs:118-139  Sampler::sample: temperature 0 -> argmax; else logits /= T, softmax in place, one coin from
 * the rng, then multinomial (topp <= 0 or >= 1) or top-p.  Mutates logits and rng_state like the reference. */
size_t q3o_sample(float* logits, size_t n, float temperature, float topp, uint64_t* rng_state);

/* ---- exporter side (format owner): model_exporter.rs ---- */
/* :321-338 */
float q3o_round_half_to_even(float x);
/* :104-162  returns 0 ok, -1 if n % group_size != 0.  max_error may be NULL */
int q3o_quantize_q80(int8_t* q, float* s, float* max_error, const float* w, size_t n, size_t group_size);
/* :47-57 */
size_t q3o_find_optimal_group_size(size_t hidden_dim, size_t requested);
/* :164-191  writes exactly 256 bytes */
void q3o_write_header(uint8_t* out256, const q3o_config* cfg, int32_t max_seq_len);
/* configuration.rs:77-146  parse + validate; returns 0 or -1 (q3o_last_error) */
int q3o_read_config(const uint8_t* data, size_t len, q3o_config* out);

/* ---- whole model: models/mod.rs:55-73, models/qwen3.rs ---- */
/* ctx_len 0 = keep the file's seq_len.  returns NULL on error */
q3o_model* q3o_create(const char* checkpoint_path, uint32_t ctx_len);
void q3o_destroy(q3o_model* m);
void q3o_get_config(const q3o_model* m, q3o_config* out);
/* models/qwen3.rs:62-79; returns pointer to vocab_size logits valid until next call, NULL if
 * token/pos out of range (the reference panics there) */
const float* q3o_forward(q3o_model* m, size_t token, size_t pos);
/* re-zero the KV cache (fresh Qwen3Transformer::new state, qwen3.rs:439-440) */
void q3o_reset(q3o_model* m);
/* debug taps for per-stage parity: copies of internal buffers after the last forward */
const float* q3o_tap_x(const q3o_model* m);          /* [dim] after final norm */
const float* q3o_key_cache(const q3o_model* m);      /* [L][seq_len][kv_dim] */
const float* q3o_value_cache(const q3o_model* m);
/* number of OpenMP threads the oracle will use (rows / heads are distributed like rayon does) */
int q3o_num_threads(void);
void q3o_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
