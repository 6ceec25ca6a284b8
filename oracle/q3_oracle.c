/*
 * q3_oracle.c -- CPU ORACLE (test infrastructure, see q3_oracle.h for the pinning status).
 *
 * Plain-C restatement of the reference's arithmetic, every function citing the file:line of
 * reinterpretcat/qwen3-rs it follows.  Build:  gcc -O3 -ffp-contract=off -fno-fast-math -fopenmp
 * (never -ffast-math / -fassociative-math: float reductions must stay strictly sequential).
 *
 * Rust semantics restated here:
 *   - `iter.sum::<f32>()` is `fold(-0.0, |a,b| a+b)` (edition 2024 => rustc >= 1.85, where the
 *     additive identity is -0.0); strictly left to right.
 *   - rustc never contracts a*b+c into an FMA and never reassociates.
 *   - f32::round = roundf (half away from zero); `as i8` saturates and maps NaN to 0.
 *   - f32::max(a,b) = fmaxf(a,b) (NaN-ignoring); powf/cos/sin/exp lower to glibc's float routines.
 *   - rayon only distributes independent output elements (rows / heads); OpenMP does the same here,
 *     so results do not depend on the thread count.
 */
#define _GNU_SOURCE
#include "q3_oracle.h"

#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define Q3O_MAGIC 0x616a6331      /* configuration.rs:8, model_exporter.rs:34 */
#define Q3O_VERSION 1             /* configuration.rs:10 */
#define Q3O_HEADER_SIZE 256       /* configuration.rs:12 */
#define Q3O_CONFIG_SIZE 52        /* 13 x i32, configuration.rs:14,35-50 */
#define Q3O_EPSILON 1e-6f         /* layers.rs:6 */
#define Q3O_ROPE_BASE 1e6f        /* layers.rs:9 */

static __thread char g_err[512];

const char* q3o_last_error(void) { return g_err; }

static void set_err(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int q3o_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void q3o_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* Rust `x as i8` for f32: saturating, NaN -> 0 */
static inline int8_t f32_as_i8(float v) {
    if (v != v) return 0;
    if (v <= -128.0f) return -128;
    if (v >= 127.0f) return 127;
    return (int8_t)v; /* truncation toward zero, in range */
}

/* ------------------------------------------------------------------ tensor.rs */

/* tensor.rs:91-119 */
void q3o_quantize(int8_t* q, float* s, const float* x, size_t size, size_t group_size) {
    const float Q_MAX = 127.0f;
    size_t num_groups = size / group_size;
    for (size_t g = 0; g < num_groups; ++g) {
        const float* xg = x + g * group_size;
        float wmax = 0.0f; /* fold(0.0, |acc,v| acc.max(v.abs())) */
        for (size_t i = 0; i < group_size; ++i) wmax = fmaxf(wmax, fabsf(xg[i]));
        float scale = wmax / Q_MAX;
        s[g] = scale;
        for (size_t i = 0; i < group_size; ++i) {
            float qv = (scale != 0.0f) ? xg[i] / scale : 0.0f;
            q[g * group_size + i] = f32_as_i8(roundf(qv));
        }
    }
}

/* tensor.rs:72-80 */
void q3o_dequantize(const int8_t* q, const float* s, float* x, size_t size, size_t group_size) {
    for (size_t i = 0; i < size; ++i) x[i] = (float)q[i] * s[i / group_size];
}

/* tensor.rs:31-62 compute_matmul_row */
static inline float matmul_row(const int8_t* xq, const float* xs, const int8_t* wq, const float* ws,
                               size_t row, size_t n, size_t group_size) {
    size_t row_off = row * n;
    size_t num_groups = n / group_size;
    float acc = -0.0f; /* Iterator::sum::<f32>() identity */
    for (size_t g = 0; g < num_groups; ++g) {
        size_t gs = g * group_size;
        const int8_t* xp = xq + gs;
        const int8_t* wp = wq + row_off + gs;
        int32_t idot = 0; /* exact: |idot| <= 64*127*127 */
        for (size_t k = 0; k < group_size; ++k) idot += (int32_t)xp[k] * (int32_t)wp[k];
        float wsc = ws[(row_off + gs) / group_size];
        float term = (float)idot * wsc; /* tensor.rs:59: ((dot as f32) * ws) * xs */
        term = term * xs[g];
        acc = acc + term;
    }
    return acc;
}

/* tensor.rs:23-29: one rayon task per output row */
void q3o_matmul(float* xout, const int8_t* xq, const float* xs, const int8_t* wq, const float* ws,
                size_t n, size_t d, size_t group_size) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)d; ++i) xout[i] = matmul_row(xq, xs, wq, ws, (size_t)i, n, group_size);
}

/* ------------------------------------------------------------------ layers.rs */

/* layers.rs:109-131 */
void q3o_rmsnorm(float* out, const float* in, const float* weight, size_t n) {
    float ss = -0.0f;
    for (size_t i = 0; i < n; ++i) {
        float sq = in[i] * in[i];
        ss = ss + sq;
    }
    float mean = ss / (float)n;
    float f = 1.0f / sqrtf(mean + Q3O_EPSILON);
    for (size_t i = 0; i < n; ++i) {
        float t = f * in[i];
        out[i] = weight[i] * t;
    }
}

/* layers.rs:161-171 */
void q3o_rope_freqs(float* freqs_cs, size_t head_dim, size_t pos) {
    size_t half = head_dim / 2;
    for (size_t i = 0; i < half; ++i) {
        float e = -((float)i) / (float)half;
        float freq = powf(Q3O_ROPE_BASE, e);
        float angle = (float)pos * freq;
        freqs_cs[2 * i] = cosf(angle);
        freqs_cs[2 * i + 1] = sinf(angle);
    }
}

/* layers.rs:173-185: rotate-half pairing (x[i], x[i+half]) */
void q3o_rope_apply(float* slice, size_t head_dim, const float* freqs_cs) {
    size_t half = head_dim / 2;
    for (size_t i = 0; i < half; ++i) {
        float c = freqs_cs[2 * i], s = freqs_cs[2 * i + 1];
        float xv = slice[i], yv = slice[i + half];
        float a = xv * c, b = yv * s;
        float e = xv * s, f = yv * c;
        slice[i] = a - b;
        slice[i + half] = e + f;
    }
}

/* layers.rs:495-506 */
void q3o_softmax(float* x, size_t n) {
    float mx = -INFINITY;
    for (size_t i = 0; i < n; ++i) mx = fmaxf(mx, x[i]);
    float sum = -0.0f;
    for (size_t i = 0; i < n; ++i) {
        x[i] = expf(x[i] - mx);
        sum = sum + x[i];
    }
    float inv = 1.0f / sum;
    for (size_t i = 0; i < n; ++i) x[i] = x[i] * inv;
}

/* layers.rs:472-475 */
void q3o_swiglu(float* hb, const float* hb2, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        float g = hb[i];
        float den = 1.0f + expf(-g);
        float sw = g * (1.0f / den);
        hb[i] = sw * hb2[i];
    }
}

/* layers.rs:346-372 (qk-norm + rope) and 374-419 (attention); one rayon task per query head */
void q3o_attention(float* xb, float* q, float* key_cache_layer, const float* value_cache_layer,
                   const float* q_norm_w, const float* k_norm_w, size_t pos, size_t n_heads,
                   size_t n_kv_heads, size_t head_dim) {
    size_t kv_dim = n_kv_heads * head_dim;
    size_t kv_mul = n_heads / n_kv_heads;
    float* freqs = (float*)malloc(sizeof(float) * head_dim);
    float* temp = (float*)malloc(sizeof(float) * head_dim);
    float* att = (float*)malloc(sizeof(float) * n_heads * (pos + 1));
    q3o_rope_freqs(freqs, head_dim, pos);

    for (size_t h = 0; h < n_heads; ++h) { /* layers.rs:353-360 */
        float* qh = q + h * head_dim;
        memcpy(temp, qh, sizeof(float) * head_dim);
        q3o_rmsnorm(qh, temp, q_norm_w, head_dim);
        q3o_rope_apply(qh, head_dim, freqs);
    }
    float* krow = key_cache_layer + pos * kv_dim;
    for (size_t h = 0; h < n_kv_heads; ++h) { /* layers.rs:363-371, in place in the cache */
        float* kh = krow + h * head_dim;
        memcpy(temp, kh, sizeof(float) * head_dim);
        q3o_rmsnorm(kh, temp, k_norm_w, head_dim);
        q3o_rope_apply(kh, head_dim, freqs);
    }

    float scale = 1.0f / sqrtf((float)head_dim); /* (head_dim as f32).sqrt().recip() */
#pragma omp parallel for schedule(static)
    for (long hh = 0; hh < (long)n_heads; ++hh) {
        size_t h = (size_t)hh;
        const float* qh = q + h * head_dim;
        size_t kvh = h / kv_mul;
        float* a = att + h * (pos + 1);
        for (size_t t = 0; t <= pos; ++t) { /* layers.rs:391-401 */
            const float* k = key_cache_layer + t * kv_dim + kvh * head_dim;
            float dot = -0.0f;
            for (size_t i = 0; i < head_dim; ++i) {
                float p = qh[i] * k[i];
                dot = dot + p;
            }
            a[t] = dot * scale;
        }
        q3o_softmax(a, pos + 1); /* layers.rs:404 */
        float* out = xb + h * head_dim;
        for (size_t i = 0; i < head_dim; ++i) out[i] = 0.0f; /* fill(0.0) */
        for (size_t t = 0; t <= pos; ++t) { /* layers.rs:407-417 */
            const float* v = value_cache_layer + t * kv_dim + kvh * head_dim;
            float w = a[t];
            for (size_t i = 0; i < head_dim; ++i) {
                float p = w * v[i];
                out[i] = out[i] + p;
            }
        }
    }
    free(att);
    free(temp);
    free(freqs);
}

/* sampler.rs:57-59: Iterator::max_by(total_cmp) returns the LAST maximum */
static inline int32_t total_key(float f) {
    int32_t b;
    memcpy(&b, &f, 4);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1); /* core::f32::total_cmp */
    return b;
}

size_t q3o_sample_argmax(const float* logits, size_t n) {
    if (n == 0) return 0; /* unwrap_or_default */
    size_t best = 0;
    int32_t bk = total_key(logits[0]);
    for (size_t i = 1; i < n; ++i) {
        int32_t k = total_key(logits[i]);
        if (k >= bk) { /* not Greater => later element wins */
            bk = k;
            best = i;
        }
    }
    return best;
}

/* sampler.rs:44-54 */
uint32_t q3o_random_u32(uint64_t* st) {
    uint64_t s = *st;
    s ^= s >> 12;
    s ^= s << 25;
    s ^= s >> 27;
    *st = s;
    return (uint32_t)((s * 0x2545F4914F6CDD1DULL) >> 32);
}
float q3o_random_f32(uint64_t* st) { return (float)(q3o_random_u32(st) >> 8) / 16777216.0f; }

/* sampler.rs:62-71 */
size_t q3o_sample_mult(const float* p, size_t n, float coin) {
    float cdf = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        cdf = cdf + p[i];
        if (coin < cdf) return i;
    }
    return n ? n - 1 : 0; /* saturating_sub */
}

typedef struct { float prob; uint32_t index; } q3o_probindex;
static int cmp_probindex_desc(const void* a, const void* b) {
    const q3o_probindex* x = (const q3o_probindex*)a;
    const q3o_probindex* y = (const q3o_probindex*)b;
    const int32_t kx = total_key(x->prob), ky = total_key(y->prob);
    if (kx != ky) return kx > ky ? -1 : 1;           /* b.prob.total_cmp(&a.prob): descending */
    return x->index < y->index ? -1 : (x->index > y->index ? 1 : 0);   /* ties: ascending index (see header) */
}

/* sampler.rs:74-112 */
size_t q3o_sample_topp(const float* p, size_t n, float topp, float coin) {
    if (n == 0) return 0;
    const size_t denom = (n - 1) > 1 ? (n - 1) : 1; /* saturating_sub(1).max(1) */
    const float cutoff = (1.0f - topp) / (float)denom;
    q3o_probindex* pi = (q3o_probindex*)malloc(sizeof(q3o_probindex) * n);
    size_t n0 = 0;
    for (size_t i = 0; i < n; ++i)
        if (p[i] >= cutoff) { pi[n0].prob = p[i]; pi[n0].index = (uint32_t)i; ++n0; }
    qsort(pi, n0, sizeof(q3o_probindex), cmp_probindex_desc);
    float cumulative = 0.0f;
    size_t last_idx = n0 ? n0 - 1 : 0;
    for (size_t i = 0; i < n0; ++i) {
        cumulative = cumulative + pi[i].prob;
        if (cumulative > topp) { last_idx = i; break; }
    }
    const float r = coin * cumulative;
    float cdf = 0.0f;
    size_t res = n0 ? pi[last_idx].index : 0;        /* n0 == 0: probindex[0] keeps its initial {0.0, 0} */
    if (n0) {
        for (size_t i = 0; i <= last_idx; ++i) {
            cdf = cdf + pi[i].prob;
            if (r < cdf) { res = pi[i].index; break; }
        }
    }
    free(pi);
    return res;
}

/* sampler.rs:118-139 */
size_t q3o_sample(float* logits, size_t n, float temperature, float topp, uint64_t* rng_state) {
    if (temperature == 0.0f) return q3o_sample_argmax(logits, n);
    for (size_t i = 0; i < n; ++i) logits[i] = logits[i] / temperature;
    q3o_softmax(logits, n);
    const float coin = q3o_random_f32(rng_state);
    if (topp <= 0.0f || topp >= 1.0f) return q3o_sample_mult(logits, n, coin);
    return q3o_sample_topp(logits, n, topp, coin);
}

/* ------------------------------------------------------------------ exporter (format owner) */

/* model_exporter.rs:321-338 */
float q3o_round_half_to_even(float x) {
    float rounded = roundf(x);
    float diff = fabsf(x - rounded);
    if (diff != 0.5f) return rounded;
    /* `rounded as i32 % 2 == 0` (saturating cast; halfway values are far below 2^31) */
    int32_t ri;
    if (rounded != rounded) ri = 0;
    else if (rounded >= 2147483648.0f) ri = INT32_MAX;
    else if (rounded <= -2147483648.0f) ri = INT32_MIN;
    else ri = (int32_t)rounded;
    if (ri % 2 == 0) return rounded;
    return (x >= 0.0f) ? rounded - 1.0f : rounded + 1.0f;
}

/* model_exporter.rs:104-162 */
int q3o_quantize_q80(int8_t* q, float* s, float* max_error, const float* w, size_t n, size_t group_size) {
    if (group_size == 0 || n % group_size != 0) {
        set_err("Weight length is not a multiple of group_size");
        return -1;
    }
    size_t num_groups = n / group_size;
    float overall = 0.0f;
#pragma omp parallel for schedule(static) reduction(max : overall)
    for (long gg = 0; gg < (long)num_groups; ++gg) {
        size_t g = (size_t)gg;
        const float* grp = w + g * group_size;
        float gmax = 0.0f;
        for (size_t i = 0; i < group_size; ++i) gmax = fmaxf(gmax, fabsf(grp[i]));
        float scale = (gmax > 0.0f) ? gmax / 127.0f : 1.0f;
        float gerr = 0.0f;
        for (size_t i = 0; i < group_size; ++i) {
            int8_t qi = 0;
            if (scale > 0.0f) {
                float r = q3o_round_half_to_even(grp[i] / scale);
                /* f32::clamp(-127,127): NaN stays NaN -> `as i8` gives 0 */
                if (r < -127.0f) r = -127.0f;
                if (r > 127.0f) r = 127.0f;
                qi = f32_as_i8(r);
            }
            q[g * group_size + i] = qi;
            float deq = (float)qi * scale;
            gerr = fmaxf(gerr, fabsf(deq - grp[i]));
        }
        s[g] = scale;
        overall = fmaxf(overall, gerr);
    }
    if (max_error) *max_error = overall;
    return 0;
}

/* model_exporter.rs:47-57 */
size_t q3o_find_optimal_group_size(size_t hidden_dim, size_t requested) {
    const size_t MIN_GROUP_SIZE = 4;
    size_t size = requested < hidden_dim ? requested : hidden_dim;
    while (size >= MIN_GROUP_SIZE && hidden_dim % size != 0) size /= 2;
    return size > MIN_GROUP_SIZE ? size : MIN_GROUP_SIZE;
}

static void put_u32(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
static int32_t get_i32(const uint8_t* p) {
    return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}

/* model_exporter.rs:164-191 */
void q3o_write_header(uint8_t* out, const q3o_config* c, int32_t max_seq_len) {
    memset(out, 0, Q3O_HEADER_SIZE);
    put_u32(out + 0, (uint32_t)Q3O_MAGIC);
    put_u32(out + 4, (uint32_t)Q3O_VERSION);
    put_u32(out + 8, (uint32_t)c->architecture_id);
    put_u32(out + 12, (uint32_t)c->dim);
    put_u32(out + 16, (uint32_t)c->hidden_dim);
    put_u32(out + 20, (uint32_t)c->n_layers);
    put_u32(out + 24, (uint32_t)c->n_heads);
    put_u32(out + 28, (uint32_t)c->n_kv_heads);
    put_u32(out + 32, (uint32_t)c->vocab_size);
    put_u32(out + 36, (uint32_t)max_seq_len);
    put_u32(out + 40, (uint32_t)c->head_dim);
    put_u32(out + 44, (uint32_t)(c->shared_classifier != 0));
    put_u32(out + 48, (uint32_t)c->group_size);
}

/* configuration.rs:77-146 */
int q3o_read_config(const uint8_t* data, size_t len, q3o_config* out) {
    if (len < Q3O_CONFIG_SIZE) {
        set_err("Insufficient data: need %d bytes, have %zu remaining", Q3O_CONFIG_SIZE, len);
        return -1;
    }
    if (len < Q3O_HEADER_SIZE) {
        set_err("Cannot skip %d bytes: insufficient data", Q3O_HEADER_SIZE - Q3O_CONFIG_SIZE);
        return -1;
    }
    int32_t magic = get_i32(data + 0), version = get_i32(data + 4);
    int32_t arch = get_i32(data + 8), dim = get_i32(data + 12), hidden = get_i32(data + 16);
    int32_t n_layers = get_i32(data + 20), n_heads = get_i32(data + 24), n_kv = get_i32(data + 28);
    int32_t vocab = get_i32(data + 32), seq_len = get_i32(data + 36), head_dim = get_i32(data + 40);
    int32_t shared = get_i32(data + 44), group = get_i32(data + 48);
    if (magic != Q3O_MAGIC) {
        set_err("Invalid model configuration: Invalid checkpoint magic number: expected %#x, got %#x",
                Q3O_MAGIC, (unsigned)magic);
        return -1;
    }
    if (version != Q3O_VERSION) {
        set_err("Invalid model configuration: Unsupported checkpoint version: expected %d, got %d",
                Q3O_VERSION, version);
        return -1;
    }
    const char* names[8] = {"architecture_id", "dim", "n_layers", "n_heads",
                            "n_kv_heads", "vocab_size", "seq_len", "head_dim"};
    int32_t vals[8] = {arch, dim, n_layers, n_heads, n_kv, vocab, seq_len, head_dim};
    for (int i = 0; i < 8; ++i)
        if (vals[i] <= 0) {
            set_err("Invalid model configuration: Invalid %s: must be positive, got %d", names[i], vals[i]);
            return -1;
        }
    out->architecture_id = arch; out->dim = dim; out->hidden_dim = hidden; out->n_layers = n_layers;
    out->n_heads = n_heads; out->n_kv_heads = n_kv; out->head_dim = head_dim; out->seq_len = seq_len;
    out->vocab_size = vocab; out->group_size = group; out->shared_classifier = (shared != 0);
    return 0;
}

/* ------------------------------------------------------------------ model (models/qwen3.rs) */

typedef struct {
    const int8_t* q;
    const float* s;
} qtensor;

struct q3o_model {
    q3o_config cfg;
    void* map;
    size_t map_len;
    /* f32 norm weights, qwen3.rs:228-232 */
    const float *rms_att, *rms_ffn, *rms_final, *q_ln, *k_ln;
    qtensor tok; /* embedding (dequantised row-by-row: same arithmetic as qwen3.rs:241-242) */
    qtensor *wq, *wk, *wv, *wo, *w1, *w2, *w3;
    qtensor wcls;
    /* TransformerBlockBuffers, qwen3.rs:414-445 */
    float *x, *xb, *xb2, *q, *hb, *hb2, *key_cache, *value_cache, *logits;
    int8_t *xq_q, *hq_q;
    float *xq_s, *hq_s;
};

typedef struct {
    const uint8_t* base;
    size_t len, off;
} cursor;

/* utils.rs:18-58 MemoryMapper */
static const void* cur_take(cursor* c, size_t bytes, const char* what) {
    if (c->off + bytes > c->len) {
        set_err("Failed to read %s: Insufficient data: need %zu bytes, have %zu remaining", what, bytes,
                c->len - c->off);
        return NULL;
    }
    const void* p = c->base + c->off;
    c->off += bytes;
    return p;
}

/* models/mod.rs:83-110 create_quantized_tensors */
static qtensor* take_qtensors(cursor* c, size_t n_tensors, size_t size_each, size_t group_size, const char* what) {
    qtensor* t = (qtensor*)calloc(n_tensors, sizeof(qtensor));
    for (size_t i = 0; i < n_tensors; ++i) {
        t[i].q = (const int8_t*)cur_take(c, size_each, what);
        if (!t[i].q) { free(t); return NULL; }
        t[i].s = (const float*)cur_take(c, (size_each / group_size) * sizeof(float), what);
        if (!t[i].s) { free(t); return NULL; }
    }
    return t;
}

void q3o_destroy(q3o_model* m) {
    if (!m) return;
    free(m->wq); free(m->wk); free(m->wv); free(m->wo); free(m->w1); free(m->w2); free(m->w3);
    free(m->x); free(m->xb); free(m->xb2); free(m->q); free(m->hb); free(m->hb2);
    free(m->key_cache); free(m->value_cache); free(m->logits);
    free(m->xq_q); free(m->hq_q); free(m->xq_s); free(m->hq_s);
    if (m->map) munmap(m->map, m->map_len);
    free(m);
}

/* models/mod.rs:55-73 TransformerBuilder::build + qwen3.rs:17-52,199-277 */
q3o_model* q3o_create(const char* path, uint32_t ctx_len) {
    g_err[0] = 0;
    int fd = open(path, O_RDONLY);
    if (fd < 0) {
        set_err("Failed to open checkpoint: %s: %s", path, strerror(errno));
        return NULL;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size == 0) {
        set_err("Failed to create memory mapping");
        close(fd);
        return NULL;
    }
    void* map = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) {
        set_err("Failed to create memory mapping");
        return NULL;
    }
    q3o_model* m = (q3o_model*)calloc(1, sizeof *m);
    m->map = map;
    m->map_len = (size_t)st.st_size;
    if (q3o_read_config((const uint8_t*)map, m->map_len, &m->cfg) != 0) { q3o_destroy(m); return NULL; }
    if (ctx_len != 0 && (int32_t)ctx_len < m->cfg.seq_len) m->cfg.seq_len = (int32_t)ctx_len; /* mod.rs:65-67 */
    if (m->cfg.architecture_id != 1) { /* mod.rs:69-72 */
        set_err("Unknown architecture_id: %d", m->cfg.architecture_id);
        q3o_destroy(m);
        return NULL;
    }
    const q3o_config* c = &m->cfg;
    size_t dim = c->dim, L = c->n_layers, hd = c->head_dim, V = c->vocab_size, H = c->hidden_dim;
    size_t G = c->group_size, ahd = (size_t)c->n_heads * hd, kvd = (size_t)c->n_kv_heads * hd;
    if (G == 0) { set_err("Invalid group_size 0"); q3o_destroy(m); return NULL; }
    cursor cur = {(const uint8_t*)map, m->map_len, Q3O_HEADER_SIZE};
    int ok = 1;
    ok = ok && (m->rms_att = (const float*)cur_take(&cur, L * dim * 4, "attention normalization weights"));
    ok = ok && (m->rms_ffn = (const float*)cur_take(&cur, L * dim * 4, "FFN normalization weights"));
    ok = ok && (m->rms_final = (const float*)cur_take(&cur, dim * 4, "final normalization weights"));
    ok = ok && (m->q_ln = (const float*)cur_take(&cur, L * hd * 4, "query layer norm weights"));
    ok = ok && (m->k_ln = (const float*)cur_take(&cur, L * hd * 4, "key layer norm weights"));
    qtensor* tok = ok ? take_qtensors(&cur, 1, V * dim, G, "quantized tensor data") : NULL;
    ok = ok && tok;
    if (tok) { m->tok = tok[0]; free(tok); }
    ok = ok && (m->wq = take_qtensors(&cur, L, dim * ahd, G, "quantized tensor data"));
    ok = ok && (m->wk = take_qtensors(&cur, L, dim * kvd, G, "quantized tensor data"));
    ok = ok && (m->wv = take_qtensors(&cur, L, dim * kvd, G, "quantized tensor data"));
    ok = ok && (m->wo = take_qtensors(&cur, L, ahd * dim, G, "quantized tensor data"));
    ok = ok && (m->w1 = take_qtensors(&cur, L, dim * H, G, "quantized tensor data"));
    ok = ok && (m->w2 = take_qtensors(&cur, L, H * dim, G, "quantized tensor data"));
    ok = ok && (m->w3 = take_qtensors(&cur, L, dim * H, G, "quantized tensor data"));
    if (ok) {
        if (c->shared_classifier) m->wcls = m->tok; /* qwen3.rs:252-253 */
        else {
            qtensor* cls = take_qtensors(&cur, 1, dim * V, G, "quantized tensor data");
            ok = ok && cls;
            if (cls) { m->wcls = cls[0]; free(cls); }
        }
    }
    if (!ok) { q3o_destroy(m); return NULL; }
    size_t S = c->seq_len;
    size_t xb_len = ahd > dim ? ahd : dim; /* xb is [all_heads_dim]; first `dim` used by norms */
    m->x = (float*)calloc(dim, 4);
    m->xb = (float*)calloc(xb_len, 4);
    m->xb2 = (float*)calloc(dim, 4);
    m->q = (float*)calloc(ahd, 4);
    m->hb = (float*)calloc(H, 4);
    m->hb2 = (float*)calloc(H, 4);
    m->xq_q = (int8_t*)calloc(xb_len, 1);
    m->xq_s = (float*)calloc(xb_len / G + 1, 4);
    m->hq_q = (int8_t*)calloc(H, 1);
    m->hq_s = (float*)calloc(H / G + 1, 4);
    m->key_cache = (float*)calloc(L * S * kvd, 4); /* zero-filled, qwen3.rs:439-440 */
    m->value_cache = (float*)calloc(L * S * kvd, 4);
    m->logits = (float*)calloc(V, 4);
    if (!m->x || !m->xb || !m->xb2 || !m->q || !m->hb || !m->hb2 || !m->xq_q || !m->xq_s || !m->hq_q ||
        !m->hq_s || !m->key_cache || !m->value_cache || !m->logits) {
        set_err("out of memory allocating run state");
        q3o_destroy(m);
        return NULL;
    }
    return m;
}

void q3o_get_config(const q3o_model* m, q3o_config* out) { *out = m->cfg; }

void q3o_reset(q3o_model* m) {
    size_t n = (size_t)m->cfg.n_layers * m->cfg.seq_len * m->cfg.n_kv_heads * m->cfg.head_dim;
    memset(m->key_cache, 0, n * 4);
    memset(m->value_cache, 0, n * 4);
}

const float* q3o_tap_x(const q3o_model* m) { return m->x; }
const float* q3o_key_cache(const q3o_model* m) { return m->key_cache; }
const float* q3o_value_cache(const q3o_model* m) { return m->value_cache; }

/* qwen3.rs:131-176 TransformerBlock::forward */
static void block_forward(q3o_model* m, size_t l, size_t pos) {
    const q3o_config* c = &m->cfg;
    size_t dim = c->dim, hd = c->head_dim, H = c->hidden_dim, G = c->group_size;
    size_t ahd = (size_t)c->n_heads * hd, kvd = (size_t)c->n_kv_heads * hd, S = c->seq_len;

    q3o_rmsnorm(m->xb, m->x, m->rms_att + l * dim, dim);                       /* :134 */
    q3o_quantize(m->xq_q, m->xq_s, m->xb, dim, G);                               /* :136 */

    /* MultiHeadAttention::forward, layers.rs:328-344 */
    float* kc = m->key_cache + l * S * kvd;
    float* vc = m->value_cache + l * S * kvd;
    q3o_matmul(m->q, m->xq_q, m->xq_s, m->wq[l].q, m->wq[l].s, dim, ahd, G);
    q3o_matmul(kc + pos * kvd, m->xq_q, m->xq_s, m->wk[l].q, m->wk[l].s, dim, kvd, G);
    q3o_matmul(vc + pos * kvd, m->xq_q, m->xq_s, m->wv[l].q, m->wv[l].s, dim, kvd, G);
    q3o_attention(m->xb, m->q, kc, vc, m->q_ln + l * hd, m->k_ln + l * hd, pos, c->n_heads, c->n_kv_heads, hd);

    q3o_quantize(m->xq_q, m->xq_s, m->xb, ahd, G);                               /* :152 full xb */
    q3o_matmul(m->xb2, m->xq_q, m->xq_s, m->wo[l].q, m->wo[l].s, ahd, dim, G);   /* :153 */
    for (size_t i = 0; i < dim; ++i) m->x[i] = m->x[i] + m->xb2[i];              /* :156 */

    q3o_rmsnorm(m->xb, m->x, m->rms_ffn + l * dim, dim);                         /* :159 */
    q3o_quantize(m->xq_q, m->xq_s, m->xb, dim, G);                               /* :161 */

    /* FeedForward::forward, layers.rs:466-480 */
    q3o_matmul(m->hb, m->xq_q, m->xq_s, m->w1[l].q, m->w1[l].s, dim, H, G);
    q3o_matmul(m->hb2, m->xq_q, m->xq_s, m->w3[l].q, m->w3[l].s, dim, H, G);
    q3o_swiglu(m->hb, m->hb2, H);
    q3o_quantize(m->hq_q, m->hq_s, m->hb, H, G);
    q3o_matmul(m->xb, m->hq_q, m->hq_s, m->w2[l].q, m->w2[l].s, H, dim, G);
    for (size_t i = 0; i < dim; ++i) m->x[i] = m->x[i] + m->xb[i];               /* :175 */
}

/* qwen3.rs:62-79 */
const float* q3o_forward(q3o_model* m, size_t token, size_t pos) {
    const q3o_config* c = &m->cfg;
    if (token >= (size_t)c->vocab_size || pos >= (size_t)c->seq_len) {
        set_err("index out of range: token %zu (vocab %d), pos %zu (seq_len %d)", token, c->vocab_size, pos,
                c->seq_len);
        return NULL;
    }
    size_t dim = c->dim, G = c->group_size;
    /* layers.rs:72-76 over the table dequantised by tensor.rs:72-80 */
    for (size_t i = 0; i < dim; ++i) {
        size_t idx = token * dim + i;
        m->x[i] = (float)m->tok.q[idx] * m->tok.s[idx / G];
    }
    for (size_t l = 0; l < (size_t)c->n_layers; ++l) block_forward(m, l, pos);
    q3o_rmsnorm(m->x, m->x, m->rms_final, dim);                                  /* :72 forward_inplace */
    q3o_quantize(m->xq_q, m->xq_s, m->x, dim, G);                                /* :75 */
    q3o_matmul(m->logits, m->xq_q, m->xq_s, m->wcls.q, m->wcls.s, dim, c->vocab_size, G); /* :76 */
    return m->logits;
}
