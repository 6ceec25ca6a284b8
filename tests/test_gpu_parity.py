"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the CPU
oracle on the same seeded inputs, against the committed golden fixtures, and -- at the full 0.6B size --
through properties that do not need the oracle to finish the whole run.

Bars:  int8 group-quant matmul, quantize, dequantize, argmax: BIT-EXACT always.
       default engine mode (reference summation order): logits BIT-IDENTICAL to the oracle.
       Q3_FLAG_FAST (tree reductions): per-op |delta| <= 4e-6 relative to unit-scale data (stated below)."""
import os

import numpy as np
import pytest

from conftest import assert_biteq, golden_path

pytestmark = pytest.mark.gpu
FAST_TOL = 4e-6   # abs tolerance for unit-scale activations in the opt-in tree-reduction mode


@pytest.fixture(scope="module")
def ops(q3):
    return q3.ops


def test_extension_is_the_code_that_runs(q3):
    """The .so in-tree is loaded and a device is present; nothing below can silently fall back."""
    assert os.path.samefile(q3.lib_path(), os.path.join(os.path.dirname(q3.__file__), "..", "libqwen3_hip.so"))
    maps = open("/proc/self/maps").read()
    assert "libqwen3_hip.so" in maps


@pytest.mark.parametrize("G", [16, 32, 64, 128])
def test_quantize_bitexact(ops, oracle, G):
    rng = np.random.default_rng(G)
    x = (rng.standard_normal(3072) * 3).astype(np.float32)
    x[:G] = 0.0
    x[G] = 1e-30
    qa, sa = ops.quantize(x, G)
    qb, sb = oracle.quantize(x, G)
    assert np.array_equal(qa, qb)
    assert_biteq(sa, sb, "scales")
    assert_biteq(ops.dequantize(qa, sa, G), oracle.dequantize(qb, sb, G), "dequantize")


@pytest.mark.parametrize("G", [64, 32])
def test_quantize_half_integer_quotients_and_extreme_scales(ops, oracle, G):
    """tensor.rs:108-111 `(x / scale).round() as i8` on the inputs where a shortcut would show: every group holds quotients AT and a
    few ulps around k + 0.5 for every k, both signs -- the exact ties that round-half-away and round-half-even disagree on, and
    their neighbours, where a reciprocal-multiply differs from the IEEE division -- under scales from 1e-38 to 1e38, plus denormal
    and huge values.  (Round 4 measured such a shortcut -- x * rcp(scale), nearest-even, exact path behind a 2^-20 |q| guard: it
    passes this test and is no faster anywhere, 1.6 % slower on the 8B shape; the device keeps the IEEE division.)"""
    rng = np.random.default_rng(7 + G)
    groups = []
    for gi in range(600):
        e = rng.uniform(-37.5, 37.5) if gi % 3 else rng.choice([-37.9, -30.0, 0.0, 30.0, 37.9])
        wmax = np.float32(10.0 ** e) * np.float32(rng.uniform(1.0, 2.0))
        scale = np.float32(wmax / np.float32(127.0))
        k = rng.integers(0, 127, G).astype(np.float32)
        q = (k + np.float32(0.5)).astype(np.float32)                       # target quotients k + 0.5
        x = (q * scale).astype(np.float32)
        steps = rng.integers(-3, 4, G)                                     # a few ulps either side of the tie
        x = np.nextafter(x, np.where(steps > 0, np.float32(np.inf), np.float32(-np.inf)).astype(np.float32)).astype(np.float32) if gi % 2 else x
        for _ in range(2):
            x = np.where(np.abs(steps) > 1, np.nextafter(x, np.where(steps > 0, np.float32(np.inf), np.float32(-np.inf)).astype(np.float32)), x).astype(np.float32)
        x *= rng.choice([-1.0, 1.0], G).astype(np.float32)
        x[0] = wmax if gi % 4 else -wmax                                   # pins the group maximum (so the scale is the one above)
        x = np.clip(x, -wmax, wmax).astype(np.float32)
        groups.append(x)
    groups.append(np.full(G, 1e-45, np.float32))                           # denormals only
    groups.append(np.concatenate([[np.float32(3e38)], rng.standard_normal(G - 1).astype(np.float32) * np.float32(1e38)]).astype(np.float32))
    groups.append(np.concatenate([[np.float32(1.0)], np.full(G - 1, 1e-42, np.float32)]).astype(np.float32))
    x = np.concatenate(groups).astype(np.float32)
    pad = (-len(x)) % 1024
    x = np.concatenate([x, np.zeros(pad, np.float32)])
    qa, sa = ops.quantize(x, G)
    qb, sb = oracle.quantize(x, G)
    assert_biteq(sa, sb, "scales")
    bad = np.nonzero(qa != qb)[0]
    assert bad.size == 0, f"{bad.size} quantized values differ, first at {bad[:5]}: x={x[bad[:5]]} device={qa[bad[:5]]} oracle={qb[bad[:5]]}"


@pytest.mark.parametrize("n,d,G", [(64, 40, 16), (1024, 2048, 64), (2048, 1024, 64), (3072, 1024, 64), (2560, 96, 64),
                                   (9728, 24, 64), (12288, 16, 64), (4096, 100, 128), (1024, 333, 32), (128, 1, 64),
                                   (1024, 151936 // 8, 64)])
def test_matmul_bitexact(ops, oracle, n, d, G):
    """tensor.rs:23-62 incl. ragged row counts, n not a multiple of 1 KiB, every model's inner dims."""
    rng = np.random.default_rng(n + d)
    xq = rng.integers(-127, 128, n).astype(np.int8)
    xs = rng.random(n // G).astype(np.float32)
    wq = rng.integers(-127, 128, n * d).astype(np.int8)
    ws = (rng.random(n * d // G) * 0.01).astype(np.float32)
    xs[0] = 0.0                                          # a zero activation group
    assert_biteq(ops.matmul(xq, xs, wq, ws, n, d, G), oracle.matmul(xq, xs, wq, ws, n, d, G), f"matmul {n}x{d}")


def test_matmul_extreme_values_and_linearity(ops, oracle):
    n, d, G = 1024, 64, 64
    xq = np.full(n, 127, np.int8)
    wq = np.full(n * d, -127, np.int8)
    xs = np.ones(n // G, np.float32)
    ws = np.ones(n * d // G, np.float32)
    out = ops.matmul(xq, xs, wq, ws, n, d, G)
    assert_biteq(out, oracle.matmul(xq, xs, wq, ws, n, d, G))
    assert out[0] == -127.0 * 127.0 * n
    # scaling every activation scale by 2 scales the output by exactly 2 (power of two: no rounding change)
    assert_biteq(ops.matmul(xq, xs * 2, wq, ws, n, d, G), out * 2)


@pytest.mark.parametrize("n", [64, 128, 1024, 2560, 4096])
def test_rmsnorm(ops, oracle, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    ref = oracle.rmsnorm(x, w)
    assert_biteq(ops.rmsnorm(x, w, strict=True), ref, "reference-order rmsnorm")
    assert np.max(np.abs(ops.rmsnorm(x, w, strict=False) - ref)) <= FAST_TOL


def test_rmsnorm_exact_sum_adversarial(ops, oracle):
    """Inputs that force ties and binade crossings inside the speculative exact sum: the result must still be
    bit-identical to the sequential fold (the verification loop, not luck, guarantees it)."""
    rng = np.random.default_rng(1)
    w = np.ones(1024, np.float32)
    cases = [np.full(1024, 1.0, np.float32),                         # every partial sum exact, many crossings
             np.full(1024, 3.0, np.float32),
             (2.0 ** rng.integers(-12, 12, 1024)).astype(np.float32),  # power-of-two squares: ties everywhere
             np.concatenate([np.full(512, 1e-4, np.float32), np.full(512, 4096.0, np.float32)]),
             np.concatenate([np.full(1, 8192.0, np.float32), np.full(1023, 1.0 / 64, np.float32)]),
             np.zeros(1024, np.float32)]
    for i, x in enumerate(cases):
        assert_biteq(ops.rmsnorm(x, w, strict=True), oracle.rmsnorm(x, w), f"case {i}")
    for seed in range(20):
        x = (np.random.default_rng(seed).standard_normal(2048) * 10.0 ** np.random.default_rng(seed).integers(-3, 3)).astype(np.float32)
        assert_biteq(ops.rmsnorm(x, np.ones(2048, np.float32), strict=True), oracle.rmsnorm(x, np.ones(2048, np.float32)))


def test_softmax_swiglu_expf(ops, oracle):
    rng = np.random.default_rng(2)
    for n in (1, 7, 777, 5000):
        a = (rng.standard_normal(n) * 4).astype(np.float32)
        assert_biteq(ops.softmax(a, strict=True), oracle.softmax(a), f"softmax {n}")
        assert np.max(np.abs(ops.softmax(a, strict=False) - oracle.softmax(a))) <= FAST_TOL
    g = np.concatenate([(rng.standard_normal(3072) * 3), [0.0, -0.0, 100.0, -100.0, 88.8, -104.0]]).astype(np.float32)
    u = rng.standard_normal(g.size).astype(np.float32)
    assert_biteq(ops.swiglu(g, u), oracle.swiglu(g, u), "swiglu")
    # device expf restates glibc's algorithm: compare against the host libm on 200k values + specials
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.expf.restype = ctypes.c_float
    libm.expf.argtypes = [ctypes.c_float]
    xs = np.concatenate([rng.uniform(-104, 89, 100000), rng.standard_normal(100000) * 3,
                         [0.0, -0.0, np.inf, -np.inf, 88.7, 88.8, -103.9, -104.1, 1e-30, -1e-30]]).astype(np.float32)
    ref = np.array([libm.expf(float(v)) for v in xs], dtype=np.float32)
    assert_biteq(ops.expf(xs), ref, "expf vs glibc")


def test_argmax_device(ops, oracle):
    rng = np.random.default_rng(4)
    lg = rng.standard_normal(151936).astype(np.float32)
    assert ops.argmax(lg) == oracle.sample_argmax(lg)
    lg[[5, 77777, 151000]] = lg.max() + 1
    assert ops.argmax(lg) == 151000 == oracle.sample_argmax(lg)       # last maximum wins
    assert ops.argmax(np.array([-0.0, 0.0, -0.0], np.float32)) == 1    # total order: -0 < +0


@pytest.mark.parametrize("nh,nkv,hd,S,pos", [(4, 2, 16, 64, 9), (16, 8, 128, 256, 200), (8, 8, 64, 32, 0),
                                             (4, 1, 32, 700, 650), (16, 8, 128, 160, 127), (16, 8, 128, 160, 128),
                                             (32, 8, 128, 300, 299)])
def test_attention(ops, oracle, nh, nkv, hd, S, pos):
    """QK-RMSNorm + RoPE + GQA attention of one layer (layers.rs:346-419), chunk boundaries included."""
    rng = np.random.default_rng(pos + hd)
    kvd = nkv * hd
    q = rng.standard_normal(nh * hd).astype(np.float32)
    K = rng.standard_normal((S, kvd)).astype(np.float32)
    V = rng.standard_normal((S, kvd)).astype(np.float32)
    K[pos + 1:] = 0
    qw = (1 + 0.1 * rng.standard_normal(hd)).astype(np.float32)
    kw = (1 + 0.1 * rng.standard_normal(hd)).astype(np.float32)
    rb, rq, rk = oracle.attention(q, K, V, qw, kw, pos, nh, nkv, hd)
    xb, q2, k2 = ops.attention(q, K, V, qw, kw, pos, nh, nkv, hd, strict=True)
    assert_biteq(xb, rb, "xb")
    assert_biteq(q2, rq, "q after norm+rope")
    assert_biteq(k2.reshape(S, kvd)[pos], rk.reshape(S, kvd)[pos], "K row written in place")
    xb, q2, k2 = ops.attention(q, K, V, qw, kw, pos, nh, nkv, hd, strict=False)
    assert np.max(np.abs(xb - rb)) <= FAST_TOL and np.max(np.abs(q2 - rq)) <= FAST_TOL


@pytest.mark.parametrize("nh,nkv,S,pos", [
    (16, 8, 256, 0), (16, 8, 256, 7), (16, 8, 256, 8), (16, 8, 256, 23), (16, 8, 256, 63), (16, 8, 256, 64), (16, 8, 256, 65),
    (16, 8, 256, 95), (16, 8, 256, 129), (16, 8, 256, 191), (16, 8, 256, 192), (16, 8, 256, 255),
    (32, 8, 136, 100),             # four query heads per kv head
    (8, 8, 72, 70),                # one query head per kv head
    (16, 8, 8, 7),                 # the smallest cache k_attn_short2 takes (one 8-row step)
    (16, 8, 100, 99), (16, 8, 20, 17),   # caches that are not a whole number of 8-row steps: k_attn_short
])
def test_attention_short_contexts_head_dim_128(ops, oracle, nh, nkv, S, pos):
    """k_attn_short2 (round 5: coalesced key / value staging, q*K product tile, one or two workgroups per head) across its
    internal switch points -- staging tiers (8-row steps), one / two / four score waves (64 / 128 / 256 positions), value rows
    through LDS (<= 64 positions) or straight to the output waves, the second workgroup per head past 64 positions -- and the
    hand-back to k_attn_short for caches it does not take.  Bit for bit against the oracle (layers.rs:346-419), stale rows
    beyond the context included in the cache."""
    hd = 128
    rng = np.random.default_rng(1000 * S + pos)
    kvd = nkv * hd
    q = (3.0 * rng.standard_normal(nh * hd)).astype(np.float32)
    K = rng.standard_normal((S, kvd)).astype(np.float32)          # rows past pos keep stale values: they must not matter
    V = rng.standard_normal((S, kvd)).astype(np.float32)
    qw = (1 + 0.1 * rng.standard_normal(hd)).astype(np.float32)
    kw = (1 + 0.1 * rng.standard_normal(hd)).astype(np.float32)
    rb, rq, rk = oracle.attention(q, K, V, qw, kw, pos, nh, nkv, hd)
    xb, q2, k2 = ops.attention(q, K, V, qw, kw, pos, nh, nkv, hd, strict=True)
    assert_biteq(xb, rb, "xb")
    assert_biteq(q2, rq, "q after norm+rope")
    assert_biteq(k2.reshape(S, kvd)[pos], rk.reshape(S, kvd)[pos], "K row written in place")


@pytest.mark.parametrize("nh,nkv,hd,S,pos", [
    (4, 2, 64, 4600, 4500),        # head_dim 64: per-query-head scores kernel, block-wide row maximum
    (8, 2, 128, 4600, 4500),       # 4 query heads per kv head: shared K chunks, 71 block maxima (more than one per lane)
    (4, 2, 128, 8500, 8400),       # 2 query heads per kv head, context > 8192: probability rows in global memory
    (8, 2, 128, 4096, 4095),       # the last position of a 4096 context: every lane's block maximum is live
])
def test_attention_long_context(ops, oracle, nh, nkv, hd, S, pos):
    """pos beyond the LDS score buffer (scores go through the global scratch) and dozens of K/V chunks."""
    rng = np.random.default_rng(9)
    kvd = nkv * hd
    q = rng.standard_normal(nh * hd).astype(np.float32)
    K = (rng.standard_normal((S, kvd)) * 0.5).astype(np.float32)
    V = rng.standard_normal((S, kvd)).astype(np.float32)
    K[1000:1100] = 0.0                       # never-written rows are attended over as zeros (generation.rs:26-29)
    V[1000:1100] = 0.0
    qw = np.ones(hd, np.float32)
    kw = np.ones(hd, np.float32)
    rb, rq, rk = oracle.attention(q, K, V, qw, kw, pos, nh, nkv, hd)
    xb, q2, k2 = ops.attention(q, K, V, qw, kw, pos, nh, nkv, hd, strict=True)
    assert_biteq(xb, rb, "xb long context")
    assert_biteq(k2.reshape(S, kvd)[pos], rk.reshape(S, kvd)[pos])
    xb, _, _ = ops.attention(q, K, V, qw, kw, pos, nh, nkv, hd, strict=False)
    assert np.max(np.abs(xb - rb)) <= FAST_TOL


def test_attention_split_path_every_exact_sum_block_length(ops, oracle):
    """k_attn_out's exact softmax sum runs on 64 lanes x blocks of 4 * ceil(np / 256) terms: every block length 8 ... 64 and the rows that
    end on / next to a block or a 256-position boundary, bit for bit (layers.rs:495-506, 406-417)."""
    nh, nkv, hd, S = 4, 2, 128, 4096
    rng = np.random.default_rng(77)
    kvd = nkv * hd
    q = rng.standard_normal(nh * hd).astype(np.float32)
    K = (rng.standard_normal((S, kvd)) * 0.5).astype(np.float32)
    V = rng.standard_normal((S, kvd)).astype(np.float32)
    qw = (1.0 + 0.1 * rng.standard_normal(hd)).astype(np.float32)
    kw = (1.0 + 0.1 * rng.standard_normal(hd)).astype(np.float32)
    for pos in [256, 300, 511, 512, 700, 767, 1023, 1024, 1100, 1290, 1535, 1800, 2047, 2048, 2300, 2303, 2304, 2600, 2815, 2816, 3000, 3300,
                3500, 3583, 3839, 3840, 4000, 4095]:
        rb, _, rk = oracle.attention(q, K, V, qw, kw, pos, nh, nkv, hd)
        xb, _, k2 = ops.attention(q, K, V, qw, kw, pos, nh, nkv, hd, strict=True)
        assert_biteq(xb, rb, f"xb at position {pos}")
        assert_biteq(k2.reshape(S, kvd)[pos], rk.reshape(S, kvd)[pos], f"K row at position {pos}")


@pytest.mark.parametrize("pos", [40, 100, 200, 700, 2300])
def test_attention_scores_outside_the_exp_main_range(ops, oracle, pos):
    """Norm weights of 6 put score differences far beyond 88 into the softmax: the exps of the attention kernels take glibc's main path
    alone only when every lane of a wave is inside |x| < 88 (q3_expf_special), and fall back to the full special-case handling otherwise
    -- underflow to +0, the subnormal tail -- bit for bit (short-context kernel: positions < 256; split path beyond)."""
    nh, nkv, hd, S = 4, 2, 128, 4096
    rng = np.random.default_rng(5)
    kvd = nkv * hd
    q = rng.standard_normal(nh * hd).astype(np.float32)
    K = rng.standard_normal((S, kvd)).astype(np.float32)
    V = rng.standard_normal((S, kvd)).astype(np.float32)
    qw = np.full(hd, 6.0, np.float32)
    kw = np.full(hd, 6.0, np.float32)
    rb, _, _ = oracle.attention(q, K, V, qw, kw, pos, nh, nkv, hd)
    xb, _, _ = ops.attention(q, K, V, qw, kw, pos, nh, nkv, hd, strict=True)
    assert np.isfinite(rb).all()
    assert_biteq(xb, rb, f"xb at position {pos}")


def test_eager_launch_mode_matches_graph(q3):
    """Q3_FLAG_NO_GRAPH launches the same kernel chain eagerly: identical logits and tokens."""
    path = golden_path("tiny-untied.bin")
    with q3.TransformerBuilder(path).build() as g, q3.TransformerBuilder(path).with_graph(False).build() as e:
        for pos in range(5):
            assert_biteq(np.array(g.forward(7, pos), copy=True), np.array(e.forward(7, pos), copy=True), f"pos {pos}")
        g.reset_kv(); e.reset_kv()
        assert g.generate_greedy(3, 2, 10) == e.generate_greedy(3, 2, 10)


# ---- whole model -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny", "tiny-untied"])
def test_forward_matches_golden_fixture(q3, name):
    """Committed fixtures (numpy restatement): logits and the KV cache, bit for bit."""
    g = np.load(golden_path(f"{name}.golden.npz"))
    with q3.TransformerBuilder(golden_path(f"{name}.bin")).build() as t:
        for (tok, pos), want in zip(g["calls"], g["logits"]):
            assert_biteq(t.forward(int(tok), int(pos)), want, f"{name} forward({tok},{pos})")
        assert_biteq(t.read_state("key"), g["key_cache"].reshape(-1), "key cache")
        assert_biteq(t.read_state("value"), g["value_cache"].reshape(-1), "value cache")
        prompt = [int(v) for v in g["prompt"]]
        t.reset_kv()
        toks, _ = q3.generate(t, prompt, max_new_tokens=12)
        assert toks == [int(v) for v in g["generate_tokens"]]
        t.reset_kv()
        ctoks, cpos, _ = q3.chat_turn(t, prompt, 0, 10)
        assert ctoks == [int(v) for v in g["chat_tokens"]] and cpos == int(g["chat_pos"])
        # device-resident greedy loop == host loop over forward()
        t.reset_kv()
        assert t.generate_greedy(prompt[-1], len(prompt) - 1, 12) == toks


@pytest.mark.parametrize("name,ctx", [("tiny-g64", 0), ("small-hd128", 0), ("small-hd128", 100)])
def test_forward_vs_oracle_live(q3, oracle, tmp_ckpt_dir, name, ctx):
    ck = q3.checkpoint
    path = os.path.join(tmp_ckpt_dir, f"{name}.bin")
    ck.ensure_synthetic_checkpoint(path, ck.SHAPES[name], seed=21, sparse_zero_groups=True)
    om = oracle.OracleModel(path, ctx)
    with q3.TransformerBuilder(path).with_ctx_length(ctx or None).build() as t:
        assert t.get_config().seq_len == om.config.seq_len
        tok = 1
        for pos in [0, 1, 2, 3, 1, 7, 8]:                      # includes a rewritten position
            a = np.array(t.forward(tok, pos), copy=True)
            b = om.forward(tok, pos)
            assert_biteq(a, b, f"{name} pos {pos}")
            assert t.forward_argmax(tok, pos) == oracle.sample_argmax(b)
            tok = oracle.sample_argmax(b)
        assert_biteq(t.read_state("x"), om.tap_x(), "x after final norm")
    # opt-in tree-reduction mode: same tokens on this short run, logits within tolerance of the oracle
    om.reset()
    with q3.TransformerBuilder(path).with_ctx_length(ctx or None).with_strict(False).build() as t:
        for pos in range(4):
            a, b = np.array(t.forward(3, pos), copy=True), om.forward(3, pos)
            assert np.max(np.abs(a - b)) <= 2e-5, np.max(np.abs(a - b))


@pytest.mark.parametrize("name", ["qwen3-0.6b-dims-l2", "qwen3-4b-dims-l2", "qwen3-8b-dims-l2"])
def test_tolerance_mode_on_the_listed_layer_dims(q3, oracle, name):
    """Q3_FLAG_FAST on the shapes whose GEMV launches take the tree-fold instantiations (k_gemv FIN = 2, round 6) next to the tree
    RMSNorm / attention sums.  Stated tolerance: max |delta logit| <= 0.15 x the standard deviation of the oracle's logits on a
    2-layer stack.  The deviation is bimodal -- ~1e-6 x the logit scale while no re-quantized activation changes its int8 value, and
    1e-3 .. 6e-2 x once one does (the flip is amplified by the following W8A8 layers; measured max 0.061 x on the 8B dims) -- which
    is why this mode is opt-in and never the benchmark's `value`."""
    ck = q3.checkpoint
    path = os.path.join(os.environ.get("Q3_CKPT_DIR", "/tmp"), f"q3_{name}.bin")
    ck.ensure_synthetic_checkpoint(path, ck.SHAPES[name], seed=1235)
    om = oracle.OracleModel(path, 256)
    with q3.TransformerBuilder(path).with_ctx_length(256).with_strict(False).build() as t:
        tok = 3
        for pos in range(6):
            a, b = np.array(t.forward(tok, pos), copy=True), om.forward(tok, pos)
            assert np.all(np.isfinite(a))
            assert np.max(np.abs(a - b)) <= 0.15 * np.std(b), (name, pos, np.max(np.abs(a - b)), np.std(b))
            tok = oracle.sample_argmax(b)


@pytest.mark.parametrize("name", ["qwen3-4b-dims-l2", "qwen3-8b-dims-l2"])
def test_big_layer_dims_vs_oracle(q3, oracle, name):
    """BASELINE configs 3-5 use the 4B / 8B layer shapes: row lengths 2560 / 9728 (not multiples of 1 KiB),
    4096 / 12288 (several tiles per row), 32 heads over 8 kv heads, untied classifier.  2-layer, reduced-vocab
    variants keep the oracle in seconds; the chat-mode pattern (every prompt token forwarded, generation.rs:116-123)
    is used so that prefill positions are covered."""
    ck = q3.checkpoint
    path = os.path.join(os.environ.get("Q3_CKPT_DIR", "/tmp"), f"q3_{name}.bin")
    ck.ensure_synthetic_checkpoint(path, ck.SHAPES[name], seed=1235)
    om = oracle.OracleModel(path, 256)
    with q3.TransformerBuilder(path).with_ctx_length(256).build() as t:
        prompt = ck.iter_prompt_tokens(ck.SHAPES[name], 1235, 5)
        seen, last = [], {}

        def check(tok, pos, logits):
            assert_biteq(logits, om.forward(tok, pos), f"{name} forward({tok},{pos})")
            seen.append(pos)
            last["logits"] = logits

        toks, pos, _ = q3.chat_turn(t, prompt, 0, 4, on_logits=check)
        assert seen == list(range(9)) and pos == 9 and len(toks) == 4
        # the device-resident loop continues the same sequence: 3 more tokens, compared with the oracle
        tok = oracle.sample_argmax(last["logits"])
        want, ot = [], tok
        for p in range(9, 12):
            ot = oracle.sample_argmax(om.forward(ot, p))
            want.append(ot)
        assert t.generate_greedy(tok, 9, 3) == want


def test_0p6b_dims_every_position_to_the_split_vs_oracle(q3, oracle):
    """The headline shape's layer dimensions (dim 1024, hidden 3072, 16 heads over 8 kv heads) with 2 layers: a 270-token
    sequence forwarded from position 0 crosses every 64-timestep wave boundary of the short-context attention kernel, its
    32-timestep V register sets and the hand-over to the split kernels at pos 256.  Tokens, the final logits and both
    KV caches are compared with the oracle bit for bit (the latency-bound GEMV variants with the register fold of the
    group terms are the ones this shape selects)."""
    ck = q3.checkpoint
    name = "qwen3-0.6b-dims-l2"
    sh = ck.SHAPES[name]
    path = os.path.join(os.environ.get("Q3_CKPT_DIR", "/tmp"), f"q3_{name}.bin")
    ck.ensure_synthetic_checkpoint(path, sh, seed=77)
    n = 270
    om = oracle.OracleModel(path, 320)
    toks = [int(v) for v in np.random.default_rng(3).integers(0, sh.vocab_size, size=n)]
    for p in range(n):                                      # chat-mode pattern: every token forwarded (generation.rs:116-123)
        lg = om.forward(toks[p], p)
    nxt = oracle.sample_argmax(lg)
    want, tk = [], nxt
    for p in range(n, n + 6):
        tk = oracle.sample_argmax(om.forward(tk, p))
        want.append(tk)
    with q3.TransformerBuilder(path).with_ctx_length(320).build() as t:
        assert t.prefill(toks, 0) == nxt
        assert t.generate_greedy(nxt, n, 6) == want
        n = n + 6
        kvd = sh.n_kv_heads * sh.head_dim
        ok, ov = om.kv_cache()
        for layer in range(sh.n_layers):
            assert_biteq(t.read_state("key", layer * 320 * kvd, n * kvd), ok.reshape(sh.n_layers, -1)[layer][:n * kvd], "key cache")
            assert_biteq(t.read_state("value", layer * 320 * kvd, n * kvd), ov.reshape(sh.n_layers, -1)[layer][:n * kvd], "value cache")
        t.reset_kv()
        om.reset()
        for p in (0, 1, 2):                                 # the forward() surface on a fresh cache
            assert_biteq(np.array(t.forward(5, p), copy=True), om.forward(5, p), f"forward(5,{p})")


def test_4b_dims_long_context_split_vs_oracle(q3, oracle):
    """BASELINE config 3's layer dimensions (dim 2560, 32 heads over 8 kv heads, head_dim 128) with 2 layers, past the
    split point: the scores kernel that stages each K chunk once for the four query heads of a kv head runs with full
    64-timestep chunks and a ragged last one that holds the new row.  Tokens, logits and both caches vs the oracle."""
    ck = q3.checkpoint
    name = "qwen3-4b-dims-l2"
    sh = ck.SHAPES[name]
    path = os.path.join(os.environ.get("Q3_CKPT_DIR", "/tmp"), f"q3_{name}.bin")
    ck.ensure_synthetic_checkpoint(path, sh, seed=78)
    n, ctx = 262, 336
    om = oracle.OracleModel(path, ctx)
    toks = [int(v) for v in np.random.default_rng(4).integers(0, sh.vocab_size, size=n)]
    for p in range(n):
        lg = om.forward(toks[p], p)
    nxt = oracle.sample_argmax(lg)
    with q3.TransformerBuilder(path).with_ctx_length(ctx).build() as t:
        assert t.prefill(toks, 0) == nxt
        tk = nxt
        for p in range(n, n + 62):                          # crosses the 320-timestep chunk boundary
            want = om.forward(tk, p)
            assert_biteq(np.array(t.forward(tk, p), copy=True), want, f"forward({tk},{p})")
            tk = oracle.sample_argmax(want)
        n = n + 62
        kvd = sh.n_kv_heads * sh.head_dim
        ok, ov = om.kv_cache()
        for layer in range(sh.n_layers):
            assert_biteq(t.read_state("key", layer * ctx * kvd, n * kvd), ok.reshape(sh.n_layers, -1)[layer][:n * kvd], "key cache")
            assert_biteq(t.read_state("value", layer * ctx * kvd, n * kvd), ov.reshape(sh.n_layers, -1)[layer][:n * kvd], "value cache")


def test_long_context_split_attention_vs_oracle(q3, oracle, tmp_ckpt_dir):
    """pos >= 256 switches the engine to the long-context launch plan (scores over heads x T-chunks, softmax + V over
    heads x element slices).  Logits stay bit-identical across the switch, at chunk boundaries and deep into the
    context; the device-resident greedy loop crosses the threshold mid-run."""
    ck = q3.checkpoint
    path = os.path.join(tmp_ckpt_dir, "small-longctx.bin")
    ck.ensure_synthetic_checkpoint(path, ck.SHAPES["small-longctx"], seed=5)
    om = oracle.OracleModel(path)
    with q3.TransformerBuilder(path).build() as t:
        tok = 7
        for pos in [0, 1, 254, 255, 256, 257, 383, 384, 385, 700, 1500, 2047, 300]:
            a = np.array(t.forward(tok, pos), copy=True)
            b = om.forward(tok, pos)
            assert_biteq(a, b, f"pos {pos}")
            tok = oracle.sample_argmax(b)
        t.reset_kv()
        om.reset()
        want, ot = [], 11
        for p in range(250, 262):
            ot = oracle.sample_argmax(om.forward(ot, p))
            want.append(ot)
        assert t.generate_greedy(11, 250, 12) == want
    with q3.TransformerBuilder(path).with_value_transposed(False).build() as t:   # Q3_FLAG_NO_VALUE_T: row-major value cache only
        om.reset()
        tok = 7
        for pos in [0, 255, 256, 257, 384, 700, 2047, 300]:
            a, b = np.array(t.forward(tok, pos), copy=True), om.forward(tok, pos)
            assert_biteq(a, b, f"no transposed value cache, pos {pos}")
            tok = oracle.sample_argmax(b)
    with q3.TransformerBuilder(path).with_strict(False).build() as t:      # opt-in tree mode through the split path
        om.reset()                                                          # (never allocates the transposed value cache)
        for pos in (300, 301, 900):
            a, b = np.array(t.forward(3, pos), copy=True), om.forward(3, pos)
            assert np.max(np.abs(a - b)) <= 2e-5


def test_context_beyond_lds_score_rows_vs_oracle(q3, oracle, tmp_path_factory):
    """seq_len > 4096: the single-kernel attention keeps its score rows in HBM instead of LDS (short positions), and the
    split plan takes over at pos >= 256.  Both stay bit-identical to the oracle; q3_profile reports every family."""
    import dataclasses
    ck = q3.checkpoint
    shape = dataclasses.replace(ck.SHAPES["small-longctx"], max_seq_len=8192)
    path = str(tmp_path_factory.mktemp("ctx8k") / "small-ctx8k.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=6)
    om = oracle.OracleModel(path)
    with q3.TransformerBuilder(path).build() as t:
        assert t.get_config().seq_len == 8192
        tok = 9
        for pos in [0, 1, 2, 100, 255, 256, 5000, 8191]:
            a = np.array(t.forward(tok, pos), copy=True)
            b = om.forward(tok, pos)
            assert_biteq(a, b, f"pos {pos}")
            tok = oracle.sample_argmax(b)
        for pos in (3, 4000):
            prof = t.profile(5, pos, 2)
            names = [n for n, _, _ in prof]
            assert names == ["qkv", "attn", "wo", "w13", "w2", "lm_head", "next"]
            L = shape.n_layers
            launches = {n: k for n, _, k in prof}
            # `next`: 0 launches when the state bookkeeping is folded into the classifier launch (round 3 default), else one
            assert launches["qkv"] == 2 * L and launches["lm_head"] == 2 and launches["next"] in (0, 2)
            assert launches["attn"] == (2 * L if pos < 256 else 4 * L)      # split plan: scores + out kernels
            assert all(ms > 0 for _, ms, k in prof if k > 0)


def test_device_prefill_matches_chat_pattern(q3, oracle):
    """q3_prefill == the prompt loop of `chat` (generation.rs:116-123): same KV rows, same first generated token, and
    decoding from there reproduces the oracle's chat turn."""
    path = golden_path("tiny-untied.bin")
    g = np.load(golden_path("tiny-untied.golden.npz"))
    prompt = [int(v) for v in g["prompt"]]
    om = oracle.OracleModel(path)
    want, pos, _ = q3.chat_turn(om, prompt, 0, 10, sample=oracle.sample_argmax)
    assert want == [int(v) for v in g["chat_tokens"]]
    with q3.TransformerBuilder(path).build() as t:
        first = t.prefill(prompt, 0)
        assert first == want[0]
        rest = t.generate_greedy(first, len(prompt), 9)
        assert [first] + rest == want
        k, v = om.kv_cache()
        n = len(prompt) + 9
        kvd = t.get_config().n_kv_heads * t.get_config().head_dim
        S = t.get_config().seq_len
        for layer in range(t.get_config().n_layers):
            assert_biteq(t.read_state("key", layer * S * kvd, n * kvd), k[layer, :n].reshape(-1), f"K layer {layer}")
        with pytest.raises(IndexError):
            t.prefill([1, 2, 999999], 0)


# ---------------------------------------------------------------------------------------------------------------
# Batched decode (include/qwen3_hip.h section 2b): every stream bit-identical to its single-stream run
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape_name,n_streams", [("tiny-g64", 3), ("tiny-g64", 32), ("small-hd128", 16), ("small-hd128", 32)])
def test_batched_decode_is_bit_identical_per_stream(q3, shape_name, n_streams, tmp_path_factory):
    """q3_forward_batch / q3_generate_greedy_batch: int8 MFMA over all streams, per-stream logits bit-identical to
    q3_forward on a fresh engine (same (token, pos) sequence), greedy tokens identical, KV rows identical."""
    ck = q3.checkpoint
    shape = ck.SHAPES[shape_name]
    path = str(tmp_path_factory.mktemp("bat") / f"{shape_name}.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=4321)
    rng = np.random.default_rng(11)
    toks0 = [int(t) for t in rng.integers(0, shape.vocab_size, n_streams)]
    pos0 = [int(p) for p in rng.integers(0, 6, n_streams)]      # ragged start positions, zero KV prefix
    steps = 5
    ref_logits, ref_tokens, ref_k = [], [], []
    for i in range(min(n_streams, 6)):                           # the first streams are checked against single-stream runs
        with q3.TransformerBuilder(path).build() as t:
            tok, ll, tt = toks0[i], [], []
            for k in range(steps):
                lg = np.array(t.forward(tok, pos0[i] + k), copy=True)
                ll.append(lg)
                tok = q3.sample_argmax(lg)
                tt.append(tok)
            ref_logits.append(ll)
            ref_tokens.append(tt)
            ref_k.append((t.read_state("key"), t.read_state("value")))
    with q3.TransformerBuilder(path).build() as t:
        t.batch_init(n_streams)
        toks = list(toks0)
        for k in range(steps):
            lg, am = t.forward_batch(toks, [p + k for p in pos0])
            for i in range(len(ref_logits)):
                assert_biteq(lg[i], ref_logits[i][k], f"stream {i} step {k} logits")
                assert am[i] == ref_tokens[i][k]
            toks = am
        for i in range(len(ref_logits)):
            assert_biteq(t.batch_read_state(i, "key"), ref_k[i][0], f"stream {i} key cache")
            assert_biteq(t.batch_read_state(i, "value"), ref_k[i][1], f"stream {i} value cache")
        t.batch_reset_kv()
        out = t.generate_greedy_batch(toks0, pos0, steps)
        for i in range(len(ref_tokens)):
            assert [int(v) for v in out[i]] == ref_tokens[i]
        # a smaller batch on the same engine re-plans and still matches; the single-stream path is untouched
        t.batch_reset_kv()
        out2 = t.generate_greedy_batch(toks0[:2], pos0[:2], steps)
        assert [int(v) for v in out2[0]] == ref_tokens[0] and [int(v) for v in out2[1]] == ref_tokens[1]
        lg1 = np.array(t.forward(toks0[0], pos0[0]), copy=True)
        assert_biteq(lg1, ref_logits[0][0], "single-stream forward after batched use")


def test_batched_decode_long_positions(q3, tmp_path_factory):
    """Positions past 256 (multi-chunk K/V staging, speculative exact softmax sum in the per-kv-head attention kernel):
    330 greedy steps per stream equal the single-stream run, and so do the logits of one more forward."""
    ck = q3.checkpoint
    shape = ck.SHAPES["small-longctx"]
    path = str(tmp_path_factory.mktemp("bat") / "small-longctx.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=77)
    toks0, pos0, steps = [3, 99, 250, 17], [0, 5, 1, 40], 330
    with q3.TransformerBuilder(path).with_ctx_length(512).build() as t:
        ref, ref_logits = [], []
        for i in range(4):
            t.reset_kv()
            ref.append(t.generate_greedy(toks0[i], pos0[i], steps))
            ref_logits.append(np.array(t.forward(ref[i][-1], pos0[i] + steps), copy=True))
        t.batch_init(4)
        out = t.generate_greedy_batch(toks0, pos0, steps)
        for i in range(4):
            assert [int(v) for v in out[i]] == ref[i], f"stream {i}"
        lg, _ = t.forward_batch([r[-1] for r in ref], [p + steps for p in pos0])
        for i in range(4):
            assert_biteq(lg[i], ref_logits[i], f"stream {i} logits at pos {pos0[i] + steps}")


@pytest.mark.parametrize("shape_name,n_prompt,first_pos", [("tiny-g64", 41, 0), ("small-hd128", 70, 3), ("small-longctx", 333, 0)])
def test_batched_prefill_is_sequential_equivalent(q3, shape_name, n_prompt, first_pos, tmp_path_factory):
    """q3_prefill_batched (up to 256 positions per weight pass, shared KV cache) == q3_prefill == the prompt loop of `chat`
    (generation.rs:116-123): bit-identical cache rows, same first generated token, same continuation."""
    ck = q3.checkpoint
    shape = ck.SHAPES[shape_name]
    path = str(tmp_path_factory.mktemp("pre") / f"{shape_name}.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=99)
    prompt = ck.iter_prompt_tokens(shape, 5, n_prompt)
    with q3.TransformerBuilder(path).with_ctx_length(512).build() as t:
        want_first = t.prefill(prompt, first_pos)
        want_rest = t.generate_greedy(want_first, first_pos + n_prompt, 6)
        want_k, want_v = t.read_state("key"), t.read_state("value")
    with q3.TransformerBuilder(path).with_ctx_length(512).build() as t:
        got_first = t.prefill(prompt, first_pos, batched=True)
        assert got_first == want_first
        rest = t.generate_greedy(got_first, first_pos + n_prompt, 6)
        assert rest == want_rest
        assert_biteq(t.read_state("key"), want_k, "key cache after batched prefill + decode")
        assert_biteq(t.read_state("value"), want_v, "value cache after batched prefill + decode")
        # the batched decode API still works on the same engine afterwards (re-plans, own KV caches)
        t.batch_init(2)
        out = t.generate_greedy_batch([prompt[0], prompt[1]], [0, 0], 3)
        assert out.shape == (2, 3)
        with pytest.raises(IndexError):
            t.prefill([1, 2, 10 ** 7], 0, batched=True)


@pytest.mark.parametrize("shape_name", ["small-longctx", "qwen3-0.6b-dims-l2"])      # head_dim 64 (k_attn_gqa) / 128 (k_attn_pf)
@pytest.mark.parametrize("block", [32, 48, 128, 256, -128, -256, 512, -512])
def test_batched_prefill_block_sizes_agree(q3, block, shape_name, tmp_path_factory, monkeypatch, dev_forms):
    """Q3_PREFILL_M picks the positions per weight pass: 32 = the batch-32 kernels (k_bgemm + LDS term tile), larger blocks
    the dense kernels (k_pgemm in-lane fold, k_attn_pf).  Every block size must give the cache rows and tokens of the
    sequential prompt loop (generation.rs:116-123) bit for bit, including a ragged last block and a non-zero start.
    Blocks of 128 / 256 run the LDS-tiled matmul k_pgemm2 with 4 x 4 position-tile workgroups by default on these shapes;
    negative block: the same size with its other forms -- -128: 2 row tiles per workgroup (Q3_PGEMM2_RT=2), -256: 4 x 8 tiles
    (Q3_PGEMM2_PT=8), -512: 8 x 8 tiles (Q3_PGEMM3_RT=8); 48 positions (3 tiles) stay on k_pgemm."""
    if block == -128:
        dev_forms({"Q3_PGEMM2_RT": "2"})
    if block == -256:
        dev_forms({"Q3_PGEMM2_PT": "8"})
    if block == -512:                                   # 8 x 8 workgroup tiles (k_pgemm3<.., 8, 8, 1>)
        dev_forms({"Q3_PGEMM2_PT": "8", "Q3_PGEMM3_RT": "8"})
    block = abs(block)
    ck = q3.checkpoint
    shape = ck.SHAPES[shape_name]
    path = str(tmp_path_factory.mktemp("preb") / "m.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=321)
    prompt = ck.iter_prompt_tokens(shape, 9, 301)
    with q3.TransformerBuilder(path).with_ctx_length(512).build() as t:
        t.prefill(prompt[:5], 0)                      # positions 0..4 by the sequential path in both runs
        want_first = t.prefill(prompt[5:], 5)
        want_rest = t.generate_greedy(want_first, len(prompt), 5)
        want_k, want_v = t.read_state("key"), t.read_state("value")
    monkeypatch.setenv("Q3_PREFILL_M", str(block))
    with q3.TransformerBuilder(path).with_ctx_length(512).build() as t:
        t.prefill(prompt[:5], 0)
        got_first = t.prefill(prompt[5:], 5, batched=True)
        assert got_first == want_first
        assert t.generate_greedy(got_first, len(prompt), 5) == want_rest
        assert_biteq(t.read_state("key"), want_k, f"key cache, block {block}")
        assert_biteq(t.read_state("value"), want_v, f"value cache, block {block}")


@pytest.mark.parametrize("n_block", [81, 97, 255, 257, 383])
def test_dense_prefill_ragged_block_lengths(q3, n_block, tmp_path_factory):
    """One dense block of n positions for n around the kernels' switch points: 81 is the shortest dense block (6 position tiles), 97 is
    odd (a ragged last position tile; 97 x 24 head vectors = 36 full workgroups + 24 vectors of k_knorm_rope_blk), 255 / 257 straddle the
    prologue launches' switch from four workgroups per position to one, 383 leaves position tiles and 8-tile groups ragged at once.  K / V
    rows, the first token and five more against the sequential prompt loop (generation.rs:116-123), bit for bit."""
    ck = q3.checkpoint
    shape = ck.SHAPES["qwen3-0.6b-dims-l2"]
    path = str(tmp_path_factory.mktemp("prer") / "m.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=77)
    prompt = ck.iter_prompt_tokens(shape, 11, 5 + n_block)
    with q3.TransformerBuilder(path).with_ctx_length(512).build() as t:
        t.prefill(prompt[:5], 0)
        want_first = t.prefill(prompt[5:], 5)
        want_rest = t.generate_greedy(want_first, len(prompt), 5)
        want_k, want_v = t.read_state("key"), t.read_state("value")
    with q3.TransformerBuilder(path).with_ctx_length(512).build() as t:
        t.prefill(prompt[:5], 0)
        got_first = t.prefill(prompt[5:], 5, batched=True)
        assert got_first == want_first
        assert t.generate_greedy(got_first, len(prompt), 5) == want_rest
        assert_biteq(t.read_state("key"), want_k, f"key cache, {n_block} positions")
        assert_biteq(t.read_state("value"), want_v, f"value cache, {n_block} positions")


def test_batched_prefill_kv_and_token_vs_oracle(q3, oracle, tmp_path_factory):
    """The batched (int8 MFMA, 32 positions per weight pass) prefill against the ORACLE directly, not only against the
    sequential HIP path: every K/V row of a 70-token chat-mode prompt (generation.rs:116-123) and the first generated
    token, bit for bit."""
    ck = q3.checkpoint
    shape = ck.SHAPES["small-hd128"]
    path = str(tmp_path_factory.mktemp("preo") / "small-hd128.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=31)
    prompt = ck.iter_prompt_tokens(shape, 8, 70)
    om = oracle.OracleModel(path, 128)
    for p, tok in enumerate(prompt):
        lg = om.forward(tok, p)
    ok, ov = om.kv_cache()
    kvd = shape.n_kv_heads * shape.head_dim
    with q3.TransformerBuilder(path).with_ctx_length(128).build() as t:
        assert t.prefill(prompt, 0, batched=True) == oracle.sample_argmax(lg)
        for layer in range(shape.n_layers):
            assert_biteq(t.read_state("key", layer * 128 * kvd, 70 * kvd), ok[layer].reshape(-1)[:70 * kvd], f"key rows layer {layer}")
            assert_biteq(t.read_state("value", layer * 128 * kvd, 70 * kvd), ov[layer].reshape(-1)[:70 * kvd], f"value rows layer {layer}")


# k_dgemm launch forms (round 4; the plan is built at batch_init, the knobs are read from the environment then): the default
# (residual launches), every family in-lane with the fused hq quantizer and both ring depths, the one-stream-tile-per-workgroup
# and three-wave W1|W3 forms, and k_bgemm everywhere
_DGEMM_FORMS = [{}, {"Q3_DGEMM_FAMILIES": "15"}, {"Q3_DGEMM_FAMILIES": "15", "Q3_DGEMM_DEEP": "0"},
                {"Q3_DGEMM_FAMILIES": "15", "Q3_DGEMM_W13_MODE": "0"}, {"Q3_DGEMM_FAMILIES": "15", "Q3_DGEMM_W13_MODE": "2"},
                {"Q3_DGEMM_FAMILIES": "15", "Q3_DGEMM_BLDS": "0"}, {"Q3_BATCH_DGEMM": "0"}]


@pytest.mark.parametrize("form", range(len(_DGEMM_FORMS)))
def test_batched_decode_8b_layer_dims_vs_forward_and_oracle(q3, oracle, form, dev_forms):
    """BASELINE config 4's matrix shapes on the batched path: row lengths 4096 / 12288 (64 and 192 groups per row), 32
    heads over 8 kv heads, untied classifier -- 2 layers, reduced vocabulary.  32 streams x 8 steps: every stream's
    logits bit-identical to q3_forward on the same (token, pos) sequence for the first 4 streams, and 2 streams
    against the oracle.  Run for every matmul form of the batched path (_DGEMM_FORMS)."""
    dev_forms(_DGEMM_FORMS[form])
    ck = q3.checkpoint
    name = "qwen3-8b-dims-l2"
    shape = ck.SHAPES[name]
    path = os.path.join(os.environ.get("Q3_CKPT_DIR", "/tmp"), f"q3_{name}.bin")
    ck.ensure_synthetic_checkpoint(path, shape, seed=1235)
    n_streams, steps = 32, 8
    rng = np.random.default_rng(5)
    toks0 = [int(t) for t in rng.integers(0, shape.vocab_size, n_streams)]
    pos0 = [int(p) for p in rng.integers(0, 9, n_streams)]
    with q3.TransformerBuilder(path).with_ctx_length(64).build() as t:
        ref = []
        for i in range(4):
            t.reset_kv()
            tok, ll = toks0[i], []
            for k in range(steps):
                lg = np.array(t.forward(tok, pos0[i] + k), copy=True)
                ll.append(lg)
                tok = q3.sample_argmax(lg)
            ref.append(ll)
        t.batch_init(n_streams, 64)
        toks, all_lg = list(toks0), []
        for k in range(steps):
            lg, am = t.forward_batch(toks, [p + k for p in pos0])
            all_lg.append(np.array(lg, copy=True))
            for i in range(4):
                assert_biteq(lg[i], ref[i][k], f"stream {i} step {k}")
            toks = am
    om = oracle.OracleModel(path, 64)
    for i in (7, 31):
        om.reset()
        tok = toks0[i]
        for k in range(steps):
            lg = om.forward(tok, pos0[i] + k)
            assert_biteq(all_lg[k][i], lg, f"stream {i} step {k} vs oracle")
            tok = oracle.sample_argmax(lg)


@pytest.mark.parametrize("form", [0, 1, 2, 6])
def test_batched_decode_4b_layer_dims_vs_forward(q3, form, dev_forms):
    """The 4B matrix shapes on the batched path: row lengths 2560 / 9728 are 40 / 152 quantization groups -- not a multiple of
    16, so k_dgemm runs its 8-group ring (and k_bgemm ragged phases) -- 20 streams (a ragged second stream tile), 6 steps: logits
    of four streams bit-identical to q3_forward.  Forms as in _DGEMM_FORMS (default, every family in-lane at both depths, k_bgemm)."""
    dev_forms(_DGEMM_FORMS[form])
    ck = q3.checkpoint
    name = "qwen3-4b-dims-l2"
    shape = ck.SHAPES[name]
    path = os.path.join(os.environ.get("Q3_CKPT_DIR", "/tmp"), f"q3_{name}.bin")
    ck.ensure_synthetic_checkpoint(path, shape, seed=1235)
    n_streams, steps = 20, 6
    rng = np.random.default_rng(6)
    toks0 = [int(t) for t in rng.integers(0, shape.vocab_size, n_streams)]
    pos0 = [int(p) for p in rng.integers(0, 9, n_streams)]
    with q3.TransformerBuilder(path).with_ctx_length(64).build() as t:
        ref = {}
        for i in (0, 7, 16, 19):
            t.reset_kv()
            tok, ll = toks0[i], []
            for k in range(steps):
                lg = np.array(t.forward(tok, pos0[i] + k), copy=True)
                ll.append(lg)
                tok = q3.sample_argmax(lg)
            ref[i] = ll
        t.batch_init(n_streams, 64)
        toks = list(toks0)
        for k in range(steps):
            lg, am = t.forward_batch(toks, [p + k for p in pos0])
            for i in ref:
                assert_biteq(lg[i], ref[i][k], f"stream {i} step {k}")
            toks = am


def test_full_size_8b_batch32_streams_equal_single_stream(q3, oracle):
    """BASELINE config 4 at FULL size (Qwen3-8B shape, 36 layers, vocab 151936, untied): 32 concurrent greedy streams x
    16 steps; two of the streams re-run single-stream on the same engine must give the same tokens (the size-independent
    property; the oracle needs seconds per 8B token, so it checks the first two tokens of one stream only)."""
    ck = q3.checkpoint
    shape = ck.SHAPES["qwen3-8b"]
    path = os.path.join(os.environ.get("Q3_CKPT_DIR", "/tmp"), "qwen3-8b-seed1236.q3bin")    # shared with tools/bench_batch.py
    ck.ensure_synthetic_checkpoint(path, shape, seed=1236)
    prompts = [ck.iter_prompt_tokens(shape, 1236 + 1000 + i, 8) for i in range(32)]
    first_tok, first_pos = [p[-1] for p in prompts], [len(p) - 1 for p in prompts]
    with q3.TransformerBuilder(path).with_ctx_length(64).build() as t:
        t.batch_init(32, 64)
        out = t.generate_greedy_batch(first_tok, first_pos, 16)
        assert out.shape == (32, 16)
        for i in (0, 19):
            t.reset_kv()
            assert t.generate_greedy(first_tok[i], first_pos[i], 16) == [int(v) for v in out[i]], f"stream {i}"
        assert len({tuple(int(v) for v in row) for row in out}) > 16      # the streams really are different sequences
    om = oracle.OracleModel(path, 64)
    tok = first_tok[19]
    for k in range(2):
        tok = oracle.sample_argmax(om.forward(tok, first_pos[19] + k))
        assert tok == int(out[19][k]), f"stream 19 token {k} vs oracle"
    om.close()


def test_full_size_4b_batched_prefill_256_tokens(q3, oracle):
    """BASELINE config 3 at FULL size (Qwen3-4B shape: dim 2560 / hidden 9728 are not multiples of 1 KiB, 36 layers): a
    256-token batched prefill writes the same K/V rows as the sequential device loop (checked on the first and last
    layer) and returns the same first token; rows 0..2 and the logits-side argmax chain are checked against the oracle
    for the first three positions."""
    ck = q3.checkpoint
    shape = ck.SHAPES["qwen3-4b"]
    path = os.path.join(os.environ.get("Q3_CKPT_DIR", "/tmp"), "qwen3-4b-seed1235.q3bin")     # shared with tools/bench_chat.py
    ck.ensure_synthetic_checkpoint(path, shape, seed=1235)
    prompt = ck.iter_prompt_tokens(shape, 1235, 256)
    kvd = shape.n_kv_heads * shape.head_dim
    L = shape.n_layers
    with q3.TransformerBuilder(path).with_ctx_length(320).build() as t:
        first_seq = t.prefill(prompt, 0)
        want = {l: (t.read_state("key", l * 320 * kvd, 256 * kvd), t.read_state("value", l * 320 * kvd, 256 * kvd)) for l in (0, L - 1)}
        t.reset_kv()
        assert t.prefill(prompt, 0, batched=True) == first_seq
        for l in (0, L - 1):
            assert_biteq(t.read_state("key", l * 320 * kvd, 256 * kvd), want[l][0], f"key rows layer {l}")
            assert_biteq(t.read_state("value", l * 320 * kvd, 256 * kvd), want[l][1], f"value rows layer {l}")
        got3 = {l: (t.read_state("key", l * 320 * kvd, 3 * kvd), t.read_state("value", l * 320 * kvd, 3 * kvd)) for l in (0, L - 1)}
    om = oracle.OracleModel(path, 320)
    for p in range(3):
        om.forward(prompt[p], p)
    ok, ov = om.kv_cache()
    for l in (0, L - 1):
        assert_biteq(got3[l][0], ok[l].reshape(-1)[:3 * kvd], f"oracle key rows layer {l}")
        assert_biteq(got3[l][1], ov[l].reshape(-1)[:3 * kvd], f"oracle value rows layer {l}")


def test_config3_decode_past_a_2048_token_prefill_vs_oracle(q3, oracle):
    """BASELINE config 3 past the prefill, against the ORACLE: the 4B layer dimensions (2 layers, reduced vocabulary) at
    ctx 4096 -- a 2,048-token batched prefill (chat pattern, generation.rs:116-123) followed by 8 decode forwards at
    positions 2,048..2,055 (the split long-context attention kernels over a prefilled cache, layers.rs:388-417).  First
    token, every decode step's logits and the K/V rows around the hand-over are bit-identical."""
    ck = q3.checkpoint
    name = "qwen3-4b-dims-l2"
    shape = ck.SHAPES[name]
    path = os.path.join(os.environ.get("Q3_CKPT_DIR", "/tmp"), f"q3_{name}.bin")
    ck.ensure_synthetic_checkpoint(path, shape, seed=1235)
    S, n_prompt, n_dec = 4096, 2048, 8
    prompt = ck.iter_prompt_tokens(shape, 1235, n_prompt)
    om = oracle.OracleModel(path, S)
    for p, tok in enumerate(prompt):
        lg = om.forward(tok, p)
    want_first = oracle.sample_argmax(lg)
    kvd = shape.n_kv_heads * shape.head_dim
    with q3.TransformerBuilder(path).with_ctx_length(S).build() as t:
        assert t.prefill(prompt, 0, batched=True) == want_first
        ok, ov = om.kv_cache()
        for layer in range(shape.n_layers):
            for lo, hi in ((0, 40), (1000, 1040), (n_prompt - 40, n_prompt)):
                assert_biteq(t.read_state("key", (layer * S + lo) * kvd, (hi - lo) * kvd),
                             ok[layer].reshape(-1)[lo * kvd:hi * kvd], f"prefill key rows {lo}..{hi} layer {layer}")
                assert_biteq(t.read_state("value", (layer * S + lo) * kvd, (hi - lo) * kvd),
                             ov[layer].reshape(-1)[lo * kvd:hi * kvd], f"prefill value rows {lo}..{hi} layer {layer}")
        tok = want_first
        for k in range(n_dec):
            pos = n_prompt + k
            a = np.array(t.forward(tok, pos), copy=True)
            b = om.forward(tok, pos)
            assert_biteq(a, b, f"decode logits at pos {pos}")
            tok = oracle.sample_argmax(b)
        ok, ov = om.kv_cache()
        for layer in range(shape.n_layers):
            lo, hi = n_prompt - 4, n_prompt + n_dec
            assert_biteq(t.read_state("key", (layer * S + lo) * kvd, (hi - lo) * kvd),
                         ok[layer].reshape(-1)[lo * kvd:hi * kvd], f"key rows across the hand-over, layer {layer}")
            assert_biteq(t.read_state("value", (layer * S + lo) * kvd, (hi - lo) * kvd),
                         ov[layer].reshape(-1)[lo * kvd:hi * kvd], f"value rows across the hand-over, layer {layer}")


def test_full_size_4b_prefill_2048_then_decode_properties(q3):
    """BASELINE config 3 at FULL size (Qwen3-4B shape, 36 layers, ctx 4096): 2,048-token batched prefill + 16 decode tokens.
    Size-independent properties: the device-resident greedy loop == the forward()+argmax loop == a second run from a
    reset cache, and the prefill's first token is reproducible."""
    ck = q3.checkpoint
    shape = ck.SHAPES["qwen3-4b"]
    path = os.path.join(os.environ.get("Q3_CKPT_DIR", "/tmp"), "qwen3-4b-seed1235.q3bin")     # shared with tools/bench_chat.py
    ck.ensure_synthetic_checkpoint(path, shape, seed=1235)
    n_prompt, n_dec = 2048, 16
    prompt = ck.iter_prompt_tokens(shape, 1235, n_prompt)
    with q3.TransformerBuilder(path).with_ctx_length(4096).build() as t:
        first = t.prefill(prompt, 0, batched=True)
        loop = t.generate_greedy(first, n_prompt, n_dec)
        t.reset_kv()
        assert t.prefill(prompt, 0, batched=True) == first
        tok, manual = first, []
        for k in range(n_dec):
            lg = np.array(t.forward(tok, n_prompt + k), copy=True)
            assert np.all(np.isfinite(lg))
            tok = int(len(lg) - 1 - np.argmax(lg[::-1]))      # last maximum (sampler.rs:57-59)
            manual.append(tok)
        assert manual == loop
        t.reset_kv()
        assert t.prefill(prompt, 0, batched=True) == first
        assert t.generate_greedy(first, n_prompt, n_dec) == loop


def test_deepseek_header_two_layer_dims_vs_oracle(q3, oracle, tmp_path_factory):
    """BASELINE config 5's checkpoint header -- DeepSeek-R1-0528-Qwen3-8B: seq_len 131072 in the file, 8B layer dimensions,
    untied classifier (README.md:37) -- on a 2-layer variant with a reduced vocabulary: the context argument clamps seq_len
    (models/mod.rs:65-67) and the logits of the generate-mode call pattern (first forward at pos 7 over a zero KV prefix,
    generation.rs:26-29), including positions in the split long-context plan, are bit-identical to the oracle."""
    import dataclasses
    ck = q3.checkpoint
    shape = dataclasses.replace(ck.SHAPES["deepseek-r1-0528-qwen3-8b"], n_layers=2, vocab_size=16384)
    path = str(tmp_path_factory.mktemp("ds") / "deepseek-l2.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=53)
    with open(path, "rb") as f:
        hdr = oracle.read_config(f.read(256))
    assert hdr.seq_len == 131072 and not hdr.shared_classifier and hdr.dim == 4096 and hdr.hidden_dim == 12288
    ctx = 768
    om = oracle.OracleModel(path, ctx)
    with q3.TransformerBuilder(path).with_ctx_length(ctx).build() as t:
        assert t.get_config().seq_len == ctx
        tok = 11
        for pos in [7, 8, 9, 10, 300, 301, ctx - 1]:
            a = np.array(t.forward(tok, pos), copy=True)
            b = om.forward(tok, pos)
            assert_biteq(a, b, f"pos {pos}")
            tok = oracle.sample_argmax(b)
        with pytest.raises(IndexError):
            t.forward(1, ctx)


def test_full_size_deepseek_8b_header_16_token_determinism(q3):
    """BASELINE config 5's replica at FULL size (8.7 GB checkpoint with the 131072-position header, ctx 1024): 16 greedy
    tokens of the device-resident loop == the forward()+argmax loop == a second run from a reset cache."""
    ck = q3.checkpoint
    name = "deepseek-r1-0528-qwen3-8b"
    shape = ck.SHAPES[name]
    path = os.path.join(os.environ.get("Q3_CKPT_DIR", "/tmp"), f"q3_{name}.bin")      # shared with bench.py other_configs
    ck.ensure_synthetic_checkpoint(path, shape, seed=1234)
    prompt = ck.iter_prompt_tokens(shape, 1234, 8)
    first_tok, first_pos = prompt[-1], len(prompt) - 1
    with q3.TransformerBuilder(path).with_ctx_length(1024).build() as t:
        assert t.get_config().seq_len == 1024
        loop = t.generate_greedy(first_tok, first_pos, 16)
        t.reset_kv()
        tok, manual = first_tok, []
        for k in range(16):
            lg = np.array(t.forward(tok, first_pos + k), copy=True)
            assert np.all(np.isfinite(lg))
            tok = int(len(lg) - 1 - np.argmax(lg[::-1]))
            manual.append(tok)
        assert manual == loop
        t.reset_kv()
        assert t.generate_greedy(first_tok, first_pos, 16) == loop
        assert len(set(loop)) > 4


@pytest.mark.parametrize("n_heads,n_kv,hd", [(4, 4, 32), (8, 1, 64), (6, 2, 128)])
def test_batched_paths_other_head_layouts(q3, n_heads, n_kv, hd, tmp_path_factory):
    """kv_mul 1 (no sharing, head_dim < 64), kv_mul 8 (falls back to the per-head attention kernel; batched prefill then
    reports UNSUPPORTED and the sequential device loop is used) and kv_mul 3: batched decode / prefill stay bit-identical."""
    ck = q3.checkpoint
    dim = 128
    shape = ck.ModelShape(dim, 256, 2, n_heads, n_kv, 512, 160, hd, True, 64)
    path = str(tmp_path_factory.mktemp("heads") / "m.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=n_heads * 100 + n_kv)
    prompt = ck.iter_prompt_tokens(shape, 9, 45)
    with q3.TransformerBuilder(path).build() as t:
        ref = []
        for i in range(3):
            t.reset_kv()
            ref.append(t.generate_greedy(prompt[i], i, 40))
        t.reset_kv()
        want_first = t.prefill(prompt, 0)
        want_k = t.read_state("key")
        t.batch_init(3)
        out = t.generate_greedy_batch(prompt[:3], [0, 1, 2], 40)
        for i in range(3):
            assert [int(v) for v in out[i]] == ref[i], f"stream {i}"
        t.reset_kv()
        try:
            got_first = t.prefill(prompt, 0, batched=True)
            assert n_heads // n_kv <= 7
            assert got_first == want_first
            assert_biteq(t.read_state("key"), want_k, "key cache after batched prefill")
        except q3.Q3Error as err:
            assert err.code == -5 and n_heads // n_kv > 7


def test_batched_decode_with_per_stream_samplers(q3, tmp_path_factory):
    """q3_batch_sampler_set: stream i of the batch draws what a single-stream engine with q3_sampler_set(seed_i) draws."""
    ck = q3.checkpoint
    shape = ck.SHAPES["small-hd128"]
    path = str(tmp_path_factory.mktemp("bsamp") / "m.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=61)
    toks0, pos0, seeds, steps = [5, 77, 1000, 3], [0, 2, 1, 4], [11, 22, 33, 44], 24
    for temperature, topp in ((0.8, 0.9), (1.1, 1.0)):
        with q3.TransformerBuilder(path).build() as t:
            ref = []
            for i in range(4):
                t.reset_kv()
                t.set_sampler(temperature, topp, seeds[i])
                ref.append(t.generate_greedy(toks0[i], pos0[i], steps))
            t.set_sampler(0.0, topp, 0)
            t.batch_init(4)
            t.set_batch_sampler(temperature, topp, seeds)
            out = t.generate_greedy_batch(toks0, pos0, steps)
            for i in range(4):
                assert [int(v) for v in out[i]] == ref[i], f"stream {i} T={temperature}"
            # back to greedy: the argmax path again, no coin drawn
            t.set_batch_sampler(0.0, topp, seeds)
            t.batch_reset_kv()
            g = t.generate_greedy_batch(toks0, pos0, 4)
            t.reset_kv()
            assert [int(v) for v in g[0]] == t.generate_greedy(toks0[0], pos0[0], 4)


def test_batched_decode_error_behaviour(q3, tmp_path_factory):
    ck = q3.checkpoint
    path = str(tmp_path_factory.mktemp("bat") / "tiny-g64.bin")
    ck.write_synthetic_checkpoint(path, ck.SHAPES["tiny-g64"], seed=1)
    with q3.TransformerBuilder(path).build() as t:
        with pytest.raises(IndexError):
            t.forward_batch([1], [0])                             # q3_batch_init not called
        t.batch_init(4)
        with pytest.raises(IndexError):
            t.forward_batch([1, 2, 3, 4, 5], [0] * 5)             # more streams than allocated
        with pytest.raises(IndexError):
            t.forward_batch([10 ** 6], [0])                       # token out of range, like forward()
        with pytest.raises(IndexError):
            t.forward_batch([1], [10 ** 6])
        with pytest.raises(q3.Q3Error):
            t.batch_init(33)
    g16 = str(tmp_path_factory.mktemp("bat") / "tiny.bin")
    ck.write_synthetic_checkpoint(g16, ck.SHAPES["tiny"], seed=1)   # group 16: below one 64-byte MFMA step
    with q3.TransformerBuilder(g16).build() as t:
        with pytest.raises(q3.Q3Error) as ei:
            t.batch_init(2)
        assert ei.value.code == -5


# ---------------------------------------------------------------------------------------------------------------
# Device-side Sampler::sample (sampler.rs:118-139)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("temperature,topp", [(0.7, 0.9), (1.0, 1.0), (1.3, 0.0), (0.3, 0.5), (2.0, 0.99)])
def test_device_sampler_matches_oracle(q3, oracle, temperature, topp, tmp_path_factory):
    """q3_sampler_set + forward_argmax/generate: the device draws token by token exactly what Sampler::sample draws from
    the same logits and the same xorshift64* stream (softmax denominator, cdf and nucleus sums in index order)."""
    ck = q3.checkpoint
    shape = ck.SHAPES["small-hd128"]                                   # vocab 2048: top-p keeps hundreds of candidates
    path = str(tmp_path_factory.mktemp("samp") / "small-hd128.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=31)
    seed = 0x1234ABCD5678EF01
    om = oracle.OracleModel(path)
    smp = oracle.Sampler(shape.vocab_size, temperature, topp, seed)
    with q3.TransformerBuilder(path).build() as t:
        t.set_sampler(temperature, topp, seed)
        tok, want = 17, []
        for pos in range(0, 40):                                       # step by step: forward + sample on the device
            nxt = smp.sample(om.forward(tok, pos))
            got = t.forward_argmax(tok, pos)
            assert got == nxt, f"pos {pos}"
            want.append(nxt)
            tok = nxt
        assert t.sampler_rng_state() == smp.rng_state.value
        lg = np.array(t.forward(3, 41), copy=True)                      # q3_forward leaves sampling (and the rng) to the caller
        assert t.sampler_rng_state() == smp.rng_state.value and np.all(np.isfinite(lg))
        # the device-resident loop draws the same sequence
        t.reset_kv()
        t.set_sampler(temperature, topp, seed)
        assert t.generate_greedy(17, 0, 40) == want
        # temperature 0 restores the argmax path and draws no coin
        t.set_sampler(0.0, topp, 99)
        t.reset_kv()
        om.reset()
        lg = om.forward(5, 0)
        assert t.forward_argmax(5, 0) == oracle.sample_argmax(lg)
        assert t.sampler_rng_state() == 99


def test_device_sampler_large_vocab_and_prefill(q3, oracle, tmp_path_factory):
    """Vocabulary 16384 (many blocks per lane, thousands of nucleus candidates incl. exact ties) on raw logits through
    the full engine, and the chat-mode prefill: one discarded coin per prompt position, sequential and batched."""
    ck = q3.checkpoint
    shape = ck.SHAPES["qwen3-4b-dims-l2"]
    path = str(tmp_path_factory.mktemp("samp") / "4b-l2.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=8)
    om = oracle.OracleModel(path)
    prompt = ck.iter_prompt_tokens(shape, 3, 37)
    for temperature, topp in [(0.8, 0.95), (1.0, 1.0)]:
        seed = 424242
        smp = oracle.Sampler(shape.vocab_size, temperature, topp, seed)
        om.reset()
        want_tokens, pos, _ = q3.chat_turn(om, prompt, 0, 6, sample=smp.sample)
        with q3.TransformerBuilder(path).with_ctx_length(256).build() as t:
            for batched in (False, True):
                t.reset_kv()
                t.set_sampler(temperature, topp, seed)
                first = t.prefill(prompt, 0, batched=batched)
                assert first == want_tokens[0], f"batched={batched}"
                rest = t.generate_greedy(first, len(prompt), 5)
                assert [first] + rest == want_tokens, f"batched={batched}"


@pytest.mark.parametrize("shape_name", ["qwen3-4b-dims-l2", "tiny-g64"])      # vocabulary 16,384 / 512 (one partial workgroup range)
@pytest.mark.parametrize("pipeline", ["1", "0"])
def test_device_sampler_pipelined_and_single_kernel_forms_agree(q3, oracle, pipeline, shape_name, tmp_path_factory, dev_forms):
    """The engine's draw runs as a pipeline (exact sum | chip-wide normalise + histogram + compaction | sort + exact walks) or,
    with Q3_SAMPLER_PIPELINE=0, as the single-workgroup kernel the batched sampler and q3_op_sample use: both must give the
    oracle's tokens (sampler.rs:118-139), for a flat-ish and a peaked temperature, nucleus and plain multinomial."""
    if pipeline != "1":
        dev_forms({"Q3_SAMPLER_PIPELINE": pipeline})
    ck = q3.checkpoint
    shape = ck.SHAPES[shape_name]
    path = str(tmp_path_factory.mktemp("samp2") / "m.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=11)
    om = oracle.OracleModel(path)
    with q3.TransformerBuilder(path).with_ctx_length(96).build() as t:
        for temperature, topp in [(1.3, 0.9), (0.25, 0.95), (0.9, 0.3), (1.0, 1.0), (4.0, 0.999)]:
            seed = 99 + int(temperature * 100)
            smp = oracle.Sampler(shape.vocab_size, temperature, topp, seed)
            om.reset()
            tok, want = 7, []
            for pos in range(12):
                logits = np.array(om.forward(tok, pos), copy=True)
                tok = smp.sample(logits)
                want.append(tok)
            t.reset_kv()
            t.set_sampler(temperature, topp, seed)
            got = t.generate_greedy(7, 0, 12)
            assert got == want, f"T {temperature} top-p {topp} pipeline {pipeline}"


def test_dense_prefill_with_the_batched_attention_kernel(q3, tmp_path_factory, dev_forms):
    """Q3_PREFILL_ATT_PF=0 keeps the dense matmuls but runs the block's attention on k_attn_gqa2 (the fallback for layouts
    k_attn_pf2 does not cover): same cache rows and tokens as the sequential prompt loop."""
    dev_forms({"Q3_PREFILL_ATT_PF": "0"})
    ck = q3.checkpoint
    shape = ck.SHAPES["qwen3-0.6b-dims-l2"]
    path = str(tmp_path_factory.mktemp("pf0") / "m.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=5)
    prompt = ck.iter_prompt_tokens(shape, 4, 150)
    with q3.TransformerBuilder(path).with_ctx_length(256).build() as t:
        want = t.prefill(prompt, 0)
        want_rest = t.generate_greedy(want, len(prompt), 4)
        want_k, want_v = t.read_state("key"), t.read_state("value")
    with q3.TransformerBuilder(path).with_ctx_length(256).build() as t:
        got = t.prefill(prompt, 0, batched=True)
        assert got == want
        assert t.generate_greedy(got, len(prompt), 4) == want_rest
        assert_biteq(t.read_state("key"), want_k, "key cache")
        assert_biteq(t.read_state("value"), want_v, "value cache")


def test_engine_runs_an_exported_hf_checkpoint(q3, oracle, tmp_path):
    """safetensors (F32 + BF16, LoRA adapter) -> export.py -> engine: logits bit-identical to the oracle on that file."""
    from qwen3_rs_amd import export
    from test_export import build_model_dir
    d = str(tmp_path / "hf")
    build_model_dir(d, tied=True, with_qk_norm=True, lora_rank=4, seed=21)
    path = str(tmp_path / "exported.bin")
    export.export_model(d, path, 64)
    om = oracle.OracleModel(path)
    with q3.TransformerBuilder(path).build() as t:
        tok = 5
        for pos in range(8):
            a, b = np.array(t.forward(tok, pos), copy=True), om.forward(tok, pos)
            assert_biteq(a, b, f"pos {pos}")
            tok = oracle.sample_argmax(b)


def _random_shapes(n, seed):
    """seeded sweep over the supported shape space (group | every inner dimension, head_dim a power of two)"""
    rng = np.random.default_rng(seed)
    ck_shapes = []
    while len(ck_shapes) < n:
        G = int(rng.choice([16, 32, 64, 128]))
        hd = int(rng.choice([8, 16, 32, 64, 128]))
        n_kv = int(rng.choice([1, 2, 3, 4]))
        kv_mul = int(rng.choice([1, 2, 3, 4, 8]))
        n_heads = n_kv * kv_mul
        dim = G * int(rng.integers(1, 6))
        hidden = G * int(rng.integers(1, 9))
        if (n_heads * hd) % G or n_heads * hd > 1024:
            continue
        vocab = 16 * int(rng.integers(2, 40))
        L = int(rng.integers(1, 4))
        seq = int(rng.choice([24, 40, 72]))
        ck_shapes.append((dim, hidden, L, n_heads, n_kv, vocab, seq, hd, bool(rng.integers(0, 2)), G))
    return ck_shapes


@pytest.mark.parametrize("spec", _random_shapes(16, 2024) + _random_shapes(16, 7), ids=lambda s: "d{}h{}L{}H{}kv{}v{}s{}hd{}t{}g{}".format(*[int(x) for x in s]))
def test_random_shapes_vs_oracle(q3, oracle, spec, tmp_path):
    """A seeded sweep of model shapes (odd head layouts, every group size, tied / untied classifier): forward logits
    bit-identical to the oracle; where the batched paths support the shape, they match the single-stream engine too."""
    ck = q3.checkpoint
    shape = ck.ModelShape(*spec)
    path = str(tmp_path / "m.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=int(sum(int(x) for x in spec)))
    om = oracle.OracleModel(path)
    with q3.TransformerBuilder(path).build() as t:
        tok, toks = 3 % shape.vocab_size, []
        for pos in range(10):
            a, b = np.array(t.forward(tok, pos), copy=True), om.forward(tok, pos)
            assert_biteq(a, b, f"pos {pos}")
            tok = oracle.sample_argmax(b)
            toks.append(tok)
        t.reset_kv()
        assert t.generate_greedy(3 % shape.vocab_size, 0, 10) == toks
        try:
            t.batch_init(3)
        except q3.Q3Error as err:
            assert err.code == -5 and (shape.group_size < 64 or shape.dim % 16 or shape.hidden_dim % 16)
            return
        out = t.generate_greedy_batch([3 % shape.vocab_size] * 3, [0, 0, 0], 10)
        for i in range(3):
            assert [int(v) for v in out[i]] == toks
        t.reset_kv()
        prompt = [3 % shape.vocab_size] + toks[:8]
        want = t.prefill(prompt, 0)
        t.reset_kv()
        try:
            assert t.prefill(prompt, 0, batched=True) == want == toks[8]
        except q3.Q3Error as err:
            assert err.code == -5


def test_op_sample_random_distributions_vs_oracle(q3, oracle):
    """q3_op_sample on synthetic logits: peaked, flat, tied, tiny and 152k-entry vocabularies, top-p from 0 to 1 (the
    histogram preselection and its full-sort fallback, multi-segment exact sums): every draw equals the oracle's."""
    rng = np.random.default_rng(17)
    cases = []
    for n in (2, 17, 1000, 4096, 50000, 151936):
        for sigma in (0.05, 1.0, 6.0):
            cases.append((n, sigma))
    state = 0xC0FFEE
    for n, sigma in cases:
        lg = rng.normal(0.0, sigma, n).astype(np.float32)
        lg[rng.integers(0, n, min(n, 40))] = lg[0]                              # exact ties
        for temperature, topp in ((1.0, 1.0), (0.7, 0.9), (1.5, 0.3), (0.2, 0.999), (1.0, 0.0)):
            smp = oracle.Sampler(n, temperature, topp, state)
            want = smp.sample(lg)
            got, new_state = q3.ops.sample(lg, temperature, topp, state)
            assert got == want, (n, sigma, temperature, topp)
            assert new_state == smp.rng_state.value
            state = new_state


def test_two_engines_are_independent(q3):
    """Replicas: engines share nothing (own stream, KV cache, scratch, graphs).  Interleaving two engines -- here on one
    device -- gives each exactly the tokens it produces alone."""
    pa, pb = golden_path("tiny.bin"), golden_path("tiny-untied.bin")
    with q3.TransformerBuilder(pa).build() as a, q3.TransformerBuilder(pb).build() as b:
        solo_a = a.generate_greedy(5, 2, 20)
        solo_b = b.generate_greedy(9, 0, 20)
        a.reset_kv(); b.reset_kv()
        ta, tb, ia, ib = 5, 9, [], []
        for k in range(20):                      # strictly alternating forward_argmax calls
            ta = a.forward_argmax(ta, 2 + k); ia.append(ta)
            tb = b.forward_argmax(tb, k); ib.append(tb)
        assert ia == solo_a and ib == solo_b
        with q3.TransformerBuilder(pa).build() as a2:      # a third engine on the same checkpoint
            assert a2.generate_greedy(5, 2, 20) == solo_a


def test_engine_error_behaviour(q3, tmp_ckpt_dir):
    """Same failure surface as TransformerBuilder::build / forward in the reference."""
    with pytest.raises(q3.Q3Error, match="Failed to open checkpoint"):
        q3.TransformerBuilder(os.path.join(tmp_ckpt_dir, "nope.bin")).build()
    data = open(golden_path("tiny.bin"), "rb").read()
    p = os.path.join(tmp_ckpt_dir, "trunc_gpu.bin")
    open(p, "wb").write(data[: len(data) - 100])
    with pytest.raises(q3.Q3Error, match="Insufficient data"):
        q3.TransformerBuilder(p).build()
    p = os.path.join(tmp_ckpt_dir, "arch_gpu.bin")
    open(p, "wb").write(data[:8] + (9).to_bytes(4, "little") + data[12:])
    with pytest.raises(q3.Q3Error, match="Unknown architecture_id: 9"):
        q3.TransformerBuilder(p).build()
    with q3.TransformerBuilder(golden_path("tiny.bin")).with_ctx_length(8).build() as t:
        assert t.get_config().seq_len == 8
        with pytest.raises(IndexError):
            t.forward(0, 8)
        with pytest.raises(IndexError):
            t.forward(256, 0)
        with pytest.raises(IndexError):
            t.generate_greedy(1, 4, 5)


# ---- full-size Qwen3-0.6B shape ------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big(q3, tmp_path_factory):
    ck = q3.checkpoint
    path = os.path.join(os.environ.get("Q3_CKPT_DIR", "/tmp"), "q3_qwen3-0.6b.bin")
    ck.ensure_synthetic_checkpoint(path, ck.SHAPES["qwen3-0.6b"], seed=1234)
    return path


def test_0p6b_logits_bit_identical_and_tokens(q3, oracle, big):
    """BASELINE config 2 vs config 1 on the first tokens of the benchmark prompt (the oracle needs ~0.3 s per
    token on 8 cores, so the bit-exact comparison covers 6 tokens; the rest is covered by properties below)."""
    prompt = q3.checkpoint.iter_prompt_tokens(q3.checkpoint.SHAPES["qwen3-0.6b"], 1234, 8)
    om = oracle.OracleModel(big, 1024)
    with q3.TransformerBuilder(big).with_ctx_length(1024).build() as t:
        tok, pos = prompt[-1], len(prompt) - 1             # generate mode: first call at pos n-1, zero KV prefix
        for _ in range(6):
            a = np.array(t.forward(tok, pos), copy=True)
            b = om.forward(tok, pos)
            assert_biteq(a, b, f"0.6B logits at pos {pos}")
            tok, pos = oracle.sample_argmax(b), pos + 1


def test_0p6b_full_run_properties(q3, big):
    """128-token greedy run at full size: (1) the device-resident loop, the forward_argmax loop and the host
    loop over full logits produce the same tokens; (2) re-running from a reset cache reproduces them
    (idempotence / determinism); (3) every KV row below the first position stays exactly zero (generate-mode
    quirk, generation.rs:26-29) and rows at and above it are written."""
    sh = q3.checkpoint.SHAPES["qwen3-0.6b"]
    prompt = q3.checkpoint.iter_prompt_tokens(sh, 1234, 8)
    first_pos, first_tok = len(prompt) - 1, prompt[-1]
    with q3.TransformerBuilder(big).with_ctx_length(1024).build() as t:
        a = t.generate_greedy(first_tok, first_pos, 128)
        t.reset_kv()
        b = t.generate_greedy(first_tok, first_pos, 128)
        assert a == b
        t.reset_kv()
        host, _ = q3.generate(t, prompt, max_new_tokens=24)
        assert host == a[:24]
        t.reset_kv()
        tok, pos, c = first_tok, first_pos, []
        for _ in range(24):
            tok = t.forward_argmax(tok, pos)
            c.append(tok)
            pos += 1
        assert c == a[:24]
        kvd = sh.n_kv_heads * sh.head_dim
        for layer in (0, sh.n_layers - 1):
            base = layer * 1024 * kvd
            head = t.read_state("key", base, (first_pos + 2) * kvd).reshape(-1, kvd)
            assert not head[:first_pos].any() and head[first_pos].any()
