"""CPU tests: pin the oracle (and the product's host-side format code) against every known-answer vector
the reference's own tests hold for this path:
qwen3-export/tests/unit/model_exporter_test.rs:27-45, 48-87, 90-101, 104-134, 137-161, 397-427.
The forward pass has no reference vectors (qwen3-inference has zero tests): parity unpinned there; the two
independent restatements are checked against each other and against committed fixtures instead."""
import numpy as np
import pytest

from conftest import assert_biteq, golden_path


# ---- model_exporter_test.rs:27-45 ------------------------------------------------------------------
@pytest.mark.parametrize("x,want", [(1.4, 1.0), (1.6, 2.0), (-1.4, -1.0), (-1.6, -2.0),
                                    (0.5, 0.0), (1.5, 2.0), (2.5, 2.0), (3.5, 4.0), (-0.5, 0.0), (-1.5, -2.0), (-2.5, -2.0)])
def test_round_half_to_even_table(oracle, q3, x, want):
    assert oracle.round_half_to_even(x) == want
    assert float(q3.checkpoint.round_half_to_even(np.float32(x))) == want


# ---- model_exporter_test.rs:48-67 ------------------------------------------------------------------
def test_quantize_q80_known_values(oracle, q3):
    w = np.array([0.0, 127.0, -127.0, 63.5], dtype=np.float32)
    for fn in (oracle.quantize_q80, q3.checkpoint.quantize_q80):
        q, s, _ = fn(w, 4)
        assert len(s) == 1 and abs(float(s[0]) - 1.0) < 1e-6
        assert list(q) == [0, 127, -127, 64]


# ---- model_exporter_test.rs:70-87 ------------------------------------------------------------------
def test_quantize_q80_zero_weights_scale_is_one(oracle, q3):
    for fn in (oracle.quantize_q80, q3.checkpoint.quantize_q80):
        q, s, err = fn(np.zeros(4, dtype=np.float32), 4)
        assert float(s[0]) == 1.0 and not q.any() and err == 0.0


# ---- model_exporter_test.rs:90-101 -----------------------------------------------------------------
def test_quantize_q80_invalid_group_size(oracle, q3):
    for fn in (oracle.quantize_q80, q3.checkpoint.quantize_q80):
        with pytest.raises(ValueError, match="multiple of group_size"):
            fn(np.array([1.0, 2.0, 3.0], dtype=np.float32), 4)


# ---- model_exporter_test.rs:104-134 ----------------------------------------------------------------
@pytest.mark.parametrize("hidden,req,want", [(128, 64, 64), (128, 32, 32), (128, 16, 16), (32, 64, 32), (128, 96, 4),
                                             (60, 40, 20), (60, 30, 30), (127, 64, 4), (15, 8, 4), (128, 2, 4)])
def test_find_optimal_group_size(oracle, q3, hidden, req, want):
    assert oracle.find_optimal_group_size(hidden, req) == want
    assert q3.checkpoint.find_optimal_group_size(hidden, req) == want


# ---- model_exporter_test.rs:137-142 ----------------------------------------------------------------
def test_header_constants(q3):
    ck = q3.checkpoint
    assert (ck.MAGIC_NUMBER, ck.VERSION, ck.HEADER_SIZE, ck.MIN_GROUP_SIZE) == (0x616A6331, 1, 256, 4)


# ---- model_exporter_test.rs:145-161 ----------------------------------------------------------------
def test_quantization_symmetry(oracle, q3):
    w = np.array([100.0, -100.0, 50.0, -50.0], dtype=np.float32)
    for fn in (oracle.quantize_q80, q3.checkpoint.quantize_q80):
        q, s, _ = fn(w, 4)
        assert abs(float(s[0]) - 100.0 / 127.0) < 1e-6
        assert q[0] == -q[1] and q[2] == -q[3]


# ---- model_exporter_test.rs:397-427 ----------------------------------------------------------------
def test_quantization_binary_consistency(oracle, q3):
    w = np.array([1, 2, 3, 4, -5, 6, -7, 8, 0.1, -0.2, 0.3, -0.4, 100, -100, 50, -25], dtype=np.float32)
    for fn in (oracle.quantize_q80, q3.checkpoint.quantize_q80):
        q, s, err = fn(w, 4)
        assert len(q) == 16 and len(s) == 4 and err >= 0
        for got, want in zip(s, [4 / 127, 8 / 127, 0.4 / 127, 100 / 127]):
            assert abs(float(got) - want) < 1e-6
        assert q.min() >= -127
        deq = q.astype(np.float32).reshape(4, 4) * s[:, None]
        assert np.max(np.abs(deq.reshape(-1) - w) / np.repeat(s, 4)) <= 0.5 + 1e-6


def test_exporter_quantizer_c_vs_numpy_bitwise(oracle, q3):
    rng = np.random.default_rng(3)
    w = (rng.standard_normal(64 * 257) * 0.03).astype(np.float32)
    w[:64] = 0
    w[64:70] = [0.5, 1.5, 2.5, -0.5, -1.5, -2.5]
    q1, s1, e1 = oracle.quantize_q80(w, 64)
    q2, s2, e2 = q3.checkpoint.quantize_q80(w, 64)
    assert np.array_equal(q1, q2)
    assert_biteq(s1, s2, "scales")
    assert abs(e1 - e2) < 1e-12


# ---- header: writer (model_exporter.rs:164-191) vs readers (configuration.rs:77-146) -----------------
def test_header_roundtrip_all_parsers(oracle, q3):
    ck = q3.checkpoint
    for name in ("qwen3-0.6b", "qwen3-4b", "qwen3-8b", "deepseek-r1-0528-qwen3-8b", "tiny"):
        sh = ck.SHAPES[name]
        raw = ck.header_bytes(sh)
        assert len(raw) == 256 and raw[52:] == b"\0" * 204
        c1 = oracle.read_config(raw)
        c2 = q3.engine.parse_header(raw)
        for c in (c1, c2):
            assert (c.dim, c.hidden_dim, c.n_layers, c.n_heads, c.n_kv_heads, c.vocab_size, c.seq_len, c.head_dim,
                    c.group_size, bool(c.shared_classifier)) == (sh.dim, sh.hidden_dim, sh.n_layers, sh.n_heads,
                                                                  sh.n_kv_heads, sh.vocab_size, sh.max_seq_len,
                                                                  sh.head_dim, sh.group_size, sh.shared_classifier)
        # the oracle's own header writer must produce the same 256 bytes
        assert oracle.write_header(c1, sh.max_seq_len) == raw


@pytest.mark.parametrize("mutate,msg", [
    (lambda b: b"\x00\x00\x00\x00" + b[4:], "magic"),
    (lambda b: b[:4] + b"\x02\x00\x00\x00" + b[8:], "version"),
    (lambda b: b[:12] + b"\x00\x00\x00\x00" + b[16:], "dim"),
    (lambda b: b[:36] + b"\xff\xff\xff\xff" + b[40:], "seq_len"),
    (lambda b: b[:40], "nsufficient"),
    (lambda b: b[:200], "nsufficient"),
])
def test_header_validation_errors(oracle, q3, mutate, msg):
    raw = mutate(q3.checkpoint.header_bytes(q3.checkpoint.SHAPES["tiny"]))
    with pytest.raises(ValueError, match=msg):
        oracle.read_config(raw)
    with pytest.raises(q3.Q3Error, match=msg):
        q3.engine.parse_header(raw)


def test_model_byte_counts_match_survey(q3):
    """SURVEY.md section 8d / BASELINE.md section 3 figures the roofline is computed from."""
    ck = q3.checkpoint
    assert ck.SHAPES["qwen3-0.6b"].file_size() == 633_495_808
    assert ck.SHAPES["qwen3-4b"].file_size() == 4_274_448_640
    assert ck.SHAPES["qwen3-8b"].file_size() == 8_703_561_984
    assert ck.SHAPES["qwen3-0.6b"].weight_bytes_per_token() == (595_984_384, 37_249_024)
    assert ck.SHAPES["qwen3-4b"].weight_bytes_per_token() == (4_022_272_000, 251_392_000)
    assert ck.SHAPES["qwen3-8b"].weight_bytes_per_token() == (7_568_097_280, 473_006_080)


# ---- op-level: C oracle vs independent numpy restatement, bit for bit ---------------------------------
def test_ops_c_vs_numpy(oracle, np_oracle):
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(256) * 2).astype(np.float32)
    x[:64] = 0.0                       # runtime quantizer stores scale 0.0 for a zero group (tensor.rs:107-114)
    x[70] = 0.5 * np.abs(x[64:128]).max() / 127 * 3  # lands near a .5 boundary: half-away rounding
    for G in (16, 32, 64):
        q1, s1 = oracle.quantize(x, G)
        q2, s2 = np_oracle.quantize(x, G)
        assert np.array_equal(q1, q2)
        assert_biteq(s1, s2)
        assert s1[0] == 0.0 and not q1[:G].any()
        assert_biteq(oracle.dequantize(q1, s1, G), np_oracle.dequantize(q1, s1, G))
    n, d, G = 192, 37, 32
    xq = rng.integers(-127, 128, n).astype(np.int8)
    xs = rng.random(n // G).astype(np.float32)
    wq = rng.integers(-127, 128, n * d).astype(np.int8)
    ws = rng.random(n * d // G).astype(np.float32)
    assert_biteq(oracle.matmul(xq, xs, wq, ws, n, d, G), np_oracle.matmul(xq, xs, wq, ws, n, d, G), "matmul")
    w = (1 + 0.1 * rng.standard_normal(256)).astype(np.float32)
    assert_biteq(oracle.rmsnorm(x, w), np_oracle.rmsnorm(x, w), "rmsnorm")
    a = (rng.standard_normal(77) * 5).astype(np.float32)
    assert_biteq(oracle.softmax(a), np_oracle.softmax(a), "softmax")
    g, u = (rng.standard_normal(99) * 4).astype(np.float32), rng.standard_normal(99).astype(np.float32)
    assert_biteq(oracle.swiglu(g, u), np_oracle.swiglu(g, u), "swiglu")
    for pos in (0, 1, 17, 40959, 131071):
        cs1, cs2 = oracle.rope_freqs(128, pos), np_oracle.rope_freqs(128, pos)
        assert_biteq(cs1, cs2, f"rope freqs pos {pos}")
        v = rng.standard_normal(128).astype(np.float32)
        assert_biteq(oracle.rope_apply(v, cs1), np_oracle.rope_apply(v, cs2), "rope apply")


def test_argmax_last_maximum_wins(oracle, np_oracle, q3):
    """sampler.rs:57-59: Iterator::max_by(total_cmp) keeps the LAST of equal maxima; -0.0 < +0.0."""
    a = np.array([1.0, 3.0, 2.0, 3.0, -1.0], dtype=np.float32)
    z = np.array([-0.0, 0.0, -0.0], dtype=np.float32)
    for fn in (oracle.sample_argmax, np_oracle.argmax_last, q3.sample_argmax):
        assert fn(a) == 3
        assert fn(z) == 1
        assert fn(np.array([-5.0, -5.0], dtype=np.float32)) == 1


# ---- whole-model: C oracle vs committed numpy-oracle fixtures ------------------------------------------
@pytest.mark.parametrize("name", ["tiny", "tiny-untied"])
def test_c_oracle_matches_golden(oracle, q3, name):
    import hashlib
    g = np.load(golden_path(f"{name}.golden.npz"))
    path = golden_path(f"{name}.bin")
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == str(g["checkpoint_sha256"])
    m = oracle.OracleModel(path)
    for (tok, pos), want in zip(g["calls"], g["logits"]):
        assert_biteq(m.forward(int(tok), int(pos)), want, f"{name} forward({tok},{pos})")
    k, v = m.kv_cache()
    assert_biteq(k, g["key_cache"], "key cache")
    assert_biteq(v, g["value_cache"], "value cache")
    # call patterns of generation.rs through the host-side mirrors
    prompt = [int(t) for t in g["prompt"]]
    toks, _ = q3.generate(oracle.OracleModel(path), prompt, max_new_tokens=12, sample=oracle.sample_argmax)
    assert toks == [int(t) for t in g["generate_tokens"]]
    ctoks, cpos, _ = q3.chat_turn(oracle.OracleModel(path), prompt, 0, 10, sample=oracle.sample_argmax)
    assert ctoks == [int(t) for t in g["chat_tokens"]] and cpos == int(g["chat_pos"])


def test_golden_checkpoint_is_reproducible(q3, tmp_ckpt_dir):
    """The committed checkpoint bytes are what checkpoint.write_synthetic_checkpoint produces today."""
    import hashlib
    import os
    g = np.load(golden_path("tiny.golden.npz"))
    p = os.path.join(tmp_ckpt_dir, "tiny_again.bin")
    q3.checkpoint.write_synthetic_checkpoint(p, q3.checkpoint.SHAPES["tiny"], seed=int(g["seed"]), sparse_zero_groups=True)
    assert hashlib.sha256(open(p, "rb").read()).hexdigest() == str(g["checkpoint_sha256"])


def test_oracle_thread_count_does_not_change_results(oracle, q3, tmp_ckpt_dir):
    """rayon only distributes independent rows/heads (tensor.rs:26, layers.rs:378): so does the oracle."""
    path = golden_path("tiny-untied.bin")
    outs = []
    for nt in (1, 3):
        oracle.set_num_threads(nt)
        m = oracle.OracleModel(path)
        outs.append([m.forward(5, p) for p in range(4)])
    oracle.set_num_threads(0)
    for a, b in zip(*outs):
        assert_biteq(a, b)


def test_oracle_error_behaviour(oracle, tmp_ckpt_dir):
    import os
    with pytest.raises(RuntimeError, match="Failed to open checkpoint"):
        oracle.OracleModel(os.path.join(tmp_ckpt_dir, "missing.bin"))
    data = open(golden_path("tiny.bin"), "rb").read()
    trunc = os.path.join(tmp_ckpt_dir, "trunc.bin")
    open(trunc, "wb").write(data[: len(data) // 2])
    with pytest.raises(RuntimeError, match="Insufficient data"):
        oracle.OracleModel(trunc)
    bad_arch = os.path.join(tmp_ckpt_dir, "arch.bin")
    open(bad_arch, "wb").write(data[:8] + (7).to_bytes(4, "little") + data[12:])
    with pytest.raises(RuntimeError, match="Unknown architecture_id: 7"):
        oracle.OracleModel(bad_arch)
    m = oracle.OracleModel(golden_path("tiny.bin"), ctx_len=8)
    assert m.config.seq_len == 8                       # models/mod.rs:65-67
    with pytest.raises(IndexError):
        m.forward(0, 8)                                # reference panics (slice index)
    with pytest.raises(IndexError):
        m.forward(256, 0)


def test_generate_mode_attends_over_zero_prefix(oracle):
    """generation.rs:26-29: prompt[0..n-2] never reach forward(); rows 0..n-2 of the KV cache stay zero but
    are attended over.  forward(t, 5) on a fresh model must therefore differ from forward(t, 0)."""
    m = oracle.OracleModel(golden_path("tiny.bin"))
    a = m.forward(9, 5)
    k, v = m.kv_cache()
    assert not k[:, :5].any() and not v[:, :5].any() and k[:, 5].any()
    b = oracle.OracleModel(golden_path("tiny.bin")).forward(9, 0)
    assert not np.array_equal(a, b)


# ---------------------------------------------------------------------------------------------------------------
# Sampler (sampler.rs): xorshift64* stream, temperature / softmax, multinomial and nucleus sampling
# ---------------------------------------------------------------------------------------------------------------
def test_sampler_rng_stream_known_answers(oracle):
    """random_u32 (sampler.rs:44-49) against an arbitrary-precision evaluation of the same recurrence."""
    M = (1 << 64) - 1
    for seed in (1, 42, 0xDEADBEEFCAFEBABE, M):
        s = oracle.Sampler(8, 1.0, 0.9, seed)
        st = seed
        for _ in range(16):
            st ^= st >> 12
            st ^= (st << 25) & M
            st ^= st >> 27
            want = ((st * 0x2545F4914F6CDD1D) & M) >> 32
            assert s.random_u32() == want
        f = s.random_f32()
        assert 0.0 <= f < 1.0 and f * 16777216.0 == int(f * 16777216.0)      # (u32 >> 8) / 2^24


def test_sampler_c_and_numpy_restatements_agree(oracle, np_oracle):
    rng = np.random.default_rng(3)
    for n, sigma, temp, topp in [(50, 1.0, 1.0, 0.9), (1000, 3.0, 0.7, 0.95), (1000, 0.01, 1.3, 0.5), (4096, 2.0, 0.6, 1.0),
                                 (4096, 2.0, 0.6, 0.0), (300, 5.0, 0.2, 0.3), (7, 1.0, 2.0, 0.99)]:
        a, b = oracle.Sampler(n, temp, topp, 1234), np_oracle.NpSampler(n, temp, topp, 1234)
        for _ in range(25):
            lg = rng.normal(0.0, sigma, n).astype(np.float32)
            lg[rng.integers(0, n, 5)] = lg[0]                                 # exact ties among the candidates
            assert a.sample(lg) == b.sample(lg)
        assert a.rng_state.value == b.state


def test_sampler_edge_cases(oracle):
    lg = np.array([0.5, 2.0, -1.0, 2.0], dtype=np.float32)
    assert oracle.Sampler(4, 0.0, 0.9, 5).sample(lg) == 3                      # temperature 0: last maximum, no coin drawn
    s = oracle.Sampler(4, 0.0, 0.9, 5)
    s.sample(lg)
    assert s.rng_state.value == 5
    p = np.array([0.1, 0.2, 0.3, 0.4], dtype=np.float32)
    assert oracle.sample_mult(p, 0.0) == 0 and oracle.sample_mult(p, 0.1) == 1     # coin < cdf is strict
    assert oracle.sample_mult(p, 0.999999) == 3 and oracle.sample_mult(p * 0.5, 0.9) == 3    # falls off the end -> n-1
    # nucleus: sorted 0.4,0.3,0.2,0.1; topp 0.5 keeps {0.4,0.3} (cum 0.7 > 0.5 at i=1); r = coin*0.7
    assert oracle.sample_topp(p, 0.5, 0.0) == 3 and oracle.sample_topp(p, 0.5, 0.99) == 2
    assert oracle.sample_topp(p, 0.5, 0.5) == 3 and oracle.sample_topp(p, 0.5, 0.6) == 2      # r = .35 < .4 ; r = .42 -> second
    tie = np.array([0.25, 0.25, 0.25, 0.25], dtype=np.float32)
    assert oracle.sample_topp(tie, 0.6, 0.0) == 0 and oracle.sample_topp(tie, 0.6, 0.4) == 1   # ties: ascending index
