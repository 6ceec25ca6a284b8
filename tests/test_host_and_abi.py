"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol the header
declares, fails loudly without a GPU, never routes through the oracle, and the host mirrors of the
reference's call patterns behave like generation.rs."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden_path


def test_library_exports_every_declared_symbol(q3):
    hdr = open(os.path.join(ROOT, "include", "qwen3_hip.h")).read()
    declared = set(re.findall(r"\b(q3_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(q3.EXPORTED_SYMBOLS), declared ^ set(q3.EXPORTED_SYMBOLS)
    lib = q3.load_library()
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"libqwen3_hip.so does not export {sym}"
    out = subprocess.check_output(["nm", "-D", "--defined-only", q3.lib_path()], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert declared <= exported
    # developer entry points are compiled in only with -DQ3_DEV (libqwen3_hip_dev.so), never exported by the product library
    assert not {e for e in exported if e.startswith("q3_dev_")}, "q3_dev_* symbols in the product library"
    assert lib.q3_abi_version() == 1
    # the library was built from the checked-out sources (q3_build_id = hash of csrc/* + the header, baked in by make)
    if "Q3_HIP_LIB" not in os.environ:
        assert lib.q3_build_id().decode() == q3.source_build_id()


def test_no_cpu_fallback_without_gpu(q3):
    """On a box without a HIP device every compute entry point must fail loudly (status -4), not fall back."""
    import ctypes as C
    lib = q3.load_library()
    ndev = C.c_int(0)
    hip = C.CDLL("libamdhip64.so")
    if hip.hipGetDeviceCount(C.byref(ndev)) == 0 and ndev.value > 0:
        pytest.skip("a GPU is present; the loud-failure path is exercised on CPU-only boxes")
    with pytest.raises(q3.Q3Error) as e:
        q3.TransformerBuilder(golden_path("tiny.bin")).build()
    assert e.value.code == -4 and "no CPU fallback" in e.value.msg
    with pytest.raises(q3.Q3Error):
        q3.ops.matmul(np.zeros(64, np.int8), np.ones(1, np.float32), np.zeros(64, np.int8), np.ones(1, np.float32), 64, 1, 64)


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    pkg = os.path.join(ROOT, "qwen3-rs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "q3_oracle" not in text and "np_oracle" not in text and "q3o_" not in text, os.path.join(dirpath, f)
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), os.path.join(dirpath, f)
    out = subprocess.check_output(["ldd", os.path.join(pkg, "libqwen3_hip.so")], text=True)
    assert "q3_oracle" not in out
    # importing the product package must not import the oracle
    code = "import sys; sys.path.insert(0, %r); import qwen3_rs_amd; assert not any('oracle' in m for m in sys.modules), sys.modules.keys()" % pkg
    subprocess.check_call([sys.executable, "-c", code])


def test_product_library_reads_only_the_documented_environment_variables(q3):
    """include/qwen3_hip.h ("Environment") lists what libqwen3_hip.so reads; every other Q3_* switch lives in the developer build
    (-DQ3_DEV).  The binary's strings must be exactly that list (at most 8 names), and no developer symbol may be exported."""
    header = open(os.path.join(ROOT, "include", "qwen3_hip.h")).read()
    env_doc = header[header.index(" * Environment:"):header.index("#ifndef QWEN3_HIP_H")]
    documented = set(re.findall(r"^ \*     (Q3_[A-Z0-9_]+)", env_doc, re.M))
    assert 0 < len(documented) <= 8, documented
    blob = open(q3.lib_path(), "rb").read()
    found = set(m.decode() for m in re.findall(rb"(?<![A-Za-z0-9_])(Q3_[A-Z][A-Z0-9_]+)\x00", blob))
    assert found == documented, (sorted(found), sorted(documented))


class _Recorder:
    """A fake Transformer recording the (token,pos) calls it receives."""

    def __init__(self, vocab=16, seq_len=32, script=None):
        self.calls = []
        self.vocab, self.seq_len = vocab, seq_len
        self.script = script or {}

    def get_config(self):
        class C:
            seq_len = self.seq_len
        return C

    def forward(self, token, pos):
        self.calls.append((token, pos))
        logits = np.zeros(self.vocab, dtype=np.float32)
        logits[self.script.get(pos, (token + 1) % self.vocab)] = 1.0
        return logits


def test_generate_call_pattern(q3):
    """generation.rs:9-48: no forward() for prompt[0..n-2]; first call is (prompt[n-1], n-1); the sampled
    token is fed back at pos+1; a stop token is counted and ends the loop; seq_len bounds pos."""
    t = _Recorder()
    toks, metrics = q3.generate(t, [4, 5, 6, 7], max_new_tokens=3)
    assert t.calls == [(7, 3), (8, 4), (9, 5)] and toks == [8, 9, 10]
    assert metrics.generated_count == 3 and metrics.elapsed > 0
    t = _Recorder(script={4: 2})
    toks, m = q3.generate(t, [5, 5, 5], stop_tokens=[2])
    assert toks == [6, 7, 2] and t.calls[-1][1] == 4 and m.generated_count == 3   # terminating token counted
    t = _Recorder(seq_len=6)
    toks, _ = q3.generate(t, [3])
    assert [p for _, p in t.calls] == [0, 1, 2, 3, 4, 5]        # while pos < seq_len
    with pytest.raises(ValueError, match="provide a prompt"):
        q3.generate(_Recorder(), [])


def test_chat_call_pattern(q3):
    """generation.rs:94-151: every prompt token is forwarded (a sample drawn and discarded for each), then
    decode feeds next_token back until a stop token."""
    t = _Recorder()
    toks, pos, m = q3.chat_turn(t, [4, 5, 6], 0, 4)
    assert t.calls[:3] == [(4, 0), (5, 1), (6, 2)]
    assert toks == [7, 8, 9, 10] and t.calls[3:] == [(7, 3), (8, 4), (9, 5), (10, 6)] and pos == 7
    t = _Recorder(script={2: 9, 3: 1})
    toks, pos, _ = q3.chat_turn(t, [4, 5, 6], 0, 10, stop_tokens=[1])
    assert toks == [9] and pos == 4                                # next_token == stop -> turn ends before output


def test_sample_argmax_host(q3):
    rng = np.random.default_rng(0)
    a = rng.standard_normal(1000).astype(np.float32)
    a[[10, 500, 999]] = 7.0
    assert q3.sample_argmax(a) == 999


def test_checkpoint_layout_offsets(q3):
    ck = q3.checkpoint
    sh = ck.SHAPES["qwen3-0.6b"]
    off = ck.tensor_offsets(sh)
    assert off["__end__"][0] == sh.file_size()
    assert off["input_layernorm"][0] == 256
    for name, v in off.items():                       # every section 16-byte aligned -> dwordx4 loads legal
        for o in v[:2]:
            assert o % 16 == 0, (name, o)
    data = open(golden_path("tiny.bin"), "rb").read()
    assert ck.read_header(golden_path("tiny.bin")) == ck.SHAPES["tiny"]
    assert len(data) == ck.SHAPES["tiny"].file_size()


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # replicas only: each rank decodes its own stream; time = max over ranks, tokens = sum (no data-path collective)
    agg = bench.aggregate_over_ranks(local_tokens=100 + rank, local_seconds=0.5 + 0.25 * rank)
    q.put((rank, agg))
    dist.destroy_process_group()


def test_replica_aggregation_two_ranks_gloo():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    ps = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for r in (0, 1):
        tokens, seconds = res[r]
        assert tokens == 201 and abs(seconds - 0.75) < 1e-9


def test_bench_gpus2_spawns_two_replica_workers_stub_engine():
    """`python bench.py --gpus 2` must itself start two worker processes (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE set,
    gloo barrier + max/sum aggregation) BEFORE anything touches HIP, and print ONE line with n_gpus 2.  The workers use
    the launcher's host stub here (no GPU in this container); on the GPU box the same launcher starts real engines."""
    import json
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "2",
                        "--stub-engine"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["replicas"] == 2
    # whole-job value: 2 replicas x 12 tokens over the slowest replica's time
    assert abs(d["value"] - 24 / (d["ms_per_step"] * 12 / 1e3)) / d["value"] < 1e-3
    assert "_tokens" not in d


def test_bench_cpu_leg_compares_equal_length_prefixes(oracle, tmp_ckpt_dir):
    """The oracle leg compares min(len(cpu), len(gpu)) tokens (a --steps below the CPU sample must not read as a
    mismatch) and reports both the all-core and the one-thread figure; a wrong token flips the verdict."""
    sys.path.insert(0, ROOT)
    import bench
    from qwen3_rs_amd import checkpoint as ck
    sh = ck.SHAPES["tiny-g64"]
    path = os.path.join(tmp_ckpt_dir, "bench_tiny.bin")
    ck.ensure_synthetic_checkpoint(path, sh, seed=5)
    m = oracle.OracleModel(path, 64)
    tok, toks = 3, []
    for k in range(20):
        tok = oracle.sample_argmax(m.forward(tok, 7 + k))
        toks.append(tok)
    m.close()
    for n in (5, 20):
        cb, ok = bench.cpu_baseline(path, 64, 3, 7, toks[:n], max_tokens=12, budget_s=5, one_thread_budget_s=2)
        assert ok and cb["tokens_match_gpu"] and cb["tokens_compared"] == min(n, 12)
        assert cb["one_thread"]["cores"] == 1 and cb["one_thread"]["value"] > 0 and cb["value"] > 0
    bad = list(toks)
    bad[2] = (bad[2] + 1) % sh.vocab_size
    cb, ok = bench.cpu_baseline(path, 64, 3, 7, bad, max_tokens=12, budget_s=5, one_thread_budget_s=2)
    assert not ok and not cb["tokens_match_gpu"]


def test_batch_entry_points_reject_prefill_only_context(q3):
    """q3_batch_reset_kv / q3_batch_read_state on an engine without q3_batch_init must fail cleanly
    (null engine here: no GPU in the CPU suite; the has_kv guard itself is exercised on the GPU box)."""
    L = q3.load_library()
    assert L.q3_batch_reset_kv(None) == -3


def test_bench_under_torchrun_two_ranks_stub_engine():
    """The driver's N > 1 launch form: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` -- every
    rank is a worker (RANK / LOCAL_RANK / WORLD_SIZE from the environment), gloo carries the barrier and the aggregate,
    rank 0 prints the ONE line."""
    import json
    import subprocess
    port = 29700 + (os.getpid() % 200)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6",
                        "--warmup", "1", "--stub-engine"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and "_tokens" not in d


def test_host_sample_argmax_last_maximum_under_total_order(q3):
    """q3_host_sample_argmax (the host half of q3_host_generate: logits.to_vec() + Sampler::sample_argmax, sampler.rs:57-59):
    index of the LAST maximum under f32::total_cmp, copy bit-identical -- ties, -0.0 < +0.0, NaNs (positive NaN above +inf,
    negative NaN below -inf), lengths that are not a multiple of the vector width, aligned and unaligned destinations."""
    import ctypes as C
    import numpy as np
    lib = q3.load_library()
    fp = C.POINTER(C.c_float)

    def total_key(a):
        b = a.view(np.int32).astype(np.int64)
        return np.where(b < 0, b ^ 0x7fffffff, b)

    rng = np.random.default_rng(11)
    cases = []
    for n in (1, 7, 8, 9, 63, 1000, 151936):
        a = rng.standard_normal(n).astype(np.float32)
        cases.append(a)
        t = a.copy(); t[rng.integers(0, n, 3)] = t.max()                  # ties: the last one wins
        cases.append(t)
    cases.append(np.array([0.0, -0.0, 0.0, -0.0], dtype=np.float32))     # +0.0 is the maximum, last occurrence index 2
    cases.append(np.array([-0.0, -0.0], dtype=np.float32))
    cases.append(np.array([1.0, np.inf, np.nan, 2.0, np.nan, -np.nan], dtype=np.float32))
    cases.append(np.array([-np.inf, -1e30, -np.inf], dtype=np.float32))
    for a in cases:
        a = np.ascontiguousarray(a)
        k = total_key(a)
        want = int(np.nonzero(k == k.max())[0][-1])
        for off in (0, 1):                                                 # aligned / unaligned destination
            store = np.zeros(a.size + 16, dtype=np.float32)
            base = (-store.ctypes.data // 4) % 8                           # first 32-byte aligned element
            dst = store[base + off: base + off + a.size]
            got = lib.q3_host_sample_argmax(a.ctypes.data_as(fp), a.size, dst.ctypes.data_as(fp))
            assert got == want, (a[:8], got, want)
            assert np.array_equal(dst.view(np.int32), a.view(np.int32))
        assert lib.q3_host_sample_argmax(a.ctypes.data_as(fp), a.size, None) == want


def test_bench_roofline_from_graph_stamps_arithmetic(monkeypatch):
    """bench.py's roofline block from graph-replay kernel durations (tools/kstamps.py): achieved = algorithmic bytes of a token's k_gemv
    launches / the sum of their stamped durations; the eager HIP-event figure is kept as eager_check; a stamped run of another build is
    refused.  (The stamp run itself needs a GPU; its arithmetic does not.)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    import kstamps
    from qwen3_rs_amd import checkpoint as ck
    shape = ck.SHAPES["qwen3-0.6b"]
    fam = {"qkv": (28, 3.0, 2.0), "attn": (28, 3.1, 1.4), "wo": (28, 1.7, 1.3), "w13": (28, 3.5, 1.3), "w2": (28, 2.4, 1.3), "lm_head": (1, 30.0, 1.4)}
    gk = {"build_id": "abc", "stamped_build_id": "abc", "product_library_us_per_token": 540.0,
          "stamped_developer_library_us_per_token": 620.0, "sum_duration_plus_gap_us_per_token": 626.0, "wall_us_per_token_this_call": 618.0,
          "stamp_overhead_us_per_launch": 0.57, "tokens_folded": 68, "launches_per_token": 141,
          "families": {k: {"launches_per_token": n, "avg_duration_us": d, "avg_gap_to_predecessor_us": g} for k, (n, d, g) in fam.items()}}
    monkeypatch.setattr(kstamps, "graph_kernel_durations", lambda **kw: dict(gk))
    args = bench.parse_args(["--steps", "20", "--warmup", "5"])
    out = {"build_id": {"library": "abc", "sources": "abc"},
           "roofline": {"bound": "hbm", "achieved": 1400.0, "frac": 0.175, "avg_launch_us": 4.2, "per_kernel": [], "note": "eager", "peak": 8000.0, "unit": "GB/s", "traffic": 5800000}}
    bench.roofline_from_graph_stamps(out, args, shape, 5, 7, "/nonexistent")
    rf = out["roofline"]
    bpl = bench.gemv_bytes_per_launch(shape)
    want = sum(bpl[k] * n for k, (n, d, g) in fam.items() if k in bpl) / (sum(d * n for k, (n, d, g) in fam.items() if k in bpl) * 1e-6)
    assert abs(rf["achieved"] - want / 1e9) < 0.1 and abs(rf["frac"] - want / 8e12) < 1e-4
    assert rf["eager_check"]["frac"] == 0.175 and rf["launches_per_token"] == 113 and rf["traffic"] == 5800000
    assert rf["graph_mode"]["stamp_overhead_us_per_launch"] == 0.57 and len(rf["per_kernel"]) == 6
    assert sum(bpl[k] * n for k, (n, d, g) in fam.items() if k in bpl) == sum(shape.weight_bytes_per_token())   # 633,233,408 B
    out2 = {"build_id": {"library": "other"}, "roofline": dict(rf)}
    with pytest.raises(RuntimeError):
        bench.roofline_from_graph_stamps(out2, args, shape, 5, 7, "/nonexistent")


def _two_replica_bench_check(extra):
    """Body shared by the CPU (stub engine) and the two-GPU (real engines) variants: `bench.py --gpus 2` is ONE line with
    n_gpus 2 whose whole-job value is (well above) one replica's."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "2", "--no-cpu-baseline",
           "--no-other-configs"] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["steps"] == 16 and d["scaling"] == "weak"
    one = subprocess.run(cmd[:3] + ["1"] + cmd[4:], capture_output=True, text=True, timeout=1200)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert d1["n_gpus"] == 1
    return d, d1


def test_bench_two_replicas_stub_engine_same_body_as_the_two_gpu_test():
    """The body of the two-GPU test below, on the launcher's host stub: a test that only ever skips must not be able to rot."""
    d, d1 = _two_replica_bench_check(["--stub-engine"])
    assert d["value"] > 1.5 * d1["value"], (d["value"], d1["value"])


@pytest.mark.gpu
def test_bench_two_real_replicas_on_two_devices():
    """`bench.py --gpus 2` with REAL engines: one process + one engine per GPU (devices 0 and 1), no collective on the data
    path, value = sum of tokens / max time.  Needs two visible GPUs (the 1-GPU test boxes skip it)."""
    import torch
    if torch.cuda.device_count() < 2:          # (counting devices does not initialise HIP in this process)
        pytest.skip("fewer than two GPUs visible")
    d, d1 = _two_replica_bench_check([])
    assert d["value"] > 1.5 * d1["value"], (d["value"], d1["value"])      # two independent replicas: ~2x, never ~1x


def _undefined_names(path):
    """Names a module reads at function or module level that nothing binds: pyflakes when importable, else an ast walk
    (module-level bindings + builtins + each function's own parameters / assignments / imports / comprehension targets)."""
    import ast
    import builtins
    src = open(path).read()
    compile(src, path, "exec")                                             # syntax
    try:
        from pyflakes import api, reporter
        import io
        out, err = io.StringIO(), io.StringIO()
        api.check(src, path, reporter.Reporter(out, err))
        return [ln for ln in out.getvalue().splitlines() if "undefined name" in ln]
    except ImportError:
        pass
    tree = ast.parse(src, path)
    bound = set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__builtins__", "__spec__", "__package__"}
    for node in ast.walk(tree):                                            # anything bound anywhere in the file, by any construct
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            bound.add(node.name)
            if not isinstance(node, ast.ClassDef):
                a = node.args
                for arg in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                    bound.add(arg.arg)
        elif isinstance(node, ast.Lambda):
            a = node.args
            for arg in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                bound.add(arg.arg)
        elif isinstance(node, (ast.Import, ast.ImportFrom)):
            for al in node.names:
                bound.add((al.asname or al.name).split(".")[0])
        elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
            bound.add(node.id)
        elif isinstance(node, ast.ExceptHandler) and node.name:
            bound.add(node.name)
        elif isinstance(node, (ast.Global, ast.Nonlocal)):
            bound.update(node.names)
    # a name imported only INSIDE one function does not exist in another: per-function check of module-like names
    module_level = set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__spec__", "__package__"}
    for node in tree.body:
        for sub in ast.walk(node) if not isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)) else [node]:
            if isinstance(sub, (ast.Import, ast.ImportFrom)):
                for al in sub.names:
                    module_level.add((al.asname or al.name).split(".")[0])
            elif isinstance(sub, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                module_level.add(sub.name)
            elif isinstance(sub, ast.Name) and isinstance(sub.ctx, ast.Store):
                module_level.add(sub.id)
    bad = []

    def visit_fn(fn, outer):
        local = set(outer)
        for sub in ast.walk(fn):
            if isinstance(sub, (ast.Import, ast.ImportFrom)):
                for al in sub.names:
                    local.add((al.asname or al.name).split(".")[0])
            elif isinstance(sub, ast.Name) and isinstance(sub.ctx, (ast.Store, ast.Del)):
                local.add(sub.id)
            elif isinstance(sub, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                local.add(sub.name)
            elif isinstance(sub, ast.ExceptHandler) and sub.name:
                local.add(sub.name)
            elif isinstance(sub, (ast.Global, ast.Nonlocal)):
                local.update(sub.names)
            elif isinstance(sub, ast.arg):
                local.add(sub.arg)
        for sub in ast.walk(fn):
            if isinstance(sub, ast.Name) and isinstance(sub.ctx, ast.Load) and sub.id not in local:
                bad.append(f"{path}:{sub.lineno}: undefined name '{sub.id}'")

    for node in ast.walk(tree):
        if isinstance(node, ast.Name) and isinstance(node.ctx, ast.Load) and node.id not in bound:
            bad.append(f"{path}:{node.lineno}: undefined name '{node.id}'")
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)):
            visit_fn(node, module_level)
        elif isinstance(node, ast.ClassDef):
            for m in node.body:
                if isinstance(m, (ast.FunctionDef, ast.AsyncFunctionDef)):
                    visit_fn(m, module_level | {node.name})
    return sorted(set(bad))


def test_python_sources_compile_and_have_no_undefined_names():
    """tests/*.py, tools/*.py, bench.py, __graft_entry__.py and the package byte-compile and read no name that nothing binds
    (the two-GPU test of round 5 called json.loads in a module that never imported json and only ever skipped)."""
    import glob
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for pat in ("tests/*.py", "tests/golden/*.py", "tools/*.py", "oracle/*.py", "qwen3-rs_amd/qwen3_rs_amd/*.py"):
        files += sorted(glob.glob(os.path.join(ROOT, pat)))
    assert len(files) > 10
    bad = []
    for f in files:
        bad += _undefined_names(f)
    assert not bad, "\n".join(bad)
