#!/usr/bin/env python3
"""Developer diagnostic (not a test, not the bench): op parity, tiny-model parity, 0.6B timing + per-kernel
profile on the GPU box.  Usage: python tools/gpu_check.py [--big] [--strict-big]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
sys.path.insert(0, ROOT)

import numpy as np

import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck
from oracle import q3_oracle as co


def biteq(a, b):
    return np.array_equal(np.ascontiguousarray(a, dtype=np.float32).view(np.int32),
                          np.ascontiguousarray(b, dtype=np.float32).view(np.int32))


def report(name, a, b):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    eq = biteq(a, b)
    err = float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0
    print(f"  {name:40s} {'BIT-EQUAL' if eq else 'differs '}  max|d|={err:.3e}")
    return eq


def ops_check():
    rng = np.random.default_rng(0)
    ops = q3.ops
    print("== ops")
    for G in (16, 32, 64, 128):
        x = (rng.standard_normal(1024) * 3).astype(np.float32)
        x[:G] = 0
        qa, sa = ops.quantize(x, G)
        qb, sb = co.quantize(x, G)
        print(f"  quantize G={G}: q equal {np.array_equal(qa, qb)} s equal {biteq(sa, sb)}")
    for (n, d, G) in [(64, 40, 16), (1024, 2048, 64), (2048, 1024, 64), (3072, 1024, 64), (2560, 96, 64), (9728, 24, 64),
                      (12288, 16, 64), (4096, 100, 128), (1024, 333, 32)]:
        xq = rng.integers(-127, 128, n).astype(np.int8)
        xs = rng.random(n // G).astype(np.float32)
        wq = rng.integers(-127, 128, n * d).astype(np.int8)
        ws = rng.random(n * d // G).astype(np.float32)
        report(f"matmul n={n} d={d} G={G}", ops.matmul(xq, xs, wq, ws, n, d, G), co.matmul(xq, xs, wq, ws, n, d, G))
    x = rng.standard_normal(1024).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(1024)).astype(np.float32)
    report("rmsnorm default", ops.rmsnorm(x, w), co.rmsnorm(x, w))
    report("rmsnorm strict", ops.rmsnorm(x, w, strict=True), co.rmsnorm(x, w))
    a = (rng.standard_normal(777) * 4).astype(np.float32)
    report("softmax default", ops.softmax(a), co.softmax(a))
    report("softmax strict", ops.softmax(a, strict=True), co.softmax(a))
    g = (rng.standard_normal(3072) * 3).astype(np.float32)
    u = rng.standard_normal(3072).astype(np.float32)
    report("swiglu", ops.swiglu(g, u), co.swiglu(g, u))
    xs_ = np.concatenate([rng.uniform(-104, 89, 200000), rng.standard_normal(200000) * 3,
                          [0.0, -0.0, np.inf, -np.inf, 88.7, 88.8, -103.9, -104.1, 1e-30]]).astype(np.float32)
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.expf.restype = ctypes.c_float
    libm.expf.argtypes = [ctypes.c_float]
    ref = np.array([libm.expf(float(v)) for v in xs_], dtype=np.float32)
    report("expf vs glibc (400k)", ops.expf(xs_), ref)
    lg = rng.standard_normal(151936).astype(np.float32)
    lg[[5, 77777, 151000]] = 9.5
    print("  argmax", ops.argmax(lg), co.sample_argmax(lg))
    for (nh, nkv, hd, S, pos) in [(4, 2, 16, 64, 9), (16, 8, 128, 256, 200), (8, 8, 64, 32, 0), (4, 1, 32, 700, 650)]:
        kvd = nkv * hd
        q = rng.standard_normal(nh * hd).astype(np.float32)
        K = rng.standard_normal((S, kvd)).astype(np.float32)
        V = rng.standard_normal((S, kvd)).astype(np.float32)
        K[pos + 1:] = 0
        qw = (1 + 0.1 * rng.standard_normal(hd)).astype(np.float32)
        kw = (1 + 0.1 * rng.standard_normal(hd)).astype(np.float32)
        rb, rq, rk = co.attention(q, K, V, qw, kw, pos, nh, nkv, hd)
        for strict in (False, True):
            xb, q2, k2 = ops.attention(q, K, V, qw, kw, pos, nh, nkv, hd, strict=strict)
            tag = f"attn nh={nh} nkv={nkv} hd={hd} pos={pos} {'strict' if strict else 'default'}"
            report(tag + " xb", xb, rb)
            report(tag + " q", q2, rq)
            report(tag + " krow", k2.reshape(S, kvd)[pos], rk.reshape(S, kvd)[pos])


def model_check(name, seed=7, ctx=0, steps=6):
    sh = ck.SHAPES[name]
    path = f"/tmp/q3_{name}.bin"
    ck.write_synthetic_checkpoint(path, sh, seed=seed, sparse_zero_groups=True)
    om = co.OracleModel(path, ctx)
    for strict in (False, True):
        t = q3.TransformerBuilder(path).with_ctx_length(ctx or None).with_strict(strict).build()
        om.reset()
        tok, ok = 3, True
        worst = 0.0
        for pos in list(range(steps)):
            a = np.array(t.forward(tok, pos), copy=True)
            b = om.forward(tok, pos)
            worst = max(worst, float(np.max(np.abs(a - b))))
            ok = ok and biteq(a, b)
            ta, tb = q3.sample_argmax(a), co.sample_argmax(b)
            assert t.forward_argmax(tok, pos) == ta, "device argmax mismatch"
            if ta != tb:
                print("   token mismatch at pos", pos, ta, tb)
            tok = tb
        print(f"  {name:12s} strict={int(strict)}  logits {'BIT-EQUAL' if ok else 'differ'}  max|d|={worst:.3e}  "
              f"std={float(b.std()):.3f}")
        t.close()


def big_check(strict=False, n_tokens=128, parity_tokens=8):
    sh = ck.SHAPES["qwen3-0.6b"]
    path = "/tmp/q3_qwen3-0.6b.bin"
    t0 = time.time()
    ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
    print(f"== 0.6B checkpoint ready in {time.time() - t0:.1f}s")
    prompt = ck.iter_prompt_tokens(sh, 1234, 8)
    t0 = time.time()
    t = q3.TransformerBuilder(path).with_ctx_length(1024).with_strict(strict).build()
    print(f"  engine create {time.time() - t0:.2f}s  strict={strict}")
    first_pos, first_tok = len(prompt) - 1, prompt[-1]
    toks = t.generate_greedy(first_tok, first_pos, 4)  # warm
    for rep in range(3):
        t.reset_kv()
        t0 = time.perf_counter()
        toks = t.generate_greedy(first_tok, first_pos, n_tokens)
        dt = time.perf_counter() - t0
        wq, wsb = sh.weight_bytes_per_token()
        tps = n_tokens / dt
        print(f"  device-resident greedy: {n_tokens} tok in {dt * 1e3:.2f} ms -> {tps:.1f} tok/s, "
              f"{dt / n_tokens * 1e6:.1f} us/tok, {tps * (wq + wsb) / 8e12 * 100:.1f}% of 8 TB/s")
    # host-synchronous forward path (full logits egress per token)
    t.reset_kv()
    t0 = time.perf_counter()
    out, metrics = q3.generate(t, prompt, max_new_tokens=32)
    print(f"  forward()+logits D2H+host argmax: {metrics.report()[2]:.1f} tok/s")
    print("  tokens equal (device loop vs host loop):", out == toks[:32])
    prof = t.profile(first_tok, first_pos + 5, reps=5)
    tot = sum(p[1] for p in prof)
    for name, ms, n in prof:
        print(f"    {name:8s} launches={n:4d} total={ms / 5 * 1e3:8.1f} us/token  avg={ms / max(n, 1) * 1e3:7.2f} us")
    print(f"    sum = {tot / 5 * 1e3:.1f} us/token (eager, events between kernels)")
    if parity_tokens:
        om = co.OracleModel(path, 1024)
        t.reset_kv()
        tok, pos = first_tok, first_pos
        for i in range(parity_tokens):
            a = np.array(t.forward(tok, pos), copy=True)
            b = om.forward(tok, pos)
            ta, tb = q3.sample_argmax(a), co.sample_argmax(b)
            srt = np.sort(b)
            print(f"   pos {pos}: biteq={biteq(a, b)} max|d|={np.max(np.abs(a - b)):.3e} tok {ta} vs {tb} "
                  f"margin={srt[-1] - srt[-2]:.4f}")
            tok, pos = tb, pos + 1
    t.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--strict-big", action="store_true")
    ap.add_argument("--no-ops", action="store_true")
    args = ap.parse_args()
    if not args.no_ops:
        ops_check()
        print("== tiny models")
        for nm in ("tiny", "tiny-untied", "tiny-g64", "small-hd128"):
            model_check(nm)
    if args.big:
        big_check(False)
    if args.strict_big:
        big_check(True, parity_tokens=4)
