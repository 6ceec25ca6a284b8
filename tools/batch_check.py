#!/usr/bin/env python3
"""Batched decode vs the single-stream engine: bit-identical logits / tokens per stream, then step timing.

    python tools/batch_check.py [shape ...]        (default: tiny-g64 small-hd128)
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck

def check(name, n_streams, steps=6, ctx=0):
    sh = ck.SHAPES[name]; path = f"/tmp/q3_{name}.bin"
    ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
    rng = np.random.default_rng(7)
    toks0 = [int(t) for t in rng.integers(0, sh.vocab_size, n_streams)]
    pos0 = [int(p) for p in rng.integers(0, 5, n_streams)]
    ref_logits, ref_tokens = [], []
    for i in range(n_streams):
        with q3.TransformerBuilder(path).with_ctx_length(ctx or 64).build() as t:
            tok, ll, tt = toks0[i], [], []
            for k in range(steps):
                lg = np.array(t.forward(tok, pos0[i] + k), copy=True)
                ll.append(lg); tok = int(q3.sample_argmax(lg)); tt.append(tok)
            ref_logits.append(ll); ref_tokens.append(tt)
    with q3.TransformerBuilder(path).with_ctx_length(ctx or 64).build() as t:
        t.batch_init(n_streams, 0)
        toks = list(toks0); bad = 0
        for k in range(steps):
            lg, am = t.forward_batch(toks, [p + k for p in pos0])
            for i in range(n_streams):
                a, b = lg[i].view(np.uint32), ref_logits[i][k].view(np.uint32)
                nd = int((a != b).sum())
                if nd or am[i] != ref_tokens[i][k]:
                    bad += 1
                    if bad <= 5:
                        print(f"  MISMATCH {name} stream {i} step {k}: {nd} logits differ (max abs {np.abs(lg[i]-ref_logits[i][k]).max():.3e}), argmax {am[i]} vs {ref_tokens[i][k]}")
            toks = am
        t.batch_reset_kv()
        out = t.generate_greedy_batch(toks0, pos0, steps)
        ok2 = all(list(out[i]) == ref_tokens[i] for i in range(n_streams))
        print(f"{name}: B={n_streams} steps={steps}: forward_batch {'OK bit-identical' if bad == 0 else f'{bad} MISMATCHES'}; generate_greedy_batch {'OK' if ok2 else 'MISMATCH'}", flush=True)
        return bad == 0 and ok2

def timing(name, n_streams, steps=64, ctx=2048):
    sh = ck.SHAPES[name]; path = f"/tmp/q3_{name}.bin"
    ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
    with q3.TransformerBuilder(path).with_ctx_length(ctx).build() as t:
        t.batch_init(n_streams, ctx)
        toks = list(range(5, 5 + n_streams)); pos = [7] * n_streams
        t.generate_greedy_batch(toks, pos, 4)
        t0 = time.perf_counter(); t.generate_greedy_batch(toks, pos, steps); dt = time.perf_counter() - t0
        w, s = sh.weight_bytes_per_token()
        print(f"{name}: B={n_streams}: {dt/steps*1e6:.1f} us/step, {n_streams*steps/dt:.0f} tok/s aggregate, weights {(w+s)/(dt/steps)/1e12:.2f} TB/s", flush=True)

if __name__ == "__main__":
    names = sys.argv[1:] or ["tiny-g64", "small-hd128"]
    ok = True
    for nm in names:
        for B in (3, 16, 32):
            ok = check(nm, B) and ok
    if os.environ.get("Q3_BATCH_TIMING", "1") != "0":
        for nm in ("qwen3-8b-dims-l2",):
            for B in (16, 32):
                timing(nm, B)
    sys.exit(0 if ok else 1)
