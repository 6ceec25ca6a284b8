#!/usr/bin/env python3
"""BASELINE.json configs[3]: Qwen3-8B Q8 on one MI355X, 32 concurrent greedy decode streams (aggregate throughput).

SURVEY.md section 8d: synthetic 8B-shape checkpoint (seed 1236, untied classifier), 32 independent streams with
distinct 8-token prompts in `generate` mode (first forward at pos 7 over a zero KV prefix, generation.rs:9-48),
ctx 2048 per stream, 256 steps; aggregate tok/s; every stream's tokens must equal its single-stream run (checked
here on `--verify` streams through q3_generate_greedy on the same engine).  Prints ONE JSON line.

    python tools/bench_batch.py [--shape qwen3-8b] [--streams 32] [--steps 256] [--ctx 2048] [--verify 2]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))

import qwen3_rs_amd as q3                                    # noqa: E402
from qwen3_rs_amd import checkpoint as ck                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="qwen3-8b")
    ap.add_argument("--streams", type=int, default=32)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--ctx", type=int, default=2048)
    ap.add_argument("--seed", type=int, default=1236)
    ap.add_argument("--verify", type=int, default=2, help="streams re-run single-stream and compared token by token")
    ap.add_argument("--ckpt-dir", default=os.environ.get("Q3_CKPT_DIR", "/tmp"))
    ap.add_argument("--tolerance", type=int, default=0, help="also run a Q3_FLAG_FAST engine for this many steps (tolerance_mode block)")
    a = ap.parse_args()

    shape = ck.SHAPES[a.shape]
    path = os.path.join(a.ckpt_dir, f"{a.shape}-seed{a.seed}.q3bin")
    t0 = time.time()
    ck.ensure_synthetic_checkpoint(path, shape, seed=a.seed)
    print(f"[bench_batch] checkpoint {path} ready in {time.time() - t0:.1f}s", file=sys.stderr)
    prompts = [ck.iter_prompt_tokens(shape, a.seed + 1000 + i, 8) for i in range(a.streams)]
    first_tok = [p[-1] for p in prompts]                      # generate mode: only the last prompt token is forwarded
    first_pos = [len(p) - 1 for p in prompts]

    with q3.TransformerBuilder(path).with_ctx_length(a.ctx).build() as t:
        t0 = time.time()
        t.batch_init(a.streams, a.ctx)
        print(f"[bench_batch] batch_init (weight repack + {a.streams} KV caches) {time.time() - t0:.1f}s", file=sys.stderr)
        t.generate_greedy_batch(first_tok, first_pos, 4)      # warm-up (graph capture)
        best, toks = None, None
        for _ in range(2):
            t.batch_reset_kv()
            t0 = time.perf_counter()
            out = t.generate_greedy_batch(first_tok, first_pos, a.steps)
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, toks = dt, out
        single_s, identical = None, True
        for i in range(min(a.verify, a.streams)):
            t.reset_kv()
            t0 = time.perf_counter()
            ref = t.generate_greedy(first_tok[i], first_pos[i], a.steps)
            single_s = time.perf_counter() - t0
            identical = identical and [int(v) for v in toks[i]] == [int(v) for v in ref]
        w, s = shape.weight_bytes_per_token()
        tol = None
        if a.tolerance > 0:
            # what the reference summation order costs on this path (bench.py `tolerance_mode`; never `value`): the same streams on a
            # Q3_FLAG_FAST engine -- tree sums in the per-stream prologues and attention -- against this (strict) engine: tok/s,
            # max |delta logit| over the first 6 steps (both engines fed the strict tokens), leading steps on which all streams agree
            import numpy as np
            n = min(a.tolerance, a.steps)
            t.batch_reset_kv()
            t0 = time.perf_counter()
            strict_tok = np.asarray(t.generate_greedy_batch(first_tok, first_pos, n))       # [streams][n]
            strict_s = time.perf_counter() - t0
            t.batch_reset_kv()
            feeds = [list(first_tok)] + [[int(strict_tok[i][k]) for i in range(a.streams)] for k in range(5)]
            sl = [np.array(t.forward_batch(feeds[k], [p + k for p in first_pos])[0], copy=True) for k in range(6)]
            with q3.TransformerBuilder(path).with_ctx_length(a.ctx).with_strict(False).build() as f:
                f.batch_init(a.streams, a.ctx)
                f.generate_greedy_batch(first_tok, first_pos, 4)
                fbest, ftok = None, None
                for _ in range(2):
                    f.batch_reset_kv()
                    t0 = time.perf_counter()
                    o = np.asarray(f.generate_greedy_batch(first_tok, first_pos, n))
                    dt = time.perf_counter() - t0
                    if fbest is None or dt < fbest:
                        fbest, ftok = dt, o
                f.batch_reset_kv()
                dl = []
                for k in range(6):
                    fl = f.forward_batch(feeds[k], [p + k for p in first_pos])[0]
                    dl.append(float(np.max(np.abs(fl - sl[k]))))
            lead = 0
            while lead < n and np.array_equal(ftok[:, lead], strict_tok[:, lead]):
                lead += 1
            tol = {"flags": "Q3_FLAG_FAST", "steps": n, "value": round(a.streams * n / fbest, 1), "unit": "tok/s",
                   "strict_value_same_run": round(a.streams * n / strict_s, 1), "ratio_to_strict": round(strict_s / fbest, 4),
                   "max_abs_delta_logit_first_steps": [round(d, 6) for d in dl],
                   "strict_logit_std_first_steps": [round(float(np.std(l)), 4) for l in sl],
                   "leading_steps_all_streams_identical_to_strict": lead}
    step_s = best / a.steps
    print(json.dumps({
        "metric": "batched_decode_tokens_per_second", "unit": "tok/s", "n_gpus": 1,
        "value": round(a.streams * a.steps / best, 1), "ms_per_step": round(1e3 * step_s, 4),
        "streams": a.streams, "steps": a.steps,
        "single_stream_tok_s": round(a.steps / single_s, 1) if single_s else None,
        "speedup_vs_single_stream": round(a.streams * single_s / best, 2) if single_s else None,
        "weights_once_per_step_hbm_frac_of_8TBps": round((w + s) / step_s / 8e12, 4),
        "streams_verified_identical_to_single_stream": min(a.verify, a.streams) if identical else 0,
        "tokens_identical": bool(identical),
        "dtype": "int8 weights x int8 activations on v_mfma_i32_16x16x64_i8, f32 group terms folded in reference order",
        "data": "synthetic", "tolerance_mode": tol,
        "config": {"workload": f"{a.shape} Q8 batch={a.streams} concurrent greedy streams, {a.steps} steps, ctx {a.ctx}, "
                               f"generate-mode call pattern", "seed": a.seed},
    }))
    return 0 if identical else 1


if __name__ == "__main__":
    sys.exit(main())
