#!/usr/bin/env python3
"""Sweep the specialised GEMV configurations (Q3_CFG_<FAMILY>=k, q3_engine.hip kGemvCfgs) of one model shape:
for every family and candidate index, tok/s of the device-resident greedy loop and the family's launch period.
usage: cfg_sweep.py [shape] [ntok]   (env Q3_SWEEP="QKV:4,W13:4,WO:2,W2:3" overrides the candidate counts)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck

name = sys.argv[1] if len(sys.argv) > 1 else "qwen3-0.6b"
ntok = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sh = ck.SHAPES[name]
path = f"/tmp/q3_{name}.bin"
ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
fams = {"QKV": "qkv", "W13": "w13", "WO": "wo", "W2": "w2", "LMHEAD": "lm_head"}
counts = {"QKV": 4, "W13": 4, "WO": 2, "W2": 3, "LMHEAD": 2}
if os.environ.get("Q3_SWEEP"):
    counts = {kv.split(":")[0]: int(kv.split(":")[1]) for kv in os.environ["Q3_SWEEP"].split(",")}


def run(env):
    for k in ("Q3_CFG_QKV", "Q3_CFG_W13", "Q3_CFG_WO", "Q3_CFG_W2", "Q3_CFG_LMHEAD"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t = q3.TransformerBuilder(path).with_ctx_length(1024).with_strict(True).build()
    t.generate_greedy(5, 7, 8)
    best = 0.0
    for _ in range(3):
        t.reset_kv()
        t0 = time.perf_counter(); toks = t.generate_greedy(5, 7, ntok); dt = time.perf_counter() - t0
        best = max(best, ntok / dt)
    prof = {n: (ms / max(cnt, 1) * 1e3) for n, ms, cnt in t.profile(5, 7 + ntok // 2, 10)}
    t.close()
    return best, prof, [int(x) for x in toks]


base, bprof, btoks = run({})
print(f"{name}: default {base:.1f} tok/s  " + " ".join(f"{k}={v:.2f}us" for k, v in bprof.items()), flush=True)
basegen, gprof, gtoks = run({"Q3_CFG_QKV": "-1", "Q3_CFG_W13": "-1", "Q3_CFG_WO": "-1", "Q3_CFG_W2": "-1"})
print(f"{name}: generic(r02) {basegen:.1f} tok/s  " + " ".join(f"{k}={v:.2f}us" for k, v in gprof.items()) + f"  tokens default==generic: {btoks == gtoks}", flush=True)
for F, n in counts.items():
    for k in range(n):
        v, prof, tk = run({f"Q3_CFG_{F}": str(k)})
        ok = tk == gtoks
        print(f"  {F}={k}: {v:.1f} tok/s  {fams[F]} {prof[fams[F]]:.2f} us  tokens_same={ok}", flush=True)
