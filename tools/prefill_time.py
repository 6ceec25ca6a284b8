#!/usr/bin/env python3
"""python tools/prefill_time.py <shape> <n_prompt> [ctx] -- batched prefill timing (for rocprofv3 runs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck
name, n = sys.argv[1], int(sys.argv[2]); ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
sh = ck.SHAPES[name]; path = f"/tmp/q3_{name}.bin"
ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
prompt = ck.iter_prompt_tokens(sh, 5, n)
with q3.TransformerBuilder(path).with_ctx_length(ctx).build() as t:
    t.prefill(prompt[:40], 0, batched=True)
    t.reset_kv()
    t0 = time.perf_counter(); t.prefill(prompt, 0, batched=True); dt = time.perf_counter() - t0
    print(f"{name}: batched prefill of {n} tokens: {dt*1e3:.1f} ms, {n/dt:.0f} tok/s, {dt/ (n/32) *1e6 / sh.n_layers:.1f} us per layer per 32-position block")
