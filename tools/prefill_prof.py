#!/usr/bin/env python3
"""Profiling target: one 2,048-token dense prefill of the 4B shape (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck
name = os.environ.get("Q3_SHAPE", "qwen3-4b")
n = int(os.environ.get("Q3_NPROMPT", "2048"))
shape = ck.SHAPES[name]
path = os.path.join("/tmp", "qwen3-4b-seed1235.q3bin" if name == "qwen3-4b" else f"q3_{name}.bin")
ck.ensure_synthetic_checkpoint(path, shape, seed=1235 if name == "qwen3-4b" else 1234)
prompt = ck.iter_prompt_tokens(shape, 1235, n)
t = q3.TransformerBuilder(path).with_ctx_length(4096).with_graph(False).build()
t.prefill(prompt[:256], 0, batched=True)          # warm-up: packs the weights, builds the plan
t.reset_kv()
t0 = time.perf_counter()
first = t.prefill(prompt, 0, batched=True)
dt = time.perf_counter() - t0
print(f"{name}: prefill {n} tokens in {dt*1e3:.1f} ms = {n/dt:.0f} tok/s, first token {first}")
t.close()
