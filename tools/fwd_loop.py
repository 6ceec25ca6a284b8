#!/usr/bin/env python3
"""Host-synchronous forward() loop (the path the Rust shim uses): tokens/s incl. logits egress + host argmax."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import numpy as np
import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck
sh = ck.SHAPES["qwen3-0.6b"]; path = "/tmp/q3_qwen3-0.6b.bin"
ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
t = q3.TransformerBuilder(path).with_ctx_length(1024).build()
prompt = ck.iter_prompt_tokens(sh, 1234, 8)
q3.generate(t, prompt, max_new_tokens=8); t.reset_kv()
toks, m = q3.generate(t, prompt, max_new_tokens=64)
print("forward()+copy+host argmax:", round(m.report()[2], 1), "tok/s")
t.reset_kv(); t0 = time.perf_counter(); tok, pos = prompt[-1], 7
for _ in range(64):
    tok = t.forward_argmax(tok, pos); pos += 1
print("forward_argmax():", round(64 / (time.perf_counter() - t0), 1), "tok/s")
t.reset_kv(); t0 = time.perf_counter(); t.generate_greedy(prompt[-1], 7, 64); print("generate_greedy:", round(64 / (time.perf_counter() - t0), 1), "tok/s")
