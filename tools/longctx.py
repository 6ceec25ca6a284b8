#!/usr/bin/env python3
"""Decode speed vs context position on the 0.6B synthetic checkpoint (ctx 4096): us/token at several start positions."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck
sh = ck.SHAPES["qwen3-0.6b"]; path = "/tmp/q3_qwen3-0.6b.bin"
ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
t = q3.TransformerBuilder(path).with_ctx_length(4096).with_strict(bool(int(os.environ.get("Q3_STRICT", "1")))).build()
t.generate_greedy(5, 0, 4)
for p0 in (8, 200, 300, 600, 1200, 2500, 4000):
    t0 = time.perf_counter(); t.generate_greedy(5, p0, 32); dt = time.perf_counter() - t0
    print(f"pos {p0:5d}..{p0+31:5d}: {dt/32*1e6:8.1f} us/token", flush=True)
