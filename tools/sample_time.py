#!/usr/bin/env python3
"""Decode speed with the device-side sampler vs greedy (0.6B synthetic checkpoint)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck
name = sys.argv[1] if len(sys.argv) > 1 else "qwen3-0.6b"
sh = ck.SHAPES[name]; path = f"/tmp/q3_{name}.bin"
ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
with q3.TransformerBuilder(path).with_ctx_length(1024).build() as t:
    t.generate_greedy(5, 0, 8)
    for temp, topp in [(0.0, 0.9), (1.0, 1.0), (0.7, 0.9), (0.7, 0.5), (0.3, 0.95)]:
        t.set_sampler(temp, topp, 1234)
        t.reset_kv()
        t0 = time.perf_counter(); toks = t.generate_greedy(5, 7, 64); dt = time.perf_counter() - t0
        print(f"{name} temperature {temp} topp {topp}: {dt/64*1e6:9.1f} us/token  first tokens {toks[:4]}", flush=True)
