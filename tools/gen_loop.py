#!/usr/bin/env python3
"""Tiny driver for profiling: device-resident greedy loop on a synthetic checkpoint.
Q3_STAMPS / Q3_ABLATE need the developer build: make -C qwen3-rs_amd dev; Q3_HIP_LIB=qwen3-rs_amd/libqwen3_hip_dev.so"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck
name = os.environ.get("Q3_SHAPE", "qwen3-0.6b")
n = int(os.environ.get("Q3_NTOK", "32"))
sh = ck.SHAPES[name]
path = f"/tmp/q3_{name}.bin"
ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
b = q3.TransformerBuilder(path).with_ctx_length(1024).with_strict(bool(int(os.environ.get("Q3_STRICT", "1"))))
b = b.with_graph(not int(os.environ.get("Q3_EAGER", "0")))
t = b.build()
t.generate_greedy(5, 7, 4)
for _ in range(int(os.environ.get("Q3_REPS", "2"))):
    t.reset_kv()
    t0 = time.perf_counter(); t.generate_greedy(5, 7, n); dt = time.perf_counter() - t0
    print(f"{name}: {n/dt:.1f} tok/s  {dt/n*1e6:.1f} us/tok")
t.close()
