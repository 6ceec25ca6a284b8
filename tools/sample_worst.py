#!/usr/bin/env python3
"""Worst case of the device sampler: a flat 151,936-entry distribution (nucleus = most of the vocabulary).
Wall time of q3_op_sample (upload + scratch allocation + one k_sample launch + read-back) for a few (T, top-p) pairs."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import qwen3_rs_amd as q3
rng = np.random.default_rng(3)
n = 151936
for sigma in (0.05, 1.0, 6.0):
    lg = rng.normal(0.0, sigma, n).astype(np.float32)
    for temperature, topp in ((1.0, 1.0), (1.0, 0.9), (1.0, 0.999), (0.7, 0.5)):
        q3.ops.sample(lg, temperature, topp, 1234)
        t0 = time.perf_counter()
        for _ in range(5):
            tok, _ = q3.ops.sample(lg, temperature, topp, 1234)
        dt = (time.perf_counter() - t0) / 5
        print(f"sigma {sigma:4.2f} T {temperature} top-p {topp}: {dt*1e3:7.3f} ms per call (token {tok})", flush=True)
