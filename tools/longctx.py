#!/usr/bin/env python3
"""Decode speed vs context position (ctx 4096): us/token at several start positions, plus the per-family profile.

    python tools/longctx.py [shape] [positions...]      e.g.  python tools/longctx.py qwen3-4b-dims-l2 300 2300
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck
name = sys.argv[1] if len(sys.argv) > 1 else "qwen3-0.6b"
positions = [int(v) for v in sys.argv[2:]] or [8, 200, 300, 600, 1200, 2500, 4000]
sh = ck.SHAPES[name]; path = f"/tmp/q3_{name}.bin"
ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
t = q3.TransformerBuilder(path).with_ctx_length(4096).with_strict(bool(int(os.environ.get("Q3_STRICT", "1")))).build()
L = t.get_config().n_layers
t.generate_greedy(5, 0, 4)
for p0 in positions:
    t0 = time.perf_counter(); t.generate_greedy(5, p0, 32); dt = time.perf_counter() - t0
    prof = t.profile(5, p0 + 16, 4)
    fam = "  ".join(f"{n}={1e3 * ms / max(k, 1):.1f}us" for n, ms, k in prof if k)
    print(f"pos {p0:5d}..{p0+31:5d}: {dt/32*1e6:8.1f} us/token ({L} layers) | per launch: {fam}", flush=True)
