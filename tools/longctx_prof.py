#!/usr/bin/env python3
"""python tools/longctx_prof.py <shape> <pos> [steps] -- decode at a long position (for rocprofv3 runs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck
name, pos = sys.argv[1], int(sys.argv[2]); steps = int(sys.argv[3]) if len(sys.argv) > 3 else 32
sh = ck.SHAPES[name]; path = f"/tmp/q3_{name}.bin"
ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
with q3.TransformerBuilder(path).with_ctx_length(4096).build() as t:
    t.generate_greedy(5, pos, 4)
    t0 = time.perf_counter(); t.generate_greedy(5, pos, steps); dt = time.perf_counter() - t0
    print(f"{name} pos {pos}: {dt/steps*1e6:.1f} us/token")
    if os.environ.get("Q3_PROFILE_FAMILIES", "1") != "0":
        for fam, ms, n in t.profile(5, pos, 8):
            print(f"  {fam:8s} {ms/8*1e3:8.1f} us/token  {n//8:4d} launches  {ms/n*1e3:6.2f} us each")
