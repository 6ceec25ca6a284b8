#!/usr/bin/env python3
"""Developer: in-kernel timelines (dev build, Q3_STAMPS=1) of a short greedy run starting at a given position.
   Q3_STAMPS=1 Q3_HIP_LIB=qwen3-rs_amd/libqwen3_hip_dev.so python tools/stamps_at.py <pos> [shape]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck
pos = int(sys.argv[1]); name = sys.argv[2] if len(sys.argv) > 2 else "qwen3-0.6b"
sh = ck.SHAPES[name]; path = f"/tmp/q3_{name}.bin"
ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
t = q3.TransformerBuilder(path).with_ctx_length(int(os.environ.get("Q3_CTX", "1024"))).build()
t.generate_greedy(5, 0, pos)          # fill the cache up to pos
print(f"--- stamps for positions {pos}..{pos + 7}", file=sys.stderr)
t.generate_greedy(5, pos, 8)
t.close()
