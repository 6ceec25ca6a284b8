#!/usr/bin/env python3
"""Developer: dense / batched prefill against the sequential prompt loop on one shape -- which (layer, position) K / V rows differ.
   Q3_HIP_LIB=qwen3-rs_amd/libqwen3_hip_dev.so python3 tools/pf_check.py [shape] [block] [n_prompt]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck
name = sys.argv[1] if len(sys.argv) > 1 else "qwen3-0.6b-dims-l2"
block = sys.argv[2] if len(sys.argv) > 2 else "48"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 301
sh = ck.SHAPES[name]; path = f"/tmp/q3_pfcheck_{name}.bin"
ck.ensure_synthetic_checkpoint(path, sh, seed=321)
prompt = ck.iter_prompt_tokens(sh, 9, n)
ctx = 512 if n < 500 else 4096
with q3.TransformerBuilder(path).with_ctx_length(ctx).build() as t:
    t.prefill(prompt[:5], 0)
    want = t.prefill(prompt[5:], 5)
    wk, wv = t.read_state("key"), t.read_state("value")
os.environ["Q3_PREFILL_M"] = block
with q3.TransformerBuilder(path).with_ctx_length(ctx).build() as t:
    t.prefill(prompt[:5], 0)
    got = t.prefill(prompt[5:], 5, batched=True)
    gk, gv = t.read_state("key"), t.read_state("value")
kvd = sh.kv_dim
print("first token", got, want)
for nm, a, b in (("key", gk, wk), ("value", gv, wv)):
    a = a.view(np.int32).reshape(sh.n_layers, ctx, kvd); b = b.view(np.int32).reshape(sh.n_layers, ctx, kvd)
    for l in range(sh.n_layers):
        bad = np.nonzero((a[l] != b[l]).any(axis=1))[0]
        print(nm, "layer", l, "bad rows", len(bad), (bad[:12].tolist(), "...", bad[-6:].tolist()) if len(bad) else "")
