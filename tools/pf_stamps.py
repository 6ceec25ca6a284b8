#!/usr/bin/env python3
"""Developer: timeline of the dense-prefill attention kernel (k_attn_pf2, dev build, Q3_STAMPS=1), wave 0 of the workgroup
(kv head 3, last 8 positions) of the last block of a prefill.  python tools/pf_stamps.py [n_prompt] [shape]"""
import os, sys, ctypes as C
os.environ["Q3_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("Q3_HIP_LIB", os.path.join(ROOT, "qwen3-rs_amd", "libqwen3_hip_dev.so"))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
name = sys.argv[2] if len(sys.argv) > 2 else "qwen3-4b-dims-l2"
sh = ck.SHAPES[name]; path = f"/tmp/q3_{name}.bin"
ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
t = q3.TransformerBuilder(path).with_ctx_length(4096).build()
prompt = ck.iter_prompt_tokens(sh, 3, n)
t.prefill(prompt, 0, batched=True)
lib = t._lib
lib.q3_dev_batch_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
for idx in range(0, 40):
    buf = (C.c_uint64 * 96)()
    if lib.q3_dev_batch_stamps(t._h, idx, buf) != 0 or buf[0] == 0 or buf[4] <= buf[0]:
        continue
    st = [buf[i] for i in range(5)]
    c = [buf[i] for i in range(8, 14)]
    print(f"launch {idx}: scores {st[1]-st[0]}  softmax {st[2]-st[1]}  V {st[3]-st[2]}  store {st[4]-st[3]}  total {st[4]-st[0]} ticks (100 MHz)")
    print(f"   chunk 8: scores commit+barrier {c[1]-c[0]}  dots {c[2]-c[1]};  V commit+barrier {c[4]-c[3]}  fold {c[5]-c[4]}")
