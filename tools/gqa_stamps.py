#!/usr/bin/env python3
"""Developer: timeline of the batched per-kv-head attention kernel (k_attn_gqa, dev build, Q3_STAMPS=1) for the LAST position of
the last 32-position block of a batched prefill (the longest context).  python tools/gqa_stamps.py [n_prompt] [shape]"""
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
names = ["norm+rope", "scores (K chunks)", "softmax", "V accumulation", "store"]
for idx in range(0, 40):
    buf = (C.c_uint64 * 96)()
    if lib.q3_dev_batch_stamps(t._h, idx, buf) != 0 or buf[0] == 0 or buf[5] != 0 and buf[6] != 0:
        continue
    st = [buf[i] for i in range(5)]
    if st[4] <= st[0] or st[4] - st[0] > 10**9:
        continue
    c8 = [buf[i] for i in range(8, 14)]
    if c8[0] and c8[3] > c8[2]:      # k_attn_gqa2: wave 0 at chunk 8 of each loop
        print("   chunk 8, wave 0: score dots %d + barrier wait %d;  V fold %d + barrier wait %d" % (c8[1] - c8[0], c8[2] - c8[1], c8[4] - c8[3], c8[5] - c8[4]))
    elif c8[0]:
        print("   chunk 8 of the scores loop: commit %d  barrier %d  issue-next %d  dots %d  barrier %d" % tuple(c8[i + 1] - c8[i] for i in range(5)))
    print(f"launch {idx}: " + "  ".join(f"{names[i]} {st[i + 1] - st[i]}" for i in range(4)) + f"  total {st[4] - st[0]} cycles")
