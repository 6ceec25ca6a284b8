#!/usr/bin/env python3
"""Developer: in-kernel timeline of the batch-32 activation prologue launches (k_bquant_split; dev build, Q3_STAMPS=1): shader cycles
after kernel entry at [loads + squares issued][barrier: vector landed][exact sum done][quantized][barrier][packed stores issued]."""
import os, sys, ctypes as C
os.environ["Q3_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("Q3_HIP_LIB", os.path.join(ROOT, "qwen3-rs_amd", "libqwen3_hip_dev.so"))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck
name = sys.argv[1] if len(sys.argv) > 1 else "qwen3-8b-dims-l2"
sh = ck.SHAPES[name]; path = f"/tmp/q3_{name}.bin"
ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
t = q3.TransformerBuilder(path).with_ctx_length(2048).build()
t.batch_init(32, 2048)
t.generate_greedy_batch(list(range(5, 37)), [7] * 32, 8)
lib = t._lib
lib.q3_dev_batch_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
for idx in range(0, 40):
    buf = (C.c_uint64 * 96)()
    if lib.q3_dev_batch_stamps(t._h, idx, buf) != 0:
        break
    if (buf[15] & ~0xf) != 0xB0:
        a = [buf[i] for i in range(5)]
        if a[0] and a[4] > a[0] and a[4] - a[0] < 10**7 and all(a[i + 1] >= a[i] for i in range(4)):
            print(f"launch {idx:2d} attention (k_attn_gqa2, stream 31 / kv head 3): norm+rope {a[1]-a[0]}  scores {a[2]-a[1]}  softmax {a[3]-a[2]}  V {a[4]-a[3]}  total {a[4]-a[0]}")
        continue
    st = [buf[i] for i in range(7)]
    print(f"launch {idx:2d} PRO {buf[15] & 0xf}: " + "  ".join(f"{st[i] - st[0]:6d}" if st[i] else "     -" for i in range(1, 7)))
t.close()
