#!/usr/bin/env python3
"""Developer: in-kernel timeline of the batched matmuls (dev build, Q3_STAMPS=1). Launch order per layer: quant,QKV,attn,quant,WO,quant,W13,quant,W2."""
import os, sys, ctypes as C
os.environ["Q3_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("Q3_HIP_LIB", os.path.join(ROOT, "qwen3-rs_amd", "libqwen3_hip_dev.so"))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck, engine
name = sys.argv[1] if len(sys.argv) > 1 else "qwen3-8b-dims-l2"
sh = ck.SHAPES[name]; path = f"/tmp/q3_{name}.bin"
ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
t = q3.TransformerBuilder(path).with_ctx_length(2048).build()
t.batch_init(32, 2048)
t.generate_greedy_batch(list(range(5, 37)), [7] * 32, 8)
lib = engine._load() if hasattr(engine, "_load") else t._lib
lib = t._lib
lib.q3_dev_batch_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
names = {1: "QKV", 4: "WO", 6: "W13", 8: "W2"}
for idx, nm in names.items():
    buf = (C.c_uint64 * 96)()
    rc = lib.q3_dev_batch_stamps(t._h, 9 + idx, buf)       # second layer
    print(nm, "rc", rc)
    for ph in range(12):
        st = [buf[ph * 8 + i] for i in range(7)]
        if st[0] == 0: break
        d = [st[i + 1] - st[i] for i in range(6)]
        gap = st[0] - prev_end if ph else 0
        prev_end = st[6]
        print(f"  phase {ph:2d}: gap {gap:6d} | mfma {d[0]:6d} issue {d[1]:5d} barrier {d[2]:6d} fold {d[3]:6d} commit {d[4]:6d} barrier {d[5]:6d}  (ticks @100MHz? see clock_probe)")
