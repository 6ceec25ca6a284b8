import os, sys
sys.path.insert(0, "qwen3-rs_amd")
import qwen3_rs_amd as q3
from qwen3_rs_amd import checkpoint as ck
sh = ck.SHAPES["qwen3-0.6b"]; path = "/tmp/q3_qwen3-0.6b.bin"
ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
t = q3.TransformerBuilder(path).with_ctx_length(1024).build()
t.generate_greedy(5, 7, 40)
for pos in (20, 70, 130, 200):
    pr = t.profile(5, pos, 20)
    print(os.environ.get("Q3_ABLATE", "0"), "pos", pos, [(n, round(ms / c * 1e3, 2)) for n, ms, c in pr if c])
