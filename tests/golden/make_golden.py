#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the build container; outputs are data only).

Writes, for each tiny shape:
  <name>.bin          synthetic checkpoint in the reference's on-disk format (qwen3_rs_amd.checkpoint)
  <name>.golden.npz   logits of a fixed forward(token,pos) call list, produced by the INDEPENDENT numpy
                      restatement oracle/np_oracle.py (not by the C oracle and not by the HIP engine),
                      plus the greedy tokens of the `generate` and `chat` call patterns.
The reference itself (Rust) cannot run here, so these pin the two restatements and the engine against
each other, not against rustc output -- see oracle/q3_oracle.h.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
sys.path.insert(0, ROOT)

from qwen3_rs_amd import checkpoint as ck  # noqa: E402
from qwen3_rs_amd.generation import generate, chat_turn  # noqa: E402
from oracle import np_oracle as no  # noqa: E402

FIXTURES = {"tiny": 7, "tiny-untied": 11}


class _NpAdapter:
    def __init__(self, m):
        self.m = m

    def forward(self, token, pos):
        return self.m.forward(token, pos)

    def get_config(self):
        class C:
            pass
        c = C()
        c.seq_len = self.m.seq_len
        return c


def main():
    for name, seed in FIXTURES.items():
        shape = ck.SHAPES[name]
        path = os.path.join(HERE, f"{name}.bin")
        ck.write_synthetic_checkpoint(path, shape, seed=seed, sparse_zero_groups=True)
        sha = hashlib.sha256(open(path, "rb").read()).hexdigest()
        # 1. a fixed call list with out-of-order and repeated positions (KV rows rewritten in place)
        calls = [(3, 0), (17, 1), (200, 2), (5, 3), (9, 1), (250, 4), (0, 5), (31, 7), (77, 6)]
        m = no.NpQwen3(path)
        logits = np.stack([m.forward(t, p) for t, p in calls]).astype(np.float32)
        key = m.key.copy()
        val = m.val.copy()
        # 2. generate pattern (generation.rs:9-48): zero KV prefix, first call at pos n-1
        prompt = ck.iter_prompt_tokens(shape, seed, 6)
        gen_tokens, _ = generate(_NpAdapter(no.NpQwen3(path)), prompt, max_new_tokens=12, sample=no.argmax_last)
        # 3. chat pattern (generation.rs:94-151): sequential prefill then decode
        chat_tokens, chat_pos, _ = chat_turn(_NpAdapter(no.NpQwen3(path)), prompt, 0, 10, sample=no.argmax_last)
        np.savez_compressed(os.path.join(HERE, f"{name}.golden.npz"), calls=np.array(calls, dtype=np.int64), logits=logits,
                            key_cache=key, value_cache=val, prompt=np.array(prompt, dtype=np.int64),
                            generate_tokens=np.array(gen_tokens, dtype=np.int64),
                            chat_tokens=np.array(chat_tokens, dtype=np.int64), chat_pos=np.int64(chat_pos),
                            checkpoint_sha256=np.array(sha), seed=np.int64(seed))
        print(name, "ckpt", os.path.getsize(path), "bytes sha", sha[:12], "logits", logits.shape, "gen", gen_tokens[:6],
              "chat", chat_tokens[:6])


if __name__ == "__main__":
    main()
