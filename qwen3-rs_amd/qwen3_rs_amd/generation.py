"""The call patterns that reach `forward` (qwen3-inference/src/generation.rs) and the tok/s definition.

Token ids in, token ids out: tokenizer encode/decode (tokenizer.rs) is outside the hot path.  Works with
any object exposing forward(token,pos)->logits and get_config() -- the HIP Transformer or the test oracle.
"""
from __future__ import annotations

import time
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import numpy as np


def sample_argmax(logits: np.ndarray) -> int:
    """Sampler::sample_argmax (sampler.rs:57-59): Iterator::max_by(total_cmp) keeps the LAST maximum."""
    bits = np.ascontiguousarray(logits, dtype=np.float32).view(np.int32).astype(np.int64)
    key = np.where(bits < 0, bits ^ 0x7FFFFFFF, bits)
    if key.size == 0:
        return 0
    return int(np.nonzero(key == key.max())[0][-1])


class TokenMetrics:
    """generation.rs:198-233: clock starts before the first generated token's forward, count++ per sample."""

    def __init__(self):
        self.start_time: Optional[float] = None
        self.generated_count = 0
        self.elapsed = 0.0

    def start_generation(self):
        if self.start_time is None:
            self.start_time = time.perf_counter()

    def increment_token(self):
        self.generated_count += 1

    def report(self) -> Tuple[int, float, float]:
        if self.start_time is not None:
            self.elapsed = time.perf_counter() - self.start_time
        tps = self.generated_count / self.elapsed if self.elapsed > 0 else 0.0
        return self.generated_count, self.elapsed, tps


def generate(transformer, prompt_tokens: Sequence[int], max_new_tokens: Optional[int] = None,
             stop_tokens: Iterable[int] = (), sample: Callable[[np.ndarray], int] = sample_argmax,
             on_logits: Optional[Callable[[int, int, np.ndarray], None]] = None):
    """`generate` (generation.rs:9-48).  Prompt tokens 0..n-2 never reach forward(): the first call is
    forward(prompt[n-1], n-1) over a zero KV prefix.  Returns (generated tokens incl. a terminating one,
    TokenMetrics).  max_new_tokens bounds the loop (the reference only stops at seq_len / BOS / EOS)."""
    if len(prompt_tokens) == 0:
        raise ValueError("Please provide a prompt")
    stop = set(stop_tokens)
    seq_len = transformer.get_config().seq_len
    metrics = TokenMetrics()
    out: List[int] = []
    pos, token = 0, prompt_tokens[0]
    while pos < seq_len:
        if pos < len(prompt_tokens) - 1:
            nxt = prompt_tokens[pos + 1]
        else:
            if max_new_tokens is not None and len(out) >= max_new_tokens:
                break
            metrics.start_generation()
            logits = np.array(transformer.forward(token, pos), copy=True)   # generation.rs:159-160
            if on_logits is not None:
                on_logits(token, pos, logits)
            nxt = sample(logits)
            metrics.increment_token()
            out.append(nxt)
            if nxt in stop:
                break
        token = nxt
        pos += 1
    metrics.report()
    return out, metrics


def chat_turn(transformer, prompt_tokens: Sequence[int], pos: int, max_new_tokens: int,
              stop_tokens: Iterable[int] = (), sample: Callable[[np.ndarray], int] = sample_argmax,
              on_logits: Optional[Callable[[int, int, np.ndarray], None]] = None):
    """One user turn + assistant turn of `chat` (generation.rs:94-151): every prompt token goes through
    forward() one at a time (sequential prefill, a sample drawn and discarded for each), then decode until
    a stop token.  Returns (generated tokens, next pos, TokenMetrics)."""
    stop = set(stop_tokens)
    seq_len = transformer.get_config().seq_len
    next_token = 0
    for tok in prompt_tokens:                       # handle_user_turn, generation.rs:116-123
        if pos >= seq_len:
            break
        logits = np.array(transformer.forward(tok, pos), copy=True)
        if on_logits is not None:
            on_logits(tok, pos, logits)
        next_token = sample(logits)
        pos += 1
    metrics = TokenMetrics()
    out: List[int] = []
    while len(out) < max_new_tokens and pos < seq_len:   # handle_assistant_turn, generation.rs:127-151
        if next_token in stop:
            break
        metrics.start_generation()
        out.append(next_token)
        logits = np.array(transformer.forward(next_token, pos), copy=True)
        if on_logits is not None:
            on_logits(next_token, pos, logits)
        next_token = sample(logits)
        metrics.increment_token()
        pos += 1
    metrics.report()
    return out, pos, metrics
