"""Byte-level BPE tokenizer of the reference: the `.tokenizer` file writer (qwen3-export/src/tokenizer_exporter.rs) and
the runtime encoder/decoder (qwen3-inference/src/tokenizer.rs).  Host-side only; nothing here touches the GPU.

Behaviour is the reference's, including its quirks: a token's merge score is looked up by the TOKEN string in a map
keyed by the raw merge strings ("a b"), so almost every score is the default -1e6 (tokenizer_exporter.rs:169-173);
`encode` resolves a string to the FIRST vocabulary entry with those bytes (tokenizer.rs:145-151) and merges the
leftmost best-scoring pair until none is left (tokenizer.rs:208-234).  The reference scans the whole vocabulary for
every lookup (O(V) each); here a dict from bytes to the first index gives the same answers in O(1).
"""
from __future__ import annotations

import ctypes
import ctypes.util
import json
import os
import struct
from typing import Dict, List, Optional, Tuple

DEFAULT_SCORE = -1e6                                   # tokenizer_exporter.rs:81

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.logf.argtypes = [ctypes.c_float]
_libm.logf.restype = ctypes.c_float


def unicode_to_byte_map() -> Dict[str, int]:
    """UnicodeToByteMap::new (tokenizer_exporter.rs:41-66): the GPT-2 byte <-> printable-character table."""
    m: Dict[str, int] = {}
    for lo, hi in ((33, 126), (161, 172), (174, 255)):
        for b in range(lo, hi + 1):
            m[chr(b)] = b
    n = 0
    taken = set(m.values())
    for b in range(256):
        if b not in taken:
            m[chr(256 + n)] = b
            n += 1
    return m


_U2B = unicode_to_byte_map()


def token_to_bytes(token: str) -> bytes:
    """tokenizer_exporter.rs:68-76: mapped characters become their byte, anything else its UTF-8 encoding."""
    out = bytearray()
    for ch in token:
        b = _U2B.get(ch)
        if b is not None:
            out.append(b)
        else:
            out += ch.encode("utf-8")
    return bytes(out)


def merge_rank_score(rank: int) -> float:
    """-((rank + 1) as f32).ln()  (tokenizer_exporter.rs:170), with the platform logf like rustc's intrinsic."""
    return -float(_libm.logf(ctypes.c_float(float(rank + 1))))


def load_token_data(model_dir: str) -> Tuple[Dict[str, int], Dict[str, int], int]:
    """tokenizer_exporter.rs:98-133,186-229: (vocab incl. added_tokens, merge ranks keyed by the merge string, max len)."""
    path = os.path.join(model_dir, "tokenizer.json")
    if not os.path.exists(path):
        raise FileNotFoundError(f"tokenizer.json not found in model directory: {model_dir}")
    try:
        with open(path, "r", encoding="utf-8") as f:
            data = json.load(f)
    except ValueError as err:
        raise ValueError(f"Failed to parse tokenizer.json from {path}: {err}")
    model = data.get("model") if isinstance(data, dict) else None
    vocab_obj = model.get("vocab") if isinstance(model, dict) else None
    if not isinstance(vocab_obj, dict):
        raise ValueError("Could not find vocabulary in tokenizer.json")
    vocab = {tok: int(i) for tok, i in vocab_obj.items() if isinstance(i, int) and not isinstance(i, bool) and i >= 0}
    added = data.get("added_tokens")
    if isinstance(added, list):
        for t in added:
            if isinstance(t, dict) and isinstance(t.get("content"), str) and isinstance(t.get("id"), int):
                vocab[t["content"]] = int(t["id"])
    merges = model.get("merges")
    ranks: Dict[str, int] = {}
    if isinstance(merges, list):
        for rank, mstr in enumerate(merges):
            if isinstance(mstr, str):
                ranks[mstr] = rank
    max_len = max((len(t.encode("utf-8")) for t in vocab), default=0)
    return vocab, ranks, max_len


def export_tokenizer(model_dir: str, output_path: str, bos_token_id: int, eos_token_id: int) -> str:
    """tokenizer_exporter.rs:88-176: writes `<output_path>.tokenizer`:
    u32 max_token_length, u32 bos, u32 eos, then per token in id order f32 score, u32 len, bytes."""
    vocab, ranks, max_len = load_token_data(model_dir)
    ordered = sorted(((i, t) for t, i in vocab.items()), key=lambda p: p[0])
    out = f"{output_path}.tokenizer"
    with open(out, "wb") as f:
        f.write(struct.pack("<III", max_len, bos_token_id, eos_token_id))
        for _, tok in ordered:
            rank = ranks.get(tok)
            score = merge_rank_score(rank) if rank is not None else DEFAULT_SCORE
            raw = token_to_bytes(tok)
            f.write(struct.pack("<fI", score, len(raw)))
            f.write(raw)
    return out


class Tokenizer:
    """tokenizer.rs:28-101.  vocab entries are raw bytes; missing trailing records become empty tokens with score 0."""

    def __init__(self, checkpoint_path: str, vocab_size: int, enable_thinking: bool = False):
        with open(f"{checkpoint_path}.tokenizer", "rb") as f:
            data = f.read()
        self.max_token_length, self.bos_token_id, self.eos_token_id = struct.unpack_from("<III", data, 0)
        off = 12
        self.vocab: List[bytes] = []
        self.merge_scores: List[float] = []
        for _ in range(vocab_size):
            if off + 4 > len(data):
                self.vocab.append(b"")
                self.merge_scores.append(0.0)
                continue
            (score,) = struct.unpack_from("<f", data, off)
            off += 4
            self.merge_scores.append(score)
            if off + 4 > len(data):
                self.vocab.append(b"")
                continue
            (n,) = struct.unpack_from("<I", data, off)
            off += 4
            if off + n > len(data):
                self.vocab.append(b"")
                off = len(data)
                continue
            self.vocab.append(data[off:off + n])
            off += n
        self.vocab_size = vocab_size
        self._first: Dict[bytes, int] = {}
        for i, b in enumerate(self.vocab):
            self._first.setdefault(b, i)                  # position(): the first entry with these bytes
        self.prompt_template = self._load_template(checkpoint_path, False, enable_thinking)
        self.system_prompt_template = self._load_template(checkpoint_path, True, enable_thinking)

    @staticmethod
    def _load_template(checkpoint_path: str, with_system: bool, enable_thinking: bool) -> str:
        """tokenizer.rs:103-122"""
        suffix = {(True, True): ".template.with-system-and-thinking", (True, False): ".template.with-system",
                  (False, True): ".template.with-thinking", (False, False): ".template"}[(with_system, enable_thinking)]
        try:
            with open(f"{checkpoint_path}{suffix}", "r", encoding="utf-8") as f:
                return f.read()
        except OSError:
            import sys
            print(f"Warning: Could not load prompt template {checkpoint_path}{suffix}", file=sys.stderr)
            return ""

    def decode_bytes(self, token: int) -> bytes:
        return self.vocab[token] if 0 <= token < len(self.vocab) else b""

    def decode(self, token: int) -> str:
        """tokenizer.rs:125-143 (partial UTF-8 sequences are kept as bytes there; here they surface as surrogates so
        that concatenating the pieces and encoding with 'surrogateescape' restores the exact bytes)."""
        return self.decode_bytes(token).decode("utf-8", errors="surrogateescape")

    def str_lookup(self, s: str) -> Optional[int]:
        return self._first.get(s.encode("utf-8"))

    def encode(self, text: str) -> List[int]:
        """tokenizer.rs:165-237"""
        tokens: List[int] = []
        chars = list(text)
        i = 0
        while i < len(chars):
            found = False
            if chars[i] == "<":
                limit = min(len(chars), i + self.max_token_length)
                end = next((j for j in range(i + 1, limit) if chars[j] == ">"), None)
                if end is not None:
                    tid = self.str_lookup("".join(chars[i:end + 1]))
                    if tid is not None:
                        tokens.append(tid)
                        i = end + 1
                        found = True
            if not found:
                tid = self.str_lookup(chars[i])
                if tid is not None:
                    tokens.append(tid)
                else:
                    print(f"Warning: unknown character '{chars[i]}' in input, skipping.")
                i += 1
        while True:
            best_score, best_id, best_idx = -1e10, None, None
            for k in range(len(tokens) - 1):
                mid = self._first.get(self.vocab[tokens[k]] + self.vocab[tokens[k + 1]])
                if mid is not None and self.merge_scores[mid] > best_score:
                    best_score, best_id, best_idx = self.merge_scores[mid], mid, k
            if best_id is None:
                break
            tokens[best_idx] = best_id
            del tokens[best_idx + 1]
        return tokens

    def render_prompt(self, pos: int, system_prompt: Optional[str], user_prompt: str) -> str:
        """generation.rs:188-195"""
        if pos == 0 and system_prompt is not None:
            return self.system_prompt_template.replace("%s", f"{system_prompt}\n{user_prompt}")
        return self.prompt_template.replace("%s", user_prompt)


# ---------------------------------------------------------------------------------------------------------------
# Prompt template files (qwen3-export/src/chat_template_exporter.rs): the reference does not render the HF Jinja
# template; it classifies it (ChatML / DeepSeek markers, :64-86) and writes fixed "%s" patterns (:179-221).
# ---------------------------------------------------------------------------------------------------------------
_QWEN3_USER = "<|im_start|>user\n%s<|im_end|>\n<|im_start|>assistant\n"
_QWEN3_SYSTEM = "<|im_start|>system\n%s<|im_end|>\n" + _QWEN3_USER
_NO_THINK_QWEN3 = "<think>\n\n</think>\n\n"
_DEEPSEEK_USER = "<｜User｜>%s<｜Assistant｜>"
_NO_THINK_DEEPSEEK = "<think>\n</think>"


def export_templates(model_dir: str, output_path: str) -> List[str]:
    """Writes `<output_path>.template[.with-thinking|.with-system|.with-system-and-thinking]`; returns the paths."""
    cfg_path = os.path.join(model_dir, "tokenizer_config.json")
    chat_template = None
    if os.path.exists(cfg_path):
        try:
            with open(cfg_path, "r", encoding="utf-8") as f:
                v = json.load(f).get("chat_template")
            chat_template = v if isinstance(v, str) else None
        except ValueError as err:
            raise ValueError(f"Failed to parse tokenizer config JSON: {err}")
    if chat_template is None:
        raise ValueError(f"No chat template found in tokenizer_config.json at {model_dir}")
    if "<|im_start|>" in chat_template and "<|im_end|>" in chat_template:
        kind = "qwen3"
        thinking = "enable_thinking" in chat_template
        system = "system" in chat_template and "messages[0].role" in chat_template
    elif "<｜User｜>" in chat_template and "<｜Assistant｜>" in chat_template:
        kind = "deepseek"
        thinking, system = "think" in chat_template, "system_prompt" in chat_template
    else:
        raise ValueError("Unknown template type, cannot render templates")

    def render(has_system: bool, enable_thinking: bool) -> str:
        if kind == "qwen3":
            return (_QWEN3_SYSTEM if has_system else _QWEN3_USER) + ("" if enable_thinking else _NO_THINK_QWEN3)
        return ("%s" if has_system else "") + _DEEPSEEK_USER + ("" if enable_thinking else _NO_THINK_DEEPSEEK)

    variants = [(".template", False, False)]
    if thinking:
        variants.append((".template.with-thinking", False, True))
    if system:
        variants.append((".template.with-system", True, False))
        if thinking:
            variants.append((".template.with-system-and-thinking", True, True))
    written = []
    for suffix, has_system, enable_thinking in variants:
        path = f"{output_path}{suffix}"
        with open(path, "w", encoding="utf-8") as f:
            f.write(render(has_system, enable_thinking))
        written.append(path)
    return written

