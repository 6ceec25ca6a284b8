"""Checkpoint format: header, tensor order, exporter quantizer, synthetic checkpoint writer.

The engine consumes the reference's Q8 group-quantized checkpoint unchanged.  No model files exist
on disk (no network), so tests and bench.py write *synthetic* checkpoints with this module, following
the reference writer byte for byte:

  header        qwen3-export/src/model_exporter.rs:164-191   (reader: qwen3-inference/src/configuration.rs:77-113)
  norm order    qwen3-export/src/models/qwen3.rs:16-22        (reader: models/qwen3.rs:228-232)
  tensor order  qwen3-export/src/models/qwen3.rs:25-47        (reader: models/qwen3.rs:235-259)
  tensor body   int8[size] then f32[size/group]               model_exporter.rs:302-303, models/mod.rs:92-101
  quantizer     quantize_q80, model_exporter.rs:104-162; round_half_to_even :321-338

numpy only; nothing here touches the GPU or the oracle.
"""
from __future__ import annotations

import dataclasses
import os
import struct
from concurrent.futures import ThreadPoolExecutor
from typing import BinaryIO, Dict, Iterator, List, Tuple

import numpy as np

MAGIC_NUMBER = 0x616A6331  # "ajc1", model_exporter.rs:34
VERSION = 1                # model_exporter.rs:35
HEADER_SIZE = 256          # model_exporter.rs:36
MIN_GROUP_SIZE = 4         # model_exporter.rs:37
ARCH_QWEN3 = 1             # models/mod.rs:70


@dataclasses.dataclass(frozen=True)
class ModelShape:
    """The 11 header fields (configuration.rs:18-30)."""
    dim: int
    hidden_dim: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    vocab_size: int
    max_seq_len: int
    head_dim: int
    shared_classifier: bool
    group_size: int = 64
    architecture_id: int = ARCH_QWEN3

    @property
    def all_heads_dim(self) -> int:
        return self.n_heads * self.head_dim

    @property
    def kv_dim(self) -> int:
        return self.n_kv_heads * self.head_dim

    def quantized_tensors(self) -> List[Tuple[str, int, int, int]]:
        """(name, count, rows, cols) in on-disk order (export/models/qwen3.rs:25-47)."""
        L, d, h = self.n_layers, self.dim, self.hidden_dim
        t = [("embed_tokens", 1, self.vocab_size, d),
             ("q_proj", L, self.all_heads_dim, d),
             ("k_proj", L, self.kv_dim, d),
             ("v_proj", L, self.kv_dim, d),
             ("o_proj", L, d, self.all_heads_dim),
             ("gate_proj", L, h, d),     # w1
             ("down_proj", L, d, h),     # w2
             ("up_proj", L, h, d)]       # w3
        if not self.shared_classifier:
            t.append(("lm_head", 1, self.vocab_size, d))
        return t

    def norm_tensors(self) -> List[Tuple[str, int]]:
        """(name, total f32 count) in on-disk order (export/models/qwen3.rs:16-22)."""
        L = self.n_layers
        return [("input_layernorm", L * self.dim), ("post_attention_layernorm", L * self.dim),
                ("norm", self.dim), ("q_norm", L * self.head_dim), ("k_norm", L * self.head_dim)]

    def file_size(self) -> int:
        g = self.group_size
        n = HEADER_SIZE + 4 * sum(c for _, c in self.norm_tensors())
        for _, cnt, r, c in self.quantized_tensors():
            n += cnt * (r * c + 4 * (r * c // g))
        return n

    def weight_bytes_per_token(self) -> Tuple[int, int]:
        """Algorithmic bytes streamed per decoded token: (int8 bytes, f32 scale bytes) of the L layers'
        seven matrices plus the classifier, each read once."""
        g = self.group_size
        q = 0
        for name, cnt, r, c in self.quantized_tensors():
            if name in ("embed_tokens", "lm_head"):
                continue
            q += cnt * r * c
        q += self.vocab_size * self.dim  # classifier (tied or not) is streamed once
        return q, 4 * (q // g)


# The four models the reference README lists (README.md:33-37); HF config.json values.
SHAPES: Dict[str, ModelShape] = {
    "qwen3-0.6b": ModelShape(1024, 3072, 28, 16, 8, 151936, 40960, 128, True),
    "qwen3-4b": ModelShape(2560, 9728, 36, 32, 8, 151936, 40960, 128, True),
    "qwen3-8b": ModelShape(4096, 12288, 36, 32, 8, 151936, 40960, 128, False),
    "deepseek-r1-0528-qwen3-8b": ModelShape(4096, 12288, 36, 32, 8, 151936, 131072, 128, False),
    # tiny shapes for tests (group | every inner dim)
    "tiny": ModelShape(64, 128, 2, 4, 2, 256, 64, 16, True, 16),
    "tiny-untied": ModelShape(128, 192, 3, 4, 4, 320, 48, 32, False, 32),
    "tiny-g64": ModelShape(256, 384, 2, 4, 2, 512, 96, 64, True, 64),
    "small-hd128": ModelShape(512, 1024, 3, 8, 4, 2048, 256, 128, True, 64),
    "small-longctx": ModelShape(256, 512, 2, 4, 2, 512, 2048, 64, True, 64),
    # the real 4B / 8B layer dimensions with 2 layers and a reduced vocabulary: parity-test cases for
    # BASELINE configs 3-5 that the CPU oracle finishes in seconds (n = 2560, 9728, 4096, 12288; kv_mul 4)
    "qwen3-0.6b-dims-l2": ModelShape(1024, 3072, 2, 16, 8, 4096, 1024, 128, True, 64),
    "qwen3-4b-dims-l2": ModelShape(2560, 9728, 2, 32, 8, 16384, 4096, 128, True, 64),
    "qwen3-8b-dims-l2": ModelShape(4096, 12288, 2, 32, 8, 16384, 4096, 128, False, 64),
}


def find_optimal_group_size(hidden_dim: int, requested: int) -> int:
    """model_exporter.rs:47-57"""
    size = min(requested, hidden_dim)
    while size >= MIN_GROUP_SIZE and hidden_dim % size != 0:
        size //= 2
    return max(size, MIN_GROUP_SIZE)


def round_half_to_even(x: np.ndarray) -> np.ndarray:
    """model_exporter.rs:321-338 -- identical to IEEE roundTiesToEven, which is what np.rint does."""
    return np.rint(np.asarray(x, dtype=np.float32))


def quantize_q80(w: np.ndarray, group_size: int) -> Tuple[np.ndarray, np.ndarray, float]:
    """model_exporter.rs:104-162.  Returns (int8[n], f32 scales[n/group], max_error)."""
    w = np.ascontiguousarray(w, dtype=np.float32).reshape(-1)
    if w.size % group_size != 0:
        raise ValueError("Weight length is not a multiple of group_size")
    g = w.reshape(-1, group_size)
    gmax = np.max(np.abs(g), axis=1)
    scale = np.where(gmax > 0, gmax / np.float32(127.0), np.float32(1.0)).astype(np.float32)
    scaled = g / scale[:, None]
    q = np.clip(np.rint(scaled), -127.0, 127.0)
    q = np.where(np.isnan(q), 0.0, q).astype(np.int8)
    err = float(np.max(np.abs(q.astype(np.float32) * scale[:, None] - g))) if w.size else 0.0
    return q.reshape(-1), scale, err


def header_bytes(shape: ModelShape) -> bytes:
    """model_exporter.rs:164-191"""
    h = struct.pack("<Ii11I", MAGIC_NUMBER, VERSION, shape.architecture_id, shape.dim, shape.hidden_dim,
                    shape.n_layers, shape.n_heads, shape.n_kv_heads, shape.vocab_size, shape.max_seq_len,
                    shape.head_dim, int(shape.shared_classifier), shape.group_size)
    return h + b"\0" * (HEADER_SIZE - len(h))


def read_header(path: str) -> ModelShape:
    """configuration.rs:77-146 (host-side twin of the engine's C++ parser; used by tools/tests)."""
    with open(path, "rb") as f:
        raw = f.read(HEADER_SIZE)
    if len(raw) < HEADER_SIZE:
        raise ValueError("Insufficient data for header")
    (magic, version, arch, dim, hidden, L, nh, nkv, vocab, seq, hd, shared, group) = struct.unpack("<13i", raw[:52])
    if magic != MAGIC_NUMBER:
        raise ValueError(f"Invalid checkpoint magic number: expected {MAGIC_NUMBER:#x}, got {magic:#x}")
    if version != VERSION:
        raise ValueError(f"Unsupported checkpoint version: expected {VERSION}, got {version}")
    return ModelShape(dim, hidden, L, nh, nkv, vocab, seq, hd, shared != 0, group, arch)


def _tensor_rng(seed: int, tensor_idx: int, item: int, chunk: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, tensor_idx, item, chunk]))


def _synth_chunk(seed: int, tidx: int, item: int, chunk: int, n: int, sigma: float, group: int):
    """One chunk of a synthetic tensor: N(0, sigma^2) f32 pushed through quantize_q80's rule (model_exporter.rs:119-136;
    same arithmetic as quantize_q80 above without its max-error pass)."""
    w = _tensor_rng(seed, tidx, item, chunk).standard_normal(n, dtype=np.float32)
    w *= np.float32(sigma)
    g = w.reshape(-1, group)
    gmax = np.max(np.abs(g), axis=1)
    scale = np.where(gmax > 0, gmax / np.float32(127.0), np.float32(1.0)).astype(np.float32)
    np.divide(g, scale[:, None], out=g)
    np.rint(g, out=g)
    np.clip(g, -127.0, 127.0, out=g)
    return g.astype(np.int8).reshape(-1), scale


def write_synthetic_checkpoint(path: str, shape: ModelShape, seed: int = 1234, *, chunk_rows: int = 4096,
                               workers: int | None = None, sparse_zero_groups: bool = False) -> int:
    """Write a synthetic checkpoint of `shape` (i.i.d. N(0, sigma^2) f32 rows pushed
    through the exporter rule quantize_q80, norm weights 1 + N(0, 0.1^2)).  Deterministic in (shape, seed)
    and independent of the worker count (the rng is keyed by tensor / item / chunk).  Returns the number of bytes written.

    sparse_zero_groups: zero a few weight groups so the exporter's scale-1.0 rule for all-zero groups
    (model_exporter.rs:122) is exercised by parity tests.
    """
    g = shape.group_size
    workers = workers or min(32, os.cpu_count() or 1)
    tmp = path + ".tmp"
    # every (tensor, item) in file order; chunk jobs of several items are in flight at once so that tensors with few
    # chunks per item (3 for a 12288 x 4096 matrix) still keep every worker busy
    items = [(tidx, name, item, rows, cols)
             for tidx, (name, cnt, rows, cols) in enumerate(shape.quantized_tensors()) for item in range(cnt)]
    with open(tmp, "wb") as f:
        f.write(header_bytes(shape))
        for ni, (_, count) in enumerate(shape.norm_tensors()):
            rng = _tensor_rng(seed, 1000 + ni, 0, 0)
            w = (1.0 + 0.1 * rng.standard_normal(count, dtype=np.float32)).astype(np.float32)
            f.write(w.tobytes())
        with ThreadPoolExecutor(max_workers=workers) as pool:
            def submit(k):
                tidx, name, item, rows, cols = items[k]
                sigma = 0.05 if name in ("embed_tokens", "lm_head") else float(cols) ** -0.5
                return [pool.submit(_synth_chunk, seed, tidx, item, ci, min(chunk_rows, rows - r0) * cols, sigma, g)
                        for ci, r0 in enumerate(range(0, rows, chunk_rows))]
            pending, nxt, in_flight = {}, 0, 0
            for k in range(len(items)):
                while nxt < len(items) and (in_flight < 2 * workers or nxt <= k):
                    pending[nxt] = submit(nxt)
                    in_flight += len(pending[nxt])
                    nxt += 1
                jobs = pending.pop(k)
                in_flight -= len(jobs)
                name = items[k][1]
                qs, ss = [], []
                for j in jobs:
                    q, sc = j.result()
                    qs.append(q)
                    ss.append(sc)
                if sparse_zero_groups and name not in ("embed_tokens",):
                    # zero group 1 of the first row: scale must be written as 1.0
                    q0 = qs[0].copy()
                    s0 = ss[0].copy()
                    if q0.size >= 2 * g:
                        q0[g:2 * g] = 0
                        s0[1] = 1.0
                    qs[0], ss[0] = q0, s0
                for q in qs:
                    f.write(q.tobytes())
                for sc in ss:
                    f.write(sc.astype("<f4").tobytes())
        n = f.tell()
    os.replace(tmp, path)
    assert n == shape.file_size(), (n, shape.file_size())
    return n


def ensure_synthetic_checkpoint(path: str, shape: ModelShape, seed: int = 1234, **kw) -> str:
    """Write the checkpoint unless a file of the right size and header already exists."""
    try:
        if os.path.getsize(path) == shape.file_size() and read_header(path) == shape:
            return path
    except (OSError, ValueError):
        pass
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    write_synthetic_checkpoint(path, shape, seed, **kw)
    return path


def tensor_offsets(shape: ModelShape) -> Dict[str, Tuple[int, ...]]:
    """Byte offsets of every section: norms -> offset; quantized -> (q_off, s_off, item_stride)."""
    off = HEADER_SIZE
    out: Dict[str, Tuple[int, ...]] = {}
    for name, count in shape.norm_tensors():
        out[name] = (off,)
        off += 4 * count
    g = shape.group_size
    for name, cnt, rows, cols in shape.quantized_tensors():
        size = rows * cols
        stride = size + 4 * (size // g)
        out[name] = (off, off + size, stride)
        off += cnt * stride
    out["__end__"] = (off,)
    return out


def iter_prompt_tokens(shape: ModelShape, seed: int, n: int) -> List[int]:
    """Seed-derived prompt token ids in [0, vocab)."""
    rng = np.random.Generator(np.random.PCG64([seed, 777]))
    return [int(t) for t in rng.integers(0, shape.vocab_size, size=n)]
