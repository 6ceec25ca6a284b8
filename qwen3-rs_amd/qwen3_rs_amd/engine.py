"""ctypes binding of libqwen3_hip.so (include/qwen3_hip.h) shaped like the reference's Rust API."""
from __future__ import annotations

import ctypes as C
import dataclasses
import os
from typing import List, Optional, Tuple

import numpy as np

FLAG_FAST = 1
FLAG_NO_GRAPH = 2
FLAG_NO_VALUE_T = 4

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_DIST_DIR = os.path.dirname(_PKG_DIR)

# every symbol include/qwen3_hip.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = [
    "q3_create", "q3_get_config", "q3_forward", "q3_destroy", "q3_last_error", "q3_forward_argmax",
    "q3_generate_greedy", "q3_host_generate", "q3_host_sample_argmax", "q3_prefill", "q3_reset_kv", "q3_read_state", "q3_batch_init", "q3_forward_batch",
    "q3_generate_greedy_batch", "q3_batch_reset_kv", "q3_batch_read_state", "q3_prefill_batched", "q3_batch_sampler_set", "q3_sampler_set", "q3_sampler_get_rng", "q3_forward_sample", "q3_generate_sampled", "q3_profile", "q3_profile_name", "q3_parse_header",
    "q3_abi_version", "q3_build_id", "q3_op_quantize", "q3_op_dequantize", "q3_op_matmul", "q3_op_rmsnorm", "q3_op_softmax",
    "q3_op_swiglu", "q3_op_expf", "q3_op_attention", "q3_op_argmax", "q3_op_sample",
]


class Q3Error(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[q3 status {code}] {msg}")
        self.code = code
        self.msg = msg


class _Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "architecture_id", "dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "head_dim", "seq_len",
        "vocab_size", "group_size", "shared_classifier")]


@dataclasses.dataclass(frozen=True)
class ModelConfig:
    """qwen3-inference/src/configuration.rs:18-30"""
    architecture_id: int
    dim: int
    hidden_dim: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    head_dim: int
    seq_len: int
    vocab_size: int
    group_size: int
    shared_classifier: bool

    @staticmethod
    def _from_c(c: _Config) -> "ModelConfig":
        return ModelConfig(c.architecture_id, c.dim, c.hidden_dim, c.n_layers, c.n_heads, c.n_kv_heads, c.head_dim,
                           c.seq_len, c.vocab_size, c.group_size, bool(c.shared_classifier))


def lib_path() -> str:
    return os.environ.get("Q3_HIP_LIB", os.path.join(_DIST_DIR, "libqwen3_hip.so"))


_lib: Optional[C.CDLL] = None


def source_build_id() -> str:
    """The id `make` bakes into the library: sha256 over csrc/* (sorted), include/qwen3_hip.h and the Makefile (compiler flags) -- first 16 hex digits.  A library built with ad-hoc `EXTRA=-D...` switches carries a different id."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_DIST_DIR, "csrc", "*"))) + [os.path.join(os.path.dirname(_DIST_DIR), "include", "qwen3_hip.h"),
                                                                          os.path.join(_DIST_DIR, "Makefile")]
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(b"\n")          # the Makefile appends its EXTRA switches (none for the product build) and a newline
    return h.hexdigest()[:16]


def dev_lib_path() -> str:
    """The developer build (`make -C qwen3-rs_amd dev`, -DQ3_DEV): the same sources plus the A/B switches (Q3_* environment
    variables beyond the documented ones), the kernel forms that lost their A/B, ablation bits and in-kernel timelines."""
    return os.path.join(_DIST_DIR, "libqwen3_hip_dev.so")


_libs: dict = {}


def load_library() -> C.CDLL:
    """Load libqwen3_hip.so.  Fails loudly: the HIP library is the product, there is nothing to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    _lib = _bind(lib_path())
    return _lib


class use_library:
    """Context manager: engines and operators created inside use the library at `path` (tests of the developer build's kernel
    forms: `with q3.use_library(q3.dev_lib_path()): ...`).  Objects keep the library they were created with."""

    def __init__(self, path: str):
        self.path = path

    def __enter__(self):
        global _lib
        self._saved = _lib
        _lib = _bind(self.path)
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        return False


def _bind(path: str) -> C.CDLL:
    path = os.path.abspath(path)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise Q3Error(-1, f"{path} not found: build it with `make -C {_DIST_DIR}` (or __graft_entry__.build())")
    L = C.CDLL(path)
    fp, i8p, u8p, sz = C.POINTER(C.c_float), C.POINTER(C.c_int8), C.POINTER(C.c_uint8), C.c_size_t
    L.q3_last_error.restype = C.c_char_p
    L.q3_abi_version.restype = C.c_uint32
    L.q3_build_id.restype = C.c_char_p
    L.q3_create.argtypes = [C.c_char_p, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
    L.q3_get_config.argtypes = [C.c_void_p, C.POINTER(_Config)]
    L.q3_forward.argtypes = [C.c_void_p, sz, sz]
    L.q3_forward.restype = fp
    L.q3_destroy.argtypes = [C.c_void_p]
    L.q3_destroy.restype = None
    L.q3_forward_argmax.argtypes = [C.c_void_p, sz, sz, C.POINTER(C.c_int32)]
    L.q3_generate_greedy.argtypes = [C.c_void_p, sz, sz, sz, C.POINTER(C.c_int32)]
    L.q3_host_generate.argtypes = [C.c_void_p, sz, sz, sz, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.q3_host_sample_argmax.argtypes = [fp, sz, fp]
    L.q3_host_sample_argmax.restype = sz
    L.q3_prefill.argtypes = [C.c_void_p, C.POINTER(C.c_int32), sz, sz, C.POINTER(C.c_int32)]
    L.q3_prefill_batched.argtypes = [C.c_void_p, C.POINTER(C.c_int32), sz, sz, C.POINTER(C.c_int32)]
    L.q3_sampler_set.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_uint64]
    L.q3_sampler_get_rng.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.q3_forward_sample.argtypes = [C.c_void_p, sz, sz, C.POINTER(C.c_int32)]
    L.q3_generate_sampled.argtypes = [C.c_void_p, sz, sz, sz, C.POINTER(C.c_int32)]
    L.q3_reset_kv.argtypes = [C.c_void_p]
    L.q3_read_state.argtypes = [C.c_void_p, C.c_int, sz, sz, fp]
    i32p = C.POINTER(C.c_int32)
    L.q3_batch_init.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
    L.q3_forward_batch.argtypes = [C.c_void_p, i32p, i32p, C.c_int, fp, i32p]
    L.q3_generate_greedy_batch.argtypes = [C.c_void_p, i32p, i32p, C.c_int, sz, i32p]
    L.q3_batch_sampler_set.argtypes = [C.c_void_p, C.c_float, C.c_float, C.POINTER(C.c_uint64)]
    L.q3_batch_reset_kv.argtypes = [C.c_void_p]
    L.q3_batch_read_state.argtypes = [C.c_void_p, C.c_int, C.c_int, sz, sz, fp]
    L.q3_profile.argtypes = [C.c_void_p, sz, sz, C.c_int, fp, C.POINTER(C.c_int32), C.c_int]
    L.q3_profile_name.argtypes = [C.c_int]
    L.q3_profile_name.restype = C.c_char_p
    L.q3_parse_header.argtypes = [u8p, sz, C.POINTER(_Config)]
    L.q3_op_quantize.argtypes = [i8p, fp, fp, sz, sz, C.c_int]
    L.q3_op_dequantize.argtypes = [i8p, fp, fp, sz, sz, C.c_int]
    L.q3_op_matmul.argtypes = [fp, i8p, fp, i8p, fp, sz, sz, sz, C.c_int]
    L.q3_op_rmsnorm.argtypes = [fp, fp, fp, sz, C.c_uint32, C.c_int]
    L.q3_op_softmax.argtypes = [fp, sz, C.c_uint32, C.c_int]
    L.q3_op_swiglu.argtypes = [fp, fp, sz, C.c_int]
    L.q3_op_expf.argtypes = [fp, sz, C.c_int]
    L.q3_op_attention.argtypes = [fp, fp, fp, fp, fp, fp, sz, sz, sz, sz, sz, C.c_uint32, C.c_int]
    L.q3_op_argmax.argtypes = [fp, sz, C.POINTER(C.c_int32), C.c_int]
    L.q3_op_sample.argtypes = [fp, sz, C.c_float, C.c_float, C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.c_int]
    _libs[path] = L
    return L


def _check(rc: int):
    if rc != 0:
        raise Q3Error(rc, load_library().q3_last_error().decode(errors="replace"))


def parse_header(data: bytes) -> ModelConfig:
    """configuration.rs:77-146 via the library's own parser (no GPU needed)."""
    L = load_library()
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data if data else b"\0")
    cfg = _Config()
    _check(L.q3_parse_header(buf, len(data), C.byref(cfg)))
    return ModelConfig._from_c(cfg)


class Transformer:
    """`trait Transformer` (models/mod.rs:13-18) implemented by the HIP engine."""

    def __init__(self, handle: int, lib: C.CDLL):
        self._h = C.c_void_p(handle)
        self._lib = lib
        cfg = _Config()
        _check(lib.q3_get_config(self._h, C.byref(cfg)))
        self._config = ModelConfig._from_c(cfg)

    # -- the reference surface ------------------------------------------------------------------
    def forward(self, token: int, pos: int) -> np.ndarray:
        """forward(token,pos) -> logits[vocab_size] (models/qwen3.rs:62-79).  The returned array is a VIEW of
        the engine's host buffer, valid until the next call -- the same borrow rule as `&[f32]` from
        `&mut self`; copy it like generation.rs:160 does."""
        if token < 0 or pos < 0:
            raise IndexError("negative index")
        p = self._lib.q3_forward(self._h, token, pos)
        if not p:
            msg = self._lib.q3_last_error().decode(errors="replace")
            if "out of range" in msg:
                raise IndexError(msg)   # the reference panics on slice indexing (layers.rs:73-75,335)
            raise Q3Error(-4, msg)
        return np.ctypeslib.as_array(p, shape=(self._config.vocab_size,))

    def get_config(self) -> ModelConfig:
        return self._config

    # -- extensions -------------------------------------------------------------------------------
    def forward_argmax(self, token: int, pos: int) -> int:
        out = C.c_int32(-1)
        rc = self._lib.q3_forward_argmax(self._h, token, pos, C.byref(out))
        if rc == -3:
            raise IndexError(self._lib.q3_last_error().decode(errors="replace"))
        _check(rc)
        return int(out.value)

    def generate_greedy(self, first_token: int, first_pos: int, n_tokens: int) -> List[int]:
        buf = (C.c_int32 * max(1, n_tokens))()
        rc = self._lib.q3_generate_greedy(self._h, first_token, first_pos, n_tokens, buf)
        if rc == -3:
            raise IndexError(self._lib.q3_last_error().decode(errors="replace"))
        _check(rc)
        return [int(buf[i]) for i in range(n_tokens)]

    def host_generate(self, first_token: int, first_pos: int, n_tokens: int) -> Tuple[List[int], float]:
        """The reference's decode loop in compiled host code on top of forward(): logits egress + host argmax per
        token (generation.rs:153-162).  Returns (tokens, TokenMetrics seconds)."""
        buf = (C.c_int32 * max(1, n_tokens))()
        secs = C.c_double(0.0)
        rc = self._lib.q3_host_generate(self._h, first_token, first_pos, n_tokens, buf, C.byref(secs))
        if rc == -3:
            raise IndexError(self._lib.q3_last_error().decode(errors="replace"))
        _check(rc)
        return [int(buf[i]) for i in range(n_tokens)], float(secs.value)

    def prefill(self, tokens, first_pos: int = 0, batched: bool = False) -> int:
        """chat-mode prompt loop on the device (generation.rs:116-123); returns the first generated token.
        batched=True walks the prompt in blocks of up to 2,048 positions per weight pass (q3_prefill_batched; Q3_PREFILL_M), same results."""
        arr = (C.c_int32 * len(tokens))(*[int(t) for t in tokens])
        out = C.c_int32(-1)
        fn = self._lib.q3_prefill_batched if batched else self._lib.q3_prefill
        rc = fn(self._h, arr, len(tokens), first_pos, C.byref(out))
        if rc == -3:
            raise IndexError(self._lib.q3_last_error().decode(errors="replace"))
        _check(rc)
        return int(out.value)

    def set_sampler(self, temperature: float, topp: float, rng_seed: int):
        """Sampler::new (sampler.rs:29-42) on the device: subsequent forward_argmax / generate_greedy / prefill calls draw
        with Sampler::sample; temperature 0 restores greedy decoding."""
        _check(self._lib.q3_sampler_set(self._h, temperature, topp, rng_seed))

    def sampler_rng_state(self) -> int:
        out = C.c_uint64(0)
        _check(self._lib.q3_sampler_get_rng(self._h, C.byref(out)))
        return int(out.value)

    def reset_kv(self):
        _check(self._lib.q3_reset_kv(self._h))

    def read_state(self, kind: str, offset: int = 0, count: Optional[int] = None) -> np.ndarray:
        c = self._config
        kinds = {"key": 0, "value": 1, "x": 2}
        total = c.dim if kind == "x" else c.n_layers * c.seq_len * c.n_kv_heads * c.head_dim
        count = total - offset if count is None else count
        out = np.zeros(count, dtype=np.float32)
        _check(self._lib.q3_read_state(self._h, kinds[kind], offset, count, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    # ---- batched decode (include/qwen3_hip.h section 2b): N concurrent generate loops, weights streamed once per step
    def batch_init(self, max_streams: int, ctx_len: int = 0):
        _check(self._lib.q3_batch_init(self._h, max_streams, ctx_len))
        self._batch_ctx = min(ctx_len, self._config.seq_len) if ctx_len else self._config.seq_len

    def _batch_rc(self, rc):
        if rc == -3:
            raise IndexError(self._lib.q3_last_error().decode(errors="replace"))
        _check(rc)

    def forward_batch(self, tokens, pos, want_logits: bool = True):
        """Stream i runs forward(tokens[i], pos[i]); returns (logits [n, vocab] or None, argmax list)."""
        n = len(tokens)
        tk = (C.c_int32 * n)(*[int(t) for t in tokens])
        ps = (C.c_int32 * n)(*[int(p) for p in pos])
        am = (C.c_int32 * n)()
        logits = np.zeros((n, self._config.vocab_size), dtype=np.float32) if want_logits else None
        lp = logits.ctypes.data_as(C.POINTER(C.c_float)) if want_logits else None
        self._batch_rc(self._lib.q3_forward_batch(self._h, tk, ps, n, lp, am))
        return logits, [int(v) for v in am]

    def generate_greedy_batch(self, first_tokens, first_pos, n_steps: int) -> np.ndarray:
        """[n_streams, n_steps] greedy tokens, the whole loop on the device."""
        n = len(first_tokens)
        tk = (C.c_int32 * n)(*[int(t) for t in first_tokens])
        ps = (C.c_int32 * n)(*[int(p) for p in first_pos])
        out = np.zeros((n, n_steps), dtype=np.int32)
        self._batch_rc(self._lib.q3_generate_greedy_batch(self._h, tk, ps, n, n_steps, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def set_batch_sampler(self, temperature: float, topp: float, rng_seeds):
        """one Sampler per stream (sampler.rs:29-42), stream i seeded with rng_seeds[i]; temperature 0 = greedy"""
        seeds = (C.c_uint64 * 32)(*([int(s) for s in rng_seeds] + [0] * (32 - len(rng_seeds))))
        _check(self._lib.q3_batch_sampler_set(self._h, temperature, topp, seeds))

    def batch_reset_kv(self):
        _check(self._lib.q3_batch_reset_kv(self._h))

    def batch_read_state(self, stream: int, kind: str, offset: int = 0, count: Optional[int] = None) -> np.ndarray:
        c = self._config
        kinds = {"key": 0, "value": 1, "x": 2}
        total = c.dim if kind == "x" else c.n_layers * self._batch_ctx * c.n_kv_heads * c.head_dim
        count = total - offset if count is None else count
        out = np.zeros(count, dtype=np.float32)
        _check(self._lib.q3_batch_read_state(self._h, stream, kinds[kind], offset, count, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def profile(self, token: int, pos: int, reps: int = 1):
        """Per kernel-family (name, total ms, launches) over `reps` eager forwards, HIP events on the engine stream."""
        cap = 16
        ms = (C.c_float * cap)()
        n = (C.c_int32 * cap)()
        k = self._lib.q3_profile(self._h, token, pos, reps, ms, n, cap)
        if k < 0:
            _check(k)
        return [(self._lib.q3_profile_name(i).decode(), float(ms[i]), int(n[i])) for i in range(k)]

    def close(self):
        if self._h:
            self._lib.q3_destroy(self._h)
            self._h = C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TransformerBuilder:
    """models/mod.rs:40-74: TransformerBuilder::new(path).with_ctx_length(Some(n)).build()"""

    def __init__(self, checkpoint_path: str):
        self.checkpoint_path = checkpoint_path
        self.ctx_length: Optional[int] = None
        self.device = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("Q3_DEVICE_FROM_RANK") else 0
        # Q3_EAGER_LAUNCH=1: launch kernels eagerly instead of replaying hipGraphs (profiling runs: rocprofv3's kernel trace
        # of this ROCm build crashes on long back-to-back graph replays)
        self.flags = FLAG_NO_GRAPH if os.environ.get("Q3_EAGER_LAUNCH") else 0

    def with_ctx_length(self, ctx_length: Optional[int]) -> "TransformerBuilder":
        self.ctx_length = ctx_length
        return self

    def with_device(self, device: int) -> "TransformerBuilder":
        self.device = device
        return self

    def with_strict(self, strict: bool = True) -> "TransformerBuilder":
        """strict (default): reference summation order, bit-identical logits.  False: tree reductions."""
        self.flags = (self.flags & ~FLAG_FAST) if strict else (self.flags | FLAG_FAST)
        return self

    def with_value_transposed(self, keep: bool = True) -> "TransformerBuilder":
        """keep (default): contexts past 256 positions hold the value cache twice (row-major + transposed, include/qwen3_hip.h
        Q3_FLAG_NO_VALUE_T has the byte counts).  False: the row-major cache only; same results, slower long-context decode."""
        self.flags = (self.flags & ~FLAG_NO_VALUE_T) if keep else (self.flags | FLAG_NO_VALUE_T)
        return self

    def with_graph(self, graph: bool = True) -> "TransformerBuilder":
        self.flags = (self.flags & ~FLAG_NO_GRAPH) if graph else (self.flags | FLAG_NO_GRAPH)
        return self

    def build(self) -> Transformer:
        L = load_library()
        h = C.c_void_p()
        rc = L.q3_create(self.checkpoint_path.encode(), int(self.ctx_length or 0), self.device, self.flags, C.byref(h))
        _check(rc)
        return Transformer(h.value, L)


class _Ops:
    """The reference's public free functions (tensor.rs / layers.rs) executed by the device kernels."""

    def __init__(self, device: int = 0):
        self.device = device

    @staticmethod
    def _fp(a):
        return a.ctypes.data_as(C.POINTER(C.c_float))

    @staticmethod
    def _i8(a):
        return a.ctypes.data_as(C.POINTER(C.c_int8))

    def quantize(self, x, group_size: int) -> Tuple[np.ndarray, np.ndarray]:
        x = np.ascontiguousarray(x, dtype=np.float32)
        q = np.zeros(x.size, dtype=np.int8)
        s = np.zeros(max(1, x.size // group_size), dtype=np.float32)
        _check(load_library().q3_op_quantize(self._i8(q), self._fp(s), self._fp(x), x.size, group_size, self.device))
        return q, s[: x.size // group_size]

    def dequantize(self, q, s, group_size: int) -> np.ndarray:
        q = np.ascontiguousarray(q, dtype=np.int8)
        s = np.ascontiguousarray(s, dtype=np.float32)
        x = np.zeros(q.size, dtype=np.float32)
        _check(load_library().q3_op_dequantize(self._i8(q), self._fp(s), self._fp(x), q.size, group_size, self.device))
        return x

    def matmul(self, xq, xs, wq, ws, n: int, d: int, group_size: int) -> np.ndarray:
        xq = np.ascontiguousarray(xq, dtype=np.int8)
        xs = np.ascontiguousarray(xs, dtype=np.float32)
        wq = np.ascontiguousarray(wq, dtype=np.int8)
        ws = np.ascontiguousarray(ws, dtype=np.float32)
        out = np.zeros(d, dtype=np.float32)
        _check(load_library().q3_op_matmul(self._fp(out), self._i8(xq), self._fp(xs), self._i8(wq), self._fp(ws), n, d,
                                           group_size, self.device))
        return out

    def rmsnorm(self, x, w, strict: bool = True) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        w = np.ascontiguousarray(w, dtype=np.float32)
        out = np.zeros_like(x)
        _check(load_library().q3_op_rmsnorm(self._fp(out), self._fp(x), self._fp(w), x.size,
                                            0 if strict else FLAG_FAST, self.device))
        return out

    def softmax(self, a, strict: bool = True) -> np.ndarray:
        a = np.array(a, dtype=np.float32, copy=True)
        _check(load_library().q3_op_softmax(self._fp(a), a.size, 0 if strict else FLAG_FAST, self.device))
        return a

    def swiglu(self, g, u) -> np.ndarray:
        g = np.array(g, dtype=np.float32, copy=True)
        u = np.ascontiguousarray(u, dtype=np.float32)
        _check(load_library().q3_op_swiglu(self._fp(g), self._fp(u), g.size, self.device))
        return g

    def expf(self, x) -> np.ndarray:
        x = np.array(x, dtype=np.float32, copy=True)
        _check(load_library().q3_op_expf(self._fp(x), x.size, self.device))
        return x

    def attention(self, q, key_layer, value_layer, q_norm_w, k_norm_w, pos, n_heads, n_kv_heads, head_dim,
                  strict: bool = True):
        q = np.array(q, dtype=np.float32, copy=True)
        k = np.array(key_layer, dtype=np.float32, copy=True)
        v = np.ascontiguousarray(value_layer, dtype=np.float32)
        seq_len = k.size // (n_kv_heads * head_dim)
        xb = np.zeros(n_heads * head_dim, dtype=np.float32)
        qw = np.ascontiguousarray(q_norm_w, dtype=np.float32)
        kw = np.ascontiguousarray(k_norm_w, dtype=np.float32)
        _check(load_library().q3_op_attention(self._fp(xb), self._fp(q), self._fp(k), self._fp(v), self._fp(qw),
                                              self._fp(kw), pos, seq_len, n_heads, n_kv_heads, head_dim,
                                              0 if strict else FLAG_FAST, self.device))
        return xb, q, k

    def argmax(self, logits) -> int:
        logits = np.ascontiguousarray(logits, dtype=np.float32)
        out = C.c_int32(-1)
        _check(load_library().q3_op_argmax(self._fp(logits), logits.size, C.byref(out), self.device))
        return int(out.value)

    def sample(self, logits, temperature: float, topp: float, rng_state: int):
        """one Sampler::sample draw on the device; returns (token, new rng_state)"""
        logits = np.ascontiguousarray(logits, dtype=np.float32)
        out, st = C.c_int32(-1), C.c_uint64(rng_state)
        _check(load_library().q3_op_sample(self._fp(logits), logits.size, temperature, topp, C.byref(st), C.byref(out), self.device))
        return int(out.value), int(st.value)


ops = _Ops()
