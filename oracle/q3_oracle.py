"""ctypes binding of the C oracle (oracle/libq3_oracle.so).  TEST INFRASTRUCTURE ONLY -- importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from the product package.
PARITY UNPINNED for forward() (see q3_oracle.h)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# Q3_ORACLE_LIB selects another build of the same source (the ASan/UBSan one: `make -C oracle libq3_oracle_asan.so`,
# run with LD_PRELOAD=$(gcc -print-file-name=libasan.so) -- see docs/HISTORY.md section 6)
_LIB_PATH = os.environ.get("Q3_ORACLE_LIB") or os.path.join(_HERE, "libq3_oracle.so")


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "architecture_id", "dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "head_dim", "seq_len",
        "vocab_size", "group_size", "shared_classifier")]


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("q3_oracle.c", "q3_oracle.h", "Makefile")]
    if os.environ.get("Q3_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libq3_oracle.so"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp, i8p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_int8), C.POINTER(C.c_uint8)
        sz = C.c_size_t
        L.q3o_last_error.restype = C.c_char_p
        L.q3o_quantize.argtypes = [i8p, fp, fp, sz, sz]
        L.q3o_dequantize.argtypes = [i8p, fp, fp, sz, sz]
        L.q3o_matmul.argtypes = [fp, i8p, fp, i8p, fp, sz, sz, sz]
        L.q3o_rmsnorm.argtypes = [fp, fp, fp, sz]
        L.q3o_rope_freqs.argtypes = [fp, sz, sz]
        L.q3o_rope_apply.argtypes = [fp, sz, fp]
        L.q3o_softmax.argtypes = [fp, sz]
        L.q3o_swiglu.argtypes = [fp, fp, sz]
        L.q3o_attention.argtypes = [fp, fp, fp, fp, fp, fp, sz, sz, sz, sz]
        L.q3o_sample_argmax.argtypes = [fp, sz]
        L.q3o_sample_argmax.restype = sz
        u64p = C.POINTER(C.c_uint64)
        L.q3o_random_u32.argtypes = [u64p]
        L.q3o_random_u32.restype = C.c_uint32
        L.q3o_random_f32.argtypes = [u64p]
        L.q3o_random_f32.restype = C.c_float
        L.q3o_sample_mult.argtypes = [fp, sz, C.c_float]
        L.q3o_sample_mult.restype = sz
        L.q3o_sample_topp.argtypes = [fp, sz, C.c_float, C.c_float]
        L.q3o_sample_topp.restype = sz
        L.q3o_sample.argtypes = [fp, sz, C.c_float, C.c_float, u64p]
        L.q3o_sample.restype = sz
        L.q3o_round_half_to_even.argtypes = [C.c_float]
        L.q3o_round_half_to_even.restype = C.c_float
        L.q3o_quantize_q80.argtypes = [i8p, fp, fp, fp, sz, sz]
        L.q3o_quantize_q80.restype = C.c_int
        L.q3o_find_optimal_group_size.argtypes = [sz, sz]
        L.q3o_find_optimal_group_size.restype = sz
        L.q3o_write_header.argtypes = [u8p, C.POINTER(Config), C.c_int32]
        L.q3o_read_config.argtypes = [u8p, sz, C.POINTER(Config)]
        L.q3o_read_config.restype = C.c_int
        L.q3o_create.argtypes = [C.c_char_p, C.c_uint32]
        L.q3o_create.restype = C.c_void_p
        L.q3o_destroy.argtypes = [C.c_void_p]
        L.q3o_get_config.argtypes = [C.c_void_p, C.POINTER(Config)]
        L.q3o_forward.argtypes = [C.c_void_p, sz, sz]
        L.q3o_forward.restype = fp
        L.q3o_reset.argtypes = [C.c_void_p]
        for n in ("q3o_tap_x", "q3o_key_cache", "q3o_value_cache"):
            getattr(L, n).argtypes = [C.c_void_p]
            getattr(L, n).restype = fp
        L.q3o_num_threads.restype = C.c_int
        L.q3o_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i8(a):
    return a.ctypes.data_as(C.POINTER(C.c_int8))


def last_error() -> str:
    return lib().q3o_last_error().decode()


def quantize(x, group_size):
    x = np.ascontiguousarray(x, dtype=np.float32)
    q = np.zeros(x.size, dtype=np.int8)
    s = np.zeros(x.size // group_size, dtype=np.float32)
    lib().q3o_quantize(_i8(q), _fp(s), _fp(x), x.size, group_size)
    return q, s


def dequantize(q, s, group_size):
    q = np.ascontiguousarray(q, dtype=np.int8)
    s = np.ascontiguousarray(s, dtype=np.float32)
    x = np.zeros(q.size, dtype=np.float32)
    lib().q3o_dequantize(_i8(q), _fp(s), _fp(x), q.size, group_size)
    return x


def matmul(xq, xs, wq, ws, n, d, group_size):
    xq = np.ascontiguousarray(xq, dtype=np.int8)
    xs = np.ascontiguousarray(xs, dtype=np.float32)
    wq = np.ascontiguousarray(wq, dtype=np.int8)
    ws = np.ascontiguousarray(ws, dtype=np.float32)
    out = np.zeros(d, dtype=np.float32)
    lib().q3o_matmul(_fp(out), _i8(xq), _fp(xs), _i8(wq), _fp(ws), n, d, group_size)
    return out


def rmsnorm(x, w):
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    out = np.zeros_like(x)
    lib().q3o_rmsnorm(_fp(out), _fp(x), _fp(w), x.size)
    return out


def rope_freqs(head_dim, pos):
    cs = np.zeros(head_dim, dtype=np.float32)
    lib().q3o_rope_freqs(_fp(cs), head_dim, pos)
    return cs.reshape(head_dim // 2, 2)


def rope_apply(v, cs):
    v = np.array(v, dtype=np.float32, copy=True)
    cs = np.ascontiguousarray(cs, dtype=np.float32)
    lib().q3o_rope_apply(_fp(v), v.size, _fp(cs))
    return v


def softmax(a):
    a = np.array(a, dtype=np.float32, copy=True)
    lib().q3o_softmax(_fp(a), a.size)
    return a


def swiglu(g, u):
    g = np.array(g, dtype=np.float32, copy=True)
    u = np.ascontiguousarray(u, dtype=np.float32)
    lib().q3o_swiglu(_fp(g), _fp(u), g.size)
    return g


def attention(q, key_layer, value_layer, q_norm_w, k_norm_w, pos, n_heads, n_kv_heads, head_dim):
    """returns (xb, q_after, key_layer_after) -- q and the K row `pos` are modified like the reference."""
    q = np.array(q, dtype=np.float32, copy=True)
    k = np.array(key_layer, dtype=np.float32, copy=True)
    v = np.ascontiguousarray(value_layer, dtype=np.float32)
    xb = np.zeros(n_heads * head_dim, dtype=np.float32)
    qw = np.ascontiguousarray(q_norm_w, dtype=np.float32)
    kw = np.ascontiguousarray(k_norm_w, dtype=np.float32)
    lib().q3o_attention(_fp(xb), _fp(q), _fp(k), _fp(v), _fp(qw), _fp(kw), pos, n_heads, n_kv_heads, head_dim)
    return xb, q, k


def sample_argmax(logits) -> int:
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    return int(lib().q3o_sample_argmax(_fp(logits), logits.size))


class Sampler:
    """sampler.rs:14-139 (Sampler::new / sample): temperature, top-p and the xorshift64* stream."""

    def __init__(self, vocab_size: int, temperature: float, topp: float, rng_seed: int):
        assert vocab_size > 0 and temperature >= 0.0 and 0.0 <= topp <= 1.0
        self.temperature, self.topp = float(temperature), float(topp)
        self.rng_state = C.c_uint64(rng_seed)

    def random_u32(self) -> int:
        return int(lib().q3o_random_u32(C.byref(self.rng_state)))

    def random_f32(self) -> float:
        return float(lib().q3o_random_f32(C.byref(self.rng_state)))

    def sample(self, logits) -> int:
        """Mutates a private copy (the reference mutates the caller's buffer, which `generate` has already copied)."""
        buf = np.array(logits, dtype=np.float32, copy=True)
        return int(lib().q3o_sample(_fp(buf), buf.size, self.temperature, self.topp, C.byref(self.rng_state)))


def sample_mult(probs, coin: float) -> int:
    probs = np.ascontiguousarray(probs, dtype=np.float32)
    return int(lib().q3o_sample_mult(_fp(probs), probs.size, coin))


def sample_topp(probs, topp: float, coin: float) -> int:
    probs = np.ascontiguousarray(probs, dtype=np.float32)
    return int(lib().q3o_sample_topp(_fp(probs), probs.size, topp, coin))


def round_half_to_even(x: float) -> float:
    return float(lib().q3o_round_half_to_even(x))


def quantize_q80(w, group_size):
    w = np.ascontiguousarray(w, dtype=np.float32)
    q = np.zeros(w.size, dtype=np.int8)
    s = np.zeros(max(1, w.size // max(1, group_size)), dtype=np.float32)
    err = C.c_float(0)
    rc = lib().q3o_quantize_q80(_i8(q), _fp(s), C.byref(err), _fp(w), w.size, group_size)
    if rc != 0:
        raise ValueError(last_error())
    return q, s[: w.size // group_size], float(err.value)


def find_optimal_group_size(hidden_dim, requested) -> int:
    return int(lib().q3o_find_optimal_group_size(hidden_dim, requested))


def read_config(data: bytes) -> Config:
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    cfg = Config()
    if lib().q3o_read_config(buf, len(data), C.byref(cfg)) != 0:
        raise ValueError(last_error())
    return cfg


def write_header(cfg: Config, max_seq_len: int) -> bytes:
    buf = (C.c_uint8 * 256)()
    lib().q3o_write_header(buf, C.byref(cfg), max_seq_len)
    return bytes(buf)


class OracleModel:
    """Qwen3Transformer on the CPU oracle: forward(token,pos) -> logits (copy)."""

    def __init__(self, path: str, ctx_len: int = 0):
        self._h = lib().q3o_create(path.encode(), ctx_len)
        if not self._h:
            raise RuntimeError(last_error())
        self.config = Config()
        lib().q3o_get_config(self._h, C.byref(self.config))

    def forward(self, token: int, pos: int) -> np.ndarray:
        p = lib().q3o_forward(self._h, token, pos)
        if not p:
            raise IndexError(last_error())
        return np.ctypeslib.as_array(p, shape=(self.config.vocab_size,)).copy()

    def get_config(self):
        """Transformer::get_config (models/mod.rs:17)"""
        return self.config

    def tap_x(self) -> np.ndarray:
        return np.ctypeslib.as_array(lib().q3o_tap_x(self._h), shape=(self.config.dim,)).copy()

    def kv_cache(self):
        c = self.config
        shp = (c.n_layers, c.seq_len, c.n_kv_heads * c.head_dim)
        k = np.ctypeslib.as_array(lib().q3o_key_cache(self._h), shape=shp).copy()
        v = np.ctypeslib.as_array(lib().q3o_value_cache(self._h), shape=shp).copy()
        return k, v

    def reset(self):
        lib().q3o_reset(self._h)

    def close(self):
        if self._h:
            lib().q3o_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def num_threads() -> int:
    return int(lib().q3o_num_threads())


def set_num_threads(n: int):
    lib().q3o_set_num_threads(n)
