"""np_oracle.py -- SECOND, independent CPU restatement of the reference forward pass (numpy float32).

TEST INFRASTRUCTURE ONLY (same rule as q3_oracle.c: tests/, smoke() and bench.py's cpu_baseline leg are
the only importers).  PARITY UNPINNED by the reference for forward(): the reference (Rust) has no tests
or golden vectors for this path and cannot be built here; this file exists so that the C oracle is
checked by a restatement written separately, from the reference text, in a different style (whole-array
numpy with sequential `cumsum` reductions instead of scalar loops).  It is slow: tiny shapes only.

Citations are into /root/reference (reinterpretcat/qwen3-rs @ 2025-09-05).

Sequential-sum rule: Rust's `iter.sum::<f32>()` is a strict left fold from -0.0.  `np.cumsum(.., dtype=
float32)` is `add.accumulate`, a strict left-to-right accumulation in float32 (no pairwise blocking), and
starting from a[0] is identical to starting from -0.0.  Transcendentals go through glibc's float
routines via ctypes (numpy's own SIMD powf/cosf/sinf/expf differ in the last bit), because Rust's std
calls the platform libm.
"""
from __future__ import annotations

import ctypes
import ctypes.util
import struct
from typing import List

import numpy as np

f32 = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _n in ("expf", "cosf", "sinf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
_libm.powf.restype = ctypes.c_float
_libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]


def _map1(fn, a: np.ndarray) -> np.ndarray:
    flat = np.asarray(a, dtype=f32).reshape(-1)
    out = np.fromiter((fn(float(v)) for v in flat), dtype=f32, count=flat.size)
    return out.reshape(np.shape(a))


def expf(a):
    return _map1(_libm.expf, a)


def seq_sum(a: np.ndarray, axis: int = -1) -> np.ndarray:
    """strict left-to-right float32 sum along axis"""
    a = np.asarray(a, dtype=f32)
    if a.shape[axis] == 0:
        return np.zeros(np.delete(a.shape, axis), dtype=f32) * f32(-1.0)  # -0.0
    return np.take(np.cumsum(a, axis=axis, dtype=f32), -1, axis=axis)


# ---------------------------------------------------------------- tensor.rs
def quantize(x: np.ndarray, group_size: int):
    """tensor.rs:91-119"""
    g = np.asarray(x, dtype=f32).reshape(-1, group_size)
    wmax = np.max(np.abs(g), axis=1).astype(f32)
    scale = (wmax / f32(127.0)).astype(f32)
    safe = np.where(scale != 0, scale, f32(1.0)).astype(f32)
    qv = np.where(scale[:, None] != 0, g / safe[:, None], f32(0.0)).astype(f32)
    # f32::round: half away from zero (x - trunc(x) is exact in float32)
    t = np.trunc(qv)
    r = np.where(np.abs(qv - t) >= f32(0.5), t + np.sign(qv), t)
    q = np.clip(r, -128, 127).astype(np.int8)  # saturating `as i8`
    return q.reshape(-1), scale


def dequantize(q: np.ndarray, s: np.ndarray, group_size: int) -> np.ndarray:
    """tensor.rs:72-80"""
    return (q.astype(f32).reshape(-1, group_size) * s[:, None].astype(f32)).astype(f32).reshape(-1)


def matmul(xq: np.ndarray, xs: np.ndarray, wq: np.ndarray, ws: np.ndarray, n: int, d: int, group_size: int):
    """tensor.rs:23-62"""
    ng = n // group_size
    w = wq[: d * n].reshape(d, ng, group_size).astype(np.int32)
    x = xq[:n].reshape(1, ng, group_size).astype(np.int32)
    idot = (w * x).sum(axis=2, dtype=np.int32)                     # exact
    term = (idot.astype(f32) * ws[: d * ng].reshape(d, ng).astype(f32)).astype(f32)
    term = (term * xs[:ng].reshape(1, ng).astype(f32)).astype(f32)
    return seq_sum(term, axis=1)


# ---------------------------------------------------------------- layers.rs
EPS = f32(1e-6)           # layers.rs:6
ROPE_BASE = 1e6           # layers.rs:9


def rmsnorm(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """layers.rs:109-131"""
    x = np.asarray(x, dtype=f32)
    ss = seq_sum((x * x).astype(f32))
    f = f32(1.0) / np.sqrt((ss / f32(x.size)).astype(f32) + EPS, dtype=f32)
    return (w.astype(f32) * (f * x).astype(f32)).astype(f32)


def rope_freqs(head_dim: int, pos: int):
    """layers.rs:161-171"""
    half = head_dim // 2
    cs = np.zeros((half, 2), dtype=f32)
    for i in range(half):
        e = f32(-f32(i) / f32(half))
        freq = f32(_libm.powf(ROPE_BASE, float(e)))
        angle = f32(f32(pos) * freq)
        cs[i, 0] = _libm.cosf(float(angle))
        cs[i, 1] = _libm.sinf(float(angle))
    return cs


def rope_apply(v: np.ndarray, cs: np.ndarray) -> np.ndarray:
    """layers.rs:173-185"""
    half = v.size // 2
    x, y = v[:half].astype(f32), v[half:].astype(f32)
    c, s = cs[:, 0], cs[:, 1]
    nx = ((x * c).astype(f32) - (y * s).astype(f32)).astype(f32)
    ny = ((x * s).astype(f32) + (y * c).astype(f32)).astype(f32)
    return np.concatenate([nx, ny])


def softmax(a: np.ndarray) -> np.ndarray:
    """layers.rs:495-506"""
    a = np.asarray(a, dtype=f32)
    mx = np.max(a) if a.size else f32(-np.inf)
    e = expf((a - mx).astype(f32))
    inv = f32(1.0) / seq_sum(e)
    return (e * inv).astype(f32)


def swiglu(g: np.ndarray, u: np.ndarray) -> np.ndarray:
    """layers.rs:472-475"""
    den = (f32(1.0) + expf((-g).astype(f32))).astype(f32)
    sw = (g * (f32(1.0) / den).astype(f32)).astype(f32)
    return (sw * u).astype(f32)


def argmax_last(logits: np.ndarray) -> int:
    """sampler.rs:57-59 (last maximum under total order; no NaNs expected in tests)"""
    bits = np.asarray(logits, dtype=f32).view(np.int32).astype(np.int64)
    key = np.where(bits < 0, bits ^ 0x7FFFFFFF, bits)
    m = key.max()
    return int(np.nonzero(key == m)[0][-1])


# ---------------------------------------------------------------- models/qwen3.rs
class NpSampler:
    """sampler.rs:14-139, independent of the C restatement: Python integers for the xorshift64* stream, np.cumsum
    (sequential f32) for the running sums, a key sort for the nucleus (ties: ascending index, see q3_oracle.h)."""
    M = (1 << 64) - 1

    def __init__(self, vocab_size: int, temperature: float, topp: float, rng_seed: int):
        self.temperature, self.topp, self.state = f32(temperature), f32(topp), int(rng_seed) & self.M

    def random_u32(self) -> int:
        s = self.state
        s ^= s >> 12
        s ^= (s << 25) & self.M
        s ^= s >> 27
        self.state = s
        return ((s * 0x2545F4914F6CDD1D) & self.M) >> 32

    def random_f32(self):
        return f32(f32(self.random_u32() >> 8) / f32(16777216.0))

    @staticmethod
    def sample_mult(p: np.ndarray, coin) -> int:
        cdf = np.cumsum(p.astype(f32), dtype=f32)            # 0.0 + p0 == p0: same partial sums as the reference loop
        hit = np.nonzero(f32(coin) < cdf)[0]
        return int(hit[0]) if hit.size else max(p.size - 1, 0)

    def sample_topp(self, p: np.ndarray, coin) -> int:
        n = p.size
        cutoff = f32(f32(1.0) - self.topp) / f32(max(n - 1, 1))
        idx = np.nonzero(p >= cutoff)[0]
        if idx.size == 0:
            return 0
        order = np.lexsort((idx, -p[idx].astype(np.float64)))   # prob descending, then index ascending
        sp, si = p[idx][order].astype(f32), idx[order]
        cum = np.cumsum(sp, dtype=f32)
        over = np.nonzero(cum > self.topp)[0]
        last = int(over[0]) if over.size else sp.size - 1
        r = f32(f32(coin) * cum[last])
        hit = np.nonzero(r < cum[: last + 1])[0]
        return int(si[hit[0]]) if hit.size else int(si[last])

    def sample(self, logits: np.ndarray) -> int:
        lg = np.asarray(logits, dtype=f32)
        if self.temperature == 0.0:
            return argmax_last(lg)
        p = softmax((lg / self.temperature).astype(f32))
        coin = self.random_f32()
        if self.topp <= 0.0 or self.topp >= 1.0:
            return self.sample_mult(p, coin)
        return self.sample_topp(p, coin)


class NpQwen3:
    """Qwen3Transformer (models/qwen3.rs) over a checkpoint file (format: model_exporter.rs:164-316)."""

    def __init__(self, path: str, ctx_len: int = 0):
        raw = np.fromfile(path, dtype=np.uint8)
        hdr = struct.unpack("<13i", raw[:52].tobytes())
        (magic, version, arch, self.dim, self.hidden, self.L, self.nh, self.nkv, self.vocab, seq, self.hd,
         shared, self.G) = hdr
        assert magic == 0x616A6331 and version == 1 and arch == 1
        self.seq_len = min(ctx_len, seq) if ctx_len else seq          # models/mod.rs:65-67
        self.shared = shared != 0
        self.ahd, self.kvd = self.nh * self.hd, self.nkv * self.hd
        off = 256

        def take_f32(n):
            nonlocal off
            a = raw[off: off + 4 * n].view("<f4").astype(f32)
            off += 4 * n
            return a

        def take_q(cnt, size):
            nonlocal off
            out = []
            for _ in range(cnt):
                q = raw[off: off + size].view(np.int8)
                off += size
                s = raw[off: off + 4 * (size // self.G)].view("<f4").astype(f32)
                off += 4 * (size // self.G)
                out.append((q, s))
            return out

        L, d = self.L, self.dim
        self.rms_att = take_f32(L * d).reshape(L, d)                  # qwen3.rs:228-232
        self.rms_ffn = take_f32(L * d).reshape(L, d)
        self.rms_final = take_f32(d)
        self.q_ln = take_f32(L * self.hd).reshape(L, self.hd)
        self.k_ln = take_f32(L * self.hd).reshape(L, self.hd)
        self.tok = take_q(1, self.vocab * d)[0]                       # qwen3.rs:235-242
        self.wq = take_q(L, d * self.ahd)
        self.wk = take_q(L, d * self.kvd)
        self.wv = take_q(L, d * self.kvd)
        self.wo = take_q(L, self.ahd * d)
        self.w1 = take_q(L, d * self.hidden)
        self.w2 = take_q(L, self.hidden * d)
        self.w3 = take_q(L, d * self.hidden)
        self.wcls = self.tok if self.shared else take_q(1, d * self.vocab)[0]
        assert off == raw.size, (off, raw.size)
        self.key = np.zeros((L, self.seq_len, self.kvd), dtype=f32)   # qwen3.rs:439-440
        self.val = np.zeros((L, self.seq_len, self.kvd), dtype=f32)
        self.x_final = None

    def forward(self, token: int, pos: int) -> np.ndarray:
        """qwen3.rs:62-79 + 131-176 + layers.rs:328-419,466-480"""
        assert 0 <= token < self.vocab and 0 <= pos < self.seq_len
        d, G, hd = self.dim, self.G, self.hd
        tq, ts = self.tok
        x = dequantize(tq[token * d:(token + 1) * d], ts[token * d // G:(token + 1) * d // G], G)
        kv_mul = self.nh // self.nkv
        scale = f32(1.0) / np.sqrt(f32(hd), dtype=f32)
        for l in range(self.L):
            xb = rmsnorm(x, self.rms_att[l])
            xq, xs = quantize(xb, G)
            q = matmul(xq, xs, *self.wq[l], d, self.ahd, G)
            self.key[l, pos] = matmul(xq, xs, *self.wk[l], d, self.kvd, G)
            self.val[l, pos] = matmul(xq, xs, *self.wv[l], d, self.kvd, G)
            cs = rope_freqs(hd, pos)
            for h in range(self.nh):
                q[h * hd:(h + 1) * hd] = rope_apply(rmsnorm(q[h * hd:(h + 1) * hd], self.q_ln[l]), cs)
            for h in range(self.nkv):
                k = self.key[l, pos, h * hd:(h + 1) * hd]
                self.key[l, pos, h * hd:(h + 1) * hd] = rope_apply(rmsnorm(k.copy(), self.k_ln[l]), cs)
            att_out = np.zeros(self.ahd, dtype=f32)
            for h in range(self.nh):
                kvh = h // kv_mul
                qh = q[h * hd:(h + 1) * hd]
                K = self.key[l, :pos + 1, kvh * hd:(kvh + 1) * hd]
                V = self.val[l, :pos + 1, kvh * hd:(kvh + 1) * hd]
                sc = (seq_sum((K * qh[None, :]).astype(f32), axis=1) * scale).astype(f32)
                a = softmax(sc)
                o = np.zeros(hd, dtype=f32)
                for t in range(pos + 1):
                    o = (o + (a[t] * V[t]).astype(f32)).astype(f32)
                att_out[h * hd:(h + 1) * hd] = o
            xq, xs = quantize(att_out, G)
            x = (x + matmul(xq, xs, *self.wo[l], self.ahd, d, G)).astype(f32)
            xb = rmsnorm(x, self.rms_ffn[l])
            xq, xs = quantize(xb, G)
            hb = matmul(xq, xs, *self.w1[l], d, self.hidden, G)
            hb2 = matmul(xq, xs, *self.w3[l], d, self.hidden, G)
            hb = swiglu(hb, hb2)
            hq, hs = quantize(hb, G)
            x = (x + matmul(hq, hs, *self.w2[l], self.hidden, d, G)).astype(f32)
        x = rmsnorm(x, self.rms_final)
        self.x_final = x
        xq, xs = quantize(x, G)
        return matmul(xq, xs, *self.wcls, d, self.vocab, G)
