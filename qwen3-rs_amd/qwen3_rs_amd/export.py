"""HF Qwen3 directory (config.json + *.safetensors [+ LoRA adapter]) -> the Q8 checkpoint the engine loads.

Host-side twin of the reference exporter's model half (qwen3-export/src): `export_model` (lib.rs:22-36) ->
`load_model_info` (config_loader.rs:49-176), `TensorReader` (tensor_reader.rs:20-150), `LoraMerger`
(lora_merger.rs:12-120), `BinaryModelExporter::export_binary_model` (model_exporter.rs:65-101,193-316) with the
Qwen3 tensor map (models/qwen3.rs:10-91).  The file it writes is byte-for-byte what those functions write for the same
inputs: same section order, same quantizer (`checkpoint.quantize_q80`, pinned by the reference's own known-answer
tests), same LoRA arithmetic (f32, rank index ascending, unfused multiply-add, then `base += scaling * delta`).
Tokenizer and chat-template files are not produced here (outside the hot path; the Rust CLI keeps them).

    python -m qwen3_rs_amd.export <model_dir> <out.bin> [--group-size 64]
"""
from __future__ import annotations

import json
import mmap
import os
import struct
import sys
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from .checkpoint import ARCH_QWEN3, ModelShape, find_optimal_group_size, header_bytes, quantize_q80

ARCH_NAME = "Qwen3ForCausalLM"                       # models/qwen3.rs:11
EMBED_TOKENS_KEY = "model.embed_tokens.weight"       # models/qwen3.rs:12
LM_HEAD_KEY = "lm_head.weight"                       # models/qwen3.rs:13
# (name pattern, layered, required)                    models/qwen3.rs:16-22
NORM_WEIGHT_LAYERS = [
    ("model.layers.{}.input_layernorm.weight", True, True),
    ("model.layers.{}.post_attention_layernorm.weight", True, True),
    ("model.norm.weight", False, True),
    ("model.layers.{}.self_attn.q_norm.weight", True, False),
    ("model.layers.{}.self_attn.k_norm.weight", True, False),
]
# component-major, layer-minor                         models/qwen3.rs:25-47
LAYER_COMPONENTS = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
                    "mlp.gate_proj", "mlp.down_proj", "mlp.up_proj"]


class ExportError(RuntimeError):
    pass


@dataclass
class LoRAConfig:                                     # config_loader.rs:41-47
    lora_alpha: float
    r: int
    target_modules: List[str]
    base_model_name_or_path: Optional[str] = None


@dataclass
class ModelInfo:                                      # config_loader.rs:19-38
    dim: int
    hidden_dim: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    vocab_size: int
    max_seq_len: int
    head_dim: int
    norm_eps: float
    bos_token_id: int
    eos_token_id: int
    lora: Optional[LoRAConfig] = None


def load_model_info(model_dir: str) -> ModelInfo:
    """config_loader.rs:49-176: config.json is required; adapter_config.json next to it marks a LoRA model."""
    cfg_path = os.path.join(model_dir, "config.json")
    adapter_path = os.path.join(model_dir, "adapter_config.json")
    has_cfg, has_adapter = os.path.exists(cfg_path), os.path.exists(adapter_path)
    if not has_cfg and has_adapter:
        raise ExportError(f"Only LoRA config is found in {model_dir}. Make sure to have base model files in the same directory")
    if not has_cfg:
        raise ExportError(f"No valid configuration files found in {model_dir}")
    try:
        with open(cfg_path, "r", encoding="utf-8") as f:
            hf = json.load(f)
        req = {k: hf[k] for k in ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                                  "num_key_value_heads", "vocab_size", "max_position_embeddings", "rms_norm_eps")}
    except (ValueError, KeyError) as err:
        raise ExportError(f"Failed to parse config.json: {err}")
    archs = hf.get("architectures")
    if not archs:
        raise ExportError("Cannot determine architecture")
    if len(archs) != 1:
        raise ExportError(f"Multiple architectures are not supported: {archs}")
    if archs[0] != ARCH_NAME:
        raise ExportError(f"Unsupported architecture: {archs[0]}")
    head_dim = hf.get("head_dim") or req["hidden_size"] // req["num_attention_heads"]
    info = ModelInfo(req["hidden_size"], req["intermediate_size"], req["num_hidden_layers"], req["num_attention_heads"],
                     req["num_key_value_heads"], req["vocab_size"], req["max_position_embeddings"], head_dim,
                     float(req["rms_norm_eps"]), hf.get("bos_token_id") or 0, hf.get("eos_token_id") or 0)
    if has_adapter:
        try:
            with open(adapter_path, "r", encoding="utf-8") as f:
                a = json.load(f)
            info.lora = LoRAConfig(float(a["lora_alpha"]), int(a["r"]), list(a["target_modules"]), a.get("base_model_name_or_path"))
        except (ValueError, KeyError) as err:
            raise ExportError(f"Failed to parse adapter_config.json: {err}")
    return info


class TensorReader:
    """tensor_reader.rs:20-150: every *.safetensors file of the directory, first file holding the name wins;
    F32 and BF16 only, converted to f32 (BF16: bits << 16)."""

    def __init__(self, model_dir: str):
        files = sorted(os.path.join(model_dir, f) for f in os.listdir(model_dir) if f.endswith(".safetensors"))
        if not files:
            raise ExportError(f"No SafeTensors files found in {model_dir}")
        self._maps: List[Tuple[mmap.mmap, Dict[str, dict], int]] = []
        self._files = []
        for path in files:
            f = open(path, "rb")
            self._files.append(f)
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
            (hlen,) = struct.unpack("<Q", mm[:8])
            header = json.loads(mm[8:8 + hlen].decode("utf-8"))
            header.pop("__metadata__", None)
            self._maps.append((mm, header, 8 + hlen))

    def close(self):
        for mm, _, _ in self._maps:
            mm.close()
        for f in self._files:
            f.close()
        self._maps, self._files = [], []

    def load_tensor(self, name: str) -> Optional[np.ndarray]:
        for mm, header, base in self._maps:
            ent = header.get(name)
            if ent is None:
                continue
            b0, b1 = ent["data_offsets"]
            n = int(np.prod(ent["shape"], dtype=np.int64)) if ent["shape"] else 1
            dt, nbytes = ent["dtype"], b1 - b0
            if dt == "F32":
                if nbytes != 4 * n:
                    raise ExportError(f"F32 tensor {name} size mismatch. Expected {4 * n} bytes, got {nbytes}")
                return np.frombuffer(mm, dtype="<f4", count=n, offset=base + b0).astype(np.float32)   # a copy: no view outlives the map
            if dt == "BF16":
                if nbytes != 2 * n:
                    raise ExportError(f"BF16 tensor {name} size mismatch. Expected {2 * n} bytes, got {nbytes}")
                bits = np.frombuffer(mm, dtype="<u2", count=n, offset=base + b0).astype(np.uint32) << 16   # tensor_reader.rs:124-133
                return bits.view(np.float32)
            raise ExportError(f"Unsupported tensor dtype {dt} for {name}")
        return None


def merge_lora(base: np.ndarray, lora_a: np.ndarray, lora_b: np.ndarray, alpha: float, rank: int) -> np.ndarray:
    """lora_merger.rs:62-120: W += (alpha / r) * (B @ A), in f32 with the rank index ascending and every multiply
    and add rounded separately (rustc does not fuse), exactly like the reference's per-element loop."""
    scaling = np.float32(np.float32(alpha) / np.float32(rank))
    if not np.isfinite(scaling):
        raise ExportError(f"Invalid scaling factor: {scaling} (must be finite). Alpha: {alpha}, Rank: {rank}")
    if base.size == 0 or lora_a.size == 0 or lora_b.size == 0:
        raise ExportError(f"Empty tensors not allowed: base={base.size}, A={lora_a.size}, B={lora_b.size}")
    if lora_a.size % rank:
        raise ExportError(f"LoRA A tensor size ({lora_a.size}) is not divisible by rank ({rank})")
    if lora_b.size % rank:
        raise ExportError(f"LoRA B tensor size ({lora_b.size}) is not divisible by rank ({rank})")
    n_in, n_out = lora_a.size // rank, lora_b.size // rank
    if n_in * n_out != base.size:
        raise ExportError(f"Dimension mismatch: base tensor size ({base.size}) doesn't match calculated dimensions "
                          f"({n_out}×{n_in} = {n_in * n_out})")
    a = lora_a.reshape(rank, n_in).astype(np.float32)
    b = lora_b.reshape(n_out, rank).astype(np.float32)
    delta = np.zeros((n_out, n_in), dtype=np.float32)
    for r in range(rank):
        delta = delta + (b[:, r:r + 1] * a[r:r + 1, :]).astype(np.float32)       # delta_val += b_val * a_val
    out = (base.reshape(n_out, n_in).astype(np.float32) + (scaling * delta).astype(np.float32)).astype(np.float32)
    if not np.all(np.isfinite(out)):
        raise ExportError("Non-finite value detected in result")
    return out.reshape(-1)


def detect_shared_classifier(reader: TensorReader) -> bool:
    """models/qwen3.rs:59-76"""
    lm, emb = reader.load_tensor(LM_HEAD_KEY), reader.load_tensor(EMBED_TOKENS_KEY)
    if lm is not None and emb is not None:
        return lm.size == emb.size and bool(np.all(np.abs(lm - emb) < np.float32(1e-6)))
    if lm is None and emb is not None:
        return True
    return False


def export_model(model_dir: str, output_path: str, group_size: int = 64, log=None) -> ModelShape:
    """lib.rs:22-36 + model_exporter.rs:65-101: writes `output_path`, returns the header that was written."""
    log = log or (lambda *_: None)
    info = load_model_info(model_dir)
    reader = TensorReader(model_dir)
    try:
        g = find_optimal_group_size(info.dim, group_size)                          # model_exporter.rs:39-44
        shared = detect_shared_classifier(reader)
        shape = ModelShape(info.dim, info.hidden_dim, info.n_layers, info.n_heads, info.n_kv_heads, info.vocab_size,
                           info.max_seq_len, info.head_dim, shared, g, ARCH_QWEN3)
        with open(output_path, "wb") as out:
            out.write(header_bytes(shape))                                         # :164-191
            for pattern, layered, required in NORM_WEIGHT_LAYERS:                  # :193-228
                names = [pattern.format(l) for l in range(info.n_layers)] if layered else [pattern]
                for name in names:
                    t = reader.load_tensor(name)
                    if t is None:
                        if required:
                            raise ExportError(f"Missing weight for tensor_name: '{name}'")
                        t = np.ones(info.head_dim, dtype=np.float32)
                    out.write(t.astype("<f4").tobytes())
            tensors: List[Tuple[str, Optional[str], Optional[int]]] = [(EMBED_TOKENS_KEY, None, None)]   # :230-316
            for comp in LAYER_COMPONENTS:
                for l in range(info.n_layers):
                    tensors.append((f"model.layers.{l}.{comp}.weight", comp, l))
            if not shared:
                tensors.append((LM_HEAD_KEY, None, None))
            worst = 0.0
            for name, comp, layer in tensors:
                w = reader.load_tensor(name)
                if w is None:
                    raise ExportError(f"Missing weight tensor: {name}")
                if info.lora is not None and comp is not None:
                    la = reader.load_tensor(f"base_model.model.model.layers.{layer}.{comp}.lora_A.weight")
                    lb = reader.load_tensor(f"base_model.model.model.layers.{layer}.{comp}.lora_B.weight")
                    if la is not None and lb is not None:
                        w = merge_lora(w, la, lb, info.lora.lora_alpha, info.lora.r)
                if w.size == 0:
                    continue
                q, s, err = quantize_q80(w, g)
                worst = max(worst, err)
                out.write(q.tobytes())
                out.write(s.astype("<f4").tobytes())
                log(f"quantized {name} ({w.size} weights, max error {err:.6g})")
        log(f"Quantized {len(tensors)} weight tensors to Q8_0 with max error: {worst:.8f}")
        return shape
    finally:
        reader.close()


def main(argv=None) -> int:
    import argparse
    ap = argparse.ArgumentParser(description="HF Qwen3 directory -> Q8 checkpoint (model half of `qwen3 export`)")
    ap.add_argument("model_dir")
    ap.add_argument("output")
    ap.add_argument("--group-size", type=int, default=64)
    a = ap.parse_args(argv)
    try:
        shape = export_model(a.model_dir, a.output, a.group_size, log=lambda m: print(m, file=sys.stderr))
    except ExportError as err:
        print(f"error: {err}", file=sys.stderr)
        return 1
    print(f"wrote {a.output}: {shape}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
