"""qwen3_rs_amd -- MI355X (gfx950) Qwen3 Q8 decode engine behind qwen3-rs's `Transformer` surface.

Host-side mirror (ctypes over the C ABI in include/qwen3_hip.h) of the reference interface for the hot
path only:  TransformerBuilder / Transformer.forward / get_config  (qwen3-inference/src/models/mod.rs),
the free functions of tensor.rs / layers.rs, the `generate` / `chat` call patterns (generation.rs) and
the checkpoint format (qwen3-export/src/model_exporter.rs).  All compute runs in libqwen3_hip.so
(hand-written HIP kernels); there is no CPU fallback -- loading fails loudly if the library is missing.
"""
from .engine import (Q3Error, ModelConfig, Transformer, TransformerBuilder, lib_path, dev_lib_path, load_library, use_library, ops,
                     FLAG_FAST, FLAG_NO_GRAPH, FLAG_NO_VALUE_T, EXPORTED_SYMBOLS, source_build_id)
from .generation import generate, chat_turn, TokenMetrics, sample_argmax
from . import checkpoint

__all__ = ["Q3Error", "ModelConfig", "Transformer", "TransformerBuilder", "lib_path", "dev_lib_path", "load_library", "use_library", "ops",
           "FLAG_FAST", "FLAG_NO_GRAPH", "FLAG_NO_VALUE_T", "EXPORTED_SYMBOLS", "source_build_id", "generate", "chat_turn", "TokenMetrics",
           "sample_argmax", "checkpoint"]
