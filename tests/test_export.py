"""The HF-directory -> Q8 checkpoint exporter (qwen3_rs_amd/export.py) against an independent assembly of the same
file from the C oracle's exporter-side functions (q3o_write_header / q3o_quantize_q80, themselves pinned by the
reference's known-answer tests) and scalar-loop restatements of the BF16 widening and the LoRA merge."""
import json
import os
import struct

import numpy as np
import pytest

f32 = np.float32


def write_safetensors(path, tensors):
    """tensors: name -> (dtype 'F32'|'BF16'|'F16', shape, raw little-endian bytes)"""
    header, blob, off = {}, b"", 0
    for name, (dt, shape, raw) in tensors.items():
        header[name] = {"dtype": dt, "shape": list(shape), "data_offsets": [off, off + len(raw)]}
        blob += raw
        off += len(raw)
    hj = json.dumps(header).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)) + hj + blob)


def to_bf16_bytes(x):
    """truncate f32 -> bf16 (the test only needs SOME bf16 payload); returns (bytes, the f32 values they widen to)"""
    bits = (np.asarray(x, dtype=f32).view(np.uint32) >> 16).astype(np.uint16)
    widened = (bits.astype(np.uint32) << 16).view(f32)
    return bits.astype("<u2").tobytes(), widened


CFG = dict(hidden_size=128, intermediate_size=192, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
           vocab_size=96, max_position_embeddings=64, rms_norm_eps=1e-6, head_dim=32, bos_token_id=1, eos_token_id=2,
           architectures=["Qwen3ForCausalLM"])
COMPONENT_SHAPES = {"self_attn.q_proj": (128, 128), "self_attn.k_proj": (64, 128), "self_attn.v_proj": (64, 128),
                    "self_attn.o_proj": (128, 128), "mlp.gate_proj": (192, 128), "mlp.down_proj": (128, 192),
                    "mlp.up_proj": (192, 128)}


def build_model_dir(d, *, tied, with_qk_norm, lora_rank=0, seed=0):
    """returns name -> f32 array of what the exporter should see (after dtype widening)"""
    rng = np.random.default_rng(seed)
    os.makedirs(d, exist_ok=True)
    json.dump(CFG, open(os.path.join(d, "config.json"), "w"))
    want, f32_file, bf16_file = {}, {}, {}

    def add(name, shape, bf16):
        x = rng.normal(0, 0.05, shape).astype(f32)
        if bf16:
            raw, x = to_bf16_bytes(x)
            bf16_file[name] = ("BF16", shape, raw)
        else:
            f32_file[name] = ("F32", shape, x.astype("<f4").tobytes())
        want[name] = x.reshape(-1)

    L = CFG["num_hidden_layers"]
    add("model.embed_tokens.weight", (96, 128), True)
    for l in range(L):
        add(f"model.layers.{l}.input_layernorm.weight", (128,), False)
        add(f"model.layers.{l}.post_attention_layernorm.weight", (128,), True)
        if with_qk_norm:
            add(f"model.layers.{l}.self_attn.q_norm.weight", (32,), False)
            add(f"model.layers.{l}.self_attn.k_norm.weight", (32,), False)
        for i, (comp, shape) in enumerate(COMPONENT_SHAPES.items()):
            add(f"model.layers.{l}.{comp}.weight", shape, bf16=(i + l) % 2 == 0)
    add("model.norm.weight", (128,), False)
    if not tied:
        add("lm_head.weight", (96, 128), False)
    if lora_rank:
        json.dump({"lora_alpha": 16.0, "r": lora_rank, "target_modules": ["q_proj", "v_proj"]},
                  open(os.path.join(d, "adapter_config.json"), "w"))
        for l in range(L):
            for comp in ("self_attn.q_proj", "self_attn.v_proj"):
                out_f, in_f = COMPONENT_SHAPES[comp]
                add(f"base_model.model.model.layers.{l}.{comp}.lora_A.weight", (lora_rank, in_f), False)
                add(f"base_model.model.model.layers.{l}.{comp}.lora_B.weight", (out_f, lora_rank), True)
    write_safetensors(os.path.join(d, "model-00001-of-00002.safetensors"), f32_file)
    write_safetensors(os.path.join(d, "model-00002-of-00002.safetensors"), bf16_file)
    return want


def scalar_lora_merge(base, a, b, alpha, rank, n_out, n_in):
    """lora_merger.rs:96-108 with explicit f32 scalars"""
    scaling = f32(f32(alpha) / f32(rank))
    out = base.copy()
    for o in range(n_out):
        for i in range(n_in):
            delta = f32(0.0)
            for r in range(rank):
                delta = f32(delta + f32(b[o * rank + r] * a[r * n_in + i]))
            out[o * n_in + i] = f32(out[o * n_in + i] + f32(scaling * delta))
    return out


def expected_file(oracle, want, *, tied, with_qk_norm, lora_rank, group):
    cfg = oracle.Config()
    cfg.architecture_id, cfg.dim, cfg.hidden_dim, cfg.n_layers, cfg.n_heads, cfg.n_kv_heads = 1, 128, 192, 2, 4, 2
    cfg.vocab_size, cfg.seq_len, cfg.head_dim, cfg.shared_classifier, cfg.group_size = 96, 64, 32, int(tied), group
    out = oracle.write_header(cfg, 64)
    L = 2
    for pat in ("model.layers.{}.input_layernorm.weight", "model.layers.{}.post_attention_layernorm.weight"):
        for l in range(L):
            out += want[pat.format(l)].astype("<f4").tobytes()
    out += want["model.norm.weight"].astype("<f4").tobytes()
    for pat in ("model.layers.{}.self_attn.q_norm.weight", "model.layers.{}.self_attn.k_norm.weight"):
        for l in range(L):
            t = want.get(pat.format(l))
            out += (t if t is not None else np.ones(32, dtype=f32)).astype("<f4").tobytes()
    names = ["model.embed_tokens.weight"]
    for comp in COMPONENT_SHAPES:
        names += [f"model.layers.{l}.{comp}.weight" for l in range(L)]
    if not tied:
        names.append("lm_head.weight")
    for name in names:
        w = want[name]
        parts = name.split(".")
        if lora_rank and len(parts) > 3 and ".".join(parts[3:5]) in ("self_attn.q_proj", "self_attn.v_proj"):
            comp, l = ".".join(parts[3:5]), parts[2]
            n_out, n_in = COMPONENT_SHAPES[comp]
            w = scalar_lora_merge(w, want[f"base_model.model.model.layers.{l}.{comp}.lora_A.weight"],
                                  want[f"base_model.model.model.layers.{l}.{comp}.lora_B.weight"], 16.0, lora_rank, n_out, n_in)
        q, s, _ = oracle.quantize_q80(w, group)
        out += q.tobytes() + s.astype("<f4").tobytes()
    return out


@pytest.mark.parametrize("tied,with_qk_norm,lora_rank,group", [(True, True, 0, 64), (False, False, 0, 32), (True, True, 4, 64)])
def test_export_is_byte_identical_to_independent_assembly(q3, oracle, tmp_path, tied, with_qk_norm, lora_rank, group):
    from qwen3_rs_amd import export
    d = str(tmp_path / "hf")
    want = build_model_dir(d, tied=tied, with_qk_norm=with_qk_norm, lora_rank=lora_rank, seed=7)
    out = str(tmp_path / "model.bin")
    shape = export.export_model(d, out, group)
    assert shape.shared_classifier == tied and shape.group_size == group and shape.max_seq_len == 64
    got = open(out, "rb").read()
    exp = expected_file(oracle, want, tied=tied, with_qk_norm=with_qk_norm, lora_rank=lora_rank, group=group)
    assert len(got) == len(exp)
    assert got == exp
    # and it is a loadable checkpoint: the CPU restatement of the reference runs it
    m = oracle.OracleModel(out)
    assert m.get_config().shared_classifier == int(tied)
    logits = m.forward(3, 0)
    assert logits.shape == (96,) and np.all(np.isfinite(logits))


def test_export_group_size_adjustment_and_tied_detection(q3, oracle, tmp_path):
    from qwen3_rs_amd import export
    d = str(tmp_path / "hf")
    want = build_model_dir(d, tied=False, with_qk_norm=True, seed=3)
    # an lm_head equal to the embedding is detected as shared (models/qwen3.rs:59-72) and not written
    emb = want["model.embed_tokens.weight"]
    write_safetensors(os.path.join(d, "model-00003-of-00003.safetensors"), {})
    f = os.path.join(d, "model-00001-of-00002.safetensors")
    os.remove(f)
    rest = {k: ("F32", (v.size,), v.astype("<f4").tobytes()) for k, v in want.items()}
    rest["lm_head.weight"] = ("F32", (96, 128), emb.astype("<f4").tobytes())
    os.remove(os.path.join(d, "model-00002-of-00002.safetensors"))
    write_safetensors(os.path.join(d, "model.safetensors"), rest)
    shape = export.export_model(d, str(tmp_path / "m.bin"), 1000)          # 1000 -> 128 -> largest divisor chain of dim
    assert shape.shared_classifier is True
    assert shape.group_size == q3.checkpoint.find_optimal_group_size(128, 1000) == 128
    assert os.path.getsize(str(tmp_path / "m.bin")) == shape.file_size()


def test_export_errors(q3, tmp_path):
    from qwen3_rs_amd import export
    d = str(tmp_path / "hf")
    os.makedirs(d)
    with pytest.raises(export.ExportError, match="No valid configuration files"):
        export.export_model(d, str(tmp_path / "x.bin"))
    json.dump({"lora_alpha": 1, "r": 1, "target_modules": []}, open(os.path.join(d, "adapter_config.json"), "w"))
    with pytest.raises(export.ExportError, match="Only LoRA config is found"):
        export.export_model(d, str(tmp_path / "x.bin"))
    os.remove(os.path.join(d, "adapter_config.json"))
    json.dump(dict(CFG, architectures=["A", "B"]), open(os.path.join(d, "config.json"), "w"))
    with pytest.raises(export.ExportError, match="Multiple architectures"):
        export.export_model(d, str(tmp_path / "x.bin"))
    json.dump(CFG, open(os.path.join(d, "config.json"), "w"))
    with pytest.raises(export.ExportError, match="No SafeTensors files"):
        export.export_model(d, str(tmp_path / "x.bin"))
    write_safetensors(os.path.join(d, "model.safetensors"),
                      {"model.embed_tokens.weight": ("F16", (96, 128), b"\0" * (2 * 96 * 128))})
    with pytest.raises(export.ExportError, match="Unsupported tensor dtype F16|Missing weight"):
        export.export_model(d, str(tmp_path / "x.bin"))
