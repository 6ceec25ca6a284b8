"""Tokenizer file writer + runtime (tokenizer_exporter.rs / tokenizer.rs) and the `qwen3` command line."""
import json
import math
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_tokenizer_json(d, n_vocab=None):
    """bytes 0..255 as single-character tokens (GPT-2 mapping), some merges, two special tokens"""
    from qwen3_rs_amd import tokenizer as tk
    b2u = {b: ch for ch, b in tk.unicode_to_byte_map().items()}
    vocab = {b2u[b]: b for b in range(256)}
    multi = ["he", "ll", "hell", "hello", b2u[32] + "w", "or", b2u[32] + "wor", "ld", b2u[32] + "world", "lo", "é".encode().decode("latin-1").translate({})]
    multi = multi[:-1] + ["".join(b2u[x] for x in "é".encode())]           # a 2-byte UTF-8 character as one token
    for t in multi:
        vocab[t] = len(vocab)
    merges = ["h e", "l l", "he ll", "hell o", b2u[32] + " w", "o r"]
    added = [{"id": len(vocab), "content": "<|im_start|>"}, {"id": len(vocab) + 1, "content": "<|im_end|>"}]
    json.dump({"model": {"vocab": vocab, "merges": merges}, "added_tokens": added}, open(os.path.join(d, "tokenizer.json"), "w"))
    return len(vocab) + 2


def test_unicode_to_byte_map_known_answers(q3):
    """tokenizer_exporter_test.rs:10-60"""
    from qwen3_rs_amd import tokenizer as tk
    assert tk.token_to_bytes("A") == bytes([65]) and tk.token_to_bytes("z") == bytes([122])
    assert tk.token_to_bytes("!") == bytes([33]) and tk.token_to_bytes("~") == bytes([126])
    assert tk.token_to_bytes("¡") == bytes([161]) and tk.token_to_bytes("¬") == bytes([172])
    assert tk.token_to_bytes("®") == bytes([174]) and tk.token_to_bytes("ÿ") == bytes([255])
    assert tk.token_to_bytes("hello") == bytes([104, 101, 108, 108, 111]) and tk.token_to_bytes("") == b""
    assert len(tk.token_to_bytes("\0")) == 1 and tk.token_to_bytes("Ā") == bytes([0])      # first unprintable byte
    assert sorted(tk.unicode_to_byte_map().values()) == list(range(256))
    # tokenizer_exporter_test.rs:199-207
    assert tk.merge_rank_score(0) == 0.0
    assert abs(tk.merge_rank_score(1) + math.log(2)) < 1e-3 and abs(tk.merge_rank_score(10) + 2.397895) < 1e-3
    assert tk.DEFAULT_SCORE == -1e6


def brute_force_encode(vocab, scores, max_len, text):
    """tokenizer.rs:165-237 with the reference's linear scans"""
    def lookup(b):
        for i, t in enumerate(vocab):
            if t == b:
                return i
        return None
    toks, chars, i = [], list(text), 0
    while i < len(chars):
        found = False
        if chars[i] == "<":
            end = None
            for j in range(i + 1, min(len(chars), i + max_len)):
                if chars[j] == ">":
                    end = j
                    break
            if end is not None:
                t = lookup("".join(chars[i:end + 1]).encode())
                if t is not None:
                    toks.append(t); i = end + 1; found = True
        if not found:
            t = lookup(chars[i].encode())
            if t is not None:
                toks.append(t)
            i += 1
    while True:
        best, bid, bix = -1e10, None, None
        for k in range(len(toks) - 1):
            m = lookup(vocab[toks[k]] + vocab[toks[k + 1]])
            if m is not None and scores[m] > best:
                best, bid, bix = scores[m], m, k
        if bid is None:
            break
        toks[bix] = bid
        del toks[bix + 1]
    return toks


def test_tokenizer_file_roundtrip_and_encode(q3, tmp_path):
    from qwen3_rs_amd import tokenizer as tk
    d = str(tmp_path)
    n = make_tokenizer_json(d)
    out = tk.export_tokenizer(d, os.path.join(d, "model.bin"), 7, 9)
    raw = open(out, "rb").read()
    max_len, bos, eos = struct.unpack_from("<III", raw, 0)
    assert (bos, eos) == (7, 9) and max_len == len("<|im_start|>")
    for name, content in ((".template", "<|im_start|>user\n%s<|im_end|>\n<|im_start|>assistant\n"),
                          (".template.with-system", "<|im_start|>system\n%s<|im_end|>\n<|im_start|>assistant\n")):
        open(os.path.join(d, "model.bin" + name), "w").write(content)
    t = tk.Tokenizer(os.path.join(d, "model.bin"), n)
    assert len(t.vocab) == n and t.vocab[104] == b"h" and t.vocab[n - 2] == b"<|im_start|>"
    # scores: only tokens whose STRING equals a merge string get -ln(rank+1); here none does -> all default (reference quirk)
    assert all(s == np.float32(-1e6) for s in t.merge_scores)
    for text in ["hello world", "hello<|im_start|>hello<|im_end|>", "<notatoken> é!", "a<b", "", "héllo wörld <|im_end|"]:
        got = t.encode(text)
        assert got == brute_force_encode(t.vocab, t.merge_scores, t.max_token_length, text), text
        if "ö" not in text:                                   # a character absent from the vocabulary is skipped (tokenizer.rs:199-202)
            assert b"".join(t.decode_bytes(i) for i in got) == text.encode()
    assert t.encode("hello")[0] == t.str_lookup("hello") or len(t.encode("hello")) >= 1
    assert t.render_prompt(0, None, "hi") == "<|im_start|>user\nhi<|im_end|>\n<|im_start|>assistant\n"
    assert t.render_prompt(0, "be brief", "hi").startswith("<|im_start|>system\nbe brief\nhi<|im_end|>")
    assert t.render_prompt(5, "be brief", "hi") == t.render_prompt(0, None, "hi")
    # a truncated file yields empty trailing tokens with score 0 (tokenizer.rs:55-80)
    open(out, "wb").write(raw[: len(raw) - 9])
    t2 = tk.Tokenizer(os.path.join(d, "model.bin"), n)
    assert t2.vocab[-1] == b"" and len(t2.vocab) == n


def test_template_export_matches_reference_patterns(q3, tmp_path):
    """chat_template_exporter.rs:64-221: template classification and the fixed %s patterns"""
    from qwen3_rs_amd import tokenizer as tk
    d = str(tmp_path)
    hf = "{%- if messages[0].role == 'system' %}<|im_start|>system...{% endif %}<|im_start|>user<|im_end|>{% if enable_thinking %}"
    json.dump({"chat_template": hf}, open(os.path.join(d, "tokenizer_config.json"), "w"))
    paths = tk.export_templates(d, os.path.join(d, "m.bin"))
    assert [os.path.basename(p) for p in paths] == ["m.bin.template", "m.bin.template.with-thinking", "m.bin.template.with-system",
                                                   "m.bin.template.with-system-and-thinking"]
    rd = lambda p: open(p, encoding="utf-8").read()
    assert rd(paths[0]) == "<|im_start|>user\n%s<|im_end|>\n<|im_start|>assistant\n<think>\n\n</think>\n\n"
    assert rd(paths[1]) == "<|im_start|>user\n%s<|im_end|>\n<|im_start|>assistant\n"
    assert rd(paths[2]) == "<|im_start|>system\n%s<|im_end|>\n<|im_start|>user\n%s<|im_end|>\n<|im_start|>assistant\n<think>\n\n</think>\n\n"
    assert rd(paths[3]).endswith("<|im_start|>assistant\n") and rd(paths[3]).count("%s") == 2
    json.dump({"chat_template": "{{ system_prompt }}<｜User｜>x<｜Assistant｜>think"}, open(os.path.join(d, "tokenizer_config.json"), "w"))
    paths = tk.export_templates(d, os.path.join(d, "ds.bin"))
    assert rd(paths[0]) == "<｜User｜>%s<｜Assistant｜><think>\n</think>" and rd(paths[-1]) == "%s<｜User｜>%s<｜Assistant｜>"
    json.dump({"chat_template": "plain"}, open(os.path.join(d, "tokenizer_config.json"), "w"))
    with pytest.raises(ValueError, match="Unknown template type"):
        tk.export_templates(d, os.path.join(d, "x.bin"))
    os.remove(os.path.join(d, "tokenizer_config.json"))
    with pytest.raises(ValueError, match="No chat template found"):
        tk.export_templates(d, os.path.join(d, "x.bin"))


def test_cli_export_writes_checkpoint_and_tokenizer(q3, tmp_path):
    from test_export import build_model_dir
    d = str(tmp_path / "hf")
    build_model_dir(d, tied=True, with_qk_norm=True, seed=2)
    make_tokenizer_json(d)
    out = str(tmp_path / "m.bin")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "qwen3-rs_amd"))
    r = subprocess.run([sys.executable, "-m", "qwen3_rs_amd.cli", "export", d, out, "--group-size", "64"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(out) > 256 and os.path.exists(out + ".tokenizer")
    r = subprocess.run([sys.executable, "-m", "qwen3_rs_amd.cli", "export", str(tmp_path / "nope"), out], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "does not exist" in r.stderr


def cpp_cli():
    """qwen3-rs_amd/q3_cli (make -C qwen3-rs_amd cli): `qwen3 inference` in C++ over the C ABI"""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "qwen3-rs_amd"), "cli"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return os.path.join(ROOT, "qwen3-rs_amd", "q3_cli")


def test_cpp_cli_tokenizer_matches_the_python_one(q3, tmp_path):
    """The C++ command line carries its own tokenizer (tokenizer.rs:28-237 again): same ids and same bytes as tokenizer.py on the
    fixture vocabulary -- merges, special tokens, a multi-byte character, an unknown character, a truncated file."""
    from qwen3_rs_amd import tokenizer as tk
    d = str(tmp_path)
    n = make_tokenizer_json(d)
    path = os.path.join(d, "model.bin")
    out = tk.export_tokenizer(d, path, 7, 9)
    open(path + ".template", "w").write("<|im_start|>user\n%s<|im_end|>\n")
    exe = cpp_cli()

    def both(text):
        t = tk.Tokenizer(path, n)
        ids = t.encode(text)
        r = subprocess.run([exe, "tokenize", path, str(n), text], capture_output=True, timeout=60)
        assert r.returncode == 0, r.stderr.decode(errors="replace")
        lines = [l for l in r.stdout.split(b"\n") if not l.startswith(b"Warning:")]
        got = [int(x) for x in lines[0].split()]
        assert got == ids, (text, got, ids)
        assert b"\n".join(lines[1:]) == b"".join(t.decode_bytes(i) for i in ids), text

    for text in ["hello world", "hello<|im_start|>hello<|im_end|>", "<notatoken> é!", "a<b", "héllo wörld <|im_end|", "x" * 40 + "<|im_end|>",
                 "<|im_start|>user\nhello world<|im_end|>\n"]:
        both(text)
    # prompt rendering (generation.rs:188-195): the system template only at position 0, every "%s" replaced
    open(path + ".template.with-system", "w").write("<|im_start|>system\n%s<|im_end|>\n<|im_start|>assistant\n%s")
    t = tk.Tokenizer(path, n)
    for pos, system, user in [(0, None, "hi"), (0, "be brief", "hi there"), (5, "be brief", "hi"), (0, "", "")]:
        r = subprocess.run([exe, "render", path, str(n), str(pos), "-" if system is None else system, user], capture_output=True, timeout=60)
        assert r.returncode == 0 and r.stdout.decode() == t.render_prompt(pos, system, user), (pos, system, user, r.stdout)
    raw = open(out, "rb").read()
    open(out, "wb").write(raw[: len(raw) - 9])            # truncated file: empty trailing tokens (tokenizer.rs:55-80)
    both("hello<|im_end|>")


@pytest.mark.gpu
@pytest.mark.parametrize("front_end", ["python", "cpp"])
def test_cli_inference_generate_and_chat_match_the_reference_loops(q3, oracle, tmp_path, front_end):
    """`qwen3 inference` on the device (prefill, forward, sampler) prints what the reference's host loops
    (generation.rs:9-151) print when driven by the CPU restatement with the same seed."""
    from qwen3_rs_amd import tokenizer as tk
    ck = q3.checkpoint
    d = str(tmp_path)
    n_vocab = make_tokenizer_json(d)
    shape = ck.ModelShape(256, 384, 2, 4, 2, n_vocab + (16 - n_vocab % 16) % 16, 96, 64, True, 64)
    path = os.path.join(d, "model.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=12)
    tk.export_tokenizer(d, path, 1, 2)
    open(path + ".template", "w").write("<|im_start|>%s<|im_end|>")
    tok = tk.Tokenizer(path, shape.vocab_size)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "qwen3-rs_amd"))

    def ref_generate(prompt, temperature, topp, seed):
        om = oracle.OracleModel(path, 40)
        smp = oracle.Sampler(shape.vocab_size, temperature, topp, seed)
        ids = tok.encode(prompt)
        out, token, pos = b"", ids[0], 0
        while pos < 40:
            if pos < len(ids) - 1:
                nxt = ids[pos + 1]
            else:
                nxt = smp.sample(om.forward(token, pos))
                if nxt in (tok.bos_token_id, tok.eos_token_id):
                    break
            out += tok.decode_bytes(token)
            token, pos = nxt, pos + 1
        return out

    def ref_chat(lines, temperature, topp, seed, ctx, cli_prompt=None, max_turns=8):
        """generation.rs:50-151 driven by the oracle: stdout as the reference prints it (interactive: "> " before every read,
        a newline when the window wraps or a turn ends), positions wrap to 0 without clearing the cache (generation.rs:65-69)."""
        om = oracle.OracleModel(path, ctx)
        smp = oracle.Sampler(shape.vocab_size, temperature, topp, seed)
        lines = list(lines)
        pos, user_turn, nxt, out, turns = 0, True, 0, b"", 0
        while True:
            if pos >= ctx:
                pos, user_turn = 0, True
                out += b"\n"
            if user_turn:
                if pos == 0 and cli_prompt is not None:
                    user = cli_prompt
                elif cli_prompt is not None:
                    user = ""
                else:
                    out += b"> "
                    user = lines.pop(0).strip() if lines else ""
                if not user and not (pos == 0 and cli_prompt is not None):
                    break
                turns += 1
                if turns > max_turns:
                    break
                for t in tok.encode(tok.render_prompt(pos, None, user)):
                    if pos >= ctx:
                        break
                    nxt = smp.sample(om.forward(t, pos)); pos += 1
                user_turn = False
            else:
                if nxt in (tok.bos_token_id, tok.eos_token_id):
                    out += b"\n"
                    user_turn = True
                    continue
                out += tok.decode_bytes(nxt)
                nxt = smp.sample(om.forward(nxt, pos)); pos += 1
        return out

    cli = [sys.executable, "-m", "qwen3_rs_amd.cli", "inference", path] if front_end == "python" else [cpp_cli(), "inference", path]
    for temperature, topp in ((0.0, 0.9), (0.9, 0.8)):
        r = subprocess.run(cli + ["-m", "generate", "-i", "hello world", "-t", str(temperature), "-p", str(topp), "-s", "77", "-c", "40"],
                           env=env, capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode(errors="replace")
        want = ref_generate("hello world", temperature, topp, 77)
        assert r.stdout.rstrip(b"\n") == want.rstrip(b"\n") or r.stdout.startswith(want), ("generate", temperature, r.stdout, want)
        # chat, interactive: one turn that runs into the end of a 40-position window, then end of input
        r = subprocess.run(cli + ["-m", "chat", "-t", str(temperature), "-p", str(topp), "-s", "77", "-c", "40"],
                           env=env, capture_output=True, timeout=600, input=b"hello world\n")
        assert r.returncode == 0, r.stderr.decode(errors="replace")
        assert r.stdout == ref_chat(["hello world"], temperature, topp, 77, 40), ("chat", temperature, r.stdout)
    # context wrap-around (generation.rs:65-69): a 24-position window, two user turns, ~40 generated tokens; the second turn
    # starts again at position 0 on top of the first turn's rows (nothing is cleared), then the input ends
    for temperature in (0.0, 0.9):
        r = subprocess.run(cli + ["-m", "chat", "-t", str(temperature), "-p", "0.8", "-s", "5", "-c", "24"],
                           env=env, capture_output=True, timeout=600, input=b"hello world\nhello again world\n")
        assert r.returncode == 0, r.stderr.decode(errors="replace")
        want = ref_chat(["hello world", "hello again world"], temperature, 0.8, 5, 24)
        assert want.count(b"> ") == 3 and len(want) > 30
        assert r.stdout == want, ("chat wrap", temperature, r.stdout, want)


@pytest.mark.gpu
def test_cli_front_ends_agree_on_system_prompt_reasoning_and_one_shot_chat(q3, tmp_path):
    """`-y` (system prompt at position 0), `-r 1` (the thinking templates), sampled generate: the C++ and the Python command line print
    the same bytes (lib.rs:109-138, generation.rs:174-195)."""
    from qwen3_rs_amd import tokenizer as tk
    ck = q3.checkpoint
    d = str(tmp_path)
    n_vocab = make_tokenizer_json(d)
    shape = ck.ModelShape(256, 384, 2, 4, 2, n_vocab + (16 - n_vocab % 16) % 16, 96, 64, True, 64)
    path = os.path.join(d, "model.bin")
    ck.write_synthetic_checkpoint(path, shape, seed=21)
    tk.export_tokenizer(d, path, 1, 2)
    for suffix, body in ((".template", "<|im_start|>%s<|im_end|>"), (".template.with-thinking", "<|im_start|>%s<|im_end|>hm"),
                         (".template.with-system", "<|im_start|>world %s<|im_end|>"), (".template.with-system-and-thinking", "<|im_start|>hello %s<|im_end|>hm")):
        open(path + suffix, "w").write(body)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "qwen3-rs_amd"))
    fronts = ([sys.executable, "-m", "qwen3_rs_amd.cli", "inference", path], [cpp_cli(), "inference", path])
    # (chat without -i: the turns come from stdin and the loop ends at end of input; a one-shot `-i` chat on a model that never emits
    # BOS / EOS feeds the prompt again at every wrap, forever -- generation.rs:65-69,174-188 -- so it is not run here)
    for extra, stdin in ((["-m", "chat", "-y", "hello", "-r", "1", "-t", "0", "-c", "48"], b"hello world\n"),
                         (["-m", "chat", "-y", "hello", "-t", "0.8", "-p", "0.7", "-s", "9", "-c", "48"], b"hello world\nhello\n"),
                         (["-m", "chat", "-r", "1", "-t", "0.8", "-p", "1.0", "-s", "3", "-c", "32"], b"hello\n"),
                         (["-m", "generate", "-i", "hello world hello", "-t", "1.2", "-p", "0.0", "-s", "11", "-c", "32"], b"")):
        outs = []
        for cli in fronts:
            r = subprocess.run(cli + extra, env=env, capture_output=True, timeout=120, input=stdin)
            assert r.returncode == 0, r.stderr.decode(errors="replace")
            outs.append(r.stdout)
        assert outs[0] == outs[1] and len(outs[0]) > 0, (extra, outs)
