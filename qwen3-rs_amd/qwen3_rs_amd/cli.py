"""`qwen3` command line on the MI355X engine: the flags of qwen3-cli/src/main.rs:18-93.

    python -m qwen3_rs_amd.cli export <MODEL_PATH> <OUTPUT_PATH> [--group-size 64]
    python -m qwen3_rs_amd.cli inference <checkpoint> [-t 1.0] [-p 0.9] [-s SEED] [-c CTX] [-m generate|chat]
                                         [-i INPUT] [-y SYSTEM] [-r 0|1]

`inference` follows generation.rs: `generate` echoes the prompt and decodes from its last token over a zero KV prefix
(:9-48); `chat` renders the template, forwards every prompt token (one rng coin each) and decodes until BOS/EOS
(:50-151).  Forward, sampling (temperature / top-p / xorshift64*) and the prompt loop all run on the device
(q3_prefill, q3_forward_sample); the host only tokenizes, prints and checks for the termination tokens.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

from . import export as export_mod
from .engine import TransformerBuilder
from .tokenizer import Tokenizer, export_templates, export_tokenizer


def _emit(tok: Tokenizer, token: int):
    sys.stdout.buffer.write(tok.decode_bytes(token))
    sys.stdout.flush()


def _out(raw: bytes):
    """(everything on stdout goes through the byte layer: decoded tokens are raw bytes, and mixing them with the text layer's
    own buffer reorders the output when stdout is a pipe)"""
    sys.stdout.buffer.write(raw)
    sys.stdout.flush()


def run_generate(t, tok: Tokenizer, prompt: str) -> int:
    """generation.rs:9-48"""
    prompt_tokens = tok.encode(prompt or "")
    if not prompt_tokens:
        raise SystemExit("Please provide a prompt")
    seq_len = t.get_config().seq_len
    for p in prompt_tokens[:-1][:seq_len]:                 # echoed, never forwarded (zero KV prefix)
        _emit(tok, p)
    token, pos, n_gen, t0 = prompt_tokens[-1], len(prompt_tokens) - 1, 0, None
    while pos < seq_len:
        if t0 is None:
            t0 = time.perf_counter()
        nxt = t.forward_argmax(token, pos)                 # Sampler::sample on the device when temperature > 0
        n_gen += 1
        if nxt in (tok.bos_token_id, tok.eos_token_id):
            break
        _emit(tok, token)
        token, pos = nxt, pos + 1
    _report(n_gen, t0)
    print()
    return 0


def _report(n_gen: int, t0):
    if t0 is not None and n_gen:
        dt = time.perf_counter() - t0
        print(f"\n[{n_gen / dt:.2f} tk/s, {n_gen} tokens in {dt:.2f}s]", file=sys.stderr)


def _prefill(t, ids, pos) -> int:
    """32 positions per weight pass when the checkpoint's shape allows it, else the sequential device loop; same result."""
    from .engine import Q3Error
    if os.environ.get("Q3_CLI_BATCHED_PREFILL", "1") != "0":
        try:
            return t.prefill(ids, pos, batched=True)
        except Q3Error as err:
            if err.code != -5:
                raise
    return t.prefill(ids, pos)


def run_chat(t, tok: Tokenizer, cli_prompt, system_prompt) -> int:
    """generation.rs:50-151, loop for loop: when the window is exhausted the position goes back to 0 and a user turn begins
    (generation.rs:65-69) -- the KV cache is NOT cleared, rows are simply rewritten from the front, and with a `-i` prompt
    the prompt is fed again exactly as the reference does (get_user_input, generation.rs:174-188)."""
    seq_len = t.get_config().seq_len
    pos, user_turn, nxt = 0, True, 0
    n_gen, t0 = 0, None
    while True:
        if pos >= seq_len:                                 # "Reset context if window exceeded"
            pos, user_turn = 0, True
            _out(b"\n")
        if user_turn:
            _report(n_gen, t0)
            n_gen, t0 = 0, None
            if pos == 0 and cli_prompt is not None:
                user = cli_prompt
            elif cli_prompt is not None:
                user = ""
            else:
                _out(b"> ")
                user = sys.stdin.readline().strip()
            if not user and not (pos == 0 and cli_prompt is not None):
                break
            ids = tok.encode(tok.render_prompt(pos, system_prompt, user))[: max(seq_len - pos, 0)]
            if ids:
                nxt = _prefill(t, ids, pos)
                pos += len(ids)
            user_turn = False
        else:
            if nxt in (tok.bos_token_id, tok.eos_token_id):
                _report(n_gen, t0)
                n_gen, t0 = 0, None
                _out(b"\n")
                user_turn = True
                continue
            if t0 is None:
                t0 = time.perf_counter()
            _emit(tok, nxt)
            nxt = t.forward_argmax(nxt, pos)
            n_gen += 1
            pos += 1
    return 0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="qwen3", description="Qwen3 CLI on the MI355X engine: export and inference")
    sub = ap.add_subparsers(dest="cmd")
    ex = sub.add_parser("export", help="Export a HuggingFace Qwen3 directory to the Q8 checkpoint (+ .tokenizer)")
    ex.add_argument("MODEL_PATH")
    ex.add_argument("OUTPUT_PATH")
    ex.add_argument("--group-size", "-g", type=int, default=64)
    inf = sub.add_parser("inference", help="Qwen3 inference")
    inf.add_argument("checkpoint")
    inf.add_argument("-t", "--temperature", type=float, default=1.0)
    inf.add_argument("-p", "--topp", type=float, default=0.9)
    inf.add_argument("-s", "--seed", type=int, default=None)
    inf.add_argument("-c", "--context", type=int, default=None)
    inf.add_argument("-m", "--mode", default="chat")
    inf.add_argument("-i", "--input", default=None)
    inf.add_argument("-y", "--system", default=None)
    inf.add_argument("-r", "--reasoning", type=int, default=0)
    a = ap.parse_args(argv)
    if a.cmd == "export":
        if not os.path.isdir(a.MODEL_PATH):
            print(f"Error: Model directory does not exist: {a.MODEL_PATH}", file=sys.stderr)
            return 1
        try:
            info = export_mod.load_model_info(a.MODEL_PATH)
            shape = export_mod.export_model(a.MODEL_PATH, a.OUTPUT_PATH, a.group_size, log=lambda m: print(m, file=sys.stderr))
            if os.path.exists(os.path.join(a.MODEL_PATH, "tokenizer.json")):
                print("wrote", export_tokenizer(a.MODEL_PATH, a.OUTPUT_PATH, info.bos_token_id, info.eos_token_id), file=sys.stderr)
            else:
                print("tokenizer.json not found: no .tokenizer written", file=sys.stderr)
            try:
                for path in export_templates(a.MODEL_PATH, a.OUTPUT_PATH):
                    print("wrote", path, file=sys.stderr)
            except ValueError as err:
                print(f"no prompt templates written: {err}", file=sys.stderr)
        except (export_mod.ExportError, ValueError, OSError) as err:
            print(f"Error: {err}", file=sys.stderr)
            return 1
        print(f"wrote {a.OUTPUT_PATH}: {shape}")
        return 0
    if a.cmd == "inference":
        if a.mode not in ("generate", "chat"):
            print(f"Error: Unknown mode: {a.mode}", file=sys.stderr)
            return 1
        b = TransformerBuilder(a.checkpoint)
        if a.context:
            b = b.with_ctx_length(a.context)
        with b.build() as t:
            tok = Tokenizer(a.checkpoint, t.get_config().vocab_size, a.reasoning != 0)
            seed = a.seed if a.seed is not None else int(time.time())      # lib.rs: SystemTime seconds when no seed is given
            t.set_sampler(max(a.temperature, 0.0), min(max(a.topp, 0.0), 1.0), seed)
            if a.mode == "generate":
                return run_generate(t, tok, a.input)
            return run_chat(t, tok, a.input, a.system)
    ap.print_help()
    return 1


if __name__ == "__main__":
    sys.exit(main())
