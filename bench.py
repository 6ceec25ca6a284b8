#!/usr/bin/env python3
"""bench.py -- decode tokens/s of the MI355X Qwen3 Q8 engine on BASELINE.json's headline config.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   -> ONE JSON line on stdout (rank 0).
  step      = one decoded token: forward(token,pos) + greedy argmax, all on the device (weights, KV cache and
              the token feedback loop are resident in HBM when the timed region starts).
  workload  = BASELINE.json configs[1]: Qwen3-0.6B Q8 group=64 single-stream greedy decode on 1xMI355X --
              synthetic checkpoint (seed 1234) in the reference's format, 8-token prompt, the reference's
              `generate` call pattern (first forward at pos 7 over a zero KV prefix, generation.rs:26-29), ctx 1024.
  N > 1     = N independent replicas, one process + one engine per GPU, no collective on the data path
              (the path does not shard: every listed model fits one GPU).  value = sum of tokens / max time.
  roofline  = the weight-streaming GEMV kernel (all instantiations of k_gemv): algorithmic weight bytes per
              launch / average launch period measured with HIP events on the engine's stream.
  cpu_baseline = the CPU oracle (C restatement of the Rust path; no rustc in this image) timed on the host
              cores for a bounded sample of the same run; also used to check the GPU tokens.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_BYTES_PER_S = 8.0e12   # MI355X spec (MI355X_MICROARCH.md: 8 TB/s; ~6.29 TB/s measured copy ceiling)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def dist_setup(n_gpus):
    """Replicas only: gloo carries the barrier and the max/sum of the timing -- never the data path."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    import torch.distributed as dist
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    return rank, world, local_rank


def barrier():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
    except ImportError:
        pass


def aggregate_over_ranks(local_tokens, local_seconds):
    """(sum of tokens over ranks, max of seconds over ranks).  Single process: identity."""
    try:
        import torch
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            t = torch.tensor([float(local_tokens)], dtype=torch.float64)
            s = torch.tensor([float(local_seconds)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            dist.all_reduce(s, op=dist.ReduceOp.MAX)
            return int(round(t.item())), float(s.item())
    except ImportError:
        pass
    return int(local_tokens), float(local_seconds)


def gemv_bytes_per_launch(shape):
    """Algorithmic bytes one launch of each GEMV family streams: int8 weights + f32 group scales, read once
    (SURVEY.md section 8d)."""
    g = shape.group_size
    f = 1.0 + 4.0 / g
    d, h, ahd, kvd, v = shape.dim, shape.hidden_dim, shape.all_heads_dim, shape.kv_dim, shape.vocab_size
    return {"qkv": (ahd + 2 * kvd) * d * f, "wo": d * ahd * f, "w13": 2 * h * d * f, "w2": d * h * f, "lm_head": v * d * f}


def cpu_baseline(path, ctx, first_tok, first_pos, gpu_tokens, budget_s=12.0, max_tokens=32):
    """Oracle leg (test infrastructure used as the reported CPU baseline, kind "port").  The thread count is
    chosen by a short sweep (a token's ~200 fork-joins make "all cores" slower than fewer threads on big hosts);
    the count actually used for the timed sample is what `cores` reports."""
    from oracle import q3_oracle as co
    co.build()
    m = co.OracleModel(path, ctx)
    m.forward(first_tok, first_pos)           # untimed: page-in of the mmap'd checkpoint
    ncpu = os.cpu_count() or 1
    cands = sorted({c for c in (8, 16, 32, 64, ncpu) if c <= ncpu})
    best_c, best_t = cands[-1], None
    for c in cands:                           # 2 tokens per candidate (~1 s total on a 128-core host)
        co.set_num_threads(c)
        m.reset()
        t0 = time.perf_counter()
        tok = first_tok
        for k in range(2):
            tok = co.sample_argmax(m.forward(tok, first_pos + k))
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_c, best_t = c, dt
    co.set_num_threads(best_c)
    m.reset()
    tok, pos, toks = first_tok, first_pos, []
    t0 = time.perf_counter()
    while len(toks) < max_tokens and (time.perf_counter() - t0 < budget_s or len(toks) < 2):
        tok = co.sample_argmax(m.forward(tok, pos))
        toks.append(tok)
        pos += 1
    dt = time.perf_counter() - t0
    match = toks == list(gpu_tokens[: len(toks)])
    m.close()
    return {"value": len(toks) / dt, "unit": "tokens/s", "cores": best_c, "kind": "port",
            "sample": f"first {len(toks)} generated tokens of the same run ({dt:.1f} s) on {best_c} of {ncpu} host threads "
                      f"(best of a {cands} sweep); C restatement of the Rust CPU path (no rustc in the image), OpenMP over "
                      f"rows/heads like rayon",
            "tokens_match_gpu": bool(match)}, match


def pmc_traffic(shape_name):
    """HBM read bytes per launch of the dominant streaming kernel from the committed PMC pass (rocprofv3 --pmc
    FETCH_SIZE, corrected x2 for gfx950 as MI355X_MICROARCH.md prescribes).  Counters cannot be read from inside
    this process, so the value comes from profiles/ (null when no profile of this shape is committed)."""
    f = os.path.join(ROOT, "profiles", "r01_pmc_fetch_size.json")
    if shape_name != "qwen3-0.6b" or not os.path.exists(f):
        return None
    try:
        k = json.load(open(f))["kernels"]
        tot = sum(v["avg_hbm_read_bytes_corrected"] * v["dispatches"] for n, v in k.items() if "k_gemv" in n)
        cnt = sum(v["dispatches"] for n, v in k.items() if "k_gemv" in n)
        return int(tot / cnt) if cnt else None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--shape", default="qwen3-0.6b")
    ap.add_argument("--ctx", type=int, default=1024)
    ap.add_argument("--fast", action="store_true", help="opt-in tree-reduction mode (not bit-exact)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ckpt-dir", default=os.environ.get("Q3_CKPT_DIR", "/tmp"))
    args = ap.parse_args()

    rank, world, local_rank = dist_setup(args.gpus)
    import qwen3_rs_amd as q3
    from qwen3_rs_amd import checkpoint as ck

    shape = ck.SHAPES[args.shape]
    seed = 1234
    path = os.path.join(args.ckpt_dir, f"q3_{args.shape}.bin")
    if rank == 0:
        t0 = time.time()
        ck.ensure_synthetic_checkpoint(path, shape, seed=seed)
        log(f"[bench] checkpoint {path} ready ({shape.file_size() / 1e6:.0f} MB, {time.time() - t0:.1f} s)")
    barrier()

    prompt = ck.iter_prompt_tokens(shape, seed, 8)
    first_tok, first_pos = prompt[-1], len(prompt) - 1
    K, W = args.steps, args.warmup
    if first_pos + max(K, W) > args.ctx:
        raise SystemExit("steps exceed ctx")

    eng = q3.TransformerBuilder(path).with_ctx_length(args.ctx).with_device(local_rank).with_strict(not args.fast).build()
    if W > 0:
        eng.generate_greedy(first_tok, first_pos, W)      # untimed warmup steps
    eng.reset_kv()

    def sync():
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except ImportError:
            pass

    barrier(); sync()
    t0 = time.perf_counter()
    tokens = eng.generate_greedy(first_tok, first_pos, K)   # exactly K steps; returns after the stream drained
    sync()
    dt = time.perf_counter() - t0
    barrier()
    total_tokens, max_dt = aggregate_over_ranks(K, dt)

    if rank != 0:
        eng.close()
        return

    value = total_tokens / max_dt
    wq, ws = shape.weight_bytes_per_token()
    bytes_per_token = wq + ws

    # ---- roofline of the dominant kernel: HIP events on the engine stream, one forward per rep (eager launches)
    reps = 20
    prof = eng.profile(first_tok, first_pos + K // 2, reps)
    bpl = gemv_bytes_per_launch(shape)
    per_kernel, gemv_ms, gemv_launches, gemv_bytes = [], 0.0, 0, 0.0
    for name, ms, n in prof:
        if n == 0:
            continue
        avg_us = ms / n * 1e3
        row = {"kernel": name, "launches_per_token": n // reps, "avg_us": round(avg_us, 3)}
        if name in bpl:
            row["bytes_per_launch"] = int(bpl[name])
            row["achieved_GBps"] = round(bpl[name] / (avg_us * 1e-6) / 1e9, 1)
            gemv_ms += ms
            gemv_launches += n
            gemv_bytes += bpl[name] * n
        per_kernel.append(row)
    achieved = gemv_bytes / (gemv_ms * 1e-3)                 # B/s over all GEMV launches
    roofline = {"bound": "hbm", "kernel": "k_gemv (W8A8 group-quant GEMV, all instantiations)",
                "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_BYTES_PER_S, 4), "traffic": pmc_traffic(args.shape),
                "bytes_per_launch": int(gemv_bytes / gemv_launches),
                "avg_launch_us": round(gemv_ms / gemv_launches * 1e3, 3),
                "launches_per_token": gemv_launches // reps, "per_kernel": per_kernel,
                "note": "avg launch period = HIP events bracketing each kernel family's launches of one forward on the engine "
                        "stream (kernel + ~1.6 us boundary, the quantity rocprofv3 per-dispatch durations sum to); traffic = avg HBM "
                        "read bytes per k_gemv launch from the committed PMC pass (profiles/r01_pmc_fetch_size.json)"}

    out = {"metric": "decode tokens/sec Qwen3-0.6B Q8 g=64 @1 GPU; % of int8 HBM roofline" if args.shape == "qwen3-0.6b"
           else f"decode tokens/sec {args.shape} Q8 g=64",
           "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": K, "warmup": W,
           "ms_per_step": round(max_dt / K * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "i8 (int8 x int8 -> i32 group dots, f32 scales/accumulate)", "data": "synthetic",
           "config": {"workload": f"{args.shape} Q8 group={shape.group_size} single-stream greedy decode, 8-token prompt, "
                                  f"reference generate-mode call pattern (first forward at pos 7, zero KV prefix), "
                                  f"ctx {args.ctx}, {K} tokens",
                      "mode": "tree-reduction (opt-in, not bit-exact)" if args.fast else "reference summation order (bit-identical logits)",
                      "replicas": world, "checkpoint_seed": seed,
                      "algorithmic_bytes_per_token": bytes_per_token},
           "pct_of_hbm_roofline_end_to_end": round(100.0 * (value / world) * bytes_per_token / HBM_PEAK_BYTES_PER_S, 2),
           "roofline": roofline}

    if world == 1 and not args.no_cpu_baseline:
        try:
            cb, match = cpu_baseline(path, args.ctx, first_tok, first_pos, tokens)
            out["cpu_baseline"] = cb
            if not match:
                log("[bench] WARNING: GPU tokens differ from the CPU oracle on the sampled prefix")
        except Exception as e:  # the baseline leg must never take the bench line down
            log(f"[bench] cpu_baseline failed: {e!r}")
            out["cpu_baseline"] = None
    eng.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
