#!/usr/bin/env python3
"""Per-kernel durations UNDER hipGraph REPLAY from in-kernel begin / end stamps (s_memrealtime, 100 MHz), where rocprofv3 of
this image cannot trace (it faults inside hipGraphLaunch).  Needs the developer build next to the product library:
   make -C qwen3-rs_amd dev && python3 tools/kstamps.py [n_tokens] [out.json] [shape]
Lane 0 of every wave of every launch stores its entry / exit time into its own slot; two small kernels behind the token's last launch
turn the slots into per-launch duration and gap-to-predecessor sums.  The same call first measures the device loop WITHOUT stamps on
the product library, so the result carries its own reconciliation: sum over a token of (duration + gap) against the stamped run's
wall clock, and the stamped run against the product (the difference / launches per token = the stamp overhead per launch).

bench.py imports graph_kernel_durations() for its `roofline` block (the parent process never touches the GPU: both measurements run in
child processes)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(n, shape_name, ctx, first_tok, first_pos, ckpt):
    sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
    import qwen3_rs_amd as q3
    from qwen3_rs_amd import checkpoint as ck
    ck.ensure_synthetic_checkpoint(ckpt, ck.SHAPES[shape_name], seed=1234)
    t = q3.TransformerBuilder(ckpt).with_ctx_length(ctx).build()
    t.generate_greedy(first_tok, first_pos, min(8, n))
    best = 1e9
    for _ in range(3):
        t.reset_kv()
        t0 = time.perf_counter()
        t.generate_greedy(first_tok, first_pos, n)
        best = min(best, (time.perf_counter() - t0) / n)
    print(json.dumps({"us_per_token": best * 1e6, "build_id": q3.load_library().q3_build_id().decode()}))
    t.close()


def graph_kernel_durations(n=128, shape_name="qwen3-0.6b", ctx=1024, first_tok=5, first_pos=7, ckpt=None, timeout=300):
    """Two child processes on the GPU: the product library (plain rate), then the developer library with Q3_KSTAMPS=1."""
    ckpt = ckpt or f"/tmp/q3_{shape_name}.bin"
    dev_lib = os.path.join(ROOT, "qwen3-rs_amd", "libqwen3_hip_dev.so")
    if not os.path.exists(dev_lib):
        raise RuntimeError("developer library missing (make -C qwen3-rs_amd dev)")
    cmd = [sys.executable, os.path.abspath(__file__), "--child", str(n), shape_name, str(ctx), str(first_tok), str(first_pos), ckpt]
    env = dict(os.environ)
    env.pop("Q3_KSTAMPS", None)
    env.pop("Q3_HIP_LIB", None)
    r0 = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    if r0.returncode != 0:
        raise RuntimeError("plain run failed: " + r0.stderr[-300:])
    plain = json.loads(r0.stdout.strip().splitlines()[-1])
    env.update(Q3_KSTAMPS="1", Q3_HIP_LIB=dev_lib)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError("stamped run failed: " + r.stderr[-300:])
    stamped = json.loads(r.stdout.strip().splitlines()[-1])
    lines = [ln for ln in r.stderr.splitlines() if ln.startswith("[q3 kstamps]")]
    if not lines:
        raise RuntimeError("the developer library printed no [q3 kstamps] line")
    k = json.loads(lines[-1][len("[q3 kstamps] "):])
    launches = sum(f["launches_per_token"] for f in k["families"].values())
    res = {"what": f"{shape_name} device loop, {n} tokens from position {first_pos}, hipGraph replay; per family: average launch duration "
                   "(first wave in .. last wave out) and average gap to the predecessor's end, in-kernel 100 MHz clock, developer "
                   "build with Q3_KSTAMPS=1",
           "build_id": plain["build_id"], "stamped_build_id": stamped["build_id"],
           "product_library_us_per_token": round(plain["us_per_token"], 2),
           "stamped_developer_library_us_per_token": round(stamped["us_per_token"], 2), **k}
    res["sum_over_product_ms_per_step"] = round(k["sum_duration_plus_gap_us_per_token"] / plain["us_per_token"], 4)
    res["launches_per_token"] = launches
    # what the instrumentation costs: (stamped - product) wall time per token over the launches of a token (the two fold launches
    # behind the token are part of it)
    res["stamp_overhead_us_per_launch"] = round((stamped["us_per_token"] - plain["us_per_token"]) / max(1, launches), 3)
    return res


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        n, shape_name, ctx, ft, fp, ckpt = int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
        _child(n, shape_name, ctx, ft, fp, ckpt)
        return
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    out_path = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
    shape_name = sys.argv[3] if len(sys.argv) > 3 else "qwen3-0.6b"
    s = json.dumps(graph_kernel_durations(n, shape_name), indent=1)
    print(s)
    if out_path:
        open(out_path, "w").write(s + "\n")


if __name__ == "__main__":
    main()
