#!/usr/bin/env python3
"""Developer: per-kernel durations UNDER hipGraph REPLAY from in-kernel begin / end stamps (s_memrealtime, 100 MHz), where rocprofv3 of
this image cannot trace.  Needs the developer build:
   make -C qwen3-rs_amd dev && python3 tools/kstamps.py [n_tokens] [out.json] [shape]
Every wave of every launch folds its entry / exit time into two cells of the launch (atomic min / max); a one-thread kernel behind the
token's last launch turns them into duration and gap-to-predecessor sums.  The same process first measures the device loop WITHOUT stamps
on the product library, so the file carries the reconciliation: sum over a token of (duration + gap) against ms_per_step."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
out_path = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
shape_name = sys.argv[3] if len(sys.argv) > 3 else "qwen3-0.6b"
if os.environ.get("Q3_KSTAMPS_CHILD"):
    import qwen3_rs_amd as q3
    from qwen3_rs_amd import checkpoint as ck
    sh = ck.SHAPES[shape_name]; path = f"/tmp/q3_{shape_name}.bin"
    ck.ensure_synthetic_checkpoint(path, sh, seed=1234)
    t = q3.TransformerBuilder(path).with_ctx_length(1024).build()
    t.generate_greedy(5, 7, 8)
    best = 1e9
    for _ in range(3):
        t.reset_kv(); t0 = time.perf_counter(); t.generate_greedy(5, 7, n); best = min(best, (time.perf_counter() - t0) / n)
    print(json.dumps({"us_per_token": best * 1e6, "build_id": q3.load_library().q3_build_id().decode()}))
    t.close()
    sys.exit(0)
env = dict(os.environ, Q3_KSTAMPS_CHILD="1")
env.pop("Q3_KSTAMPS", None); env.pop("Q3_HIP_LIB", None)
plain = json.loads(subprocess.run([sys.executable, __file__, str(n), "-", shape_name], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1])
env.update(Q3_KSTAMPS="1", Q3_HIP_LIB=os.path.join(ROOT, "qwen3-rs_amd", "libqwen3_hip_dev.so"))
r = subprocess.run([sys.executable, __file__, str(n), "-", shape_name], env=env, capture_output=True, text=True)
stamped = json.loads(r.stdout.strip().splitlines()[-1])
lines = [l for l in r.stderr.splitlines() if l.startswith("[q3 kstamps]")]
k = json.loads(lines[-1][len("[q3 kstamps] "):])
res = {"what": shape_name + " device loop, %d tokens from position 7, hipGraph replay; per family: average launch duration (first wave in .. last wave out) "
               "and average gap to the predecessor's end, in-kernel 100 MHz clock, developer build with Q3_KSTAMPS=1" % n,
       "build_id": plain["build_id"], "product_library_us_per_token": round(plain["us_per_token"], 2),
       "stamped_developer_library_us_per_token": round(stamped["us_per_token"], 2), **k}
res["sum_over_product_ms_per_step"] = round(k["sum_duration_plus_gap_us_per_token"] / plain["us_per_token"], 4)
s = json.dumps(res, indent=1)
print(s)
if out_path:
    open(out_path, "w").write(s + "\n")
