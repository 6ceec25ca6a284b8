#!/bin/bash
# round-5 closing rocprofv3 evidence (every run under `timeout`; the traced program follows `--` directly).  Kernel traces are EAGER runs of the
# product library (Q3_EAGER_LAUNCH=1 makes the Python layer pass Q3_FLAG_NO_GRAPH; rocprofv3 on this image faults inside hipGraphLaunch).
out=gpurun_out/r05_final; mkdir -p $out; export TMPDIR=/tmp

python3 - <<'PYEOF'
import sys; sys.path.insert(0,"qwen3-rs_amd")
from qwen3_rs_amd import checkpoint as ck
for s in ("qwen3-0.6b", "qwen3-4b"):
    ck.ensure_synthetic_checkpoint(f"/tmp/q3_{s}.bin", ck.SHAPES[s], seed=1234)
PYEOF
keep_stats() { f=$(find $out/$1 -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/$1_kernel_stats.csv; rm -rf $out/$1; }
# 1. 0.6B, 128 tokens (the bench worker)
Q3_EAGER_LAUNCH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o b -- python3 bench.py --worker --steps 128 --warmup 8 > $out/bench_worker.json 2> $out/bench_worker.err; echo "bench trace rc=$?"
keep_stats bench
# 2. 4B decode at position 2,300 (config 3's decode leg): k_attn_scores_kv / k_attn_out
Q3_EAGER_LAUNCH=1 Q3_PROFILE_FAMILIES=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/long4b -o l -- python3 tools/longctx_prof.py qwen3-4b 2300 32 > $out/long4b.out 2> $out/long4b.err; echo "long trace rc=$?"
keep_stats long4b
# 3. HBM bytes of the 0.6B token (product library, graph replay)
Q3_NTOK=8 Q3_REPS=1 Q3_SHAPE=qwen3-0.6b timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o p -- python3 tools/gen_loop.py > /dev/null 2> $out/pmc_fetch.err; echo "pmc rc=$?"
f=$(find $out/pmc_fetch -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python3 tools/pmc_collect.py $f $out/pmc_fetch.json > /dev/null; rm -rf $out/pmc_fetch
ls -la $out
