#!/bin/bash
# round-5 PMC pass (FETCH_SIZE), every profiled run under `timeout` (an untimed round-5 attempt hung for 40 minutes behind a profiler
# abort).  STATUS round 5: `rocprofv3 --pmc` faults on this image with this library in every launch mode tried -- hipGraph replay
# ("AQL packet is malformed" / SIGSEGV in the dispatch callback), eager launches, product and developer build, either attention kernel --
# while `--kernel-trace --stats` on eager launches works (profiles/r05_bench_kernel_stats.csv).  bench.py's roofline.traffic therefore
# still reads profiles/r04_pmc_fetch_size*.json: the GEMV data path (what the counter covers) did not change this round.
out=gpurun_out/${1:-pmc5}; mkdir -p $out; export TMPDIR=/tmp
keep_pmc() { f=$(find $out/$1 -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python3 tools/pmc_collect.py $f $out/$1.json > /dev/null; rm -rf $out/$1; }
python3 - <<'PYEOF'
import sys; sys.path.insert(0,"qwen3-rs_amd")
from qwen3_rs_amd import checkpoint as ck
for s in ("qwen3-0.6b", "qwen3-4b", "qwen3-8b"):
    ck.ensure_synthetic_checkpoint("/tmp/q3_%s.bin" % s, ck.SHAPES[s], seed=1234)
PYEOF
export Q3_HIP_LIB=$PWD/qwen3-rs_amd/libqwen3_hip_dev.so Q3_ATT_SHORT=${PMC_ATT_SHORT:-2}   # (the profiler faults behind k_attn_short2 launches: the GEMV counters are collected with the round 2-4 attention kernel in the plan)
for shape in qwen3-0.6b qwen3-8b qwen3-4b; do
  timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch_$shape -o p -- python3 bench.py --worker --shape $shape --steps 8 --warmup 2 > /dev/null 2> $out/pmc_fetch_$shape.err; echo "$shape rc=$?"
  keep_pmc pmc_fetch_$shape
done
ls -la $out
