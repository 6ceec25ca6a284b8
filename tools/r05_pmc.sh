#!/bin/bash
# round-5 PMC pass (FETCH_SIZE), every profiled run under `timeout` (an untimed attempt hung for 40 minutes behind a profiler abort).
# The profiled program is tools/gen_loop.py (device loop, hipGraph replay): `bench.py --worker` faults under `rocprofv3 --pmc` on this
# image this round (SIGSEGV in the profiler's dispatch callback, graph and eager launches alike), the bare device loop does not.
out=gpurun_out/${1:-pmc5}; mkdir -p $out; export TMPDIR=/tmp
keep_pmc() { f=$(find $out/$1 -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python3 tools/pmc_collect.py $f $out/$1.json > /dev/null; rm -rf $out/$1; }
python3 - <<'PYEOF'
import sys; sys.path.insert(0,"qwen3-rs_amd")
from qwen3_rs_amd import checkpoint as ck
for s in ("qwen3-0.6b", "qwen3-4b", "qwen3-8b"):
    ck.ensure_synthetic_checkpoint("/tmp/q3_%s.bin" % s, ck.SHAPES[s], seed=1234)
PYEOF
export Q3_NTOK=8 Q3_REPS=1
for shape in qwen3-0.6b qwen3-8b qwen3-4b; do
  Q3_SHAPE=$shape timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch_$shape -o p -- python3 tools/gen_loop.py > /dev/null 2> $out/pmc_fetch_$shape.err; echo "$shape rc=$?"
  keep_pmc pmc_fetch_$shape
done
ls -la $out
