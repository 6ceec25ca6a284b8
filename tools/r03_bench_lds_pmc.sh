#!/bin/bash
# LDS counters of the single-stream kernels (own rocprofv3 --pmc run, eager launches)
out=gpurun_out/${1:-ldspmc}; mkdir -p $out; export TMPDIR=/tmp
python3 - <<'PYEOF'
import sys; sys.path.insert(0,"qwen3-rs_amd")
from qwen3_rs_amd import checkpoint as ck
for s in ("qwen3-0.6b", "qwen3-8b"): ck.ensure_synthetic_checkpoint("/tmp/q3_%s.bin" % s, ck.SHAPES[s], seed=1234)
PYEOF
for shape in qwen3-0.6b qwen3-8b; do
Q3_EAGER_LAUNCH=1 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/p_$shape -o p -- python3 bench.py --worker --shape $shape --steps 16 --warmup 2 > /dev/null 2> $out/p_$shape.err
f=$(find $out/p_$shape -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 tools/pmc_collect.py $f $out/lds_$shape.json > /dev/null; rm -rf $out/p_$shape
python3 - <<PYEOF
import json
d=json.load(open("$out/lds_$shape.json"))
for k,v in d["kernels"].items(): print("$shape", k[:60], {a.replace('avg_',''):round(b) for a,b in v.items()})
PYEOF
done
