#!/bin/bash
# the headline bench under rocprofv3 with hipGraph replay (as bench.py runs it); eager fallback is in r02_profile.sh
out=gpurun_out/${1:-profg}; mkdir -p $out; export TMPDIR=/tmp
python3 - <<'PYEOF'
import sys; sys.path.insert(0,"qwen3-rs_amd")
from qwen3_rs_amd import checkpoint as ck
ck.ensure_synthetic_checkpoint("/tmp/q3_qwen3-0.6b.bin", ck.SHAPES["qwen3-0.6b"], seed=1234)
PYEOF
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o b -- python3 bench.py --worker --steps 128 --warmup 8 > $out/bench_worker.json 2> $out/bench_worker.err
f=$(find $out/bench -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/bench_kernel_stats.csv; rm -rf $out/bench
head -12 $out/bench_kernel_stats.csv; python3 -c "
import json; b=json.load(open('$out/bench_worker.json')); print(b['value'], b['roofline']['frac'], b['roofline']['avg_launch_us'], [(k['kernel'],k['avg_us']) for k in b['roofline']['per_kernel']])"
tail -3 $out/bench_worker.err
