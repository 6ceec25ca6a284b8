#!/bin/bash
# FETCH_SIZE pass for the 8B and 4B single-stream shapes (own runs, kernel trace only), summaries kept
out=gpurun_out/${1:-pmc8b}; mkdir -p $out; export TMPDIR=/tmp
for shape in qwen3-8b qwen3-4b; do
  python3 -c "import sys; sys.path.insert(0,'qwen3-rs_amd'); from qwen3_rs_amd import checkpoint as ck; ck.ensure_synthetic_checkpoint('/tmp/q3_$shape.bin', ck.SHAPES['$shape'], seed=1234)"   # the worker expects its parent to have written the checkpoint
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_$shape -o p -- python3 bench.py --worker --shape $shape --steps 8 --warmup 2 > /dev/null 2> $out/pmc_$shape.err
  f=$(find $out/pmc_$shape -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 tools/pmc_collect.py $f $out/pmc_fetch_$shape.json; rm -rf $out/pmc_$shape
done
ls -la $out
