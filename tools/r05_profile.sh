#!/bin/bash
# round-4 rocprofv3 evidence: kernel-trace stats and PMC passes in SEPARATE runs; summaries land in gpurun_out/<dir>/ and are
# copied into profiles/r05_* by hand.  Every profiled program is a single process that owns the GPU itself.
out=gpurun_out/${1:-prof5}; mkdir -p $out; export TMPDIR=/tmp
what=${2:-all}
PY=python3
$PY - <<'PYEOF'
import sys; sys.path.insert(0,"qwen3-rs_amd")
from qwen3_rs_amd import checkpoint as ck
ck.ensure_synthetic_checkpoint("/tmp/q3_qwen3-0.6b.bin", ck.SHAPES["qwen3-0.6b"], seed=1234)
PYEOF
keep_stats() { f=$(find $out/$1 -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/$1_kernel_stats.csv; rm -rf $out/$1; }
keep_pmc() { f=$(find $out/$1 -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python3 tools/pmc_collect.py $f $out/$1.json > /dev/null; rm -rf $out/$1; }
if [ $what = all ] || [ $what = bench ]; then
  # graph replay first (as bench.py runs it); if this image's rocprofv3 faults inside hipGraphLaunch, the eager trace below stands in
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_graph -o b -- $PY bench.py --worker --steps 128 --warmup 8 > $out/bench_graph_worker.json 2> $out/bench_graph_worker.err; echo "graph trace rc=$?"
  keep_stats bench_graph
  Q3_EAGER_LAUNCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o b -- $PY bench.py --worker --steps 128 --warmup 8 > $out/bench_worker.json 2> $out/bench_worker.err; echo "eager trace rc=$?"
  keep_stats bench
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o p -- $PY bench.py --worker --steps 8 --warmup 2 > /dev/null 2> $out/pmc_fetch.err
  keep_pmc pmc_fetch
fi
if [ $what = all ] || [ $what = shapes ]; then
  for shape in qwen3-8b qwen3-4b; do
    $PY -c "import sys; sys.path.insert(0,'qwen3-rs_amd'); from qwen3_rs_amd import checkpoint as ck; ck.ensure_synthetic_checkpoint('/tmp/q3_$shape.bin', ck.SHAPES['$shape'], seed=1234)"
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch_$shape -o p -- $PY bench.py --worker --shape $shape --steps 8 --warmup 2 > /dev/null 2> $out/pmc_fetch_$shape.err
    keep_pmc pmc_fetch_$shape
    Q3_EAGER_LAUNCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_$shape -o b -- $PY bench.py --worker --shape $shape --steps 32 --warmup 4 > $out/bench_$shape.json 2> $out/bench_$shape.err
    keep_stats bench_$shape
  done
fi
if [ $what = all ] || [ $what = chat ]; then
  Q3_EAGER_LAUNCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/chat -o c -- $PY tools/bench_chat.py --decode 128 > $out/chat.json 2> $out/chat.err
  keep_stats chat
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $out/pmc_mfma_prefill -o m -- $PY tools/prefill_prof.py > /dev/null 2> $out/pmc_mfma_prefill.err
  keep_pmc pmc_mfma_prefill
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch_prefill -o p -- $PY tools/prefill_prof.py > /dev/null 2> $out/pmc_fetch_prefill.err
  keep_pmc pmc_fetch_prefill
fi
if [ $what = all ] || [ $what = batch ]; then
  Q3_EAGER_LAUNCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/batch -o b -- $PY tools/bench_batch.py --steps 32 --verify 0 > $out/batch.json 2> $out/batch.err
  keep_stats batch
fi
for f in $out/*_kernel_stats.csv; do echo "== $f"; head -12 $f | cut -c1-170; done
ls -la $out
tail -2 $out/*.err | tail -30
