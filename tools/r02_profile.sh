#!/bin/bash
# rocprofv3 evidence for the round: kernel-trace stats + PMC passes (separate runs), summaries copied into profiles/ by hand.
# Every profiled program is a single process that owns the GPU itself (no child processes under the profiler).
out=gpurun_out/${1:-prof}; mkdir -p $out; export TMPDIR=/tmp
PY=python3
$PY - <<'PYEOF'
import sys; sys.path.insert(0,"qwen3-rs_amd")
from qwen3_rs_amd import checkpoint as ck
ck.ensure_synthetic_checkpoint("/tmp/q3_qwen3-0.6b.bin", ck.SHAPES["qwen3-0.6b"], seed=1234)
ck.ensure_synthetic_checkpoint("/tmp/qwen3-4b-seed1235.q3bin", ck.SHAPES["qwen3-4b"], seed=1235)
ck.ensure_synthetic_checkpoint("/tmp/qwen3-8b-seed1236.q3bin", ck.SHAPES["qwen3-8b"], seed=1236)
PYEOF
what=${2:-all}
if [ $what = all ] || [ $what = bench ]; then
Q3_EAGER_LAUNCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o b -- $PY bench.py --worker --steps 128 --warmup 8 > $out/bench_worker.json 2> $out/bench_worker.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o p -- $PY bench.py --worker --steps 8 --warmup 2 > /dev/null 2> $out/pmc_fetch.err
fi
if [ $what = all ] || [ $what = chat ]; then
Q3_EAGER_LAUNCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/chat -o c -- $PY tools/bench_chat.py --decode 128 > $out/chat.json 2> $out/chat.err
fi
if [ $what = all ] || [ $what = batch ]; then
Q3_EAGER_LAUNCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/batch -o b -- $PY tools/bench_batch.py --steps 32 --verify 0 > $out/batch.json 2> $out/batch.err
rocprofv3 -L > $out/counters.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $out/pmc_mfma -o m -- $PY tools/bench_batch.py --steps 8 --verify 0 > /dev/null 2> $out/pmc_mfma.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch_batch -o p -- $PY tools/bench_batch.py --steps 8 --verify 0 > /dev/null 2> $out/pmc_fetch_batch.err
fi
# keep the summaries only (the traces / sqlite files are tens of MB)
for d in bench chat batch; do f=$(find $out/$d -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/${d}_kernel_stats.csv; rm -rf $out/$d; done
for d in pmc_fetch pmc_mfma pmc_fetch_batch; do f=$(find $out/$d -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python3 tools/pmc_collect.py $f $out/${d}.json; rm -rf $out/$d; done
for f in $out/*_kernel_stats.csv; do echo "== $f"; head -14 $f; done
tail -3 $out/*.err | tail -40
