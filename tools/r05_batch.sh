#!/bin/bash
# batch-32 A/B: the library in the tree against gpurun_out/lib_old.so (alternating), then the batched parity tests
for rep in 1 2; do
  for L in qwen3-rs_amd/libqwen3_hip.so qwen3-rs_amd/lib_old.so; do
    [ -f $L ] && echo "$(basename $L): $(Q3_HIP_LIB=$L python3 tools/bench_batch.py --steps 128 --verify 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'tok/s', d['ms_per_step'], 'ms')")"
  done
done
python3 -m pytest tests -m gpu -x -q -k "batch" 2>&1 | tail -2
