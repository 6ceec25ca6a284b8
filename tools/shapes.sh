#!/bin/bash
# untraced per-family launch periods (graph replay) of the three single-stream shapes + in-kernel timelines (dev build)
out=gpurun_out/${1:-shapes3}; mkdir -p $out
python3 -c "
import sys; sys.path.insert(0,'qwen3-rs_amd'); from qwen3_rs_amd import checkpoint as ck
for s in ('qwen3-0.6b','qwen3-4b','qwen3-8b'): ck.ensure_synthetic_checkpoint('/tmp/q3_%s.bin' % s, ck.SHAPES[s], seed=1234)"
python3 bench.py --worker --steps 128 --warmup 8 > $out/worker_qwen3-0.6b.json 2> $out/w06.err
for shape in qwen3-4b qwen3-8b; do python3 bench.py --worker --shape $shape --steps 32 --warmup 4 > $out/worker_$shape.json 2> $out/w_$shape.err; done
for shape in qwen3-0.6b qwen3-4b qwen3-8b; do
  Q3_STAMPS=1 Q3_STRICT=1 Q3_SHAPE=$shape Q3_HIP_LIB=qwen3-rs_amd/libqwen3_hip_dev.so Q3_NTOK=24 timeout 300 python tools/gen_loop.py > $out/stamps_$shape.log 2>&1
done
python3 -c "
import json
for s in ('qwen3-0.6b','qwen3-4b','qwen3-8b'):
    d=json.load(open('$out/worker_%s.json' % s)); print(s, d['value'], d['roofline']['frac'], [(k['kernel'],k['avg_us']) for k in d['roofline']['per_kernel']])"
