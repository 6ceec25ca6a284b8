#!/bin/bash
# round-3 iteration pass: GPU tests, config sweep, in-kernel timelines (dev build); usage: r03_iter.sh <outdir> [shape ...]
out=gpurun_out/${1:-iter}; mkdir -p $out; shift
shapes=${@:-qwen3-0.6b}
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
for shape in $shapes; do
  nt=64; [ "$shape" != "qwen3-0.6b" ] && nt=24
  timeout 900 python tools/cfg_sweep.py $shape $nt > $out/sweep_$shape.log 2>&1; cat $out/sweep_$shape.log
  Q3_STAMPS=1 Q3_STRICT=1 Q3_SHAPE=$shape Q3_HIP_LIB=qwen3-rs_amd/libqwen3_hip_dev.so Q3_NTOK=24 timeout 300 python tools/gen_loop.py > $out/stamps_$shape.log 2>&1
  tail -9 $out/stamps_$shape.log
done
