#!/bin/bash
# round 5 baseline: primitive costs, 0.6B in-kernel timelines (dev build) at two positions, device-loop rate of the release build
out=gpurun_out/r05_base; mkdir -p $out
tools/prim_probe > $out/prim.txt 2>&1
for pos in 17 70; do
  Q3_STAMPS=1 Q3_HIP_LIB=qwen3-rs_amd/libqwen3_hip_dev.so python3 tools/stamps_at.py $pos > $out/stamps_$pos.txt 2>&1
done
Q3_STRICT=1 Q3_NTOK=128 Q3_REPS=6 python3 tools/gen_loop.py > $out/loop128.txt 2>&1
Q3_STRICT=1 Q3_NTOK=20 Q3_REPS=6 python3 tools/gen_loop.py > $out/loop20.txt 2>&1
cat $out/prim.txt; tail -12 $out/stamps_17.txt; tail -12 $out/stamps_70.txt; cat $out/loop128.txt $out/loop20.txt
