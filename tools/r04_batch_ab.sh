#!/bin/bash
# same-box A/B of config 4 (graph replay, 128 steps, two repetitions each): usage r04_batch_ab.sh <name> "<ENV..>" ...
out=gpurun_out/${1:-r04_bab}; shift; mkdir -p $out
for rep in 1 2; do i=0; for envs in "$@"; do i=$((i+1))
  ( export $envs; timeout 600 python3 tools/bench_batch.py --steps 128 --verify 1 > $out/r${rep}_$i.json 2> $out/r${rep}_$i.err )
  echo "rep $rep [$envs] $(python3 -c "import json;d=json.load(open('$out/r${rep}_$i.json'));print(d['value'],d['ms_per_step'],d['tokens_identical'])" 2>&1 | tail -1)"
done; done
