#!/bin/bash
# HIP runtime knobs against the 0.6B device loop (128 tokens, graph replay): usage r04_env_sweep.sh "<ENV=..>" ...   (Q3_X=0 = defaults)
for rep in 1 2; do for envs in "$@"; do
  echo "rep $rep [$envs] $( ( export $envs; Q3_STRICT=1 Q3_NTOK=128 Q3_REPS=3 timeout 300 python3 tools/gen_loop.py 2>&1 | tail -1 ) )"
done; done
