#!/bin/bash
# A/B two builds of the library on one box: tools/ab_libs.sh <libA> <libB>  (alternating runs, same GPU)
A=$1; B=$2
for rep in 1 2 3; do
  for L in $A $B; do
    echo "== $L"
    Q3_HIP_LIB=$L Q3_STRICT=1 Q3_NTOK=128 Q3_REPS=3 python3 tools/gen_loop.py 2>&1 | tail -1
    Q3_HIP_LIB=$L Q3_SHAPE=qwen3-8b Q3_STRICT=1 Q3_NTOK=32 Q3_REPS=2 python3 tools/gen_loop.py 2>&1 | tail -1
  done
done
for L in $A $B; do echo "== $L"; Q3_HIP_LIB=$L Q3_PROFILE_FAMILIES=0 python3 tools/longctx_prof.py qwen3-4b 2300 32 | head -1; done
