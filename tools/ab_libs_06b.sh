#!/bin/bash
# A/B two library builds on the 0.6B device loop only: tools/ab_libs_06b.sh <libA> <libB> [rounds]   (alternating, 10 x 128 tokens per run)
A=$1; B=$2; N=${3:-4}
for rep in $(seq 1 $N); do for L in $A $B; do
  echo "$(basename $L) $(Q3_HIP_LIB=$L Q3_STRICT=1 Q3_NTOK=128 Q3_REPS=10 python3 tools/gen_loop.py 2>&1 | grep 'tok/s' | awk '{s+=$4; n++} END{printf "%.1f us/tok avg of %d", s/n, n}')"
done; done
