#!/bin/bash
# round 5 A/B on one box: GPU parity tests (attention-related first), then the 0.6B device loop with the round 2-4 attention kernel
# (Q3_ATT_SHORT=2) against k_attn_short2, alternating.
out=gpurun_out/r05_ab; mkdir -p $out
python3 -m pytest tests -m gpu -x -q -k "attention or attn or forward or golden or generate" > $out/pytest_attn.txt 2>&1; tail -3 $out/pytest_attn.txt
for rep in 1 2 3; do
  for form in 2 1; do
    echo "Q3_ATT_SHORT=$form 128: $(Q3_ATT_SHORT=$form Q3_STRICT=1 Q3_NTOK=128 Q3_REPS=6 python3 tools/gen_loop.py 2>&1 | grep 'tok/s' | awk '{s+=$4; n++} END{printf "%.1f us/tok avg of %d", s/n, n}')"
    echo "Q3_ATT_SHORT=$form  20: $(Q3_ATT_SHORT=$form Q3_STRICT=1 Q3_NTOK=20 Q3_REPS=6 python3 tools/gen_loop.py 2>&1 | grep 'tok/s' | awk '{s+=$4; n++} END{printf "%.1f us/tok avg of %d", s/n, n}')"
  done
done
