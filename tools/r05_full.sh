#!/bin/bash
# full GPU suite + device-loop rate of the release build
out=gpurun_out/r05_full; mkdir -p $out
python3 -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
for n in 128 20; do
  echo "$n: $(Q3_STRICT=1 Q3_NTOK=$n Q3_REPS=8 python3 tools/gen_loop.py 2>&1 | grep 'tok/s' | awk '{s+=$4; n++} END{printf "%.1f us/tok avg of %d", s/n, n}')"
done
