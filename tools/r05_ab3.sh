#!/bin/bash
# same-box A/B of two library builds on the 0.6B device loop (alternating), then the parity tests that exercise the exact sums
for rep in 1 2 3; do
  for L in qwen3-rs_amd/libqwen3_hip.so qwen3-rs_amd/lib_old.so; do
    echo "$(basename $L) 128: $(Q3_HIP_LIB=$L Q3_STRICT=1 Q3_NTOK=128 Q3_REPS=8 python3 tools/gen_loop.py 2>&1 | grep 'tok/s' | awk '{s+=$4; n++} END{printf "%.2f", s/n}')  20: $(Q3_HIP_LIB=$L Q3_STRICT=1 Q3_NTOK=20 Q3_REPS=8 python3 tools/gen_loop.py 2>&1 | grep 'tok/s' | awk '{s+=$4; n++} END{printf "%.2f", s/n}')  8B: $(Q3_HIP_LIB=$L Q3_SHAPE=qwen3-8b Q3_STRICT=1 Q3_NTOK=32 Q3_REPS=3 python3 tools/gen_loop.py 2>&1 | grep 'tok/s' | awk '{s+=$4; n++} END{printf "%.1f", s/n}')"
  done
done
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
