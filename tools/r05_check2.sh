#!/bin/bash
# full GPU suite (variant forms through the developer build), ticket-order A/B, graph-mode kernel durations
out=gpurun_out/r05_check2; mkdir -p $out
python3 -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
echo "== ticket order A/B (device loop, 128 tokens x 8 runs, alternating)"
for rep in 1 2 3; do
  for L in qwen3-rs_amd/libqwen3_hip.so qwen3-rs_amd/libq3_ticket_relaxed.so; do
    echo "$(basename $L): $(Q3_HIP_LIB=$L Q3_STRICT=1 Q3_NTOK=128 Q3_REPS=8 python3 tools/gen_loop.py 2>&1 | grep 'tok/s' | awk '{s+=$4; n++} END{printf "%.2f us/tok avg of %d", s/n, n}')"
  done
done
python3 tools/kstamps.py 128 $out/kstamps_128.json > $out/kstamps.log 2>&1; tail -40 $out/kstamps.log
python3 tools/kstamps.py 20 $out/kstamps_20.json > /dev/null 2>&1; grep "sum_\|product_lib\|stamped" $out/kstamps_20.json
