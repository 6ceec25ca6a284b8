#!/bin/bash
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
D=qwen3-rs_amd/libqwen3_hip_dev.so
for rep in 1 2; do
  echo "V^T on : $(Q3_HIP_LIB=$D python3 tools/longctx_prof.py qwen3-4b 2300 32 2>/dev/null | head -3 | tr '\n' ' ')"
  echo "V^T off: $(Q3_HIP_LIB=$D Q3_VALUE_T=0 python3 tools/longctx_prof.py qwen3-4b 2300 32 2>/dev/null | head -3 | tr '\n' ' ')"
done
echo "release: $(python3 tools/longctx_prof.py qwen3-4b 2300 32 2>/dev/null | head -3 | tr '\n' ' ')"
for n in 128 20; do echo "$n: $(Q3_STRICT=1 Q3_NTOK=$n Q3_REPS=8 python3 tools/gen_loop.py 2>&1 | grep 'tok/s' | awk '{s+=$4; n++} END{printf "%.1f us/tok avg of %d", s/n, n}')"; done
python3 tools/bench_chat.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('config3', d['prefill_tok_s'], d['decode_tok_s'], d['batched_prefill_identical_to_sequential'])"
