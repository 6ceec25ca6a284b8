#!/bin/bash
python3 -m pytest tests -m gpu -x -q -k "long or attention or chat or 4b or config3" 2>&1 | tail -2
for rep in 1 2; do echo "$(python3 tools/longctx_prof.py qwen3-4b 2300 32 2>/dev/null | head -3 | tr '\n' ' ')"; done
for n in 128 20; do echo "$n: $(Q3_STRICT=1 Q3_NTOK=$n Q3_REPS=8 python3 tools/gen_loop.py 2>&1 | grep 'tok/s' | awk '{s+=$4; n++} END{printf "%.1f us/tok avg of %d", s/n, n}')"; done
