#!/bin/bash
# Same-box A/B of two library builds over every BASELINE config (alternating, 2-3 repetitions): tools/ab_rounds.sh <libA> <libB>
A=$1; B=$2
j() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' '.join(str(d.get(k)) for k in sys.argv[1:]))" "$@"; }
for rep in 1 2 3; do for L in $A $B; do
  export Q3_HIP_LIB=$L Q3_SKIP_BUILD_ID=1
  echo "[$(basename $L)] 0.6B 128 tok: $(Q3_STRICT=1 Q3_NTOK=128 Q3_REPS=3 python3 tools/gen_loop.py 2>&1 | sort -k3 -n | tail -1) | 20 tok: $(Q3_STRICT=1 Q3_NTOK=20 Q3_REPS=4 python3 tools/gen_loop.py 2>&1 | sort -k3 -n | tail -1)"
  echo "[$(basename $L)] 8B: $(Q3_SHAPE=qwen3-8b Q3_STRICT=1 Q3_NTOK=32 Q3_REPS=3 python3 tools/gen_loop.py 2>&1 | sort -k3 -n | tail -1) | 4B: $(Q3_SHAPE=qwen3-4b Q3_STRICT=1 Q3_NTOK=32 Q3_REPS=3 python3 tools/gen_loop.py 2>&1 | sort -k3 -n | tail -1)"
  echo "[$(basename $L)] config3 prefill / decode: $(python3 tools/bench_chat.py 2>/dev/null | j prefill_tok_s decode_tok_s) | config4: $(python3 tools/bench_batch.py --steps 256 --verify 0 2>/dev/null | j value)"
done; done
