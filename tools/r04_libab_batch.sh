#!/bin/bash
# same-box A/B of library builds on config 4 (graph replay, 128 steps) : usage r04_libab_batch.sh <lib> [<lib> ...]   (two rounds)
for rep in 1 2; do for lib in "$@"; do
  echo "rep $rep $(basename $lib) $( Q3_HIP_LIB=$PWD/$lib timeout 600 python3 tools/bench_batch.py --steps 128 --verify 1 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['tokens_identical'])" )"
done; done
