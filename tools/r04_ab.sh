#!/bin/bash
# A/B of environment knobs on config 4 (batch-32 decode) under rocprofv3 kernel stats (eager launches).
# usage: r04_ab.sh <name> "<ENV=.. ENV=..>" ["<ENV=..>" ...]      (Q3_X=0 = defaults)
name=$1; shift
out=gpurun_out/$name; mkdir -p $out; export TMPDIR=/tmp
i=0
for envs in "$@"; do
  i=$((i+1))
  ( export $envs Q3_EAGER_LAUNCH=1; rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace$i -o t -- python3 tools/bench_batch.py --steps 16 --verify 0 > $out/run$i.out 2> $out/run$i.err )
  f=$(find $out/trace$i -name "*kernel_stats.csv" | head -1)
  echo "== [$envs] $(python3 -c "import json;d=json.loads(open('$out/run$i.out').read().strip().splitlines()[-1]);print(d['value'],'tok/s',d['ms_per_step'],'ms')" 2>&1 | tail -1)"
  [ -n "$f" ] && cp $f $out/stats$i.csv && python3 - "$f" <<'PYEOF'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('gemm', 'bquant', 'attn')):
        print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PYEOF
  rm -rf $out/trace$i
done
