#!/bin/bash
# prefill A/B (config 3, 4B shape): env knob sets -> prefill tok/s (bench_chat, decode shortened) + kernel stats of one traced prefill
out=gpurun_out/${1:-r04_pfab}; shift; mkdir -p $out; export TMPDIR=/tmp
i=0
for envs in "$@"; do i=$((i+1))
  ( export $envs; python3 tools/bench_chat.py --decode 64 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('== [$envs] prefill',d['prefill_tok_s'],'identical',d['batched_prefill_identical_to_sequential'])" )
  ( export $envs Q3_EAGER_LAUNCH=1; rocprofv3 --kernel-trace --stats --output-format csv -d $out/t$i -o t -- python3 tools/prefill_prof.py > $out/run$i.out 2> $out/run$i.err )
  f=$(find $out/t$i -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $out/stats$i.csv && python3 - "$f" <<'PYEOF'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'pgemm' in r['Name'] or 'attn_pf' in r['Name']:
        print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PYEOF
  rm -rf $out/t$i
done
