#!/bin/bash
# same-box A/B of library builds on the 4B dense prefill: usage r04_libab_prefill.sh <name> <lib> [<lib> ...]
# per build: bench_chat prefill tok/s (identity to the sequential loop checked) and the traced kernel averages
out=gpurun_out/${1:-r04_libab}; shift; mkdir -p $out; export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename $lib .so)
  ( export Q3_HIP_LIB=$PWD/$lib Q3_SKIP_BUILD_ID=1; python3 tools/bench_chat.py --decode 64 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('== $n prefill',d['prefill_tok_s'],'identical',d['batched_prefill_identical_to_sequential'])" )
  ( export Q3_HIP_LIB=$PWD/$lib Q3_EAGER_LAUNCH=1; rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_$n -o t -- python3 tools/prefill_prof.py > $out/run_$n.out 2> $out/run_$n.err )
  f=$(find $out/t_$n -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $out/stats_$n.csv && python3 - "$f" <<'PYEOF'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'attn_pf' in r['Name'] or 'pgemm3<2' in r['Name']:
        print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PYEOF
  rm -rf $out/t_$n
done
