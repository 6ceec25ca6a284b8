#!/bin/bash
# dense prefill: block size / attention geometry sweep (per-kernel averages of one 2,048-token prefill, 4B shape)
out=gpurun_out/${1:-pfm}; mkdir -p $out
for cfg in ${CFGS:-"128 4" "256 8" "256 4"}; do
  set -- $(echo $cfg | tr _ " ")
  cd /tmp && export TMPDIR=/tmp
  Q3_PREFILL_M=$1 Q3_PREFILL_ATT_NP=$2 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/p -o pf -- python3 $GRAFT_REPO_ROOT/tools/prefill_prof.py > $GRAFT_REPO_ROOT/$out/p.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find $out/p -name "*kernel_stats.csv" | head -1)
  echo "== M=$1 NP=$2"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'k_pgemm' in n or 'k_attn_pf' in n or 'bquant' in n or 'knorm' in n: print('  %-60s calls %5s total %7.1f ms avg %8.1f us' % (n[:60], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
"; grep "tok/s" $out/p.log
  rm -rf $out/p
done
