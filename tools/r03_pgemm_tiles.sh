#!/bin/bash
# k_pgemm tile sweep: per-kernel averages of one dense 2,048-token prefill for forced (RT, PT)
out=gpurun_out/${1:-pgt}; mkdir -p $out
for t in ${TILES:-0 24 22 14 12 11}; do
  cd /tmp && export TMPDIR=/tmp
  Q3_PGEMM_TILE=$t rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/p$t -o pf -- python3 $GRAFT_REPO_ROOT/tools/prefill_prof.py > $GRAFT_REPO_ROOT/$out/p$t.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find $out/p$t -name "*kernel_stats.csv" | head -1)
  echo "== TILE=$t"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'k_pgemm' in n or 'k_attn_pf' in n or 'bquant' in n: print('  %-60s calls %5s avg %8.1f us  min %7.1f max %7.1f' % (n[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
"; grep "tok/s" $out/p$t.log
  rm -rf $out/p$t
done
