#!/bin/bash
# rocprofv3 kernel-trace summary of one dense 2,048-token prefill (4B shape)
out=gpurun_out/${1:-pf}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o pf -- python3 $GRAFT_REPO_ROOT/tools/prefill_prof.py > $GRAFT_REPO_ROOT/$out/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"; head -25 "$f" | cut -c1-200
tail -3 $out/prof.log
