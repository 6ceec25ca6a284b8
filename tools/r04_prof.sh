#!/bin/bash
# rocprofv3 kernel stats of one command (eager launches: this image's rocprofv3 crashes inside hipGraphLaunch with kernel tracing)
# usage: r04_prof.sh <name> <program> [args...]   -> gpurun_out/<name>_kernel_stats.csv + the top rows on stdout
name=$1; shift
out=gpurun_out/$name; mkdir -p $out; export TMPDIR=/tmp
Q3_EAGER_LAUNCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- "$@" > $out/run.out 2> $out/run.err
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/${name}_kernel_stats.csv && python3 - "$f" <<'PYEOF'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us  {r['Percentage']:>6s}%")
PYEOF
rm -rf $out/trace
tail -2 $out/run.out
