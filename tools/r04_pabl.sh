#!/bin/bash
# compile-time ablations of k_pgemm2 (libq3_pabl<bits>.so built with -DQ3_PABLATE=<bits>): kernel stats of one traced 4B prefill per build
out=gpurun_out/${1:-r04_pabl}; mkdir -p $out; export TMPDIR=/tmp
for lib in qwen3-rs_amd/libqwen3_hip.so qwen3-rs_amd/libq3_pabl*.so; do
  n=$(basename $lib .so)
  ( export Q3_HIP_LIB=$PWD/$lib Q3_EAGER_LAUNCH=1; rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_$n -o t -- python3 tools/prefill_prof.py > $out/run_$n.out 2> $out/run_$n.err )
  f=$(find $out/t_$n -name "*kernel_stats.csv" | head -1)
  echo "== $n"
  [ -n "$f" ] && python3 - "$f" <<'PYEOF'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'pgemm2' in r['Name']:
        print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PYEOF
  rm -rf $out/t_$n
done
