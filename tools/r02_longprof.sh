#!/bin/bash
out=gpurun_out/${1:-longprof}; mkdir -p $out; export TMPDIR=/tmp
python3 tools/longctx_prof.py qwen3-4b 2300 8 > /dev/null 2>&1
Q3_EAGER_LAUNCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -o l -- python3 tools/longctx_prof.py qwen3-4b 2300 32 > $out/long.txt 2> $out/long.err
f=$(find $out/p -name "*kernel_stats.csv" | head -1); cp $f $out/long_kernel_stats.csv; rm -rf $out/p
cat $out/long.txt; head -12 $out/long_kernel_stats.csv
python3 tools/longctx_prof.py qwen3-4b 2300 64
