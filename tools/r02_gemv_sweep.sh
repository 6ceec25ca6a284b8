#!/bin/bash
out=gpurun_out/${1:-sweep}; mkdir -p $out
for pf in 1 0; do for fin in 1 0; do
echo "=== Q3_GEMV_PF=$pf Q3_GEMV_FIN=$fin"
Q3_GEMV_PF=$pf Q3_GEMV_FIN=$fin timeout 600 python tools/bench_gemv.py 2>&1 | grep -v "lm_head 0.6B"
done; done > $out/gemv_sweep.txt 2>&1
cat $out/gemv_sweep.txt
