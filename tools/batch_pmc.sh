#!/bin/bash
# PMC passes for the batch-32 kernels: instruction mix, issue-busy cycles and wait reasons (each pass its own run)
out=gpurun_out/${1:-bpmc}; mkdir -p $out; export TMPDIR=/tmp
keep_pmc() { f=$(find $out/$1 -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python3 tools/pmc_collect.py $f $out/$1.json > /dev/null; rm -rf $out/$1; }
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  Q3_EAGER_LAUNCH=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pass$i -o p -- python3 tools/bench_batch.py --steps 8 --verify 0 > /dev/null 2> $out/pass$i.err
  keep_pmc pass$i
done
python3 - <<PYEOF
import json,glob
for f in sorted(glob.glob("$out/pass*.json")):
    d=json.load(open(f))
    for k,v in d["kernels"].items():
        if "gemm" in k or "bquant" in k or "attn" in k: print(f.split('/')[-1], k[:44], {a.replace('avg_',''):round(b) for a,b in v.items()})
PYEOF
