#!/bin/bash
# PMC passes for the dense-prefill attention kernel: instruction mix, issue-busy cycles and wait reasons (each pass its own run)
out=gpurun_out/${1:-pfpmc}; mkdir -p $out; export TMPDIR=/tmp
PY=python3
keep_pmc() { f=$(find $out/$1 -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python3 tools/pmc_collect.py $f $out/$1.json > /dev/null; rm -rf $out/$1; }
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pass$i -o p -- $PY tools/prefill_prof.py > /dev/null 2> $out/pass$i.err
  keep_pmc pass$i
done
python3 - <<PYEOF
import json,glob
for f in sorted(glob.glob("$out/pass*.json")):
    d=json.load(open(f))
    for k,v in d["kernels"].items():
        if "attn_pf" in k or "pgemm" in k: print(f.split('/')[-1], k[:40], {a:round(b) for a,b in v.items()})
PYEOF
