#!/bin/bash
# rocprofv3 evidence, one section per argument; every profiled run under `timeout`, the traced program directly behind `--`.
#   tools/profile.sh <outdir-under-gpurun_out> <section> [section ...]
# sections: bench (0.6B device loop, eager kernel trace) | bench8b | batch | prefill | long4b | pmc_fetch | pmc_batch | pmc_prefill
# Kernel traces are EAGER runs of the product library (Q3_EAGER_LAUNCH=1 -> Q3_FLAG_NO_GRAPH: rocprofv3 of this image faults inside
# hipGraphLaunch); PMC passes run alone (--pmc with --kernel-trace only), FETCH_SIZE on the bare device loop (tools/gen_loop.py).
out=gpurun_out/$1; shift; mkdir -p $out; export TMPDIR=/tmp
keep_stats() { f=$(find $out/$1 -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/$1_kernel_stats.csv; rm -rf $out/$1; }
keep_pmc() { f=$(find $out/$1 -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python3 tools/pmc_collect.py $f $out/$1.json > /dev/null; rm -rf $out/$1; }
trace() { name=$1; shift; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -o t -- "$@" > $out/$name.out 2> $out/$name.err; echo "$name trace rc=$?"; keep_stats $name; }
pmc() { name=$1; ctrs=$2; shift; shift; timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/$name -o p -- "$@" > /dev/null 2> $out/$name.err; echo "$name pmc rc=$?"; keep_pmc $name; }
ensure() { python3 -c "
import sys; sys.path.insert(0,'qwen3-rs_amd')
from qwen3_rs_amd import checkpoint as ck
ck.ensure_synthetic_checkpoint('/tmp/q3_$1.bin', ck.SHAPES['$1'], seed=1234)"; }
for sec in "$@"; do
  case $sec in
    bench)    ensure qwen3-0.6b; Q3_EAGER_LAUNCH=1 trace bench python3 bench.py --worker --steps 128 --warmup 8 --no-tolerance-mode ;;
    bench8b)  ensure qwen3-8b; Q3_EAGER_LAUNCH=1 trace bench_qwen3-8b python3 bench.py --worker --shape qwen3-8b --steps 32 --warmup 4 --no-tolerance-mode ;;
    batch)    Q3_EAGER_LAUNCH=1 trace batch python3 tools/bench_batch.py --steps 16 --verify 0 ;;
    prefill)  Q3_EAGER_LAUNCH=1 trace prefill python3 tools/prefill_prof.py ;;
    long4b)   Q3_EAGER_LAUNCH=1 Q3_PROFILE_FAMILIES=0 trace chat_decode_pos2300 python3 tools/longctx_prof.py qwen3-4b 2300 32 ;;
    pmc_fetch)
      for shape in qwen3-0.6b qwen3-8b qwen3-4b; do
        Q3_NTOK=8 Q3_REPS=1 Q3_SHAPE=$shape pmc pmc_fetch_$shape FETCH_SIZE python3 tools/gen_loop.py
      done ;;
    pmc_batch)
      i=0
      for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" \
                 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM" \
                 "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES"; do
        i=$((i+1)); Q3_EAGER_LAUNCH=1 pmc pmc_batch_pass$i "$set" python3 tools/bench_batch.py --steps 8 --verify 0
      done ;;
    pmc_prefill)
      i=0
      for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
                 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
                 "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"; do
        i=$((i+1)); pmc pmc_prefill_pass$i "$set" python3 tools/prefill_prof.py
      done ;;
    *) echo "unknown section $sec" ;;
  esac
done
ls -la $out
