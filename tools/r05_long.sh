#!/bin/bash
# long-context decode A/B (4B dims, position 2,300): chain-wave priority on / off (developer build, Q3_ABLATE=128 = off)
D=qwen3-rs_amd/libqwen3_hip_dev.so
for rep in 1 2; do
  echo "prio on : $(Q3_HIP_LIB=$D python3 tools/longctx_prof.py qwen3-4b 2300 32 | head -8 | tr '\n' ' ')"
  echo "prio off: $(Q3_HIP_LIB=$D Q3_ABLATE=128 python3 tools/longctx_prof.py qwen3-4b 2300 32 | head -8 | tr '\n' ' ')"
done
python3 tools/kstamps.py 128 gpurun_out/r05_kstamps_128.json | tail -45
