#!/bin/bash
D=qwen3-rs_amd/libqwen3_hip_dev.so
for abl in 0 2048 4096 6144; do
  echo "ablate $abl: $(Q3_HIP_LIB=$D Q3_ABLATE=$abl python3 tools/longctx_prof.py qwen3-4b 2300 16 2>/dev/null | head -3 | tr '\n' ' ')"
done
