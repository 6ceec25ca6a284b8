#!/bin/bash
# usage: ab_cfg.sh shape ntok "ENV_A" "ENV_B" rounds
s=$1; n=$2; A="$3"; B="$4"; R=${5:-3}
for rep in $(seq 1 $R); do for E in "$A" "$B"; do
  echo "[$E] $(env $E Q3_HIP_LIB=qwen3-rs_amd/libqwen3_hip_dev.so Q3_SKIP_BUILD_ID=1 Q3_SHAPE=$s Q3_STRICT=1 Q3_NTOK=$n Q3_REPS=5 python3 tools/gen_loop.py 2>&1 | grep 'tok/s' | sort -k4 -n | head -1)"
done; done
