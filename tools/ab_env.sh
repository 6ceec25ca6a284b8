#!/bin/bash
# A/B two environments of the DEVELOPER library on one box, alternating: tools/ab_env.sh "<cmd>" "ENV_A" "ENV_B" [rounds]
# e.g. tools/ab_env.sh "python3 tools/bench_batch.py --steps 128 --verify 0" "Q3_BQUANT_SPEC=0" "Q3_BQUANT_SPEC=1" 3
cmd=$1; A="$2"; B="$3"; R=${4:-3}
for rep in $(seq 1 $R); do for E in "$A" "$B"; do
  echo "[$E] $(env $E Q3_HIP_LIB=qwen3-rs_amd/libqwen3_hip_dev.so Q3_SKIP_BUILD_ID=1 $cmd 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('value'), d.get('unit'), d.get('ms_per_step'), d.get('prefill_tok_s'), d.get('decode_tok_s'))")"
done; done
