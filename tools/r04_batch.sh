#!/bin/bash
# round-4 batch-32 iteration: GPU tests, then config 4 with the k_dgemm / fused-hq switches A/B, then config 3 (the MFMA VGPR-form flag touches k_pgemm)
out=gpurun_out/${1:-r04_batch}; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -4 $out/pytest.log
for cfg in "1 1" "1 0" "0 0"; do
  set -- $cfg
  Q3_BATCH_DGEMM=$1 Q3_BATCH_FUSE_HQ=$2 timeout 600 python tools/bench_batch.py --steps 128 > $out/batch_d$1_f$2.json 2> $out/batch_d$1_f$2.err
  echo "dgemm=$1 fuse_hq=$2: $(python3 -c "import json;d=json.load(open('$out/batch_d$1_f$2.json'));print(d['value'],d['ms_per_step'],d['tokens_identical'])" 2>&1 | tail -1)"
done
timeout 600 python tools/bench_chat.py > $out/chat.json 2> $out/chat.err; python3 -c "import json;d=json.load(open('$out/chat.json'));print('chat',d['prefill_tok_s'],d['decode_tok_s'],d['batched_prefill_identical_to_sequential'])"
