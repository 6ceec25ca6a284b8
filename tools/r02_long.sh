#!/bin/bash
out=gpurun_out/${1:-long}; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -4 $out/pytest.log
timeout 900 python tools/bench_chat.py > $out/chat.json 2> $out/chat.err; cat $out/chat.json | python3 -c "import sys,json; d=json.load(sys.stdin); print({k:d[k] for k in ('prefill_tok_s','prefill_sequential_tok_s','decode_tok_s','batched_prefill_identical_to_sequential','decode_hbm_frac_of_8TBps')})"
