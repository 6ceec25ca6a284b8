#!/bin/bash
out=gpurun_out/${1:-batch}; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "batch or prefill or sampler" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -6 $out/pytest.log
timeout 900 python tools/bench_batch.py --steps 64 > $out/batch.json 2> $out/batch.err; python3 -c "
import json; d=json.load(open('$out/batch.json')); print({k:d[k] for k in ('value','ms_per_step','single_stream_tok_s','weights_once_per_step_hbm_frac_of_8TBps','tokens_identical')})"
Q3_BQUANT_SPLIT=0 timeout 900 python tools/bench_batch.py --steps 64 --verify 0 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('single-workgroup prologues (round-1 form)', {k:d[k] for k in ('value','ms_per_step')})"
