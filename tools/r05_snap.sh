#!/bin/bash
# evidence snapshot: GPU suite, graph-mode kernel durations, the default bench line and the driver-form line
out=gpurun_out/r05_snap; mkdir -p $out
python3 -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
python3 tools/kstamps.py 128 $out/kstamps_128.json > /dev/null 2>&1
python3 tools/kstamps.py 20 $out/kstamps_20.json > /dev/null 2>&1
python3 -c "
import json
for n in (128, 20):
    d=json.load(open('$out/kstamps_%d.json'%n)); print(n, 'product', d['product_library_us_per_token'], 'stamped', d['stamped_developer_library_us_per_token'], 'sum', d['sum_duration_plus_gap_us_per_token']); print({k:(v['avg_duration_us'],v['avg_gap_to_predecessor_us']) for k,v in d['families'].items()})
"
python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_form.json 2> $out/bench_driver_form.err
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 -c "
import json
for f in ('bench_driver_form','bench_default'):
    d=json.load(open('$out/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d.get('parity'), d['roofline']['frac'], d.get('value_128',{}).get('value'), d.get('forward_surface',{}).get('value'), d.get('forward_surface_128',{}).get('value'))
    for k,v in (d.get('other_configs') or {}).items(): print('   ', k, v.get('value'), v.get('prefill_tok_s'), v.get('decode_tok_s'))
"
