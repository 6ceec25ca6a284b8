#!/bin/bash
out=gpurun_out/r05_ab2; mkdir -p $out
python3 -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
D=qwen3-rs_amd/libqwen3_hip_dev.so
for rep in 1 2 3; do
  for r in 0 1; do
    echo "dev ranges=$r  20: $(Q3_HIP_LIB=$D Q3_ATT_RANGES=$r Q3_STRICT=1 Q3_NTOK=20 Q3_REPS=8 python3 tools/gen_loop.py 2>&1 | grep 'tok/s' | awk '{s+=$4; n++} END{printf "%.2f us/tok", s/n}')   128: $(Q3_HIP_LIB=$D Q3_ATT_RANGES=$r Q3_STRICT=1 Q3_NTOK=128 Q3_REPS=6 python3 tools/gen_loop.py 2>&1 | grep 'tok/s' | awk '{s+=$4; n++} END{printf "%.2f us/tok", s/n}')"
  done
done
echo "release 20: $(Q3_STRICT=1 Q3_NTOK=20 Q3_REPS=8 python3 tools/gen_loop.py 2>&1 | grep 'tok/s' | awk '{s+=$4; n++} END{printf "%.2f us/tok", s/n}')   128: $(Q3_STRICT=1 Q3_NTOK=128 Q3_REPS=6 python3 tools/gen_loop.py 2>&1 | grep 'tok/s' | awk '{s+=$4; n++} END{printf "%.2f us/tok", s/n}')"
python3 tools/kstamps.py 128 $out/kstamps_128.json > $out/kstamps.log 2>&1; grep -v "^ *\"launches\|^  }\|^  \"" $out/kstamps.log | head -30; python3 -c "
import json; d=json.load(open('$out/kstamps_128.json'))
for k,v in d['families'].items(): print(k, v)
"
