#!/bin/bash
# iteration pass: GPU tests, 128-token bench line, in-kernel timelines (dev build)
out=gpurun_out/${1:-iter}; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
timeout 600 python bench.py --no-other-configs > $out/bench128.json 2> $out/bench128.err; echo "bench128 rc=$?"
python3 - <<PY
import json
try:
    d=json.load(open("$out/bench128.json"))
    print("tok/s", d["value"], "parity", d.get("parity"), "fs", d.get("forward_surface",{}).get("value"), "frac", d["roofline"]["frac"])
    print([(k["kernel"],k["avg_us"]) for k in d["roofline"]["per_kernel"]])
except Exception as e: print("bench parse failed", e)
PY
Q3_STAMPS=1 Q3_STRICT=1 Q3_HIP_LIB=qwen3-rs_amd/libqwen3_hip_dev.so Q3_NTOK=32 timeout 300 python tools/gen_loop.py > $out/stamps.log 2>&1
tail -9 $out/stamps.log
for w in 2 3; do echo "Q3_WG_PER_CU_SMALL=$w"; Q3_WG_PER_CU_SMALL=$w Q3_STRICT=1 Q3_NTOK=128 timeout 300 python tools/gen_loop.py 2>&1 | tail -1; done
echo "default, 128 tok"; Q3_STRICT=1 Q3_NTOK=128 timeout 300 python tools/gen_loop.py 2>&1 | tail -1
timeout 600 python bench.py --shape qwen3-8b --steps 32 --warmup 4 --no-cpu-baseline --no-other-configs > $out/bench8b.json 2> $out/bench8b.err; echo "bench8b rc=$?"
python3 - <<PY
import json
try:
    d=json.load(open("$out/bench8b.json"))
    print("8B tok/s", d["value"], "frac", d["roofline"]["frac"], [(k["kernel"],k["avg_us"],k.get("achieved_GBps")) for k in d["roofline"]["per_kernel"]])
except Exception as e: print("bench8b parse failed", e)
PY
