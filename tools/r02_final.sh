#!/bin/bash
# end-of-round evidence pass: driver bench line, config 3/4 benches, in-kernel timelines, rocprofv3 stats + PMC
out=gpurun_out/${1:-final}; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err; echo "driver bench rc=$?"
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default bench rc=$?"
timeout 600 python bench.py --shape qwen3-8b --steps 32 --warmup 4 --no-other-configs > $out/bench8b.json 2> $out/bench8b.err; echo "bench8b rc=$?"
timeout 600 python bench.py --shape qwen3-4b --steps 32 --warmup 4 --no-other-configs --no-cpu-baseline > $out/bench4b.json 2> $out/bench4b.err; echo "bench4b rc=$?"
timeout 900 python tools/bench_chat.py > $out/chat.json 2> $out/chat.err; echo "chat rc=$?"
timeout 900 python tools/bench_batch.py --steps 64 > $out/batch.json 2> $out/batch.err; echo "batch rc=$?"
Q3_STAMPS=1 Q3_STRICT=1 Q3_HIP_LIB=qwen3-rs_amd/libqwen3_hip_dev.so Q3_NTOK=32 timeout 300 python tools/gen_loop.py > $out/stamps_0p6b.log 2>&1
Q3_STAMPS=1 Q3_STRICT=1 Q3_SHAPE=qwen3-8b Q3_HIP_LIB=qwen3-rs_amd/libqwen3_hip_dev.so Q3_NTOK=16 timeout 300 python tools/gen_loop.py > $out/stamps_8b.log 2>&1
Q3_STAMPS=1 Q3_HIP_LIB=qwen3-rs_amd/libqwen3_hip_dev.so timeout 300 python3 tools/longctx_prof.py qwen3-4b 2300 8 > $out/stamps_attn_out.log 2>&1
Q3_STAMP_SCORES=1 Q3_STAMPS=1 Q3_PROFILE_FAMILIES=0 Q3_HIP_LIB=qwen3-rs_amd/libqwen3_hip_dev.so timeout 300 python3 tools/longctx_prof.py qwen3-4b 2300 8 > $out/stamps_attn_scores.log 2>&1
timeout 300 python3 tools/longctx_prof.py qwen3-4b 2300 32 > $out/longctx_4b.log 2>&1
bash tools/r02_profile.sh ${1:-final}/prof all > $out/prof.log 2>&1
tail -5 $out/*.log | tail -80
