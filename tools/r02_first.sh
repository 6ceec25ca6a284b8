#!/bin/bash
# round-2 first GPU pass: tests, the driver's bench command, in-kernel timelines of every kernel family
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log
tail -3 gpurun_out/r02a/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02a/bench20.json 2> gpurun_out/r02a/bench20.err; echo "bench rc=$?"
python bench.py --no-other-configs > gpurun_out/r02a/bench128.json 2> gpurun_out/r02a/bench128.err; echo "bench128 rc=$?"
Q3_STAMPS=1 Q3_STRICT=1 Q3_HIP_LIB=qwen3-rs_amd/libqwen3_hip_dev.so Q3_NTOK=32 python tools/gen_loop.py > gpurun_out/r02a/stamps.log 2>&1
tail -20 gpurun_out/r02a/stamps.log
nproc; free -g | head -2; df -h /tmp | tail -1
