#!/bin/bash
# repeat the multi-process / multi-shard GPU tests (sharded driver over one-rank RCCL, two ranks sharing the GPU over gloo,
# the C++ multi-device host) to catch rare hangs: bash tools/gpu_stress.sh [repeats]
n=${1:-6}
mkdir -p gpurun_out
for i in $(seq 1 $n); do
  timeout 300 python -m pytest tests/test_gpu_cphd.py tests/test_gpu_dist.py tests/test_gpu_driver.py tests/test_gpu_multi.py \
    tests/test_gpu_rccl_one_rank.py -m gpu -q --timeout 60 -x > gpurun_out/stress_$i.log 2>&1
  grep -E "passed|failed" gpurun_out/stress_$i.log
  if grep -q failed gpurun_out/stress_$i.log; then tail -60 gpurun_out/stress_$i.log; break; fi
done
