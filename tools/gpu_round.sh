#!/bin/bash
# one GPU-box visit: tests, bench (configs 2 and 3), rocprofv3 kernel trace of the bench command.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag>
tag=${1:-r01}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 600 --durations=8 > gpurun_out/gpu_tests_$tag.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gpu_tests_$tag.log
tail -15 gpurun_out/gpu_tests_$tag.log
python bench.py --steps 400 --warmup 40 > gpurun_out/bench_cfg2_$tag.json 2> gpurun_out/bench_cfg2_$tag.err
cat gpurun_out/bench_cfg2_$tag.json; tail -3 gpurun_out/bench_cfg2_$tag.err
python bench.py --config 3 --steps 50 --warmup 5 --cpu-seconds 10 > gpurun_out/bench_cfg3_$tag.json 2> gpurun_out/bench_cfg3_$tag.err
cat gpurun_out/bench_cfg3_$tag.json; tail -3 gpurun_out/bench_cfg3_$tag.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 40 --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_$tag -name "*kernel_stats*" | head -3
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/kernel_stats_$tag.csv && head -12 $f
