#!/bin/bash
# one GPU-box visit: tests, smoke, bench (configs 2 and 3), rocprofv3 kernel trace of the bench command,
# PMC traffic passes.  usage (repo root on the GPU box): bash tools/gpu_round.sh <tag>
tag=${1:-r01}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 600 --durations=5 > gpurun_out/gpu_tests_$tag.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gpu_tests_$tag.log
tail -8 gpurun_out/gpu_tests_$tag.log
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py > gpurun_out/bench_cfg2_$tag.json 2> gpurun_out/bench_cfg2_$tag.err
cat gpurun_out/bench_cfg2_$tag.json
python bench.py --config 3 --steps 100 --warmup 10 --cpu-seconds 10 > gpurun_out/bench_cfg3_$tag.json 2> gpurun_out/bench_cfg3_$tag.err
cat gpurun_out/bench_cfg3_$tag.json
python bench.py --config 5 --steps 100 --warmup 10 --cpu-seconds 10 > gpurun_out/bench_cfg5_$tag.json 2> gpurun_out/bench_cfg5_$tag.err
cat gpurun_out/bench_cfg5_$tag.json
# the N > 1 step on a one-rank RCCL group (what the multi-rank path costs before any link): both exchange forms
for ex in gathered alltoall; do
  PHD_BENCH_EXCHANGE=$ex PHD_BENCH_ONE_RANK_RCCL=1 python bench.py --steps 400 --warmup 40 --cpu-seconds 0 2> /dev/null | grep metric > gpurun_out/bench_cfg2_onerank_${ex}_$tag.json
  cat gpurun_out/bench_cfg2_onerank_${ex}_$tag.json | cut -c1-160
done
python tools/phase_profile.py 2 3 > gpurun_out/phase_$tag.log 2>&1
(python tools/e2e_run.py 256; python tools/e2e_run.py 4096) 2>&1 | grep -v amdgpu.ids > gpurun_out/e2e_$tag.log
cat gpurun_out/e2e_$tag.log
cd /tmp && export TMPDIR=/tmp
for cfg in 2 3 5; do
  st=2000; wu=200; [ $cfg != 2 ] && st=50 && wu=40
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cfg${cfg}_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps $st --warmup $wu --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_cfg${cfg}_$tag.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_cfg${cfg}_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_cfg${cfg}_$tag.csv && head -5 $f
done
cd $GRAFT_REPO_ROOT
bash tools/pmc_traffic.sh 2 $tag
bash tools/pmc_traffic.sh 3 $tag
bash tools/pmc_sq.sh 3 $tag > gpurun_out/sq_counters_$tag.txt 2>&1
bash tools/pmc_sq.sh 2 $tag >> gpurun_out/sq_counters_$tag.txt 2>&1
