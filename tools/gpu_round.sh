#!/bin/bash
# one GPU-box visit: tests, smoke, the driver's bench invocation + a long run, phase stamps, rocprofv3 kernel trace of the
# bench command, PMC passes (HBM traffic, SQ issue counters), the N > 1 dry runs.
# usage (repo root on the GPU box): bash tools/gpu_round.sh <tag> [quick]
tag=${1:-r02}; quick=$2
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 900 --durations=8 -s > gpurun_out/gpu_tests_$tag.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gpu_tests_$tag.log
# the multi-device tests' bodies, every shard on device 0 (a dry run of the test file on a one-GPU box, not a measurement)
echo "== PHD_TEST_SHARE_DEVICE=1 python -m pytest tests/test_gpu_multi_devices.py -m gpu -q" >> gpurun_out/gpu_tests_$tag.log
PHD_TEST_SHARE_DEVICE=1 python -m pytest tests/test_gpu_multi_devices.py -m gpu -q --timeout 900 2>&1 | tail -2 >> gpurun_out/gpu_tests_$tag.log
grep -E "sampled particles|passed|failed|error" gpurun_out/gpu_tests_$tag.log | tail -12
python __graft_entry__.py smoke 2>&1 | tail -1
# the PMC passes first: bench.py prints `roofline.traffic` / `roofline_valu` only from counters recorded with THIS build
# (tools/collect_profiles.sh copies the same files into profiles/ afterwards)
if [ -z "$quick" ]; then
  bash tools/pmc_traffic.sh 3 $tag
  bash tools/pmc_traffic.sh 2 $tag
  bash tools/pmc_traffic.sh 5 $tag
  bash tools/pmc_sq.sh 3 $tag > gpurun_out/sq_counters_$tag.txt 2>&1
  bash tools/pmc_sq.sh 2 $tag >> gpurun_out/sq_counters_$tag.txt 2>&1
  bash tools/pmc_sq.sh 5 $tag >> gpurun_out/sq_counters_$tag.txt 2>&1
  tail -3 gpurun_out/sq_counters_$tag.txt | cut -c1-600
  cp gpurun_out/pmc_traffic_cfg2.json gpurun_out/pmc_traffic_cfg3.json gpurun_out/pmc_traffic_cfg5.json gpurun_out/pmc_sq_cfg2.json gpurun_out/pmc_sq_cfg3.json gpurun_out/pmc_sq_cfg5.json profiles/
fi
# exactly what the driver runs
# (the LAST stdout line is the compact record the driver parses; the full record — riders, notes, counters — is the file named by PHD_BENCH_RECORD)
PHD_BENCH_RECORD=$PWD/gpurun_out/bench_driver_record_$tag.json python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_$tag.json 2> gpurun_out/bench_driver_$tag.err
python - <<PY
import json
last = open("gpurun_out/bench_driver_$tag.json").read().strip().splitlines()[-1]
d = json.loads(last)
print("driver-style: %.1f steps/s (%.4f ms), general instantiation %.1f | line %d bytes | cpu %s | riders %s" % (
    d["value"], d["ms_per_step"], d["value_general"] or 0, len(last), (d["cpu_baseline"] or {}).get("value"), d["riders_steps_per_s"]))
PY
# a long run of the headline for comparison (the 3 % criterion)
PHD_BENCH_RECORD=$PWD/gpurun_out/bench_cfg3_long_record_$tag.json python bench.py --steps 400 --warmup 40 --no-secondary --cpu-seconds 0 > gpurun_out/bench_cfg3_long_$tag.json 2> gpurun_out/bench_cfg3_long_$tag.err
python -c "import json;d=json.loads(open('gpurun_out/bench_cfg3_long_$tag.json').read().strip().splitlines()[-1]);print('long run: %.1f steps/s (general %.1f), kernel %.2f us' % (d['value'], d['value_general'] or 0, d['roofline']['kernel_avg_us']))"
python tools/phase_profile.py 2 3 5 > gpurun_out/phase_$tag.log 2>&1
[ -n "$quick" ] && exit 0
(python tools/e2e_run.py 256; python tools/e2e_run.py 4096; echo "-- PHD_DRIVER_PROFILE=1 (the pipelined loop: what the HOST does per iteration)"; PHD_DRIVER_PROFILE=1 python tools/e2e_run.py 4096;
 echo "-- PHD_DRIVER_SYNC=1: the step-synchronous loop of rounds 1-5 (run_synth's own structure)"; PHD_DRIVER_SYNC=1 python tools/e2e_run.py 4096;
 echo "-- PHD_DRIVER_SYNC=1 PHD_DRIVER_PROFILE=1 (the update is synchronised for the attribution)"; PHD_DRIVER_SYNC=1 PHD_DRIVER_PROFILE=1 python tools/e2e_run.py 4096) 2>&1 | grep -v amdgpu.ids > gpurun_out/e2e_$tag.log
cat gpurun_out/e2e_$tag.log
# the N > 1 paths on this one GPU: `--gpus 2` with no launcher (the C++ multi-device host, both shards on device 0), two ranks
# sharing device 0 under the launcher (gloo transport) on the configs[3] split, one-rank RCCL
PHD_BENCH_RECORD=$PWD/gpurun_out/bench_gpus2_$tag.json PHD_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 20 --warmup 5 2> gpurun_out/bench_gpus2_$tag.err | tail -1 > gpurun_out/bench_gpus2_line_$tag.json
cut -c1-300 gpurun_out/bench_gpus2_line_$tag.json
for ex in pull alltoall; do
  PHD_BENCH_RECORD=$PWD/gpurun_out/bench_cfg4_cpp_multi_onerank_${ex}_$tag.json PHD_BENCH_EXCHANGE=$ex PHD_BENCH_CPP_MULTI=1 python bench.py --config 4 --steps 100 --warmup 10 2>/dev/null | tail -1 | cut -c1-200
done
PHD_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 5 --warmup 2 --preroll-ms 0 2> gpurun_out/bench_share2_$tag.err | grep metric > gpurun_out/bench_share2_line_$tag.json; cp profiles/bench_last.json gpurun_out/bench_share2_$tag.json
cut -c1-400 gpurun_out/bench_share2_line_$tag.json
for ex in gathered alltoall; do
  PHD_BENCH_RECORD=$PWD/gpurun_out/bench_cfg2_onerank_${ex}_$tag.json PHD_BENCH_EXCHANGE=$ex PHD_BENCH_ONE_RANK_RCCL=1 python bench.py --steps 400 --warmup 40 --cpu-seconds 0 2> /dev/null | grep metric | cut -c1-200
done
# the same N > 1 step driven by the C++ multi-device host (libphdslam_multi.so): one-rank RCCL, both exchange forms
for ex in gathered alltoall; do
  PHD_BENCH_RECORD=$PWD/gpurun_out/bench_cfg2_cpp_multi_onerank_${ex}_$tag.json PHD_BENCH_EXCHANGE=$ex PHD_BENCH_CPP_MULTI=1 python bench.py --config 2 --steps 2000 --warmup 100 2> /dev/null | grep metric | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
for cfg in 3 2 5; do
  st=50; wu=40; [ $cfg = 2 ] && st=2000 && wu=200
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cfg${cfg}_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --bare --steps $st --warmup $wu > $GRAFT_REPO_ROOT/gpurun_out/prof_cfg${cfg}_$tag.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_cfg${cfg}_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_cfg${cfg}_$tag.csv && head -4 $f
done
cd $GRAFT_REPO_ROOT
bash tools/shard_sizes.sh $tag > /dev/null 2>&1
tail -20 gpurun_out/shard_sizes_$tag.txt
