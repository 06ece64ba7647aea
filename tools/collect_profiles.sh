#!/bin/bash
# copy the summaries of one tools/gpu_round.sh visit from gpurun_out/ (scratch) into profiles/ (tracked): bash tools/collect_profiles.sh <tag> [prefix]
tag=${1:?tag}; pre=${2:-r06}
g=gpurun_out; p=profiles
cp $g/pmc_traffic_cfg2.json $g/pmc_traffic_cfg3.json $g/pmc_traffic_cfg5.json $g/pmc_sq_cfg2.json $g/pmc_sq_cfg3.json $g/pmc_sq_cfg5.json $p/
cp $g/gpu_tests_$tag.log $p/${pre}_gpu_tests.log
cp $g/bench_driver_$tag.json $p/${pre}_bench_driver_invocation_line.json      # the compact line the driver parses
cp $g/bench_driver_record_$tag.json $p/${pre}_bench_driver_invocation.json   # the full record of the same run (profiles/bench_last.json on the box)
cp $g/bench_driver_record_$tag.json $p/bench_last.json
cp $g/bench_cfg3_long_record_$tag.json $p/${pre}_bench_cfg3_long.json
cp $g/phase_$tag.log $p/${pre}_phase_stamps.txt
for c in 2 3 5; do n=$c; [ $c = 5 ] && n=5_cphd; cp $g/kernel_stats_cfg${c}_$tag.csv $p/${pre}_rocprofv3_kernel_stats_cfg$n.csv; done
cp $g/sq_counters_$tag.txt $p/${pre}_sq_counters.txt
cp $g/bench_share2_$tag.json $p/${pre}_bench_share_gpu_2ranks_cfg4.json
cp $g/e2e_$tag.log $p/${pre}_e2e_driver.txt
for ex in gathered alltoall; do
  cp $g/bench_cfg2_onerank_${ex}_$tag.json $p/${pre}_bench_cfg2_onerank_rccl_python_host_$ex.json
  cp $g/bench_cfg2_cpp_multi_onerank_${ex}_$tag.json $p/${pre}_bench_cfg2_onerank_rccl_cpp_host_$ex.json
done
cp $g/bench_gpus2_$tag.json $p/${pre}_bench_gpus2_unlaunched_share_gpu.json
for ex in pull alltoall; do cp $g/bench_cfg4_cpp_multi_onerank_${ex}_$tag.json $p/${pre}_bench_cfg4_onerank_rccl_cpp_host_$ex.json; done
cp $g/shard_sizes_$tag.txt $p/${pre}_shard_sizes.txt
ls $p | wc -l
