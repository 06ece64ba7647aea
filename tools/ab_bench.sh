#!/bin/bash
# A/B throughput runs on the GPU box: every cuda-phdslam_amd/libphdslam_<name>.so given (built here with
# `make -C cuda-phdslam_amd/csrc variant NAME=<name> VEXTRA="-D..."`) against the product library, same bench command.
# usage: bash tools/ab_bench.sh <config> <steps> name1 name2 ...
cfg=${1:-3}; steps=${2:-200}; shift 2
mkdir -p gpurun_out
run() {
  PHD_LIB=$2 python bench.py --config $cfg --bare --steps $steps --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s %9.1f steps/s  %8.2f us/step  kernel %8.2f us' % ('$1', d['value'], 1e3*d['ms_per_step'], d['roofline']['kernel_avg_us']))"
}
run product ""
for n in "$@"; do run $n $PWD/cuda-phdslam_amd/libphdslam_$n.so; done
run product-again ""
