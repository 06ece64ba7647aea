#!/bin/bash
# A/B on scans of real lengths (bench.py --meas): the product library against a variant (cuda-phdslam_amd/libphdslam_<name>.so)
# usage: bash tools/ab_meas.sh <variant name> "<cfg list>" "<meas list>"
v=${1:?variant}; cfgs=${2:-3}; meas=${3:-"27 44 61 64"}
run() { PHD_LIB=$3 python bench.py --config $1 --meas $2 --bare --steps 300 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg $1 meas %3d %-10s %9.1f steps/s  kernel %8.2f us  inst %s' % ($2, '$4', d['value'], d['roofline']['kernel_avg_us'], d['config']['instantiation']))"; }
for c in $cfgs; do for m in $meas; do
  run $c $m "" product; run $c $m $PWD/cuda-phdslam_amd/libphdslam_$v.so $v; run $c $m "" product
done; done
