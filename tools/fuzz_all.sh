#!/bin/bash
# every randomised sweep, one after the other, summary lines into gpurun_out/fuzz_<tag>.txt   usage: bash tools/fuzz_all.sh <tag> [seconds each]
tag=${1:-r03}; t=${2:-90}
mkdir -p gpurun_out
out=gpurun_out/fuzz_$tag.txt
: > $out
run() { echo "== $*" >> $out; "$@" 2>&1 | grep -v amdgpu.ids | tail -6 >> $out; }
run python tools/fuzz_parity.py $t 1000
PHD_FUZZ_SPILL=1 run python tools/fuzz_parity.py $t 5000
run python tools/fuzz_fused.py $t 1
run python tools/fuzz_multi.py $t 1
run python tools/fuzz_resample.py 40 1
run python tools/fuzz_gm_reduce.py 40 1
run python tools/fuzz_cphd.py $t 1
run python tools/fuzz_layout.py $t 1
run python tools/determinism_check.py
cat $out
