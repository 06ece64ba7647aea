#!/bin/bash
# the bundled simulation with filter_type = 1 (CPHD) through phdslam: pipelined loop against the step-synchronous one (loop time per step,
# instantiation histogram, a log file's hash)   usage (repo root on the GPU box): bash tools/cphd_e2e.sh
# CPHD drop-in: the bundled run with filter_type = 1, pipelined vs synchronous loop
D=$(mktemp -d /tmp/phd_cphd_XXXX)
python3 tools/e2e_run.py 4096 $D > /dev/null 2>&1
sed -i 's/^filter_type *=.*/filter_type = 1/' $D/config.cfg
B=$GRAFT_REPO_ROOT/cuda-phdslam_amd/bin/phdslam
for mode in pipelined sync; do
  rm -rf $D/o; mkdir -p $D/o
  if [ $mode = sync ]; then export PHD_DRIVER_SYNC=1; else unset PHD_DRIVER_SYNC; fi
  $B $D/config.cfg synth --out $D/o --seed 7 --capacity 512 > $D/stdout_$mode.txt 2>&1
  python3 -c "
import numpy as np
t=np.loadtxt('$D/o/loopTime.log'); print('CPHD phdslam, 4096 particles, $mode loop: median %.3f ms, p90 %.3f ms, mean %.3f over %d steps' % (np.median(t), np.percentile(t,90), t.mean(), len(t)))"
  grep -o "inst=[0-9-]*" $D/stdout_$mode.txt | sort | uniq -c | head -3
  md5sum $D/o/state_estimate00100.log | cut -c1-32
done
