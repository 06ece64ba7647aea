#!/bin/bash
# Soak of the pipelined driver loop (round 6): N consecutive runs of `phdslam` on the bundled simulation, each under a time-out
# (a lost wake-up between the helper thread, the main loop and the log writers would hang), every run's state logs hashed and
# compared with the first run's (the loop is deterministic: same seed -> same files) and with ONE run of the step-synchronous loop.
# usage (repo root on the GPU box): bash tools/driver_soak.sh [runs = 40] [particles = 256]  -> gpurun_out/driver_soak.txt
runs=${1:-40}; np=${2:-256}
O=$GRAFT_REPO_ROOT/gpurun_out/driver_soak.txt
D=$(mktemp -d /tmp/phd_soak_XXXX)
python3 tools/e2e_run.py $np $D > /dev/null 2>&1 || { echo "e2e_run failed" | tee $O; exit 1; }
B=$GRAFT_REPO_ROOT/cuda-phdslam_amd/bin/phdslam
hash_logs() { cat $1/state_estimate*.log | md5sum | cut -d' ' -f1; }
mkdir -p $D/sync; PHD_DRIVER_SYNC=1 timeout 120 $B $D/config.cfg synth --out $D/sync --seed 7 --capacity 512 > /dev/null 2>&1
ref=$(hash_logs $D/sync)
ok=0; bad=0; hung=0
for i in $(seq 1 $runs); do
  rm -rf $D/run; mkdir -p $D/run
  timeout 120 $B $D/config.cfg synth --out $D/run --seed 7 --capacity 512 > /dev/null 2>&1
  rc=$?
  if [ $rc = 124 ]; then hung=$((hung+1)); continue; fi
  h=$(hash_logs $D/run)
  if [ "$h" = "$ref" ] && [ $rc = 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); fi
done
echo "driver soak: $runs runs of the pipelined loop, $np particles, 331 steps each: $ok identical to the synchronous loop's logs (md5 $ref), $bad different or failed, $hung hung (120 s time-out)" | tee $O
rm -rf $D
