#!/bin/bash
# quick GPU check of a kernel change: the parity tests that exercise the merge, the bench at configs 3 / 2 / 5, phase stamps.
# Everything goes to gpurun_out/quick_<tag>.log (merged back by gpurun); the summary lines are echoed.
# usage: bash tools/quick_check.sh [full|fast] [tag]
mode=${1:-fast}; tag=${2:-q}
mkdir -p gpurun_out
log=gpurun_out/quick_$tag.log
{
if [ "$mode" = "full" ]; then
  python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -15
else
  python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_cphd.py -m gpu -q -x --timeout 900 2>&1 | tail -15
fi
for cfg in 3 2 5; do
  st=200; [ $cfg = 2 ] && st=2000
  python bench.py --config $cfg --bare --steps $st --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config $cfg: %9.1f steps/s  %8.2f us/step' % (d['value'], 1e3*d['ms_per_step']))"
done
python tools/phase_profile.py 3 2 2>&1 | grep -v amdgpu.ids
} > $log 2>&1
grep -E "passed|failed|error|^config [0-9]:|per-workgroup" $log
