#!/bin/bash
# CPHD variant after a kernel change: its tests, the bench at configs[4], the block's phase stamps, the fuzz (60 s)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_cphd.py -m gpu -q -x --timeout 600 2>&1 | tail -3
python bench.py --config 5 --bare --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 5: %.1f steps/s %.2f us' % (d['value'], 1e3*d['ms_per_step']))"
python tools/phase_profile.py 5 2>&1 | grep -E "per-workgroup|CPHD block|pass1 normalisers"
python tools/fuzz_cphd.py 2>&1 | tail -1
