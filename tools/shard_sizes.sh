#!/bin/bash
# What ONE GPU can tell about eight: the per-shard step of BASELINE.json configs[3] (16384 x 256 x 64 sharded
# 2 / 4 / 8 ways = 8192 / 4096 / 2048 particles per GPU) measured on one device: the fused single-filter step, the one-rank RCCL
# C++ host (local step + all-gather + weights + pull, every phase of the sharded step except the links), both builds of the
# update kernel (three / two workgroups per CU).   usage (repo root on the GPU box): bash tools/shard_sizes.sh <tag>
tag=${1:-r06}
O=$GRAFT_REPO_ROOT/gpurun_out/shard_sizes_$tag.txt
: > $O
for n in 2048 4096 8192 16384; do
  for build in 3 2; do
    v=$(PHD_UPDATE_BUILD=$build PHD_BENCH_RECORD=/tmp/shard_rec.json python3 bench.py --config 4 --particles $n --steps 300 --warmup 30 --bare 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f steps/s  %.1f us/step  kernel %.1f us' % (d['value'], 1e3*d['ms_per_step'], d['roofline']['kernel_avg_us']))")
    echo "single filter, fused step       n=$n build=$build-per-CU : $v" | tee -a $O
  done
  v=$(PHD_BENCH_CPP_MULTI=1 PHD_BENCH_RECORD=/tmp/shard_rec.json python3 bench.py --config 4 --particles $n --steps 300 --warmup 30 --bare 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f steps/s  %.1f us/step' % (d['value'], 1e3*d['ms_per_step']))")
  echo "one-rank RCCL C++ host (sharded step) n=$n              : $v" | tee -a $O
  python3 - <<PY | tee -a $O
import json
d = json.load(open('/tmp/shard_rec.json'))
ph = d['config'].get('multi_gpu_phase_us_shard0')
if ph: print('   phases (us, shard 0, drained after every step):', {k: round(v, 1) for k, v in ph.items() if isinstance(v, float)})
PY
done
python3 tools/weights_time.py 2048 4096 8192 16384 2>&1 | tail -6 | tee -a $O
