run() { PHD_LIB=$3 python bench.py --config $1 --meas $2 --bare --steps 400 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg $1 meas %3d %-10s %9.1f steps/s  kernel %8.2f us  inst %s' % ($2, '$4', d['value'], d['roofline']['kernel_avg_us'], d['config']['instantiation']))"; }
for m in 27 44 61 64; do
  run 3 $m "" product; run 3 $m $PWD/cuda-phdslam_amd/libphdslam_raggedr5.so r5form; run 3 $m "" product
done
for m in 27 61; do run 5 $m "" product; run 5 $m $PWD/cuda-phdslam_amd/libphdslam_raggedr5.so r5form; done
for m in 13 27; do run 2 $m "" product; run 2 $m $PWD/cuda-phdslam_amd/libphdslam_raggedr5.so r5form; done
