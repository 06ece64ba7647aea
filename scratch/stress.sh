cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
  timeout 200 python -m pytest tests/test_gpu_cphd.py tests/test_gpu_dist.py tests/test_gpu_driver.py -m gpu -q --timeout 40 -x > gpurun_out/stress_$i.log 2>&1
  grep -E "passed|failed" gpurun_out/stress_$i.log
  grep -q failed gpurun_out/stress_$i.log && break
done
