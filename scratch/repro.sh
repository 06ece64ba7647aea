cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os
sys.path.insert(0, "tests")
import importlib
t = importlib.import_module("test_gpu_driver")
os.makedirs("/tmp/ds", exist_ok=True)
print(t.write_dataset("/tmp/ds", n_steps=6, n_particles=48))
PY
mkdir -p /tmp/ds/o1
timeout -s KILL 60 cuda-phdslam_amd/bin/phdslam /tmp/ds/config.cfg synth --out /tmp/ds/o1 --seed 9 --capacity 256 --devices 1 > gpurun_out/repro_out.txt 2>&1
echo "rc=$?" >> gpurun_out/repro_out.txt
tail -20 gpurun_out/repro_out.txt
