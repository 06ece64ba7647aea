cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
which gdb rocgdb 2>&1 | head -2
(timeout 400 python -m pytest tests -m gpu -q --timeout 150 -x > gpurun_out/hang_pytest.log 2>&1) &
for i in $(seq 1 120); do
  sleep 2
  pid=$(pgrep -x phdslam | head -1)
  if [ -n "$pid" ]; then
    et=$(ps -o etimes= -p $pid | tr -d ' ')
    if [ -n "$et" ] && [ "$et" -gt 25 ]; then
      echo "hung pid $pid after $et s" > gpurun_out/hang_bt.txt
      cat /proc/$pid/cmdline | tr '\0' ' ' >> gpurun_out/hang_bt.txt
      (gdb -p $pid -batch -ex "thread apply all bt" 2>&1 || rocgdb -p $pid -batch -ex "thread apply all bt" 2>&1) | tail -150 >> gpurun_out/hang_bt.txt
      kill -9 $pid
      break
    fi
  fi
done
wait
tail -5 gpurun_out/hang_pytest.log
head -120 gpurun_out/hang_bt.txt
