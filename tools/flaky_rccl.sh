cd $GRAFT_REPO_ROOT
n=${1:-50}
bad=0
for i in $(seq 1 $n); do
  timeout 300 python -m pytest tests/test_rccl_world1_gpu.py -m gpu -q > gpurun_out/flaky_rccl_$i.log 2>&1
  if grep -q failed gpurun_out/flaky_rccl_$i.log; then bad=$((bad+1)); echo "run $i FAILED"; grep -n "^E " gpurun_out/flaky_rccl_$i.log | head -5 | cut -c1-400; else rm gpurun_out/flaky_rccl_$i.log; fi
done
echo "rccl world-1: $bad failures of $n"
