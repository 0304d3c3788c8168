# Development: repeat the cfg-4 bit-reproducibility test in FRESH processes (a ~1 % per-process corruption with the opt-in second stream
# only shows this way).  The A/B arms of the round-3 hunt (second stream forced on / off, start step, pre-filled allocator)
# were temporary environment switches in engine.py, removed again: this loop now runs the default engine.
cd $GRAFT_REPO_ROOT
n=${1:-150}
bad=0
for i in $(seq 1 $n); do
  timeout 300 python -m pytest tests/test_cfg4_full_shape_gpu.py::test_bucket_1024_batch_of_40_trains -m gpu -q > gpurun_out/flaky_$i.log 2>&1
  if grep -q failed gpurun_out/flaky_$i.log; then bad=$((bad+1)); echo "run $i FAILED"; grep -n "^E " gpurun_out/flaky_$i.log | head -5 | cut -c1-600; else rm gpurun_out/flaky_$i.log; fi
done
echo "$bad failing processes of $n"
