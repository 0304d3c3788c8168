cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_train_step_gpu.py tests/test_train_kernels_gpu.py -m gpu -q -x -k "hybrid or direct or launch_table or full_direct" > gpurun_out/g5_pytest.log 2>&1; tail -4 gpurun_out/g5_pytest.log
for k in 1 2; do python tools/train_bench.py --steps 20 --warmup 5 --ctc-weight 0.3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hybrid ms', d['ms_per_step'], 'host', d.get('host_enqueue_ms_fwd_bwd'))"; done
python tools/train_bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ctc ms', d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/train_prof_hyb6; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tb -- python3 $R/tools/train_bench.py --steps 5 --warmup 2 --ctc-weight 0.3 > $OUT/train_prof.log 2>&1
python3 $R/tools/train_census.py $OUT/trace $OUT/census.txt | head -30 | cut -c1-150
rm -rf $OUT/trace
