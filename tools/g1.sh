cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conformer_train_script.py tests/test_train_step_gpu.py -m gpu -x -q -k "manifest or launch_table" > gpurun_out/g1_pytest.log 2>&1; tail -5 gpurun_out/g1_pytest.log
python tools/train_bench.py --steps 20 --warmup 5 > gpurun_out/g1_train.json 2>&1; tail -2 gpurun_out/g1_train.json | cut -c1-400
