cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/g4_pytest_full.log 2>&1; tail -25 gpurun_out/g4_pytest_full.log
