#!/bin/bash
# usage (on the GPU box): bash tools/round_check.sh <tag> [pytest-args]  -> gpurun_out/check_<tag>/{pytest.log,bench.log,trace/}
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/check_$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q "$@" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.log 2> $OUT/bench.err
echo "bench rc=$?"; tail -c 6000 $OUT/bench.log; tail -5 $OUT/bench.err
