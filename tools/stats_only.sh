#!/bin/bash
# usage (on the GPU box): bash tools/stats_only.sh <tag>  -> per-kernel stats of one bench run in gpurun_out/stats_<tag>.csv
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/stats_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.log 2>&1
cp $OUT/bench/*/*kernel_stats.csv $R/gpurun_out/stats_$1.csv 2>/dev/null || cp $OUT/bench/*kernel_stats.csv $R/gpurun_out/stats_$1.csv
cut -c1-150 $R/gpurun_out/stats_$1.csv | head -14
