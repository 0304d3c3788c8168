#!/bin/bash
# usage (GPU box): bash tools/loader_trace.sh -> per-kernel time of the data loader's device work (tools/loader_bench.py under rocprofv3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/loader_tr; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o lt -- python3 $R/tools/loader_bench.py > $OUT/log 2>&1
python3 - <<'PY'
import csv, glob, os
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/loader_tr/trace/*kernel_stats.csv')+glob.glob(R+'/gpurun_out/loader_tr/trace/*/*kernel_stats.csv')
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]: print('%-80s calls %5s avg %9.1f us  %5.1f %%'%(r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
print('total kernel time %.1f ms over the run (24 batches)'%(tot/1e6))
PY
tail -1 $OUT/log
rm -rf $OUT/trace
