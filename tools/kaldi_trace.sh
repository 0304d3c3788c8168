#!/bin/bash
# usage (GPU box): bash tools/kaldi_trace.sh -> per-kernel durations of the Kaldi fbank path (64 x 10 s) under rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/kaldi_tr; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o kt -- python3 $R/tools/feat_bench.py > $OUT/log 2>&1
python3 - <<'PY'
import csv, glob, os
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/kaldi_tr/trace/*kernel_stats.csv')+glob.glob(R+'/gpurun_out/kaldi_tr/trace/*/*kernel_stats.csv')
for r in csv.DictReader(open(f[0])):
    if any(k in r['Name'] for k in ('kaldi','feat512','topdb')): print('%-90s calls %5s avg %9.1f us'%(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
tail -1 $OUT/log
rm -rf $OUT/trace
