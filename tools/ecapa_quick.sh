#!/bin/bash
# usage: bash tools/ecapa_quick.sh <tag>: tests + per-kernel stats of both cfg-5 sizes
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ecapaq_$TAG; mkdir -p $OUT
cd $R && timeout 900 python -m pytest tests/test_ecapa_gpu.py -x -q 2>&1 | tail -5 > $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
for C in 512 1024; do
  python3 $R/tools/ecapa_bench.py $C > $OUT/plain$C.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$C -o s -- python3 $R/tools/ecapa_bench.py $C > $OUT/bench$C.log 2>&1
done
python3 - <<PY
import csv, glob
out="$OUT"
print(open(out+"/tests.log").read())
for C in (512,1024):
    print(open(out+"/plain%d.log"%C).read().strip()[-200:])
    for r in list(csv.DictReader(open(glob.glob(out+"/stats%d/*kernel_stats.csv"%C)[0])))[:14]:
        print("   %-64s %5s %9.1f us %6s %%"%(r["Name"][:64],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"][:5]))
PY
