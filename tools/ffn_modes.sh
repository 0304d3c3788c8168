#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ffn_modes; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $R/tools/ffn_modes.py > $OUT/log 2>&1
python3 - <<PY
import csv, glob
rows=[r for r in csv.DictReader(open(glob.glob("$OUT/*kernel_trace.csv")[0])) if "ffn_packed_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
names=["mode0","LN bf16","LN f32","LN2 bf16"]
for rep in range(2):
    for k in range(4):
        seg=d[rep*80+k*20: rep*80+k*20+20]
        print(rep, names[k], "mean %.1f min %.1f max %.1f us"%(sum(seg)/len(seg), min(seg), max(seg)))
PY
