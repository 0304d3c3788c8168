#!/bin/bash
# usage (GPU box): bash tools/step_timeline.sh [--ctc-weight 0.3]  -> gpurun_out/step_timeline.txt: every launch of ONE training step
# (kernel trace under rocprofv3: start offset, duration, name) + per-kernel sums of the decoder span for the hybrid step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/step_tl; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o tb -- python3 $R/tools/train_bench.py --steps 5 --warmup 2 "$@" > $OUT/log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/step_tl/trace/*kernel_trace.csv')+glob.glob(R+'/gpurun_out/step_tl/trace/*/*kernel_trace.csv')
rows=sorted(csv.DictReader(open(f[0])), key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if 'subsample_conv1' in n]
s,e=idx[-2],idx[-1]
t0=int(rows[s]['Start_Timestamp'])
out=['step: %d launches, %.1f us'%(e-s,(int(rows[e]['Start_Timestamp'])-t0)/1e3)]
for r in rows[s:e]:
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    out.append('%8.1f dur %6.1f  %s'%((st-t0)/1e3,(en-st)/1e3,r['Kernel_Name'][:90]))
open(R+'/gpurun_out/step_timeline.txt','w').write('\n'.join(out)+'\n')
d0=[i for i in range(s,e) if 'embed_fwd_kernel' in names[i]]
if d0:
    a=d0[0]; b=[i for i in range(a,e) if 'embed_bwd_kernel' in names[i]][0]
    agg=collections.defaultdict(lambda:[0,0])
    for r in rows[a:b+1]:
        k=r['Kernel_Name'][:60]; agg[k][0]+=1; agg[k][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    print('decoder span: %d launches, %.1f us'%(b-a+1,(int(rows[b]['End_Timestamp'])-int(rows[a]['Start_Timestamp']))/1e3))
    for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]): print('  %3d x %6.1f us = %7.1f  %s'%(v[0],v[1]/v[0]/1e3,v[1]/1e3,k))
print(out[0])
PY
rm -rf $OUT/trace
