#!/bin/bash
# usage: tools/pmc_run.sh <tag> <mode>   (run on the GPU box; writes gpurun_out/pmc_<tag>/)
set -u
TAG=$1; MODE=${2:-fbank}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/tools/prof_fbank.py $MODE 50 > $OUT/stats.log 2>&1
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -o c -- python3 $R/tools/prof_fbank.py $MODE 5 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, os
out="$OUT"
for f in sorted(glob.glob(out+"/stats/**/*kernel_stats.csv", recursive=True)):
    print(open(f).read())
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out+"/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out+"/summary.txt","w") as fh:
    for k,v in agg.items():
        if "feat512" not in k and "topdb" not in k: continue
        fh.write(k+"\n"); print(k)
        for c,vals in sorted(v.items()):
            line="  %-28s mean %.4g  (n=%d)"%(c,sum(vals)/len(vals),len(vals))
            fh.write(line+"\n"); print(line)
PY
