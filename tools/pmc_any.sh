#!/bin/bash
# usage (GPU box): bash tools/pmc_any.sh <tag> <prof_target workload> <kernel name substring>
set -u
TAG=$1; W=$2; KN=$3
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/tools/prof_target.py $W 20 > $OUT/stats.log 2>&1
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_IFETCH" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -o c -- python3 $R/tools/prof_target.py $W 5 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
out="$OUT"; kn="$KN"
for f in glob.glob(out+"/stats/*kernel_stats.csv"):
    for l in open(f).read().splitlines()[:6]: print(l[:200])
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out+"/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if kn in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out+"/summary.txt","w") as fh:
    for k,v in agg.items():
        fh.write(k+"\n"); print(k)
        for c,vals in sorted(v.items()):
            line="  %-30s mean per launch %.5g  (n=%d)"%(c,sum(vals)/len(vals),len(vals))
            fh.write(line+"\n"); print(line)
PY
