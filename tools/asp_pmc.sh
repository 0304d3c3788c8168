#!/bin/bash
# usage (GPU box): bash tools/asp_pmc.sh <tag>: SQ counters of ma_asp_fused_bf16 at C = 1536 (tools/asp_bench.py), one pass per counter set
TAG=$1; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/asp_pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS" \
         "SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_LDS_ADDR_CONFLICT" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -o c -- python3 $R/tools/asp_bench.py 1536 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "asp_fused" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c,v in sorted(agg.items()):
    print("  %-32s mean per launch %.5g (n=%d)"%(c,sum(v)/len(v),len(v)))
PY
