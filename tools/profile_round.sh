#!/bin/bash
# usage (on the GPU box): bash tools/profile_round.sh <tag>   -> gpurun_out/profile_<tag>/
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.log 2>&1
for W in fbank ffnpair; do
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${W}_p$i -o c -- python3 $R/tools/prof_target.py $W 5 > $OUT/${W}_p$i.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections
out="$OUT"
txt=open(glob.glob(out+"/bench/*kernel_stats.csv")[0]).read()
open(out+"/bench_kernel_stats.csv","w").write(txt)
lines=txt.splitlines()
print("\n".join(l[:170] for l in lines[:12]))
with open(out+"/pmc_summary.txt","w") as fh:
    for w in ("fbank","ffnpair"):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for f in sorted(glob.glob(out+"/%s_p*/*counter_collection.csv"%w)):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items():
            if not ("feat512" in k or "ffn_packed" in k or "topdb" in k): continue
            fh.write("[%s] %s\n"%(w,k)); print("[%s] %s"%(w,k))
            for c,vals in sorted(v.items()):
                line="  %-30s mean per launch %.5g  (n=%d)"%(c,sum(vals)/len(vals),len(vals))
                fh.write(line+"\n"); print(line)
PY
tail -1 $OUT/bench.log | cut -c1-400
