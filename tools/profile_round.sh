#!/bin/bash
# usage (on the GPU box): bash tools/profile_round.sh <tag>   -> gpurun_out/profile_<tag>/
#   bench kernel stats (rocprofv3 --kernel-trace --stats), separate --pmc passes for the two roofline kernels, kernel-trace
#   durations of the roofline kernels launched alone, and the blit census of one step.
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (--step-only: the warm-up + timed steps and nothing else, so that the per-kernel stats are the step's launches - VERDICT r3 hygiene)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --step-only > $OUT/bench.log 2>&1
for W in fbank ffnpair; do
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${W}_p$i -o c -- python3 $R/tools/prof_target.py $W 5 > $OUT/${W}_p$i.log 2>&1
  done
  rocprofv3 --kernel-trace --output-format csv -d $OUT/${W}_trace -o t -- python3 $R/tools/prof_target.py $W 30 > $OUT/${W}_trace.log 2>&1
done
python3 - <<PY
import csv, glob, collections
out="$OUT"
# per-kernel stats of the TIMED steps only (from the first launch of timed step 1 = the 4th fbank launch, 3 warm-up steps, to the end
# of the trace): no set-up blits, no warm-up, no roofline loops
allrows=sorted(csv.DictReader(open(glob.glob(out+"/bench/*kernel_trace.csv")[0])), key=lambda r:int(r["Start_Timestamp"]))
fb0=[i for i,r in enumerate(allrows) if "feat512_kernel" in r["Kernel_Name"]]
timed=allrows[fb0[3]:]
agg=collections.defaultdict(lambda:[0,0])
for r in timed:
    a=agg[r["Kernel_Name"]]; a[0]+=1; a[1]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
tot=sum(v[1] for v in agg.values())
nsteps=len(fb0)-3
wall=(int(timed[-1]["End_Timestamp"])-int(timed[0]["Start_Timestamp"]))
lines=['"Name","Calls","TotalDurationNs","AverageNs","Percentage"']
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    lines.append('"%s",%d,%d,%.1f,%.2f'%(k.replace('"',"'"),v[0],v[1],v[1]/v[0],100.0*v[1]/tot))
lines.append('"# %d timed steps: %d launches per step, sum of kernel time %.3f ms per step, wall %.3f ms per step (first launch to last end)",,,,'%(nsteps,len(timed)//nsteps,tot/nsteps/1e6,wall/nsteps/1e6))
txt="\n".join(lines)+"\n"
open(out+"/bench_kernel_stats.csv","w").write(txt)
print("\n".join(l[:170] for l in txt.splitlines()[:14])); print(lines[-1])
# blit census: __amd_rocclr_copyBuffer launches between consecutive fbank launches of the timed steps
rows=sorted(csv.DictReader(open(glob.glob(out+"/bench/*kernel_trace.csv")[0])), key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
fb=[i for i,n in enumerate(names) if "feat512_kernel" in n]
per=[sum(1 for n in names[a:b] if "copyBuffer" in n) for a,b in zip(fb[:-1],fb[1:])]
total=sum(1 for n in names if "copyBuffer" in n)
with open(out+"/blit_census.txt","w") as fh:
    line="copyBuffer launches: %d in the whole run; between consecutive fbank launches (one step each): %s"%(total, collections.Counter(per).most_common(4))
    fh.write(line+"\n"); print(line)
    before=sum(1 for n in names[:fb[0]] if "copyBuffer" in n) if fb else 0
    line="  before the first step (model .to(device), prepare(), synthetic batch): %d"%before
    fh.write(line+"\n"); print(line)
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
# traffic.json: HBM bytes per launch of the two roofline kernels (FETCH_SIZE doubled: gfx950 tallies 128-B read requests at 64 B,
# MI355X_MICROARCH.md), stamped with the hashes of the kernel sources they were measured on; tools/stamp_traffic.py adds the commit
import hashlib, json
def sha16(rel): return hashlib.sha256(open("$R/"+rel,"rb").read()).hexdigest()[:16]
def mean(kernel_sub, counter):
    vals=[v for k,d in allagg.items() if kernel_sub in k for v in d.get(counter,[])]
    return sum(vals)/len(vals) if vals else None
allagg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out+"/*_p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        allagg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
tr={"_comment":"HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_round.sh), counters are in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B). bench.py reports traffic_source_current = false when the kernel source no longer hashes to source_sha16."}
for key,sub,src,form in (("ffn_packed_kernel","ffn_packed_kernel","mindaudio_amd/csrc/ffn_packed.hip","pair + qkv launch (tools/prof_target.py ffnpair)"),("feat512_kernel","feat512_kernel","mindaudio_amd/csrc/features.hip","fbank, 64 x 10 s (tools/prof_target.py fbank)")):
    f,w=mean(sub,"FETCH_SIZE"),mean(sub,"WRITE_SIZE")
    if f is None or w is None: continue
    tr[key]={"fetch_kib":round(f),"write_kib":round(w),"bytes":int((2*f+w)*1024),"form":form,"source":src,"source_sha16":sha16(src)}
json.dump(tr,open(out+"/traffic.json","w"),indent=1)
with open(out+"/roofline_kernels_trace.txt","w") as fh:
    for w,kn in (("fbank","feat512_kernel"),("ffnpair","ffn_packed_kernel")):
        d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(glob.glob(out+"/%s_trace/*kernel_trace.csv"%w)[0])) if kn in r["Kernel_Name"]]
        d=d[5:]
        line="%s launched alone (tools/prof_target.py %s): %d launches, kernel-trace duration mean %.1f us, min %.1f, max %.1f"%(kn,w,len(d),sum(d)/len(d),min(d),max(d))
        fh.write(line+"\n"); print(line)
PY
grep '^{"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json; cut -c1-600 $OUT/bench_line.json
