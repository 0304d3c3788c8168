# usage (GPU box): bash tools/pmc_train.sh "<counters>" [kernel name substring ...]  -> per-launch means of the counters for the named kernels of
# the cfg-4 training step (tools/train_bench.py, 3 steps under rocprofv3 --pmc; separate run per counter set)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=$1; shift
D=/tmp/pmc_train_$$
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o c -- python3 $R/tools/train_bench.py --steps 3 --warmup 1 > /dev/null 2>&1
python3 - "$D" "$@" <<'PY'
import csv, glob, sys, collections
d=sys.argv[1]; subs=sys.argv[2:]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if subs and not any(s in k for s in subs): continue
        agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()): print("   %-32s mean per launch %.5g (n=%d)"%(c,sum(vals)/len(vals),len(vals)))
PY
rm -rf $D
