# usage (GPU box): bash tools/run_train_prof.sh [tag]  -> gpurun_out/train_prof_<tag>/{train_bench.json, census.txt, *kernel_stats.csv}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}
OUT=$R/gpurun_out/train_prof_$TAG
mkdir -p $OUT
python3 $R/tools/train_bench.py --steps 10 --warmup 3 > $OUT/train_bench.json 2> $OUT/train_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tb -- python3 $R/tools/train_bench.py --steps 5 --warmup 2 > $OUT/train_prof.log 2>&1
python3 $R/tools/train_census.py $OUT/trace $OUT/census.txt | head -70
cp $(ls $OUT/trace/*/*kernel_stats.csv $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1) $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace/*/*kernel_trace.csv 2>/dev/null  # (large; the census and the stats are what is kept)
cat $OUT/train_bench.json
