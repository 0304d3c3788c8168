# usage (GPU box): bash tools/train_census.sh <tag> [train_bench flags]  -> gpurun_out/train_prof_<tag>/census.txt (per-kernel time of the training step)
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/train_prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tb -- python3 $R/tools/train_bench.py --steps 5 --warmup 2 "$@" > $OUT/train_prof.log 2>&1
python3 $R/tools/train_census.py $OUT/trace $OUT/census.txt | head -${HEAD:-30} | cut -c1-150
rm -rf $OUT/trace
