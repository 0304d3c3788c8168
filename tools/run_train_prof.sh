cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/train_prof
python3 $R/tools/train_bench.py --steps 10 --warmup 3 > $R/gpurun_out/train_bench.json 2> $R/gpurun_out/train_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/train_prof -o tb -- python3 $R/tools/train_bench.py --steps 5 --warmup 2 > $R/gpurun_out/train_prof.log 2>&1
