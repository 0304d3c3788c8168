# usage (GPU box): bash tools/rccl_trace.sh [tag] -> gpurun_out/rccl_trace_<tag>.json : the training bench with its 14 gradient buckets
# per step through RCCL at world size 1 (ReduceOp.AVG: a real RCCL kernel per bucket) under rocprofv3 --kernel-trace; how many RCCL
# kernels ran, and how many of them concurrently with one of the library's kernels on another hardware queue.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}
D=/tmp/rccl_trace_$TAG
rm -rf $D
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((20000 + RANDOM % 20000))
EXTRA=${2:-}
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --gpus 1 --train --steps 3 --warmup 1 --no-cpu-baseline --force-collective $EXTRA > $R/gpurun_out/rccl_trace_$TAG.log 2>&1
python3 $R/tools/rccl_trace_check.py $D --timeline > $R/gpurun_out/rccl_trace_$TAG.json
cat $R/gpurun_out/rccl_trace_$TAG.json | cut -c1-1800
