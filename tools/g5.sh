cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_kernels_gpu.py -m gpu -q -x -k "direct_weight or conv2_weight" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/train_prof_d8; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tb -- python3 $R/tools/train_bench.py --steps 5 --warmup 2 > $OUT/train_prof.log 2>&1
python3 $R/tools/train_census.py $OUT/trace $OUT/census.txt | grep -E "census|gemm_tn8"
rm -rf $OUT/trace
