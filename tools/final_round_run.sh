cd $GRAFT_REPO_ROOT
TAG=${1:-r5}
bash tools/profile_round.sh ${TAG:=r5} > gpurun_out/profile_$TAG.log 2>&1; tail -25 gpurun_out/profile_$TAG.log | cut -c1-220
bash tools/ecapa_profile.sh $TAG > gpurun_out/ecapa_$TAG.log 2>&1; tail -30 gpurun_out/ecapa_$TAG.log | cut -c1-200
bash tools/run_train_prof.sh $TAG > gpurun_out/train_prof_$TAG.log 2>&1; head -8 gpurun_out/train_prof_$TAG/census.txt | cut -c1-160
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/train_prof_${TAG}hyb; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tb -- python3 $R/tools/train_bench.py --steps 5 --warmup 2 --ctc-weight 0.3 > $OUT/train_prof.log 2>&1
python3 $R/tools/train_census.py $OUT/trace $OUT/census.txt | head -45 | cut -c1-150
rm -rf $OUT/trace
cd $R
python tools/block_table_ab.py > gpurun_out/block_table_ab.json 2> gpurun_out/block_table_ab.err; cat gpurun_out/block_table_ab.json
python bench.py > gpurun_out/bench_${TAG}_final.json 2> gpurun_out/bench_${TAG}_final.err; tail -c 1500 gpurun_out/bench_${TAG}_final.json
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_final.log 2>&1; tail -4 gpurun_out/pytest_gpu_final.log
