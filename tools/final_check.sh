# usage (GPU box): bash tools/final_check.sh  -> full GPU suite, smoke, the driver's bench line, the hybrid step's census
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_final3.log 2>&1; tail -3 gpurun_out/pytest_gpu_final3.log
python bench.py > gpurun_out/bench_r4_final3.json 2> gpurun_out/bench_r4_final3.err; tail -c 300 gpurun_out/bench_r4_final3.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/train_prof_r4hyb2; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tb -- python3 $R/tools/train_bench.py --steps 5 --warmup 2 --ctc-weight 0.3 > $OUT/train_prof.log 2>&1
python3 $R/tools/train_census.py $OUT/trace $OUT/census.txt | head -12 | cut -c1-150
rm -rf $OUT/trace
