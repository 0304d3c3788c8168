cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_train_kernels_gpu.py tests/test_train_step_gpu.py tests/test_conformer_encoder_gpu.py -m gpu -q -x -k "decoder_attention or hybrid" 2>&1 | tail -2
python tools/mha_small_bench.py 2>&1 | tail -6
for k in 1 2; do python tools/train_bench.py --steps 20 --warmup 5 --ctc-weight 0.3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hybrid ms', d['ms_per_step'])"; done
