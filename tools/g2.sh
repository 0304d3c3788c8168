cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ecapa_gpu.py -x -q 2>&1 | tail -2
for v in 0 1 0 1; do for C in 512 1024; do MA_G8_BLOCK=$v python tools/ecapa_bench.py $C 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('block=$v C=$C ms', d['ms_per_batch'])"; done; done
