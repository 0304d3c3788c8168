cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/g4_pytest_full.log 2>&1; tail -3 gpurun_out/g4_pytest_full.log
python bench.py > gpurun_out/bench_r6_mid.json 2> gpurun_out/bench_r6_mid.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r6_mid.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], 'train', d['train_dp']['ms_per_step'], 'hybrid', d['train_dp_hybrid']['ms_per_step'], 'cfg5', d['cfg5']['c512']['ms'], d['cfg5']['c1024']['ms'], 'cfg3', d['cfg3']['eval']['ms'], d['cfg3']['train_mode_forward'])
PY
