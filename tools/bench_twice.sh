for i in 1 2; do python bench.py --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], 'ffn_ms', d['roofline']['kernel_ms'], 'fbank_ms', d['roofline_fbank']['kernel_ms'])"; done
python tools/gemm_bench.py 2>&1 | tail -12
