# same-box A/B of the training step: HEAD's library (tools/build_head_lib.sh) vs the working tree's
run() { python tools/train_bench.py --steps 10 --warmup 3 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do echo -n "HEAD: "; MINDAUDIO_AMD_LIB=$PWD/mindaudio_amd/lib/libma_head.so run; echo -n "new:  "; run; done
