# same-box A/B: HEAD's library (tools/build_head_lib.sh) vs the working tree's
run() { python bench.py --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do echo -n "HEAD: "; MINDAUDIO_AMD_LIB=$PWD/mindaudio_amd/lib/libma_head.so run; echo -n "new:  "; run; done
