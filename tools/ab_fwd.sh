# same-box A/B of the evaluation forward: HEAD's library (tools/build_head_lib.sh <files>) against the working tree's, three rounds
run() { python bench.py --step-only --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do echo -n "HEAD: "; MINDAUDIO_AMD_LIB=$PWD/mindaudio_amd/lib/libma_head.so run; echo -n "new:  "; run; done
