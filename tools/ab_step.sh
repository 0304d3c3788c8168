#!/bin/bash
# same-box A/B of the whole north-star step: libma_head.so (tools/build_head_lib.sh <changed .hip files>) vs the working tree's library
run() { python bench.py --no-cpu-baseline --no-train-leg --no-sustained --steps 40 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], 'roofline', d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
for i in 1 2 3; do echo -n "HEAD: "; MINDAUDIO_AMD_LIB=$PWD/mindaudio_amd/lib/libma_head.so run; echo -n "new:  "; run; done
