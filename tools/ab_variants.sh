#!/bin/bash
# same-box A/B of the north-star step over every library in mindaudio_amd/lib/variants/ (tools/lib_variant.sh), three rounds
run() { python bench.py --no-cpu-baseline --no-train-leg --no-sustained --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%.1f utt/s  %.4f ms  ffn %.2f us  fbank %.2f us' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'] * 1e3, d['roofline_fbank']['kernel_ms'] * 1e3))"; }
for i in 1 2 3; do
  for so in mindaudio_amd/lib/variants/*.so; do printf "%-12s " "$(basename $so .so)"; MINDAUDIO_AMD_LIB=$PWD/$so run; done
done
