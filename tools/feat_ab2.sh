#!/bin/bash
# same-box A/B of fbank-kernel build variants: the default library against every mindaudio_amd/lib/variants/*.so (tools/lib_variant.sh)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for so in default mindaudio_amd/lib/variants/*.so; do
    for B in 64 512; do
      if [ "$so" = default ]; then unset MINDAUDIO_AMD_LIB; else export MINDAUDIO_AMD_LIB=$PWD/$so; fi
      echo -n "$(basename $so .so) B=$B: "; B=$B timeout 300 python tools/feat_bench.py 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' '.join('%s=%.1f'%(k,v) for k,v in d.items() if k in ('fbank_main_only_us','fbank_with_topdb_us','stft_us','kaldi_us','fbank_main_GBs')))"
    done
  done
done
