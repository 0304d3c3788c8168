#!/bin/bash
# same-box A/B of the fbank kernel geometries (MA_FEAT_CFG = <waves per workgroup><waves per SIMD>) + the parity tests
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_features_gpu.py tests/test_collate_gpu.py -x -q 2>&1 | tail -5
for cfg in default; do
  for B in 64 512; do
    echo -n "cfg $cfg B $B: "; B=$B timeout 300 python tools/feat_bench.py 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' '.join('%s=%.1f'%(k,v) for k,v in d.items()))"
  done
done
