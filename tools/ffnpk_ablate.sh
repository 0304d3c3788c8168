#!/bin/bash
# usage (GPU box): bash tools/ffnpk_ablate.sh  -- times ffn_packed_kernel under each development ablation
for a in 0 1 2 3 4 5 6 7; do
  echo -n "ablate=$a : "; MA_FFNPK_ABLATE=$a timeout 120 python tools/ffn_bench.py 2>&1 | grep "packed ffn"
done
