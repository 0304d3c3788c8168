#!/bin/bash
# same-box A/B of the FFN pair launch: HEAD's ffn_packed.hip (tools/build_head_lib.sh ffn_packed.hip) vs the working tree's
for i in 1 2 3; do
  echo -n "HEAD: "; MINDAUDIO_AMD_LIB=$PWD/mindaudio_amd/lib/libma_head.so HID=2048 python tools/ffn_pair_scan.py 2>&1 | tail -1
  echo -n "new:  "; HID=2048 python tools/ffn_pair_scan.py 2>&1 | tail -1
done
