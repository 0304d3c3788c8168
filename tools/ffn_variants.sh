#!/bin/bash
# Development A/B of ffn_packed.hip build variants on one box.
#   build (here, no GPU):   bash tools/ffn_variants.sh build "name1:-DMA_FFN_WT=1" "name2:-DMA_FFN_PROF" ...
#   run (GPU box):          bash tools/ffn_variants.sh run [script.py]      -> one line per variant (default tools/ffn_pair_scan.py)
#   SRC=ffn_train.hip (default ffn_packed.hip) picks the translation unit that is rebuilt.
# Variants are libmindaudio_amd.so with only $SRC rebuilt under the extra flags: mindaudio_amd/lib/variants/<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p mindaudio_amd/lib/variants
SRC=${SRC:-ffn_packed.hip}
if [ "$1" = build ]; then
  shift
  rm -f mindaudio_amd/lib/variants/*.so
  others=$(ls mindaudio_amd/lib/obj/*.o | grep -v "/$SRC.o")
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    [ "$flags" = "$spec" ] && flags=""
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $flags -c mindaudio_amd/csrc/$SRC -o /tmp/ffnv_$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mindaudio_amd/lib/variants/$name.so $others /tmp/ffnv_$name.o
    echo "built $name ($flags)"
  done
else
  script=${2:-tools/ffn_pair_scan.py}
  for rep in 1 2; do
    for so in mindaudio_amd/lib/variants/*.so; do
      echo "== $(basename $so .so)"; MINDAUDIO_AMD_LIB=$PWD/$so HID=${HID:-2048} timeout 300 python $script 2>&1 | tail -${TAIL:-1}
    done
  done
fi
