#!/bin/bash
# Development A/B of convmid_pw2.hip build variants (as tools/ffn_variants.sh):  build "name:-Dflags" ... | run [script.py]
set -e
cd "$(dirname "$0")/.."
mkdir -p mindaudio_amd/lib/variants
if [ "$1" = build ]; then
  shift
  rm -f mindaudio_amd/lib/variants/*.so
  others=$(ls mindaudio_amd/lib/obj/*.o | grep -v "/convmid_pw2.hip.o")
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    [ "$flags" = "$spec" ] && flags=""
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $flags -c mindaudio_amd/csrc/convmid_pw2.hip -o /tmp/cmv_$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mindaudio_amd/lib/variants/$name.so $others /tmp/cmv_$name.o
    echo "built $name ($flags)"
  done
else
  script=${2:-tools/convmod_timeline.py}
  for rep in 1 2; do
    for so in mindaudio_amd/lib/variants/*.so; do
      echo "== $(basename $so .so)"; MINDAUDIO_AMD_LIB=$PWD/$so timeout 300 python $script 2>&1 | grep -v amdgpu.ids | head -${HEAD:-1}
    done
  done
fi
