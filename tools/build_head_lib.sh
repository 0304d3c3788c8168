#!/bin/bash
# usage: tools/build_head_lib.sh <file.hip> [...]  -> mindaudio_amd/lib/libma_head.so = the current objects with the named sources
# taken from git HEAD instead of the working tree (same-box A/B of a kernel change: MINDAUDIO_AMD_LIB=$PWD/mindaudio_amd/lib/libma_head.so)
set -e
cd "$(dirname "$0")/.."
objs=$(ls mindaudio_amd/lib/obj/*.o)
for f in "$@"; do
  git show HEAD:mindaudio_amd/csrc/$f > mindaudio_amd/csrc/_head_$f
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -c mindaudio_amd/csrc/_head_$f -o /tmp/_head_$f.o
  rm mindaudio_amd/csrc/_head_$f
  objs=$(echo "$objs" | grep -v "/$f.o"); objs="$objs /tmp/_head_$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mindaudio_amd/lib/libma_head.so $objs
ls -la mindaudio_amd/lib/libma_head.so
