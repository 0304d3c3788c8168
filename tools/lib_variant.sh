#!/bin/bash
# Development build variants of the library for same-box A/B runs:
#   tools/lib_variant.sh <name> "<extra hipcc flags>" <file.hip> [...]   -> mindaudio_amd/lib/variants/<name>.so
# = the current objects with the named sources recompiled under the extra flags.  (tools/ab_variants.sh runs the step on each.)
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; shift 2
mkdir -p mindaudio_amd/lib/variants
objs=$(ls mindaudio_amd/lib/obj/*.o)
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $flags -c mindaudio_amd/csrc/$f -o /tmp/var_${name}_$f.o
  objs=$(echo "$objs" | grep -v "/$f.o"); objs="$objs /tmp/var_${name}_$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mindaudio_amd/lib/variants/$name.so $objs
echo "built variants/$name.so ($flags: $*)"
