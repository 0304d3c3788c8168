#!/bin/bash
# builds every micro-benchmark next to its source: tools/ubench/<name> from tools/ubench/<name>.hip (gfx950)
cd "$(dirname "$0")" && for f in *.hip; do hipcc --offload-arch=gfx950 -O3 -o "${f%.hip}" "$f" || exit 1; done
