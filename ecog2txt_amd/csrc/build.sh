#!/usr/bin/env bash
# Build libecog2txt_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
inc="$here/../../include"
out="$here/../libecog2txt_hip.so"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I"$inc" -I"$here" \
    "$here/runtime.hip" "$here/gemm.hip" "$here/lstm.hip" "$here/elementwise.hip" \
    -o "$out" "$@"
echo "built $out"
