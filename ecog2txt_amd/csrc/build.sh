#!/usr/bin/env bash
# Build libecog2txt_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
# One object per source, compiled in parallel and only when the source (or a header) is newer; then one link.
# E2T_DEBUG=1: the diagnostics build instead -- libecog2txt_hip_dbg.so, compiled with -DE2T_DEBUG: the E2T_* environment
# switches that select kernel variants and the phase-stamp buffer of the recurrences exist only there (csrc/common.h);
# the Python side loads it when E2T_DEBUG_LIB=1 (hip_lib.py) -- scripts/ only, never the product path.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
inc="$here/../../include"
out="$here/../libecog2txt_hip.so"
obj="$here/build"
dbg=()
if [ "${E2T_DEBUG:-0}" = 1 ]; then out="$here/../libecog2txt_hip_dbg.so"; obj="$here/build_dbg"; dbg=(-DE2T_DEBUG); fi
mkdir -p "$obj"
srcs=(runtime gemm lstm lstm_big elementwise comm)
pids=()
for s in "${srcs[@]}"; do
    o="$obj/$s.o"
    if [ ! -f "$o" ] || [ "$here/$s.hip" -nt "$o" ] || [ "$here/common.h" -nt "$o" ] || [ "$inc/ecog2txt_hip.h" -nt "$o" ]; then
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$inc" -I"$here" -c "$here/$s.hip" -o "$o" -Rpass-analysis=kernel-resource-usage "${dbg[@]}" "$@" > "$obj/$s.log" 2>&1 || { cat "$obj/$s.log" | grep -v "remark:" | head -40; exit 1; } &
        pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
objs=()
for s in "${srcs[@]}"; do objs+=("$obj/$s.o"); done
hipcc --offload-arch=gfx950 -fPIC -shared "${objs[@]}" -ldl -o "$out"
echo "built $out"
# test helper (tests/native/occupy.hip: a CU-occupying spin kernel for the RCCL-contention test) -- not part of the product library
th="$here/../../tests/native"
if [ -f "$th/occupy.hip" ] && { [ ! -f "$th/libe2t_test_occupy.so" ] || [ "$th/occupy.hip" -nt "$th/libe2t_test_occupy.so" ]; }; then
    hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -shared "$th/occupy.hip" -o "$th/libe2t_test_occupy.so" > "$obj/occupy.log" 2>&1 || { cat "$obj/occupy.log"; exit 1; }
fi
if [ -f "$th/segv_bt.c" ] && { [ ! -f "$th/libe2t_test_segv_bt.so" ] || [ "$th/segv_bt.c" -nt "$th/libe2t_test_segv_bt.so" ]; }; then
    gcc -O1 -g -shared -fPIC "$th/segv_bt.c" -o "$th/libe2t_test_segv_bt.so" || exit 1
fi
