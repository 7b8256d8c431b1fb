#!/bin/bash
# Build libcimrgp.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
OBJS=()
pids=()
for f in api gemm_nt potrf gram solve misc reduced layer comm; do
    if [ ! -f "$f.o" ] || [ "$f.hip" -nt "$f.o" ] || [ common.hpp -nt "$f.o" ] || [ gemm_tile.hpp -nt "$f.o" ] || [ ../../include/cimrgp.h -nt "$f.o" ]; then
        $HIPCC $FLAGS -c "$f.hip" -o "$f.o" &
        pids+=($!)
    fi
    OBJS+=("$f.o")
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libcimrgp.so "${OBJS[@]}" -ldl
echo "built $(cd .. && pwd)/libcimrgp.so"
