#!/bin/bash
# Measurement build of the library: the schedule's thresholds (common.hpp: Knobs) can be overridden by
# CIMRGP_* environment variables.  Output: cimrgp_amd/libcimrgp_tuning.so, selected with
# CIMRGP_LIB_PATH=cimrgp_amd/libcimrgp_tuning.so.  The product library (build.sh) reads no environment.
set -euo pipefail
cd "$(dirname "$0")/../cimrgp_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DCIMRGP_TUNING"
mkdir -p tuning_obj
pids=()
OBJS=()
for f in api gemm_nt potrf gram solve misc reduced layer comm; do
    if [ ! -f "tuning_obj/$f.o" ] || [ "$f.hip" -nt "tuning_obj/$f.o" ] || [ common.hpp -nt "tuning_obj/$f.o" ] || [ gemm_tile.hpp -nt "tuning_obj/$f.o" ] || [ ../../include/cimrgp.h -nt "tuning_obj/$f.o" ]; then
        $HIPCC $FLAGS -c "$f.hip" -o "tuning_obj/$f.o" &
        pids+=($!)
    fi
    OBJS+=("tuning_obj/$f.o")
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libcimrgp_tuning.so "${OBJS[@]}" -ldl
echo "built $(cd .. && pwd)/libcimrgp_tuning.so"
