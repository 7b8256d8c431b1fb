#!/bin/bash
# carried rows beside a fused one-queue tail: parity first, then the step under the knobs
cd "$(dirname "$0")/../.."
export CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so
CFGS="${CFGS}" bash tools/exp_bench.sh ${1:-r03ad}
