#!/bin/bash
# whole potrf at several sizes under schedule knobs (tuning library): CFGS = one env assignment list per line
set -uo pipefail
cd "$(dirname "$0")/../.."
export CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so
SIZES=${SIZES:-"8192"}
while read -r cfg; do
  [ -z "$cfg" ] && continue
  echo "== $cfg"
  for n in $SIZES; do env $cfg timeout -k 10 100 python3 tools/potrf_time.py $n 5 2>&1 | grep -v amdgpu.ids; done
done <<CFGS
${CFGS:-CIMRGP_GEMM_PERS=256}
CFGS
