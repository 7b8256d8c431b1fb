#!/bin/bash
# whole potrf (no carried rows) under tuning knobs.  Usage: bash tools/lab/exp_potrf_knobs.sh "ENV=.. ENV=.." ...   [SIZES="4096 8192"]
cd "$(dirname "$0")/../.."
for cfg in "$@"; do
  r=""
  for n in ${SIZES:-8192}; do
    v=$(env CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so $cfg python3 tools/potrf_time.py $n 8 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['potrf_ms'])")
    r="$r $n:$v"
  done
  echo "$cfg ->$r"
done
