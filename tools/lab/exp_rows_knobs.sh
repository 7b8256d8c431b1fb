#!/bin/bash
# carried rows with the one-launch chain (k_rows_step): where to start them, how many compute units for their far
# updates, where the factorisation's one-queue tail begins.  Usage: bash tools/lab/exp_rows_knobs.sh "ENV=.. ENV=.." ...
cd "$(dirname "$0")/../.."
for cfg in "$@"; do
  v=$(env CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so $cfg python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3), round(d['stage_ms']['potrf_with_carried_rows'],3))")
  echo "$cfg -> $v"
done
