#!/bin/bash
# the N = 8192 step (with carried rows) under schedule knobs: tuning library
set -uo pipefail
cd "$(dirname "$0")/../.."
OUT=gpurun_out; TAG=${1:-x}
export CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so
L=$OUT/${TAG}_benchknobs.log; : > $L
while read -r cfg; do
  [ -z "$cfg" ] && continue
  echo "== $cfg" >> $L
  env $cfg timeout -k 10 200 python3 bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(value=round(d['value'],2), ms=round(d['ms_per_step'],3), potrf_rows=round(d['stage_ms']['potrf_with_carried_rows'],3), potrf_alone=round(d['stage_ms']['potrf_alone'],3), roof=round(d['roofline']['frac'],3))))" >> $L 2>&1 || echo FAILED >> $L
done <<CFGS
${CFGS:-CIMRGP_GEMM_PERS=0}
CFGS
cat $L
