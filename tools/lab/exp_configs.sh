#!/bin/bash
cd "$(dirname "$0")/../.."
export CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so
for fm in 0 768 100000000; do
  for c in 3 4; do
    echo "== fused_max=$fm config $c"
    CIMRGP_FUSED_MAX=$fm timeout -k 10 300 python3 bench.py --config $c --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['fit_s'],3), round(d['predict_s'],3), [round(v,1) for v in d['layer_fit_ms']])"
  done
done
