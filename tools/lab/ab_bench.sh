#!/bin/bash
# A/B of two library builds on one box, alternating:  bash tools/lab/ab_bench.sh libA.so libB.so [rounds]
cd "$(dirname "$0")/../.."
for r in $(seq 1 ${3:-3}); do
  for lib in "$1" "$2"; do
    v=$(CIMRGP_LIB_PATH=$PWD/cimrgp_amd/$lib python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['stage_ms']['potrf_alone'],3), round(d['stage_ms']['potrf_with_carried_rows'],3))")
    echo "$lib -> $v"
  done
done
