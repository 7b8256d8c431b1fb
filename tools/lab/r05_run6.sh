#!/bin/bash
# round-5 GPU call 6: the flexible grid of the persistent update inside the factorisation (same box, alternating)
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "0 7" "16 9" "16 8" "16 7" "16 6" "8 7" "24 8"; do
  set -- $cfg
  CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so CIMRGP_PERS_FLEX=$1 CIMRGP_PERS_FLEX_MIN=$2 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python3 -c "import json; d=json.load(open('/tmp/b.json')); print('flex $1 min_rounds $2:', round(d['value'],2), round(d['ms_per_step'],3), round(d['cholesky_frac_of_peak'],4), round(d['roofline']['frac'],4), round(d['stage_ms']['potrf_alone'],3))"
done; done | tee gpurun_out/r05_flex_scan.txt
