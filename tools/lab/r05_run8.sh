#!/bin/bash
# round-5 GPU call 8: head tiles stored directly at the end of their own pass, by launch length (same box, two alternating passes)
mkdir -p gpurun_out
CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so CIMRGP_HEAD_DIRECT=99 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "potrf" 2>&1 | tail -3
for rep in 1 2; do
for hd in 0 5 6 7 8 99; do
  CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so CIMRGP_HEAD_DIRECT=$hd python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python3 -c "import json; d=json.load(open('/tmp/b.json')); print('head_direct_max_rounds $hd:', round(d['value'],2), round(d['ms_per_step'],3), round(d['cholesky_frac_of_peak'],4), round(d['roofline']['frac'],4), round(d['stage_ms']['potrf_alone'],3))"
done; done | tee gpurun_out/r05_head_direct_scan.txt
