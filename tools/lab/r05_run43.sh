#!/bin/bash
# round-5 GPU call 43: the carried rows' start in a factorisation that began with early panels (CIMRGP_ROWS_START_EARLY), one box
mkdir -p gpurun_out
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
one() { echo -n "$* : "; env CIMRGP_LIB_PATH=$T "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms']['potrf_with_carried_rows'])"; }
{
for rep in 1 2 3; do
one CIMRGP_ROWS_START_EARLY=6144
one CIMRGP_ROWS_START_EARLY=5632
done
one CIMRGP_ROWS_START_EARLY=5376
one CIMRGP_ROWS_START_EARLY=5888
} | tee gpurun_out/r05_rows_start_early.txt
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "staged or posterior or rows" 2>&1 | tail -2
