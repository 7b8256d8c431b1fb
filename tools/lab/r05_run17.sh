#!/bin/bash
# round-5 GPU call 17: the combined head + bulk launch also while the carried rows run (CIMRGP_HEADS_ROWS=1), one box
mkdir -p gpurun_out
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
one() { echo -n "$* : "; env CIMRGP_LIB_PATH=$T "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms']['potrf_alone'], d.get('selfcheck'))"; }
{
one CIMRGP_NONE=0
one CIMRGP_HEADS_ROWS=1
one CIMRGP_HEADS_ROWS=1 CIMRGP_ROWS_CUS=176
one CIMRGP_HEADS_ROWS=1 CIMRGP_ROWS_CUS=160
one CIMRGP_NONE=0
one CIMRGP_HEADS_ROWS=1
one CIMRGP_HEADS_ROWS=1 CIMRGP_ROWS_BESIDE=2048
one CIMRGP_HEADS_ROWS=1 CIMRGP_ROWS_START=6656
} | tee gpurun_out/r05_heads_rows.txt
CIMRGP_LIB_PATH=$T CIMRGP_HEADS_ROWS=1 timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "rows" 2>&1 | tail -2
