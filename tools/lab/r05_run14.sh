#!/bin/bash
# round-5 GPU call 14: around CIMRGP_ROWS_CUS=192, one box
mkdir -p gpurun_out
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
one() { echo -n "$* : "; env CIMRGP_LIB_PATH=$T "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms']['potrf_alone'])"; }
{
one CIMRGP_NONE=0
one CIMRGP_ROWS_CUS=192
one CIMRGP_ROWS_CUS=184
one CIMRGP_ROWS_CUS=200
one CIMRGP_ROWS_CUS=192 CIMRGP_ROWS_START=6656
one CIMRGP_ROWS_CUS=192 CIMRGP_ROWS_START=5888
one CIMRGP_ROWS_CUS=192 CIMRGP_CHAIN_CUS=24
one CIMRGP_ROWS_CUS=192 CIMRGP_CHAIN_CUS=40
one CIMRGP_ROWS_CUS=192 CIMRGP_CHAIN_CUS=48
one CIMRGP_ROWS_CUS=192
one CIMRGP_ROWS_CUS=192 CIMRGP_ROWS_BESIDE=2048
one CIMRGP_ROWS_CUS=192 CIMRGP_ROWS_BESIDE=3072
one CIMRGP_ROWS_CUS=192 CIMRGP_ROWS_PAIR=4096
one CIMRGP_ROWS_CUS=192 CIMRGP_ROWS_STEP=0
one CIMRGP_ROWS_CUS=192 CIMRGP_PERS_MIN_TILES=768
one CIMRGP_ROWS_CUS=192
one CIMRGP_NONE=0
} | tee gpurun_out/r05_knob_scan3.txt
