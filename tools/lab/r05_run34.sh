#!/bin/bash
# round-5 GPU call 34: the thresholds once more with the early panels in place, one box
mkdir -p gpurun_out
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
one() { echo -n "$* : "; env CIMRGP_LIB_PATH=$T "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step']['frac'])"; }
{
one CIMRGP_NONE=0
one CIMRGP_ROWS_CUS=160
one CIMRGP_ROWS_CUS=224
one CIMRGP_ROWS_START=5632
one CIMRGP_ROWS_START=6656
one CIMRGP_ROWS_START=7168
one CIMRGP_NONE=0
one CIMRGP_CHAIN_CUS=24
one CIMRGP_CHAIN_CUS=40
one CIMRGP_ROWS_BESIDE=2048
one CIMRGP_ROWS_BESIDE=3072
one CIMRGP_ROWS_BESIDE=0
one CIMRGP_EARLY_CUS=256
one CIMRGP_NONE=0
} | tee gpurun_out/r05_knob_scan4.txt
