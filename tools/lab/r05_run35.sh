#!/bin/bash
# round-5 GPU call 35: rows' start and early units with the early panels in place, one box
mkdir -p gpurun_out
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
one() { echo -n "$* : "; env CIMRGP_LIB_PATH=$T "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step']['frac'])"; }
{
one CIMRGP_NONE=0
one CIMRGP_ROWS_START=5632
one CIMRGP_ROWS_START=5120
one CIMRGP_ROWS_START=5888
one CIMRGP_ROWS_START=5632 CIMRGP_EARLY_CUS=256
one CIMRGP_ROWS_START=5632 CIMRGP_ROWS_CUS=224
one CIMRGP_ROWS_START=5632 CIMRGP_ROWS_CUS=208
one CIMRGP_NONE=0
one CIMRGP_ROWS_START=5632
one CIMRGP_ROWS_START=5632 CIMRGP_EARLY_PANELS=10
one CIMRGP_ROWS_START=5632 CIMRGP_EARLY_PANELS=6
one CIMRGP_ROWS_START=5632 CIMRGP_TAIL_BELOW=5376
} | tee gpurun_out/r05_knob_scan5.txt
