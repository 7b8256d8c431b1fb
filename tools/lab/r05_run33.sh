#!/bin/bash
# round-5 GPU call 33: CIMRGP_EARLY_PANELS 6..14 and the early updates' compute units, one box
mkdir -p gpurun_out
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
one() { echo -n "$* : "; env CIMRGP_LIB_PATH=$T "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step']['frac'], d['stage_ms']['potrf_alone'])"; }
{
one CIMRGP_EARLY_PANELS=0
one CIMRGP_EARLY_PANELS=6
one CIMRGP_EARLY_PANELS=7
one CIMRGP_EARLY_PANELS=8
one CIMRGP_EARLY_PANELS=10
one CIMRGP_EARLY_PANELS=12
one CIMRGP_EARLY_PANELS=14
one CIMRGP_EARLY_PANELS=6 CIMRGP_EARLY_CUS=256
one CIMRGP_EARLY_PANELS=6 CIMRGP_EARLY_CUS=240
one CIMRGP_EARLY_PANELS=6 CIMRGP_EARLY_CUS=208
one CIMRGP_EARLY_PANELS=8 CIMRGP_EARLY_CUS=256
one CIMRGP_EARLY_PANELS=7 CIMRGP_ROWS_START=6400
one CIMRGP_EARLY_PANELS=0
} | tee gpurun_out/r05_early_panels3.txt
