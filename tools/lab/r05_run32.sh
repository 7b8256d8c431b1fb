#!/bin/bash
# round-5 GPU call 32: CIMRGP_EARLY_PANELS 3..8, one box
mkdir -p gpurun_out
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
one() { echo -n "$* : "; env CIMRGP_LIB_PATH=$T "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launches'], d['roofline']['whole_step']['frac'], d['stage_ms']['potrf_alone'])"; }
{
one CIMRGP_EARLY_PANELS=0
one CIMRGP_EARLY_PANELS=3
one CIMRGP_EARLY_PANELS=4
one CIMRGP_EARLY_PANELS=5
one CIMRGP_EARLY_PANELS=6
one CIMRGP_EARLY_PANELS=8
one CIMRGP_EARLY_PANELS=0
one CIMRGP_EARLY_PANELS=4
one CIMRGP_EARLY_PANELS=5
one CIMRGP_EARLY_PANELS=6
one CIMRGP_EARLY_PANELS=4 CIMRGP_ROWS_BESIDE=3072
one CIMRGP_EARLY_PANELS=4 CIMRGP_ROWS_START=5632
} | tee gpurun_out/r05_early_panels2.txt
