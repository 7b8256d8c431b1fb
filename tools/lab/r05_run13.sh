#!/bin/bash
# round-5 GPU call 13: compute units of the carried rows' far updates x the rows' start, one box
mkdir -p gpurun_out
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
one() { echo -n "$* : "; env CIMRGP_LIB_PATH=$T "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms']['potrf_alone'])"; }
{
one CIMRGP_NONE=0
for rs in 6144 5632 5120; do for rc in 128 160 176 192 208; do one CIMRGP_ROWS_START=$rs CIMRGP_ROWS_CUS=$rc; done; one CIMRGP_NONE=0; done
} | tee gpurun_out/r05_knob_scan2.txt
