#!/bin/bash
# round-5 GPU call 15: wave priority of the factorisation's own persistent update above the carried rows' (s_setprio 1 / 2 in the LOWER form), one box
mkdir -p gpurun_out
one() { echo -n "$* : "; env "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms']['potrf_alone'])"; }
L=$PWD/cimrgp_amd
{
for rep in 1 2; do
one CIMRGP_LIB_PATH=$L/libcimrgp_tuning.so
one CIMRGP_LIB_PATH=$L/libcimrgp_tuning_p1.so
one CIMRGP_LIB_PATH=$L/libcimrgp_tuning_p2.so
one CIMRGP_LIB_PATH=$L/libcimrgp_tuning_p1.so CIMRGP_ROWS_CUS=224
one CIMRGP_LIB_PATH=$L/libcimrgp_tuning_p2.so CIMRGP_ROWS_CUS=224
one CIMRGP_LIB_PATH=$L/libcimrgp_tuning_p1.so CIMRGP_ROWS_CUS=256
one CIMRGP_LIB_PATH=$L/libcimrgp_tuning_p1.so CIMRGP_ROWS_CUS=224 CIMRGP_ROWS_START=7168
done
} | tee gpurun_out/r05_prio_scan.txt
