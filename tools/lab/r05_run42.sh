#!/bin/bash
# round-5 GPU call 42: two-panel far updates (K = 512 on the persistent kernel's two-chunk form) below 8192 trailing columns too (CIMRGP_FAR_PAIR), one box
mkdir -p gpurun_out
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
one() { echo -n "$* : "; env CIMRGP_LIB_PATH=$T "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms']['potrf_alone'], d['stage_ms']['potrf_with_carried_rows'])"; }
{
one CIMRGP_NONE=0
one CIMRGP_FAR_PAIR=6144
one CIMRGP_FAR_PAIR=4096
one CIMRGP_FAR_PAIR=2048
one CIMRGP_NONE=0
one CIMRGP_FAR_PAIR=5120
} | tee gpurun_out/r05_far_pair.txt
