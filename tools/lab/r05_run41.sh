#!/bin/bash
# round-5 GPU call 41: the early panels' chains on the four-wave kernels (CIMRGP_EARLY_ALONE=0), and a few neighbours, one box
mkdir -p gpurun_out
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
one() { echo -n "$* : "; env CIMRGP_LIB_PATH=$T "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step']['frac'])"; }
{
for rep in 1 2 3; do
one CIMRGP_EARLY_ALONE=1
one CIMRGP_EARLY_ALONE=0
done
one CIMRGP_EARLY_ALONE=0 CIMRGP_EARLY_PANELS=10
one CIMRGP_EARLY_ALONE=0 CIMRGP_EARLY_PANELS=6
} | tee gpurun_out/r05_early_alone.txt
