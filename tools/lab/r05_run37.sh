#!/bin/bash
# round-5 GPU call 37: the carried rows catching up during the early panels (CIMRGP_EARLY_ROWS = first early panel they follow), one box
mkdir -p gpurun_out
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
CIMRGP_LIB_PATH=$T CIMRGP_EARLY_ROWS=0 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "posterior or staged" 2>&1 | tail -2 || exit 1
one() { echo -n "$* : "; env CIMRGP_LIB_PATH=$T "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step']['frac'])"; }
{
one CIMRGP_EARLY_ROWS=-1
one CIMRGP_EARLY_ROWS=0
one CIMRGP_EARLY_ROWS=2
one CIMRGP_EARLY_ROWS=4
one CIMRGP_EARLY_ROWS=6
one CIMRGP_EARLY_ROWS=-1
one CIMRGP_EARLY_ROWS=4 CIMRGP_EARLY_PANELS=10
one CIMRGP_EARLY_ROWS=4 CIMRGP_EARLY_PANELS=12
one CIMRGP_EARLY_ROWS=6 CIMRGP_EARLY_PANELS=12
} | tee gpurun_out/r05_early_rows.txt
CIMRGP_LIB_PATH=$T CIMRGP_EARLY_ROWS=4 python3 bench.py --steps 6 --warmup 2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('parity', d.get('parity_ok'), d.get('parity_rel_err_mean'), d.get('parity_rel_err_var'), d['value'])" | tee -a gpurun_out/r05_early_rows.txt
