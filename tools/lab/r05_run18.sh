#!/bin/bash
# round-5 GPU call 18: the solves' panel inverses on the solve queue (CIMRGP_INV_ON_SOLVE=1) against the factorisation's stream (0), one box
mkdir -p gpurun_out
T=$PWD/cimrgp_amd/libcimrgp_tuning.so
one() { echo -n "$* : "; env CIMRGP_LIB_PATH=$T "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step']['frac'], d['stage_ms']['potrf_alone'])"; }
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -x -q -m gpu -k "posterior or staged or bench" 2>&1 | tail -2 || exit 1
{
for rep in 1 2 3; do
one CIMRGP_INV_ON_SOLVE=0
one CIMRGP_INV_ON_SOLVE=1
done
} | tee gpurun_out/r05_inv_on_solve.txt
