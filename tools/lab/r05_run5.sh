#!/bin/bash
# round-5 GPU call 5: closed-form decode one pass ahead + flexible grid; whole GPU test suite
mkdir -p gpurun_out
python3 tools/gemm_bench.py --m 7936,6912,5888,4864 --k 256 --reps 20 --check 2>/dev/null | tee gpurun_out/r05_gemm_d.jsonl
CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning_s0.so CIMRGP_GEMM_PERS=256 python3 tools/lab/pers_stamps.py 7936 2>/dev/null | tee gpurun_out/r05_pers_stamps4.json
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_d.json; python3 -c "import json; d=json.load(open('gpurun_out/r05_bench_d.json')); print(d['value'], d['ms_per_step'], d['cholesky_frac_of_peak'], d['roofline']['frac'], d['stage_ms']['potrf_alone'])"
CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so CIMRGP_PERS_FLEX=0 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_d_noflex.json; python3 -c "import json; d=json.load(open('gpurun_out/r05_bench_d_noflex.json')); print('noflex', d['value'], d['ms_per_step'], d['cholesky_frac_of_peak'], d['roofline']['frac'], d['stage_ms']['potrf_alone'])"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
