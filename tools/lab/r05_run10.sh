#!/bin/bash
# round-5 GPU call 10: prologue of the persistent update (LDS-only barrier, first pass through a descriptor without extent)
mkdir -p gpurun_out
python3 tools/gemm_bench.py --m 7936,6912,5888,4864 --k 256 --reps 20 --check 2>/dev/null | tee gpurun_out/r05_gemm_e.jsonl
CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning_s0.so CIMRGP_GEMM_PERS=256 python3 tools/lab/pers_stamps.py 7936 2>/dev/null | tee gpurun_out/r05_pers_stamps5.json
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_e.json; python3 -c "import json; d=json.load(open('gpurun_out/r05_bench_e.json')); print(d['value'], d['ms_per_step'], d['cholesky_frac_of_peak'], d['roofline']['frac'], d['stage_ms']['potrf_alone'])"
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3
