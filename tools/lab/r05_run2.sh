#!/bin/bash
# round-5 GPU call 2: the update kernel after the vector-instruction diet (operand negation in the multiply, buffer-addressed C stream)
mkdir -p gpurun_out
python3 tools/gemm_bench.py --m 7936,6912,5888,4864 --k 256 --reps 20 --check 2>/dev/null | tee gpurun_out/r05_gemm_b.jsonl
CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning_e20.so CIMRGP_GEMM_PERS=256 python3 tools/lab/pers_stamps.py 7936 2>/dev/null | tee gpurun_out/r05_pers_stamps2.json
bash tools/lab/exp_variants.sh run20 2>&1 | tee gpurun_out/r05_pers_variants.txt
python3 tools/gemm_bench.py --m 8192,16384 --k 256 --reps 10 --check --dtype f32 2>/dev/null | tee -a gpurun_out/r05_gemm_b.jsonl
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -4
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_b.json
python3 -c "import json; d=json.load(open('gpurun_out/r05_bench_b.json')); print(d['value'], d['ms_per_step'], d['cholesky_frac_of_peak'], d['roofline']['frac'], d['stage_ms']['potrf_alone'])"
