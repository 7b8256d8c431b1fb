#!/bin/bash
# round-5 GPU call 4: flag-synchronised stages of the persistent update against the barrier form (same box)
mkdir -p gpurun_out
for v in "" _nf; do echo "== tuning$v (nf = s_barrier form)"; CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning$v.so CIMRGP_GEMM_PERS=256 python3 tools/gemm_bench.py --m 7936,6912,5888,4864 --k 256 --reps 20 --check 2>/dev/null; done | tee gpurun_out/r05_gemm_c.txt
CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning_s0.so CIMRGP_GEMM_PERS=256 python3 tools/lab/pers_stamps.py 7936 2>/dev/null | tee gpurun_out/r05_pers_stamps3.json
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -4
for v in "" _nf; do CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning$v.so python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_c$v.json; python3 -c "import json; d=json.load(open('gpurun_out/r05_bench_c$v.json')); print('$v', d['value'], d['ms_per_step'], d['cholesky_frac_of_peak'], d['roofline']['frac'], d['stage_ms']['potrf_alone'])"; done
