#!/bin/bash
# round-5 GPU call 9: the persistent kernel with the chunk loop (K = 256 nkc) against the commit before it, same box, alternating
mkdir -p gpurun_out
for rep in 1 2; do for v in libcimrgp_prev.so libcimrgp.so; do echo "== $v"; CIMRGP_LIB_PATH=$PWD/cimrgp_amd/$v python3 tools/gemm_bench.py --m 7936,6912,5888,4864 --k 256 --reps 20 2>/dev/null; done; done | tee gpurun_out/r05_chunk_ab.txt
for v in libcimrgp_prev.so libcimrgp.so; do CIMRGP_LIB_PATH=$PWD/cimrgp_amd/$v python3 tools/potrf_sweep.py --sizes 8192,12288,16384 2>/dev/null; done | tee -a gpurun_out/r05_chunk_ab.txt
for ch in 1 3; do echo "== tuning, CIMRGP_PERS_CHUNKS=$ch"; CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so CIMRGP_PERS_CHUNKS=$ch python3 tools/potrf_sweep.py --sizes 12288,16384,32768 2>/dev/null; CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so CIMRGP_PERS_CHUNKS=$ch CIMRGP_GEMM_PERS_F32=1 python3 tools/potrf_sweep.py --sizes 8192,16384 --dtype f32 2>/dev/null; done | tee -a gpurun_out/r05_chunk_ab.txt
python3 tools/gemm_bench.py --m 7936,12032 --k 512 --reps 10 --check 2>/dev/null | tee -a gpurun_out/r05_chunk_ab.txt
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -x -q -m gpu -k "far_updates or n16384 or 18432 or 32768 or syrk or potrf" 2>&1 | tail -3
