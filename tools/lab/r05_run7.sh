#!/bin/bash
# round-5 GPU call 7: FP32 -- the persistent update (8-stage passes, staged negation, buffer-addressed C, flags) against the tile kernel
mkdir -p gpurun_out
for f32 in 0 1; do
  echo "== CIMRGP_GEMM_PERS_F32=$f32"
  CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so CIMRGP_GEMM_PERS_F32=$f32 python3 tools/gemm_bench.py --m 7936,8192,16128 --k 256 --reps 20 --check --dtype f32 2>/dev/null
  CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so CIMRGP_GEMM_PERS_F32=$f32 python3 tools/potrf_sweep.py --sizes 8192,16384 --dtype f32 2>/dev/null
done | tee gpurun_out/r05_f32_pers.txt
