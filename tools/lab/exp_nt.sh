#!/bin/bash
# non-temporal C-stream variants of the persistent kernel (9 stores, 10 both, 11 loads): correct results
cd "$(dirname "$0")/../.."
for v in "" _e9 _e10 _e11; do
  echo "== variant ${v:-base}"
  CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning$v.so CIMRGP_GEMM_PERS=256 timeout -k 10 100 python3 tools/gemm_bench.py --m 7936,5888,3840 --k 256 --reps 20 --check 2>&1 | grep -v amdgpu.ids
done
