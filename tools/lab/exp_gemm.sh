#!/bin/bash
# stand-alone trailing-update rate, old kernel against the persistent one, then whole potrf with the split
set -uo pipefail
cd "$(dirname "$0")/../.."
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-x}
export CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning.so
L=$OUT/${TAG}_gemm.log; : > $L
for pers in 0 256 224; do
  echo "== pers=$pers k=256" >> $L
  CIMRGP_GEMM_PERS=$pers timeout -k 10 120 python3 tools/gemm_bench.py --m 7936,6912,5888,4864,4096 --k 256 --reps 20 --check 2>&1 | grep -v amdgpu.ids >> $L || { echo "FAILED pers=$pers" >> $L; cat $L; exit 1; }
done
L2=$OUT/${TAG}_potrf.log; : > $L2
for cfg in "0 0" "256 0" "256 32" "256 64" "256 96"; do
  set -- $cfg
  echo "== pers=$1 chain_cus=$2" >> $L2
  for n in 8192 16384; do
    CIMRGP_GEMM_PERS=$1 CIMRGP_CHAIN_CUS=$2 timeout -k 10 120 python3 tools/potrf_time.py $n 4 2>&1 | grep -v amdgpu.ids >> $L2 || { echo "FAILED $cfg" >> $L2; cat $L2; exit 1; }
  done
done
cat $L $L2
